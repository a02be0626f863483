"""Trainer-facing surface of the device-side training step (BASELINE configs[4]): what connects `v3d.train` to
`loss.backward()` and to checkpoints in the reference's own key names.

The reference trains `LlavaQwenForCausalLM` under the HF Trainer: `loss = model(input_ids, labels=..., images=..., video_dict=...).loss;
loss.backward(); optimizer.step()` (llava_qwen.py:121-205, llava_trainer.py, train_3d.py:1572).  `LlavaQwenTrainable` is an `nn.Module`
whose parameters are `nn.Parameter`s holding the tensors of `v3d.train`'s layouts (fused q|k|v rows, stacked gate|up rows, the tower's
72-wide heads zero-padded to 128); `forward(...)` returns the loss as a tensor with a `grad_fn`: the explicit device-side
forward + backward (`train.sample_forward_backward`, every operation a C-ABI call) runs inside a `torch.autograd.Function`, and
`loss.backward()` hands its gradients to the parameters' `.grad` - so any torch optimizer (or `train.AdamW` / `train.ZeroAdamW` on
`module.param_tree()`) and a Trainer-style loop work unchanged.  `from_reference_state_dict` / `reference_state_dict` convert between the
checkpoint's keys (`model.layers.N.self_attn.q_proj.weight`, `model.vision_tower...`, `model.mm_projector.{0,2}`, ...) and the training
layouts, exactly (the padding is zeros and is dropped on the way back).

One sample per call, as the reference trains (`per_device_train_batch_size 1`, train_multi.sh:58).  Not here: DeepSpeed's engine, LoRA,
the `mlp` / `score` grounding heads.
"""
import torch
import torch.nn as nn

from . import ops, train
from ._native import V3DError
from .token_ids import IGNORE_INDEX, IMAGE_TOKEN_INDEX

VIT = "model.vision_tower.vision_tower.vision_model."


def _flatten(tree, prefix=""):
    out = []
    if isinstance(tree, dict):
        for k in tree:
            out += _flatten(tree[k], f"{prefix}.{k}" if prefix else str(k))
    elif isinstance(tree, list):
        for i, v in enumerate(tree):
            out += _flatten(v, f"{prefix}.{i}")
    else:
        out.append((prefix, tree))
    return out


class _SampleLoss(torch.autograd.Function):
    """forward: the whole sample's forward AND backward on the device (the gradients are a by-product of the one pass that exists);
    backward: hands them over, scaled by the incoming gradient of the loss (1 for a plain loss.backward(), 1 / steps under the
    Trainer's gradient accumulation)."""

    @staticmethod
    def forward(ctx, module, args, *flat_params):
        loss, grads = train.sample_forward_backward(module.param_tree(), *args, recompute=module.recompute)
        by_name = dict(_flatten(grads))
        if set(by_name) != set(module.names):
            raise V3DError("the gradient tree does not match the parameter tree")
        ctx.grads = [by_name[n] for n in module.names]
        return loss.detach().clone()

    @staticmethod
    def backward(ctx, grad_loss):
        scale = float(grad_loss)
        out = []
        for g in ctx.grads:
            if scale != 1.0:
                if g.dtype in (torch.bfloat16, torch.float16):
                    z = torch.zeros_like(g)
                    ops.axpy(z, g.contiguous(), scale)      # 16-bit scale on the device (v3d_axpy)
                    g = z
                else:
                    g = g * scale                           # the few f32 leaves (image_newline: one row)
            out.append(g)
        ctx.grads = None
        return (None, None, *out)


class LlavaQwenTrainable(nn.Module):
    def __init__(self, tree, n_q, n_kv, head_dim=128, vit_heads=16, vit_inter=4304, rope_theta=1e6, max_pos=8192, recompute=False,
                 min_xyz=(-15, -15, -5), max_xyz=(15, 15, 5), voxel_size=0.1):
        super().__init__()
        flat = _flatten(tree)
        self.names = [n for n, _ in flat]
        self._params = nn.ParameterList([nn.Parameter(t, requires_grad=True) for _, t in flat])
        self._shape = tree
        self.n_q, self.n_kv, self.hd, self.vit_heads, self.vit_inter = n_q, n_kv, head_dim, vit_heads, vit_inter
        self.recompute = recompute
        self.min_xyz, self.max_xyz, self.voxel_size = tuple(min_xyz), tuple(max_xyz), voxel_size
        first = flat[0][1]
        self.rope = train.RopeTables(head_dim, max_pos, rope_theta, first.dtype, first.device)
        H = tree["newline"].shape[0]
        n_ids = int(round(max(b - a for a, b in zip(min_xyz, max_xyz)) / voxel_size)) + 1
        self.pe_table = ops.Sin3DTable(H, n_ids, first.dtype, first.device)

    # ------------------------------------------------------------------ parameter tree <-> nn.Parameters
    def param_tree(self):
        """The parameters as the dict / list tree `v3d.train` takes (the tensors ARE the nn.Parameters' storage)."""
        it = iter(self._params)

        def build(t):
            if isinstance(t, dict):
                return {k: build(v) for k, v in t.items()}
            if isinstance(t, list):
                return [build(v) for v in t]
            return next(it).data
        return build(self._shape)

    # ------------------------------------------------------------------ checkpoints in the reference's keys
    @classmethod
    def from_reference_state_dict(cls, sd, n_q, n_kv, vit_heads=16, dtype=torch.bfloat16, device="cuda", kpad=640, **kw):
        """sd: the reference's state dict (llava_qwen.py / siglip_encoder.py / builder.py keys) -> the module."""
        t = lambda k: sd[k].to(device=device, dtype=dtype)       # noqa: E731
        n_vit = 1 + max(int(k[len(VIT + "encoder.layers."):].split(".")[0]) for k in sd if k.startswith(VIT + "encoder.layers."))
        n_llm = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("model.layers."))
        pw = t(VIT + "embeddings.patch_embedding.weight")
        Hv = pw.shape[0]
        patch_w = torch.zeros((Hv, kpad), dtype=dtype, device=device)
        patch_w[:, : pw[0].numel()] = pw.reshape(Hv, -1)
        vit_layers = []
        for i in range(n_vit):
            p = VIT + f"encoder.layers.{i}."
            raw = {"ln1_w": t(p + "layer_norm1.weight"), "ln1_b": t(p + "layer_norm1.bias"), "ln2_w": t(p + "layer_norm2.weight"),
                   "ln2_b": t(p + "layer_norm2.bias"), "o_w": t(p + "self_attn.out_proj.weight"), "o_b": t(p + "self_attn.out_proj.bias"),
                   "fc1_w": t(p + "mlp.fc1.weight"), "fc1_b": t(p + "mlp.fc1.bias"), "fc2_w": t(p + "mlp.fc2.weight"), "fc2_b": t(p + "mlp.fc2.bias")}
            for n in ("q", "k", "v"):
                raw[n + "_w"], raw[n + "_b"] = t(p + f"self_attn.{n}_proj.weight"), t(p + f"self_attn.{n}_proj.bias")
            vit_layers.append(train.siglip_pad_layer(raw, heads=vit_heads, inter_pad=(raw["fc1_w"].shape[0] + 127) // 128 * 128))
        llm_layers = []
        for i in range(n_llm):
            p = f"model.layers.{i}."
            llm_layers.append({
                "ln1": t(p + "input_layernorm.weight"),
                "qkv": torch.cat([t(p + f"self_attn.{n}_proj.weight") for n in ("q", "k", "v")], 0).contiguous(),
                "qkv_bias": torch.cat([t(p + f"self_attn.{n}_proj.bias") for n in ("q", "k", "v")], 0).contiguous(),
                "o": t(p + "self_attn.o_proj.weight").contiguous(), "ln2": t(p + "post_attention_layernorm.weight"),
                "gate_up": torch.cat([t(p + "mlp.gate_proj.weight"), t(p + "mlp.up_proj.weight")], 0).contiguous(),
                "down": t(p + "mlp.down_proj.weight").contiguous()})
        tree = {"vision": {"patch_w": patch_w, "patch_b": t(VIT + "embeddings.patch_embedding.bias"),
                           "pos": t(VIT + "embeddings.position_embedding.weight").contiguous(), "layers": vit_layers},
                "projector": {"w1": t("model.mm_projector.0.weight").contiguous(), "b1": t("model.mm_projector.0.bias"),
                              "w2": t("model.mm_projector.2.weight").contiguous(), "b2": t("model.mm_projector.2.bias")},
                "newline": t("model.image_newline"), "embed": t("model.embed_tokens.weight").contiguous(),
                "llm": {"layers": llm_layers, "norm": t("model.norm.weight"), "lm_head": t("lm_head.weight").contiguous()}}
        hd = tree["llm"]["layers"][0]["qkv"].shape[0] // (n_q + 2 * n_kv)
        m = cls(tree, n_q, n_kv, head_dim=hd, vit_heads=vit_heads, vit_inter=sd[VIT + "encoder.layers.0.mlp.fc1.weight"].shape[0], **kw)
        m._patch_shape = tuple(pw.shape)
        # tensors a real LlavaQwen checkpoint holds and this module does not train - vision_model.post_layernorm (stays in the tower after
        # siglip_encoder.py:570-571), the vision_model.head.* pooling head, the ground_head* tensors, rotary inv_freq buffers, ...: kept
        # as they came (frozen, on the host) and emitted unchanged by reference_state_dict(), so that a checkpoint saved through this
        # surface loads strictly on the reference's side
        m._passthrough = {k: v.detach().clone().cpu() for k, v in sd.items() if k not in m._modelled_keys(n_vit, n_llm)}
        return m

    @staticmethod
    def _modelled_keys(n_vit, n_llm):
        keys = {VIT + "embeddings.patch_embedding.weight", VIT + "embeddings.patch_embedding.bias", VIT + "embeddings.position_embedding.weight",
                "model.mm_projector.0.weight", "model.mm_projector.0.bias", "model.mm_projector.2.weight", "model.mm_projector.2.bias",
                "model.image_newline", "model.embed_tokens.weight", "model.norm.weight", "lm_head.weight"}
        for i in range(n_vit):
            p = VIT + f"encoder.layers.{i}."
            for n in ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.out_proj", "mlp.fc1", "mlp.fc2", "layer_norm1", "layer_norm2"):
                keys |= {p + n + ".weight", p + n + ".bias"}
        for i in range(n_llm):
            p = f"model.layers.{i}."
            keys |= {p + f"self_attn.{n}_proj.{w}" for n in ("q", "k", "v") for w in ("weight", "bias")}
            keys |= {p + "self_attn.o_proj.weight", p + "mlp.gate_proj.weight", p + "mlp.up_proj.weight", p + "mlp.down_proj.weight",
                     p + "input_layernorm.weight", p + "post_attention_layernorm.weight"}
        return keys

    def reference_state_dict(self):
        """Back to the reference's keys and shapes (fused / stacked / padded layouts undone; the padding holds zeros)."""
        tr = self.param_tree()
        out = {}
        v = tr["vision"]
        shape = getattr(self, "_patch_shape", None) or (v["patch_w"].shape[0], 3, 14, 14)
        out[VIT + "embeddings.patch_embedding.weight"] = v["patch_w"][:, : shape[1] * shape[2] * shape[3]].reshape(shape).clone()
        out[VIT + "embeddings.patch_embedding.bias"] = v["patch_b"].clone()
        out[VIT + "embeddings.position_embedding.weight"] = v["pos"].clone()
        for i, L in enumerate(v["layers"]):
            raw = train.siglip_unpad_grads(L, heads=self.vit_heads, inter=self.vit_inter)       # (the same un-padding serves weights)
            p = VIT + f"encoder.layers.{i}."
            for n in ("q", "k", "v"):
                out[p + f"self_attn.{n}_proj.weight"], out[p + f"self_attn.{n}_proj.bias"] = raw[n + "_w"].clone(), raw[n + "_b"].clone()
            out[p + "self_attn.out_proj.weight"], out[p + "self_attn.out_proj.bias"] = raw["o_w"].clone(), raw["o_b"].clone()
            out[p + "mlp.fc1.weight"], out[p + "mlp.fc1.bias"] = raw["fc1_w"].clone(), raw["fc1_b"].clone()
            out[p + "mlp.fc2.weight"], out[p + "mlp.fc2.bias"] = raw["fc2_w"].clone(), raw["fc2_b"].clone()
            out[p + "layer_norm1.weight"], out[p + "layer_norm1.bias"] = raw["ln1_w"].clone(), raw["ln1_b"].clone()
            out[p + "layer_norm2.weight"], out[p + "layer_norm2.bias"] = raw["ln2_w"].clone(), raw["ln2_b"].clone()
        pj = tr["projector"]
        out["model.mm_projector.0.weight"], out["model.mm_projector.0.bias"] = pj["w1"].clone(), pj["b1"].clone()
        out["model.mm_projector.2.weight"], out["model.mm_projector.2.bias"] = pj["w2"].clone(), pj["b2"].clone()
        out["model.image_newline"], out["model.embed_tokens.weight"] = tr["newline"].clone(), tr["embed"].clone()
        nq, nkv, hd = self.n_q * self.hd, self.n_kv * self.hd, self.hd
        for i, L in enumerate(tr["llm"]["layers"]):
            p = f"model.layers.{i}."
            for n, a, b in (("q", 0, nq), ("k", nq, nq + nkv), ("v", nq + nkv, nq + 2 * nkv)):
                out[p + f"self_attn.{n}_proj.weight"], out[p + f"self_attn.{n}_proj.bias"] = L["qkv"][a:b].clone(), L["qkv_bias"][a:b].clone()
            out[p + "self_attn.o_proj.weight"] = L["o"].clone()
            inter = L["gate_up"].shape[0] // 2
            out[p + "mlp.gate_proj.weight"], out[p + "mlp.up_proj.weight"] = L["gate_up"][:inter].clone(), L["gate_up"][inter:].clone()
            out[p + "mlp.down_proj.weight"] = L["down"].clone()
            out[p + "input_layernorm.weight"], out[p + "post_attention_layernorm.weight"] = L["ln1"].clone(), L["ln2"].clone()
        out["model.norm.weight"], out["lm_head.weight"] = tr["llm"]["norm"].clone(), tr["llm"]["lm_head"].clone()
        for k, v in getattr(self, "_passthrough", {}).items():      # unmodelled tensors of the source checkpoint, unchanged
            out[k] = v.clone()
        return out

    # ------------------------------------------------------------------ forward (llava_qwen.py:121-205 for one video sample with labels)
    def forward(self, input_ids, labels, images, world_coords, coord_token_id=None, box_input=None):
        """input_ids / labels: 1-D, the collator's (one IMAGE_TOKEN_INDEX; labels aligned to input_ids, IGNORE_INDEX where masked -
        train_3d.py:1329-1366); images [F,3,S,S] pixel values; world_coords [F,S,S,3].  The visual rows take IGNORE_INDEX labels
        (llava_arch.py:736-741).  box_input [3] + coord_token_id: Scan2Cap's box-centre PE on the <coord> rows (:416-417, 697-700).
        Returns the loss (f32 scalar with a grad_fn)."""
        tr = self.param_tree()
        dev, dt = tr["newline"].device, tr["newline"].dtype
        ids = input_ids.reshape(-1).cpu()
        lab = labels.reshape(-1).cpu()
        at = (ids == IMAGE_TOKEN_INDEX).nonzero().flatten()
        if at.numel() != 1:
            raise V3DError("exactly one <image> token per sample")
        at = int(at[0])
        frames = images.shape[0]
        n = 14
        n_vis = frames * n * (n + 1)
        pre_ids, post_ids = ids[:at].to(dev), ids[at + 1:].to(dev)
        full_labels = torch.cat([lab[:at], torch.full((n_vis,), IGNORE_INDEX, dtype=torch.int64), lab[at + 1:]]).to(dev)
        patches = ops.patchify(images.to(device=dev, dtype=dt), 14, tr["vision"]["patch_w"].shape[1])
        _, _, vox = ops.coord_pool_voxel(world_coords.to(device=dev, dtype=dt), 27, self.min_xyz, self.max_xyz, self.voxel_size, want_avg=False, want_vox=False)
        coord_rows = coord_pe = None
        if coord_token_id is not None and box_input is not None:
            rows = [r if r < at else r + n_vis - 1 for r, t in enumerate(ids.tolist()) if t == coord_token_id]
            if rows:
                centre = ops.discrete_coords(torch.as_tensor(box_input, dtype=torch.float32).reshape(1, 3).to(device=dev, dtype=dt),
                                             self.min_xyz, self.max_xyz, self.voxel_size)
                coord_pe = ops.sin3d_pe(centre[None], tr["newline"].shape[0], dim_t=self.pe_table.dim_t)[0, 0]
                coord_rows = torch.tensor(rows, dtype=torch.int64, device=dev)
        args = (patches, vox, self.pe_table, pre_ids, post_ids, full_labels, self.rope, frames, self.n_q, self.n_kv, self.hd)
        kw_args = args + (729, 27, 14, coord_rows, coord_pe)
        return _SampleLoss.apply(self, kw_args, *self._params)
