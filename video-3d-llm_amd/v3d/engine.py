"""Device-resident engine for the position-aware video -> LLM forward path.

Holds the SigLIP tower, the mlp2x_gelu projector and the Qwen2 decoder as HIP-ready weight
layouts in HBM (fused QKV, tile-interleaved gate/up, zero-padded SigLIP heads and MLP width) plus
pre-allocated workspaces, and runs the forward by launching kernels through the C ABI.
PyTorch supplies memory and the stream only; there is no torch compute on the path.

State-dict keys are the reference's (llava_qwen.py / siglip_encoder.py / builder.py):
  model.vision_tower.vision_tower.vision_model.{embeddings,encoder.layers.N}.*
  model.mm_projector.{0,2}.{weight,bias}, model.image_newline, model.embed_tokens.weight,
  model.layers.N.{self_attn.{q,k,v,o}_proj,mlp.{gate,up,down}_proj,input_layernorm,post_attention_layernorm}.*
  model.norm.weight, lm_head.weight
"""
import math
import os
from dataclasses import dataclass, field

import torch

from . import ops
from ._native import V3DError


def _up(x, m):
    return (x + m - 1) // m * m


@dataclass
class VitConfig:
    hidden: int = 1152
    inter: int = 4304
    layers: int = 26          # 27 in the checkpoint, last one deleted (siglip_encoder.py:570)
    heads: int = 16
    image: int = 384
    patch: int = 14
    eps: float = 1e-6


@dataclass
class LlmConfig:
    hidden: int = 3584
    inter: int = 18944
    layers: int = 28
    heads: int = 28
    kv_heads: int = 4
    vocab: int = 152064
    eps: float = 1e-6
    rope_theta: float = 1000000.0
    max_pos: int = 8192       # rotary table / KV-cache capacity


@dataclass
class EngineConfig:
    vit: VitConfig = field(default_factory=VitConfig)
    llm: LlmConfig = field(default_factory=LlmConfig)
    pool_out: int = 14                     # 27 -> 14 bilinear pool (mm_spatial_pool_stride 2)
    min_xyz: tuple = (-15, -15, -5)
    max_xyz: tuple = (15, 15, 5)
    voxel_size: float = 0.1
    ground_head_type: str = "infonce"      # llava_qwen.py:57-104: 'infonce' (the shipped checkpoints), 'mlp', 'score'
    object_feature_type: str = "patch14-pe"  # llava_arch.py:367-376: 'patch14-*' (ViT patches) or 'patch27-*' (pooled tokens)


def random_state_dict(cfg: EngineConfig, dtype, device, seed=0, std=0.02, ground_head=False):
    """Random-init weights at the configured widths under the reference's state-dict keys."""
    g = torch.Generator(device=device).manual_seed(seed)

    def rn(*shape, s=std):
        return (torch.randn(*shape, generator=g, device=device, dtype=torch.float32) * s).to(dtype)

    sd = {}
    v, l = cfg.vit, cfg.llm
    vp = "model.vision_tower.vision_tower.vision_model."
    n_patches = (v.image // v.patch) ** 2
    sd[vp + "embeddings.patch_embedding.weight"] = rn(v.hidden, 3, v.patch, v.patch)
    sd[vp + "embeddings.patch_embedding.bias"] = rn(v.hidden)
    sd[vp + "embeddings.position_embedding.weight"] = rn(n_patches, v.hidden)
    for i in range(v.layers):
        p = vp + f"encoder.layers.{i}."
        for nme in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[p + f"self_attn.{nme}.weight"] = rn(v.hidden, v.hidden)
            sd[p + f"self_attn.{nme}.bias"] = rn(v.hidden)
        sd[p + "layer_norm1.weight"] = 1 + rn(v.hidden)
        sd[p + "layer_norm1.bias"] = rn(v.hidden)
        sd[p + "layer_norm2.weight"] = 1 + rn(v.hidden)
        sd[p + "layer_norm2.bias"] = rn(v.hidden)
        sd[p + "mlp.fc1.weight"] = rn(v.inter, v.hidden)
        sd[p + "mlp.fc1.bias"] = rn(v.inter)
        sd[p + "mlp.fc2.weight"] = rn(v.hidden, v.inter)
        sd[p + "mlp.fc2.bias"] = rn(v.hidden)
    sd["model.mm_projector.0.weight"] = rn(l.hidden, v.hidden)
    sd["model.mm_projector.0.bias"] = rn(l.hidden)
    sd["model.mm_projector.2.weight"] = rn(l.hidden, l.hidden)
    sd["model.mm_projector.2.bias"] = rn(l.hidden)
    sd["model.image_newline"] = rn(l.hidden)
    sd["model.embed_tokens.weight"] = rn(l.vocab, l.hidden)
    hd = l.hidden // l.heads
    for i in range(l.layers):
        p = f"model.layers.{i}."
        sd[p + "self_attn.q_proj.weight"] = rn(l.hidden, l.hidden)
        sd[p + "self_attn.q_proj.bias"] = rn(l.hidden)
        sd[p + "self_attn.k_proj.weight"] = rn(l.kv_heads * hd, l.hidden)
        sd[p + "self_attn.k_proj.bias"] = rn(l.kv_heads * hd)
        sd[p + "self_attn.v_proj.weight"] = rn(l.kv_heads * hd, l.hidden)
        sd[p + "self_attn.v_proj.bias"] = rn(l.kv_heads * hd)
        sd[p + "self_attn.o_proj.weight"] = rn(l.hidden, l.hidden)
        sd[p + "mlp.gate_proj.weight"] = rn(l.inter, l.hidden)
        sd[p + "mlp.up_proj.weight"] = rn(l.inter, l.hidden)
        sd[p + "mlp.down_proj.weight"] = rn(l.hidden, l.inter)
        sd[p + "input_layernorm.weight"] = 1 + rn(l.hidden)
        sd[p + "post_attention_layernorm.weight"] = 1 + rn(l.hidden)
    sd["model.norm.weight"] = 1 + rn(l.hidden)
    sd["lm_head.weight"] = rn(l.vocab, l.hidden)
    if ground_head:
        for pfx in ("ground_head_obj.", "ground_head_query."):
            sd[pfx + "0.weight"], sd[pfx + "0.bias"] = rn(l.hidden, l.hidden), rn(l.hidden)
            sd[pfx + "2.weight"], sd[pfx + "2.bias"] = 1 + rn(l.hidden), rn(l.hidden)
            sd[pfx + "3.weight"], sd[pfx + "3.bias"] = rn(l.hidden, l.hidden), rn(l.hidden)
        sd["ground_head_zero_target"] = rn(l.hidden, s=1.0)
    return sd


def _pad2(w, rows, cols):
    out = torch.zeros((rows, cols), dtype=w.dtype, device=w.device)
    out[: w.shape[0], : w.shape[1]] = w
    return out


def _pad1(b, n):
    out = torch.zeros(n, dtype=b.dtype, device=b.device)
    out[: b.shape[0]] = b
    return out


class SceneContext:
    """Buffers that belong to one (scene, question) in flight: inputs_embeds / residual stream, KV cache,
    last-row hidden + logits, the decode step's row buffers and split-KV workspace."""


class Engine:
    VIT_DP = 96      # the attention kernel's tile width for SigLIP's head dim 72

    def __init__(self, cfg: EngineConfig, state_dict, dtype=torch.bfloat16, device="cuda", max_frames=32, llm_fp8=False):
        """llm_fp8 (BASELINE configs[3]): the decoder linears and the LM head use OCP e4m3 weights, quantised once per
        output row; the prefill also quantises the activations per token row (W8A8, MFMA), the decode step keeps them
        16-bit (W8A16, weight streaming).  The ViT, attention, norms and the residual stream stay in `dtype`."""
        self.cfg, self.dtype, self.device, self.llm_fp8 = cfg, dtype, device, bool(llm_fp8)
        v, l = cfg.vit, cfg.llm
        self.hd = l.hidden // l.heads
        if self.hd != 128:
            raise V3DError("the Qwen2 path needs head_dim 128 (mrope_section [32,16,16]*2, modeling_qwen2.py:162)")
        self.vhd = v.hidden // v.heads
        if self.vhd not in (72,):
            raise V3DError("the SigLIP path needs head_dim 72")
        if l.hidden % 128 or l.inter % 64:
            raise V3DError("LLM hidden must be a multiple of 128 and intermediate of 64")
        if self.llm_fp8 and (l.hidden % 256 or l.inter % 128):
            raise V3DError("llm_fp8 needs LLM hidden to be a multiple of 256 and intermediate of 128")
        sd = {k: t.to(device=device, dtype=dtype) for k, t in state_dict.items()}
        self._prep_vit(sd)
        self._prep_llm(sd)
        self.ground = None
        if any(k.startswith("ground_head") for k in sd):         # grounding head (llava_qwen.py:57-104), optional
            if cfg.ground_head_type not in ("infonce", "mlp", "score"):
                raise V3DError(f"ground_head_type {cfg.ground_head_type!r}: the reference has 'infonce', 'mlp' and 'score'")
            need = {"infonce": "ground_head_obj.0.weight", "mlp": "ground_head.0.weight", "score": "ground_head_score.0.weight"}[cfg.ground_head_type]
            if need not in sd:
                raise V3DError(f"ground_head_type {cfg.ground_head_type!r} but the state dict has no {need}")
            self.ground = {k: sd[k].contiguous() for k in sd if k.startswith("ground_head")}
        if "patch27" not in cfg.object_feature_type and "patch14" not in cfg.object_feature_type:
            raise V3DError(f"object_feature_type {cfg.object_feature_type!r}: the reference has 'patch14*' and 'patch27*' (llava_arch.py:367-376)")
        self.newline = sd["model.image_newline"].contiguous()
        self.embed = sd["model.embed_tokens.weight"].contiguous()
        n_ids = int(round((max(cfg.max_xyz[0] - cfg.min_xyz[0], cfg.max_xyz[1] - cfg.min_xyz[1],
                               cfg.max_xyz[2] - cfg.min_xyz[2])) / cfg.voxel_size)) + 1
        self.pe_table = ops.Sin3DTable(l.hidden, n_ids, dtype, device)
        self.rope = ops.RopeTable(self.hd, l.max_pos, l.rope_theta, dtype, device)
        self._alloc(max_frames)

    # ------------------------------------------------------------------ weight layouts
    def _prep_vit(self, sd):
        v = self.cfg.vit
        H, Hp, Hk = v.hidden, _up(v.hidden, 128), _up(v.hidden, 64)
        I, Ip = v.inter, _up(v.inter, 128)
        DP, nh, hd = self.VIT_DP, v.heads, self.vhd
        self.v_Hp, self.v_Hk, self.v_Ip = Hp, Hk, Ip
        # residual-stream row stride / fc2 output width: the next multiple of 256 when that costs < 15 % (1152 -> 1280), so that
        # fc2 (K = 4352) runs on the 256-wide GEMM tile (234 vs 265 us); the pad columns are zero weights -> stay exactly zero
        Hx = _up(Hp, 256)
        self.v_Hx = Hx if Hx <= 1.15 * Hp else Hp
        # Q | K | V heads packed at their TRUE stride (72): the attention kernel works on 96-wide tiles but treats dims >= 72 as
        # not belonging to the head (v3d_attention's d_out), so the GEMM does not compute 25 % padding columns; rows padded to a
        # multiple of 256 for the 256-wide GEMM tile (3456 -> 3584); + DP - hd readable columns behind the last head
        self.v_nqkv = _up(3 * nh * hd + (DP - hd), 256)
        self.v_kpatch = _up(3 * v.patch * v.patch, 64)
        vp = "model.vision_tower.vision_tower.vision_model."
        self.v_pe_w = _pad2(sd[vp + "embeddings.patch_embedding.weight"].reshape(H, -1), Hp, self.v_kpatch)
        self.v_pe_b = _pad1(sd[vp + "embeddings.patch_embedding.bias"], Hp)
        self.v_pos = _pad2(sd[vp + "embeddings.position_embedding.weight"], (v.image // v.patch) ** 2, Hp)
        self.v_layers = []
        for i in range(v.layers):
            p = vp + f"encoder.layers.{i}."
            wqkv = torch.zeros((self.v_nqkv, Hk), dtype=self.dtype, device=self.device)
            bqkv = torch.zeros(self.v_nqkv, dtype=self.dtype, device=self.device)
            for part, nme in enumerate(("q_proj", "k_proj", "v_proj")):
                wqkv[part * H:(part + 1) * H, :H] = sd[p + f"self_attn.{nme}.weight"]
                bqkv[part * H:(part + 1) * H] = sd[p + f"self_attn.{nme}.bias"]
            self.v_layers.append(dict(
                ln1_w=sd[p + "layer_norm1.weight"].contiguous(), ln1_b=sd[p + "layer_norm1.bias"].contiguous(),
                ln2_w=sd[p + "layer_norm2.weight"].contiguous(), ln2_b=sd[p + "layer_norm2.bias"].contiguous(),
                wqkv=wqkv, bqkv=bqkv,
                # out_proj writes the residual stream at ITS width (zero pad rows -> zero pad columns): N = 1280 runs on the 256-wide
                # tile, N = 1152 only on the 128 x 128 one (91 -> ~70 us per layer)
                wo=_pad2(sd[p + "self_attn.out_proj.weight"], self.v_Hx, Hk), bo=_pad1(sd[p + "self_attn.out_proj.bias"], self.v_Hx),
                w1=_pad2(sd[p + "mlp.fc1.weight"], Ip, Hk), b1=_pad1(sd[p + "mlp.fc1.bias"], Ip),
                w2=_pad2(sd[p + "mlp.fc2.weight"], self.v_Hx, Ip), b2=_pad1(sd[p + "mlp.fc2.bias"], self.v_Hx)))
        self.p_w0 = _pad2(sd["model.mm_projector.0.weight"], self.cfg.llm.hidden, Hk)
        self.p_b0 = sd["model.mm_projector.0.bias"].contiguous()
        self.p_w2 = sd["model.mm_projector.2.weight"].contiguous()
        self.p_b2 = sd["model.mm_projector.2.bias"].contiguous()

    def _prep_llm(self, sd):
        l = self.cfg.llm
        self.l_layers = []
        for i in range(l.layers):
            p = f"model.layers.{i}."
            wqkv = torch.cat([sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.k_proj.weight"],
                              sd[p + "self_attn.v_proj.weight"]], 0)
            bqkv = torch.cat([sd[p + "self_attn.q_proj.bias"], sd[p + "self_attn.k_proj.bias"],
                              sd[p + "self_attn.v_proj.bias"]], 0)
            nq = _up(wqkv.shape[0], 256 if self.llm_fp8 else 128)
            self.l_layers.append(dict(
                ln1=sd[p + "input_layernorm.weight"].contiguous(), ln2=sd[p + "post_attention_layernorm.weight"].contiguous(),
                wqkv=_pad2(wqkv, nq, l.hidden), bqkv=_pad1(bqkv, nq),
                wo=sd[p + "self_attn.o_proj.weight"].contiguous(),
                wgu=ops.interleave_gate_up(sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"]),
                wd=sd[p + "mlp.down_proj.weight"].contiguous()))
            if self.llm_fp8:
                L = self.l_layers[-1]
                for k in ("wqkv", "wo", "wgu", "wd"):
                    L[k + "8"] = ops.quantize_fp8_rows(L[k])
        self.l_nqkv = self.l_layers[0]["wqkv"].shape[0] if self.l_layers else 0
        self.l_norm = sd["model.norm.weight"].contiguous()
        self.l_head = _pad2(sd["lm_head.weight"], _up(l.vocab, 128), l.hidden)
        if self.llm_fp8:
            self.l_head8 = ops.quantize_fp8_rows(self.l_head)

    def grow_vocab(self, n):
        """resize_token_embeddings(n) with n > vocab (llava_qwen mirror): n - vocab rows equal to the mean of the existing rows are
        appended to the embedding and LM-head tables; per-context logits buffers are re-made."""
        l = self.cfg.llm
        if n <= l.vocab:
            return
        add = n - l.vocab
        emb = self.embed[: l.vocab]
        self.embed = torch.cat([emb, emb.float().mean(0, keepdim=True).to(emb.dtype).expand(add, -1)], 0).contiguous()
        head = self.l_head[: l.vocab]
        self.l_head = _pad2(torch.cat([head, head.float().mean(0, keepdim=True).to(head.dtype).expand(add, -1)], 0), _up(n, 128), l.hidden)
        if self.llm_fp8:
            self.l_head8 = ops.quantize_fp8_rows(self.l_head)
        l.vocab = n
        for k in ("_group_ctxs", "_group_rows", "_answer_st"):
            self.__dict__.pop(k, None)
        self.ctx = self.new_context()

    # ------------------------------------------------------------------ workspaces
    _WS_NAMES = ("v_x", "v_h", "v_qkv", "v_att", "v_mlp", "p_h", "feat", "l_h", "l_qkv", "l_att", "l_act", "l_q8", "l_s8")

    def new_prefill_workspace(self):
        """The scratch of ONE prefill in flight: the ViT's residual / LN / QKV / attention / MLP rows, the projector's rows, the
        decoder's norm / QKV / attention / SwiGLU rows (about 1.6 GB at the 7B shapes).  A second set lets a second stream prefill
        another scene at the same time (use_workspace), so that one scene's kernels fill the idle CUs of the other's last GEMM rounds."""
        v, l = self.cfg.vit, self.cfg.llm
        dt, dev = self.dtype, self.device
        T = self.max_frames * (v.image // v.patch) ** 2
        S = l.max_pos
        z = lambda *s: torch.zeros(s, dtype=dt, device=dev)     # noqa: E731
        w = SceneContext()
        w.v_x, w.v_h = z(T, self.v_Hx), z(T, self.v_Hp)
        w.v_qkv, w.v_att, w.v_mlp = z(T, self.v_nqkv), z(T, self.v_Hp), z(T, self.v_Ip)
        w.p_h, w.feat = z(T, l.hidden), z(T, l.hidden)
        w.l_h = z(S, l.hidden)
        w.l_qkv, w.l_att, w.l_act = z(S, self.l_nqkv), z(S, l.hidden), z(S, l.inter)
        w.l_q8 = w.l_s8 = None
        if self.llm_fp8:                               # e4m3 image + row scales of the current GEMM's activations
            w.l_q8 = torch.zeros((S, max(l.hidden, l.inter)), dtype=torch.uint8, device=dev)
            w.l_s8 = torch.zeros(S, dtype=torch.float32, device=dev)
        return w

    def use_workspace(self, w):
        """Select the prefill scratch subsequent launches use (host-side pointer switch, like use(ctx))."""
        self.ws = w
        return w

    def _alloc(self, max_frames):
        self.max_frames = max_frames
        self.ws = self.new_prefill_workspace()
        self.ctx = self.new_context()                  # per-scene state (sequence buffer, KV cache, decode rows)

    def new_context(self):
        """Per-scene state.  A second context lets the decode of scene i (HBM-bound weight streaming) run on
        another stream while scene i+1's ViT / prefill (MFMA-bound) occupies the matrix cores."""
        l = self.cfg.llm
        dt, dev = self.dtype, self.device
        z = lambda *s: torch.zeros(s, dtype=dt, device=dev)
        c = SceneContext()
        c.l_x = z(l.max_pos, l.hidden)
        c.kv = [z(l.max_pos, 2 * l.kv_heads * self.hd) for _ in range(l.layers)]
        c.l_last, c.logits = z(8, l.hidden), z(8, self.l_head.shape[0])
        c.d_qkv, c.d_att, c.d_act = z(self.l_nqkv), z(l.hidden), z(l.inter)
        c.dec_ws = ops.decode_workspace(l.heads, l.kv_heads, dev)
        c.kv_len = 0
        return c

    def use(self, ctx):
        """Select the context subsequent launches read/write (host-side pointer switch only)."""
        self.ctx = ctx
        return ctx

    v_x = property(lambda self: self.ws.v_x)
    v_h = property(lambda self: self.ws.v_h)
    v_qkv = property(lambda self: self.ws.v_qkv)
    v_att = property(lambda self: self.ws.v_att)
    v_mlp = property(lambda self: self.ws.v_mlp)
    p_h = property(lambda self: self.ws.p_h)
    feat = property(lambda self: self.ws.feat)
    l_h = property(lambda self: self.ws.l_h)
    l_qkv = property(lambda self: self.ws.l_qkv)
    l_att = property(lambda self: self.ws.l_att)
    l_act = property(lambda self: self.ws.l_act)
    l_q8 = property(lambda self: self.ws.l_q8)
    l_s8 = property(lambda self: self.ws.l_s8)
    l_x = property(lambda self: self.ctx.l_x)
    kv = property(lambda self: self.ctx.kv)
    l_last = property(lambda self: self.ctx.l_last)
    logits = property(lambda self: self.ctx.logits)
    dec_ws = property(lambda self: self.ctx.dec_ws)

    @property
    def kv_len(self):
        return self.ctx.kv_len

    @kv_len.setter
    def kv_len(self, v):
        self.ctx.kv_len = v

    # ------------------------------------------------------------------ ViT + projector (a9, a10)
    def encode_images(self, images):
        """[F,3,S,S] -> [F, (S/patch)^2, llm.hidden]   (encode_images, llava_arch.py:275-280)."""
        v = self.cfg.vit
        F_ = images.shape[0]
        if F_ > self.max_frames:
            raise V3DError(f"{F_} frames > engine capacity {self.max_frames}")
        n = (v.image // v.patch) ** 2
        T = F_ * n
        H, Hk, DP, nh = v.hidden, self.v_Hk, self.VIT_DP, v.heads
        x, h, qkv, att, mlp = self.v_x[:T], self.v_h[:T], self.v_qkv[:T], self.v_att[:T], self.v_mlp[:T]
        patches = ops.patchify(images.to(self.dtype), v.patch, self.v_kpatch)
        ops.gemm(patches, self.v_pe_w, bias=self.v_pe_b, res=self.v_pos, res_mod=n, epilogue=ops.EPI_BIAS_RES, out=x)
        scale = self.vhd ** -0.5
        for L in self.v_layers:
            ops.layernorm(x[:, :H], L["ln1_w"], L["ln1_b"], v.eps, out=h[:, :H])
            ops.gemm(h[:, :Hk], L["wqkv"], bias=L["bqkv"], epilogue=ops.EPI_BIAS, out=qkv)
            ld = qkv.stride(0)
            ops.attention(qkv, qkv[:, H:], qkv[:, 2 * H:], att, F_, n, n, nh, nh, DP, self.vhd,
                          ld, ld, ld, att.stride(0), n * ld, n * ld, n * att.stride(0), self.vhd, self.vhd, self.vhd, False, 0, scale)
            ops.gemm(att[:, :Hk], L["wo"], bias=L["bo"], res=x, epilogue=ops.EPI_BIAS_RES, out=x)
            ops.layernorm(x[:, :H], L["ln2_w"], L["ln2_b"], v.eps, out=h[:, :H])
            ops.gemm(h[:, :Hk], L["w1"], bias=L["b1"], epilogue=ops.EPI_BIAS_GELU_TANH, out=mlp)
            ops.gemm(mlp, L["w2"], bias=L["b2"], res=x, epilogue=ops.EPI_BIAS_RES, out=x)
        ph, feat = self.p_h[:T], self.feat[:T]
        ops.gemm(x[:, :Hk], self.p_w0, bias=self.p_b0, epilogue=ops.EPI_BIAS_GELU_ERF, out=ph)
        ops.gemm(ph, self.p_w2, bias=self.p_b2, epilogue=ops.EPI_BIAS, out=feat)
        return feat.view(F_, n, self.cfg.llm.hidden)

    def vit_hidden(self, n_frames):
        """SigLipVisionTower output of the last encode_images call (hidden_states[-1]), [F, n, hidden]."""
        n = (self.cfg.vit.image // self.cfg.vit.patch) ** 2
        return self.v_x[: n_frames * n, : self.cfg.vit.hidden].reshape(n_frames, n, -1)

    # ------------------------------------------------------------------ fusion (a11-a17)
    def voxel_ids(self, world_coords):
        """[F,384,384,3] (model dtype) -> int32 [F,14,14,3]   (llava_arch.py:403, 416)."""
        c = self.cfg
        _, _, ids = ops.coord_pool_voxel(world_coords, 27, c.min_xyz, c.max_xyz, c.voxel_size, want_avg=False, want_vox=False)
        return ids

    def build_inputs_embeds(self, input_ids, feats, ids, image_token=-200, box_input=None, coord_token_id=None, stamp=None):
        """prepare_inputs_labels_for_multimodal for one video sample (llava_arch.py:336-836, eval branch):
        text embeddings around the <image> slot, visual tokens written in place.  Returns [S, hidden] view.
        box_input [1,3] + coord_token_id: the PE of the discretised box centre is added to the rows of every <coord> text token
        (llava_arch.py:416-417, 697-700; the Scan2Cap prompt)."""
        F_ = feats.shape[0]
        n = self.cfg.pool_out
        side = int(math.isqrt(feats.shape[1]))
        n_vis = F_ * n * (n + 1)
        ids_list = input_ids.tolist()
        if ids_list.count(image_token) != 1:
            raise V3DError("exactly one <image> token per prompt (model_scanqa.py:61)")
        at = ids_list.index(image_token)
        pre = input_ids[:at]
        post = input_ids[at + 1:]
        S = len(pre) + n_vis + len(post)
        if S > self.cfg.llm.max_pos:
            raise V3DError(f"sequence {S} exceeds engine capacity {self.cfg.llm.max_pos}")
        x = self.l_x[:S]
        up = lambda t: (t if t.is_cuda else t.pin_memory().to(self.device, non_blocking=True))      # noqa: E731  (no host-blocking pageable copy)
        if len(pre):
            ops.embed_gather(self.embed, up(pre), out=x[: len(pre)])
        vt = lambda: ops.visual_tokens(feats, ids, self.pe_table, self.newline, side=side, n=n, pool=True, out=x[len(pre): len(pre) + n_vis])      # noqa: E731
        if stamp is not None:
            stamp(vt)             # bench: HIP events around the north-star kernel
        else:
            vt()
        if len(post):
            ops.embed_gather(self.embed, up(post), out=x[len(pre) + n_vis:])
        if coord_token_id is not None and box_input is not None and len(box_input):
            rows = [r if r < at else r + n_vis - 1 for r, t in enumerate(ids_list) if t == coord_token_id]
            if rows:
                c = self.cfg
                centre = ops.discrete_coords(box_input.to(device=self.device, dtype=self.dtype).reshape(-1, 3)[:1].contiguous(),
                                             c.min_xyz, c.max_xyz, c.voxel_size)
                pe = ops.sin3d_pe(centre[None], c.llm.hidden, dim_t=self.pe_table.dim_t)[0, 0]
                ops.add_row(x, torch.tensor(rows, dtype=torch.int64, device=self.device), pe)
        return x

    # ------------------------------------------------------------------ Qwen2 decoder (a18-a22)
    def llm_forward(self, x, pos0, stamps=None, head=True, batch=None, last_rows=None):
        """Runs the decoder over rows x [S, hidden] (in place) at positions pos0.., appends K/V to the
        cache and returns the f32 logits of the LAST row only ([vocab]); the reference materialises all
        S rows of logits (modeling_qwen2.py:1190-1192) but generation reads only the last.
        batch (answer_group): x holds B sequences of Sq rows each, all starting at position pos0 behind the same cached prefix;
        batch.kv[i] is [B, max_pos, 2*kv_width] (sequence b appends to and attends over its own slice), batch.positions /
        batch.dst_rows give every row's position and flat cache row.
        last_rows (r03; None = every row): the rows of x whose LAST-layer output the caller will read - [S - 1] for generation,
        [ground row] for the grounding head, [] for a scene prefill that only leaves K/V behind.  The last layer's keys and values
        are still formed for every row (the decode steps attend to them), but its attention, o_proj and MLP run for the listed rows
        only: no later computation reads the other rows' outputs (the reference computes and drops them,
        modeling_qwen2.py:1048-1070 then :1188 reads hidden_states[:, -1] in effect) - 1 / 28 of the decoder's attention and MLP
        work.  The listed rows go through the one-row decode kernels (f32 sums in another order than the GEMM tiles: rounding-level)."""
        l = self.cfg.llm
        S = x.shape[0]
        hd, nh, nkv = self.hd, l.heads, l.kv_heads
        h, qkv, att, act = self.l_h[:S], self.l_qkv[:S], self.l_att[:S], self.l_act[:S]
        kvw = nkv * hd
        scale = 1.0 / math.sqrt(hd)
        fp8 = self.llm_fp8 and S > 8
        if fp8:
            s8 = self.l_s8[:S]

            def lin(a, L, key, out, **kw):           # quantise the activation rows, then the e4m3 GEMM
                q8 = self.l_q8[:S, : a.shape[1]]
                ops.quantize_fp8_rows(a, q8, s8)
                qw, sw = L[key + "8"]
                return ops.gemm_fp8(q8, s8, qw, sw, self.dtype, out=out, **kw)

            def norm_lin(ln, L, key, out, **kw):     # RMSNorm + quantisation in one pass (the 16-bit rows are never stored)
                q8 = self.l_q8[:S, : l.hidden]
                ops.rmsnorm_quantize_fp8(x, ln, l.eps, q8, s8)
                qw, sw = L[key + "8"]
                return ops.gemm_fp8(q8, s8, qw, sw, self.dtype, out=out, **kw)
        else:
            def lin(a, L, key, out, **kw):
                return ops.gemm(a, L[key], out=out, **kw)

            def norm_lin(ln, L, key, out, **kw):
                ops.rmsnorm(x, ln, l.eps, out=h)
                return ops.gemm(h, L[key], out=out, **kw)
        n_layers = len(self.l_layers)
        for i, L in enumerate(self.l_layers):
            norm_lin(L["ln1"], L, "wqkv", qkv, bias=L["bqkv"], epilogue=ops.EPI_BIAS)
            if last_rows is not None and batch is None and i == n_layers - 1 and S > 1:
                cache = self.kv[i]
                ops.rope_kv_store(qkv, nh, nkv, hd, self.rope, cache, pos0=pos0)
                self._last_layer_rows(L, x, qkv, cache, [int(r) for r in last_rows], pos0)
                continue
            if batch is not None:
                c3 = batch.kv[i]
                cache = c3.view(-1, c3.shape[-1])
                ops.rope_kv_store(qkv, nh, nkv, hd, self.rope, cache, positions=batch.positions, dst_rows=batch.dst_rows)
                B, Sq = batch.B, batch.Sq
                shared = getattr(batch, "shared_kv", None)
                if shared is not None:   # answer_group: key tiles below shared_len from the scene's ONE cache (the questions' caches hold no copy)
                    sc = shared[i]
                    ops.attention_shared_prefix(qkv, cache, cache[:, kvw:], sc, sc[:, kvw:], batch.shared_len, att, B, Sq, pos0 + Sq, nh, nkv,
                                                qkv.stride(0), cache.stride(0), cache.stride(0), att.stride(0), Sq * qkv.stride(0), c3.stride(0),
                                                Sq * att.stride(0), hd, hd, hd, pos0, scale)
                else:
                    ops.attention(qkv, cache, cache[:, kvw:], att, B, Sq, pos0 + Sq, nh, nkv, hd, hd, qkv.stride(0), cache.stride(0),
                                  cache.stride(0), att.stride(0), Sq * qkv.stride(0), c3.stride(0), Sq * att.stride(0), hd, hd, hd, True, pos0, scale)
                lin(att, L, "wo", x, res=x, epilogue=ops.EPI_RES)
                norm_lin(L["ln2"], L, "wgu", act, epilogue=ops.EPI_SWIGLU)
                lin(act, L, "wd", x, res=x, epilogue=ops.EPI_RES)
                continue
            cache = self.kv[i]
            ops.rope_kv_store(qkv, nh, nkv, hd, self.rope, cache, pos0=pos0)       # K14 + cache append (rows pos0 .. pos0+S)
            attn = lambda: ops.attention(qkv, cache, cache[:, kvw:], att, 1, S, pos0 + S, nh, nkv, hd, hd, qkv.stride(0),
                                         cache.stride(0), cache.stride(0), att.stride(0), 0, 0, 0, hd, hd, hd, True, pos0, scale)
            if S == 1:
                ops.attention_decode(qkv, cache, cache[:, kvw:], att, pos0 + 1, nh, nkv, scale, self.dec_ws)
            elif stamps is not None and i == 1:
                stamps["attn"](attn)      # bench: HIP events around one layer's launch
            else:
                attn()
            lin(att, L, "wo", x, res=x, epilogue=ops.EPI_RES)
            if stamps is not None and i == 1 and not fp8:      # bench: HIP events around the gate/up GEMM alone
                ops.rmsnorm(x, L["ln2"], l.eps, out=h)
                stamps["gemm"](lambda: ops.gemm(h, L["wgu"], epilogue=ops.EPI_SWIGLU, out=act))
            elif stamps is not None and i == 1:                # e4m3: norm + quantisation pass, then the stamped GEMM
                ops.rmsnorm_quantize_fp8(x, L["ln2"], l.eps, self.l_q8[:S, : l.hidden], s8)
                stamps["gemm"](lambda: ops.gemm_fp8(self.l_q8[:S, : l.hidden], s8, *L["wgu8"], self.dtype, epilogue=ops.EPI_SWIGLU, out=act))
            else:
                norm_lin(L["ln2"], L, "wgu", act, epilogue=ops.EPI_SWIGLU)
            lin(act, L, "wd", x, res=x, epilogue=ops.EPI_RES)
        if batch is not None:
            return None
        self.kv_len = pos0 + S
        return self._head(x[S - 1:]) if head else None

    def _last_layer_rows(self, L, x, qkv, cache, rows, pos0):
        """The last decoder layer after its K/V rows are in the cache, for single rows of the sequence: attention of row r over keys
        0 .. pos0 + r (the decode kernel), o_proj + residual, RMSNorm + gate/up + SwiGLU, down + residual - decode_forward's per-layer
        launches with the query taken from the prefill's QKV rows."""
        l = self.cfg.llm
        hd, nh, nkv = self.hd, l.heads, l.kv_heads
        kvw = nkv * hd
        scale = 1.0 / math.sqrt(hd)
        att, act = self.ctx.d_att, self.ctx.d_act
        for r in rows:
            xr = x[r]
            ops.attention_decode(qkv[r], cache, cache[:, kvw:], att, pos0 + r + 1, nh, nkv, scale, self.dec_ws)
            if self.llm_fp8:
                row = x[r: r + 1]
                ops.linear_decode_fp8_rows(att[None], *L["wo8"], row, res=row, epilogue=ops.DEC_RES)
                h1 = ops.rmsnorm(row, L["ln2"], l.eps, out=self.l_h[:1])
                ops.linear_decode_fp8_rows(h1, *L["wgu8"], act[None], epilogue=ops.DEC_SWIGLU)
                ops.linear_decode_fp8_rows(act[None], *L["wd8"], row, res=row, epilogue=ops.DEC_RES)
            else:
                ops.linear_decode(att, L["wo"], xr, res=xr, epilogue=ops.DEC_RES)
                ops.linear_decode(xr, L["wgu"], act, norm_weight=L["ln2"], eps=l.eps, epilogue=ops.DEC_SWIGLU)
                ops.linear_decode(act, L["wd"], xr, res=xr, epilogue=ops.DEC_RES)

    def _head(self, x_row):
        """final RMSNorm + LM head on ONE row (K18: only the last position feeds generation)."""
        l = self.cfg.llm
        ops.rmsnorm(x_row, self.l_norm, l.eps, out=self.l_last[:1])
        if self.llm_fp8:
            ops.linear_decode_fp8_rows(self.l_last[:1], *self.l_head8, self.logits[:1])
        else:
            ops.linear_decode(self.l_last[0], self.l_head, self.logits[0])
        return self.logits[0, : l.vocab]

    def decode_forward(self, x_row, pos):
        """One new token (x_row [1, hidden], in place) at position `pos`: 7 launches per layer, no host sync.
        RMSNorms are fused into the QKV / gate-up linears, rotary + cache append into one kernel."""
        l = self.cfg.llm
        if self.llm_fp8:                  # e4m3 weights: the row goes through the (unfused-norm) rows path as a group of one
            g = getattr(self.ctx, "g1", None)
            if g is None:
                g = self.ctx.g1 = self.new_group(1)
            g.x[:1].copy_(x_row)
            logits = self.decode_forward_rows(g, [self.ctx], [pos])[0]
            x_row.copy_(g.x[:1])
            self.l_last[:1].copy_(g.last[:1])
            return logits
        hd, nh, nkv = self.hd, l.heads, l.kv_heads
        kvw = nkv * hd
        scale = 1.0 / math.sqrt(hd)
        xr, qkv, att, act = x_row[0], self.ctx.d_qkv, self.ctx.d_att, self.ctx.d_act
        for i, L in enumerate(self.l_layers):
            cache = self.kv[i]
            ops.linear_decode(xr, L["wqkv"], qkv, norm_weight=L["ln1"], eps=l.eps, bias=L["bqkv"], epilogue=ops.DEC_BIAS)
            ops.rope_kv_append(qkv, nh, nkv, hd, self.rope, pos, cache[pos])
            ops.attention_decode(qkv, cache, cache[:, kvw:], att, pos + 1, nh, nkv, scale, self.dec_ws)
            ops.linear_decode(att, L["wo"], xr, res=xr, epilogue=ops.DEC_RES)
            ops.linear_decode(xr, L["wgu"], act, norm_weight=L["ln2"], eps=l.eps, epilogue=ops.DEC_SWIGLU)
            ops.linear_decode(act, L["wd"], xr, res=xr, epilogue=ops.DEC_RES)
        self.kv_len = pos + 1
        return self._head(x_row)

    def last_hidden(self):
        return self.l_last[0]

    def logits_all_rows(self, x):
        """final norm + LM head over EVERY row of the residual stream x [S, hidden] -> f32 [S, vocab]: what
        Qwen2ForCausalLM.forward returns (modeling_qwen2.py:1188-1192: 16-bit linear output, then .float()).  Generation never
        needs it (only the last row is read); the plain `forward` of the mirror class does."""
        l = self.cfg.llm
        S = x.shape[0]
        h = ops.rmsnorm(x, self.l_norm, l.eps, out=self.l_h[:S])
        out = ops.gemm(h, self.l_head)
        return out[:, : l.vocab].float()

    def _check_room(self, S, max_new_tokens):
        """The KV cache / residual stream hold llm.max_pos rows: a generation that would run past them is refused up front
        (the last generated token is never fed back, so S + max_new_tokens - 1 rows are written)."""
        if max_new_tokens < 1:
            raise V3DError("max_new_tokens must be at least 1")
        if S + max_new_tokens - 1 > self.cfg.llm.max_pos:
            raise V3DError(f"prompt of {S} rows + {max_new_tokens} new tokens exceeds the engine capacity of {self.cfg.llm.max_pos} "
                           "positions (LlmConfig.max_pos): lower max_new_tokens or build the engine with a larger max_pos")

    # ------------------------------------------------------------------ grounding (a24)
    @torch.no_grad()
    def ground_scores(self, input_ids, ground_index, images, world_coords, objects):
        """ScanRefer / Multi3DRefer forward (model_scanrefer.py:165-173 -> llava_qwen.forward(use_object_proposals=True)
        -> predict_box, infonce head): one prefill, scores [n_obj + 1] (last = the zero-target).
        ground_index: index IN input_ids of the <ground> label token whose hidden state is the query."""
        if self.ground is None:
            raise V3DError("this engine was built without ground_head_* weights")
        coords = world_coords.to(self.dtype)
        boxes = objects.to(device=self.device, dtype=self.dtype).contiguous()
        feats = self.encode_images(images)
        ids = self.voxel_ids(coords)
        x = self.build_inputs_embeds(input_ids, feats, ids)
        at = input_ids.tolist().index(-200)
        n_vis = feats.shape[0] * self.cfg.pool_out * (self.cfg.pool_out + 1)
        gpos = ground_index if ground_index < at else ground_index + n_vis - 1
        self.llm_forward(x, 0, head=False, last_rows=[gpos])
        return self.predict_box(x, gpos, self.object_features(feats, coords, boxes))

    def predict_box(self, x, gpos, object_features):
        """predict_box (llava_qwen.py:280-300): x = the residual stream after the decoder (before the final norm), gpos = the row
        of the <ground> label token, object_features [n, H] -> scores ([n + 1] for 'infonce': the last is the zero-target)."""
        if self.ground is None:
            raise V3DError("this engine was built without ground_head_* weights")
        query = ops.rmsnorm(x[gpos: gpos + 1], self.l_norm, self.cfg.llm.eps, out=self.l_last[1:2])        # outputs[0][ground_locations]
        return self.ground_head(query, object_features)

    def ground_head(self, query, object_features):
        """The grounding head on the normed hidden state of the <ground> token (query [1, H]) and the object features [n, H]:
        'infonce' (llava_qwen.py:87-104, 294-300) cosine of two MLPs, zero-target appended; 'mlp' (:59-71, 283-285) one MLP on the
        query, product-and-sum with the raw object features; 'score' (:72-86, 286-292) two MLPs, their product, a scoring MLP."""
        g, kind = self.ground, self.cfg.ground_head_type

        def relu_ln(xin, pfx):                       # Linear, ReLU, LayerNorm, Linear
            h = ops.gemm(xin, g[pfx + "0.weight"], bias=g[pfx + "0.bias"], epilogue=ops.EPI_BIAS_RELU)
            hn = ops.layernorm(h, g[pfx + "2.weight"], g[pfx + "2.bias"], 1e-5)
            return ops.gemm(hn, g[pfx + "3.weight"], bias=g[pfx + "3.bias"], epilogue=ops.EPI_BIAS)

        def ln_relu(xin, pfx, last=True):            # Linear, LayerNorm, ReLU (, Linear)
            h = ops.gemm(xin, g[pfx + "0.weight"], bias=g[pfx + "0.bias"], epilogue=ops.EPI_BIAS)
            hn = ops.relu_mul_rows(ops.layernorm(h, g[pfx + "1.weight"], g[pfx + "1.bias"], 1e-5))
            return ops.gemm(hn, g[pfx + "3.weight"], bias=g[pfx + "3.bias"], epilogue=ops.EPI_BIAS) if last else hn

        if kind == "infonce":
            of = torch.cat([object_features, g["ground_head_zero_target"][None]], 0).contiguous()
            return ops.ground_scores(relu_ln(of, "ground_head_obj."), relu_ln(query, "ground_head_query.")[0])      # K20
        if kind == "mlp":
            return ops.row_dots(object_features.contiguous(), relu_ln(query, "ground_head.")[0].contiguous(), products_rounded=True)
        obj = ln_relu(object_features.contiguous(), "ground_head_obj.")
        qf = ln_relu(query, "ground_head_query.")
        mul = ops.relu_mul_rows(obj, row=qf[0].contiguous(), relu=False)                      # obj_feat * query_feat
        hn = ln_relu(mul, "ground_head_score.", last=False)
        return ops.row_dots(hn, g["ground_head_score.3.weight"][0].contiguous(), bias=g["ground_head_score.3.bias"])

    def object_features(self, feats, coords, boxes):
        """Object-proposal features (llava_arch.py:351-376, 479-501): per box, the mean of the rows whose image cell has enough of
        its pixels inside the box, plus the PE of the discretised box centre.  'patch14': the projector rows of the ViT patches
        (27 x 27 cells of 14 pixels, >= 98 of 196 inside); 'patch27': the POOLED tokens before the PE (14 x 14 cells of 27 pixels,
        >= 182 of 729).  feats [F,729,H] (encode_images), coords [F,384,384,3] and boxes [n,6] in the model dtype -> [n, H]."""
        H = self.cfg.llm.hidden
        centres = ops.discrete_coords(boxes[:, :3].contiguous(), self.cfg.min_xyz, self.cfg.max_xyz, self.cfg.voxel_size)
        pe = ops.sin3d_pe(centres[None], H, dim_t=self.pe_table.dim_t)[0]
        if "patch27" in self.cfg.object_feature_type:
            mask = ops.object_patch_mask(coords, boxes, cell=27, thresh=int(27 * 27 * 0.25))
            rows = ops.visual_tokens(feats, side=27, n=self.cfg.pool_out, pool=True)          # get_2dPool alone: [F, 196, H]
            return ops.masked_mean(rows.reshape(-1, H), mask.view(mask.shape[0], -1), add=pe)
        mask = ops.object_patch_mask(coords, boxes)                                               # K19
        return ops.masked_mean(feats.reshape(-1, H), mask.view(mask.shape[0], -1), add=pe)

    # ------------------------------------------------------------------ generate (a23)
    @torch.no_grad()
    def generate(self, input_ids, images, world_coords, max_new_tokens=16, eos_token_id=None, stopping=None, box_input=None,
                 coord_token_id=None):
        """Greedy decoding of one (scene, question): LlavaQwenForCausalLM.generate (llava_qwen.py:208-236)
        with do_sample=False, num_beams=1 as model_scanqa.py:173-185 calls it.  eos_token_id: an int or a collection of ints;
        stopping: optional host callback(token_ids_so_far: LongTensor [n]) -> bool, checked after every token (HF's
        stopping_criteria; forces one host synchronisation per step)."""
        feats = self.encode_images(images)
        ids = self.voxel_ids(world_coords.to(self.dtype))
        x = self.build_inputs_embeds(input_ids, feats, ids, box_input=box_input, coord_token_id=coord_token_id)
        S = x.shape[0]
        self._check_room(S, max_new_tokens)
        logits = self.llm_forward(x, 0, last_rows=[S - 1])
        return self.decode_loop(logits, S, max_new_tokens, eos_token_id, stopping)

    # ------------------------------------------------------------------ scene-level reuse
    @torch.no_grad()
    def prefill_scene(self, prefix_ids, images, world_coords):
        """Everything of a (scene, question) pass that does not depend on the question, done once per scene: ViT + projector,
        voxel ids, the visual tokens and the decoder over the prompt prefix [system | user | <image>] - the reference's eval loop
        recomputes all of it for every question of a scene (model_scanqa.py:130-185; the prefix is question-independent,
        :46-60).  prefix_ids: the prompt ids up to and INCLUDING the <image> placeholder (-200) (further question-independent
        ids may follow it).  The K/V rows of the prefix stay in the current context's cache; returns the prefix length P.
        Follow with answer(question_ids) any number of times."""
        feats = self.encode_images(images)
        ids = self.voxel_ids(world_coords.to(self.dtype))
        x = self.build_inputs_embeds(prefix_ids, feats, ids)
        self.llm_forward(x, 0, head=False, last_rows=[])          # the prefix only leaves its K/V rows behind
        self.ctx.prefix_len = x.shape[0]
        return self.ctx.prefix_len

    @torch.no_grad()
    def answer(self, question_ids, max_new_tokens=16, eos_token_id=None, stopping=None):
        """Greedy answer to one question about the scene prefilled by prefill_scene: only the question's rows (the ids that follow
        the prefix in the full prompt) run through the decoder, at positions P.., attending to the cached prefix; then the
        usual decode loop.  Causality makes the prefix rows of the uncached pass independent of the question, so the tokens are
        those of generate(prefix_ids + question_ids, ...) (tests/test_gpu_scene_reuse.py).  The prefix rows are left intact."""
        P = getattr(self.ctx, "prefix_len", 0)
        if not P or self.ctx.kv_len < P:
            raise V3DError("answer() needs prefill_scene() on this context first")
        q = question_ids.to(self.device)
        Q = q.numel()
        if Q < 1:
            raise V3DError("answer() needs at least one question token")
        self._check_room(P + Q, max_new_tokens)
        x = ops.embed_gather(self.embed, q, out=self.l_x[P: P + Q])
        logits = self.llm_forward(x, P, last_rows=[Q - 1])
        return self.decode_loop(logits, P + Q, max_new_tokens, eos_token_id, stopping)

    MAX_GROUP = 32          # scenes / questions whose decode steps share a pass over the weights (r04: 32; csrc DEC_MAX_ROWS)

    def _answer_state(self, n):
        """Buffers of answer_group, allocated once: per layer ONE K/V allocation [MAX_GROUP, max_pos, 2*kv_width] whose slices are the
        caches of the questions answered together (so one batched attention launch can address them by a batch stride), the
        contexts that view them, the residual stream of the batched question rows and the decode group's row buffers."""
        st = self.__dict__.get("_answer_st")
        if st is None:
            l = self.cfg.llm
            st = self._answer_st = SceneContext()
            st.n = self.MAX_GROUP
            st.kv = [torch.zeros((st.n, l.max_pos, 2 * l.kv_heads * self.hd), dtype=self.dtype, device=self.device) for _ in range(l.layers)]
            st.x = torch.zeros((l.max_pos, l.hidden), dtype=self.dtype, device=self.device)
            st.ctxs = []
            for g in range(st.n):
                c = SceneContext()
                c.kv = [k[g] for k in st.kv]
                c.kv_len = 0
                st.ctxs.append(c)
            st.rows = self.new_group(st.n)
        if n > st.n:
            raise V3DError(f"answer_group takes 1 to {st.n} questions")
        return st

    @torch.no_grad()
    def answer_group(self, questions, max_new_tokens=16, eos_token_id=None):
        """Up to 32 questions about the scene prefilled by prefill_scene, answered together: the cached prefix K/V is handed to
        each question's own cache (one broadcast copy per layer), the questions' rows run through the decoder as ONE batch
        (one pass over the weights; rows padded to the longest question, which causality keeps from affecting the real rows),
        then all answers decode as one group (decode_group).  Returns a list of token-id tensors, each cut after its first EOS.
        A question's rows do not depend on the others (tests/test_gpu_scene_reuse.py: equal to answer() on it alone)."""
        scene = self.ctx
        P = getattr(scene, "prefix_len", 0)
        if not P or scene.kv_len < P:
            raise V3DError("answer_group() needs prefill_scene() on this context first")
        G = len(questions)
        st = self._answer_state(G)
        lens = [int(q.numel()) for q in questions]
        if G < 1 or min(lens) < 1:
            raise V3DError(f"answer_group() needs 1 to {self.MAX_GROUP} non-empty questions")
        Sq = max(lens)
        l = self.cfg.llm
        if G * Sq > l.max_pos:
            raise V3DError(f"{G} questions x {Sq} rows exceed the engine's {l.max_pos}-row workspaces")
        self._check_room(P + Sq, max_new_tokens)
        # r04: the questions' caches no longer receive a copy of the prefix.  Their rows run over the scene's ONE cache for every key tile
        # below P0 = 64 floor(P / 64) (v3d_attention_shared_prefix: same tiles, same order - the bits of a full copy), and the decode steps
        # read the prefix from it as well; only the rows of the tile that straddles P are copied (at most 63).  V3D_SHARED_PREFIX=0: the
        # r03 form, every question with its own full copy (A/B and tests).
        share = os.environ.get("V3D_SHARED_PREFIX", "1") != "0" and scene.kv[0].stride(0) == st.kv[0].stride(1)
        P0 = (P // 64) * 64 if share else 0
        for i in range(l.layers):
            if not share:
                ops.copy_rows_bcast(scene.kv[i][:P], st.kv[i][:G])
            elif P > P0:
                ops.copy_rows_bcast(scene.kv[i][P0:P], st.kv[i][:G, P0:])
        ids = torch.empty((G, Sq), dtype=torch.int64)
        for g, q in enumerate(questions):
            ids[g, : lens[g]] = q.cpu()
            ids[g, lens[g]:] = q[-1]                        # pad rows: any valid id; they sit AFTER the real rows (causal)
        x = ops.embed_gather(self.embed, ids.reshape(-1).to(self.device), out=st.x[: G * Sq])
        b = SceneContext()
        b.B, b.Sq, b.kv = G, Sq, st.kv
        j = torch.arange(Sq)
        b.positions = (P + j).repeat(G).to(torch.int32).to(self.device)
        b.dst_rows = (torch.arange(G)[:, None] * l.max_pos + P + j[None, :]).reshape(-1).to(self.device)
        if share:
            b.shared_kv, b.shared_len = scene.kv, P0
        self.llm_forward(x, P, head=False, batch=b)
        last = torch.tensor([g * Sq + lens[g] - 1 for g in range(G)], dtype=torch.int64, device=self.device)
        rows = st.rows
        ops.embed_gather(x, last, out=rows.x[:G])                                     # each question's last real row
        ops.rmsnorm(rows.x[:G], self.l_norm, l.eps, out=rows.last[:G])
        if self.llm_fp8:
            ops.linear_decode_fp8_rows(rows.last[:G], *self.l_head8, rows.logits[:G])
        else:
            ops.linear_decode_rows(rows.last[:G], self.l_head, rows.logits[:G])
        ctxs = st.ctxs[:G]
        for c, n in zip(ctxs, lens):
            c.kv_len = P + n
        toks = self.decode_group(rows, ctxs, [P + n for n in lens], max_new_tokens, logits_ready=True, shared_prefix=P if share else 0,
                                 eos_token_id=eos_token_id, prefix_kv=scene.kv if share else None)
        self.use(scene)
        return [r.to(self.device) for r in self.trim_at_eos(toks, eos_token_id)]

    @torch.no_grad()
    def generate_group(self, samples, max_new_tokens=16, eos_token_id=None):
        """Throughput form of `generate` for up to 32 (input_ids, images, world_coords) samples: each is prefilled into its
        own context, then all decode together (one pass over the weights per step).  Returns a list of token-id tensors,
        each cut after its first EOS like `generate` does.  The contexts and the row buffers are cached on the engine."""
        n = len(samples)
        if not 1 <= n <= self.MAX_GROUP:
            raise V3DError(f"generate_group takes 1 to {self.MAX_GROUP} samples")
        pool = self.__dict__.setdefault("_group_ctxs", [])
        while len(pool) < n:
            pool.append(self.new_context())
        grp = self.__dict__.get("_group_rows")
        if grp is None or grp.n < n:
            grp = self._group_rows = self.new_group(self.MAX_GROUP if n > 16 else (16 if n > 4 else 4))
        keep = self.ctx
        lens = []
        try:
            for c, (input_ids, images, world_coords) in zip(pool, samples):
                self.use(c)
                feats = self.encode_images(images)
                ids = self.voxel_ids(world_coords.to(self.dtype))
                x = self.build_inputs_embeds(input_ids, feats, ids)
                self._check_room(x.shape[0], max_new_tokens)
                self.llm_forward(x, 0, last_rows=[x.shape[0] - 1])
                lens.append(x.shape[0])
        finally:
            self.use(keep)
        toks = self.decode_group(grp, pool[:n], lens, max_new_tokens, eos_token_id=eos_token_id)
        return [r.to(self.device) for r in self.trim_at_eos(toks, eos_token_id)]

    # ------------------------------------------------------------------ scenes decoding together
    def new_group(self, n_scenes):
        """Row buffers for up to 32 scenes whose decode steps run as ONE pass over the weights (the decode step is
        HBM-bound on the 15 GB of weights and the weights feed the matrix cores directly, so M scenes cost about
        what one does; a row's arithmetic does not depend on the other rows, so grouping never changes a token)."""
        if not 1 <= n_scenes <= self.MAX_GROUP:
            raise V3DError(f"a decode group holds 1 to {self.MAX_GROUP} scenes")
        l = self.cfg.llm
        z = lambda *s: torch.zeros(s, dtype=self.dtype, device=self.device)
        g = SceneContext()
        g.n = n_scenes
        g.x, g.h, g.att = z(n_scenes, l.hidden), z(n_scenes, l.hidden), z(n_scenes, l.hidden)
        g.qkv, g.act = z(n_scenes, self.l_nqkv), z(n_scenes, l.inter)
        g.last, g.logits = z(n_scenes, l.hidden), z(n_scenes, self.l_head.shape[0])
        one = ops.decode_workspace(l.heads, l.kv_heads, self.device)
        g.ws = torch.empty(one.numel() * n_scenes, dtype=torch.float32, device=self.device)
        g.amax_ws = torch.empty(256 * n_scenes, dtype=torch.float32, device=self.device)
        return g

    def decode_forward_rows(self, g, ctxs, positions, shared_prefix=0, prefix_kv=None):
        """One new token for each of the M scenes (rows of g.x, in place) at its own position: per layer 4 weight-streaming
        linears over all rows + one rotary/append and one split-KV attention launch pair covering all scenes.
        shared_prefix > 0 (answer_group): the caches start with the same rows; the attention reads them once for all rows - from
        prefix_kv[i] (the scene's own per-layer caches) when given: the rows' caches then need not hold those rows at all."""
        l = self.cfg.llm
        M = len(ctxs)
        hd, nh, nkv = self.hd, l.heads, l.kv_heads
        kvw = nkv * hd
        scale = 1.0 / math.sqrt(hd)
        x, h, qkv, att, act = g.x[:M], g.h[:M], g.qkv[:M], g.att[:M], g.act[:M]
        sk = [p + 1 for p in positions]
        if self.llm_fp8:                  # configs[3]: e4m3 weights, 16-bit activations (W8A16) - half the bytes per step
            def lin(a, L, key, out, **kw):
                return ops.linear_decode_fp8_rows(a, *L[key + "8"], out, **kw)
        else:
            def lin(a, L, key, out, **kw):
                return ops.linear_decode_rows(a, L[key], out, **kw)
        # r04: with more than four rows the persistent decode linear forms the RMSNorm of its input rows itself (the bits of rmsnorm() +
        # the unfused call: ops.linear_decode_rows_fuses_norm) - two launches fewer per layer and one before the LM head
        def norm_lin(a, nw, w, out, **kw):
            if not self.llm_fp8 and ops.linear_decode_rows_fuses_norm(M, w.shape[0], w.shape[1], kw.get("epilogue", ops.DEC_NONE)):
                return ops.linear_decode_rows(a, w, out, norm_weight=nw, eps=l.eps, **kw)
            return None
        for i, L in enumerate(self.l_layers):
            caches = [c.kv[i] for c in ctxs]
            if norm_lin(x, L["ln1"], L.get("wqkv"), qkv, bias=L["bqkv"], epilogue=ops.DEC_BIAS) is None:
                ops.rmsnorm(x, L["ln1"], l.eps, out=h)
                lin(h, L, "wqkv", qkv, bias=L["bqkv"], epilogue=ops.DEC_BIAS)
            ops.rope_kv_append_rows(qkv, nh, nkv, hd, self.rope, positions, [c[p] for c, p in zip(caches, positions)])
            ops.attention_decode_rows(qkv, caches, [c[:, kvw:] for c in caches], att, sk, nh, nkv, scale, g.ws, prefix=shared_prefix,
                                      prefix_kv=(prefix_kv[i], prefix_kv[i][:, kvw:]) if prefix_kv is not None and shared_prefix else None)
            lin(att, L, "wo", x, res=x, epilogue=ops.DEC_RES)
            if norm_lin(x, L["ln2"], L.get("wgu"), act, epilogue=ops.DEC_SWIGLU) is None:
                ops.rmsnorm(x, L["ln2"], l.eps, out=h)
                lin(h, L, "wgu", act, epilogue=ops.DEC_SWIGLU)
            lin(act, L, "wd", x, res=x, epilogue=ops.DEC_RES)
        for c, p in zip(ctxs, positions):
            c.kv_len = p + 1
        if self.llm_fp8:
            ops.rmsnorm(x, self.l_norm, l.eps, out=g.last[:M])
            ops.linear_decode_fp8_rows(g.last[:M], *self.l_head8, g.logits[:M])
        elif norm_lin(x, self.l_norm, self.l_head, g.logits[:M]) is None:
            ops.rmsnorm(x, self.l_norm, l.eps, out=g.last[:M])
            ops.linear_decode_rows(g.last[:M], self.l_head, g.logits[:M])
        return g.logits[:M, : l.vocab]

    def decode_group(self, g, ctxs, prompt_lens, max_new_tokens, logits_ready=False, shared_prefix=0, eos_token_id=None, lookahead=2, prefix_kv=None):
        """Greedy decoding of M prefilled scenes together (their prefill logits are in ctx.logits[0], or already in g.logits
        with logits_ready); returns the token ids [M, n_steps] (device), n_steps <= max_new_tokens.
        eos_token_id (int or ints): the stop test runs on the device (v3d_eos_update: a done mask per row and their count); the
        host reads the count through a pinned copy `lookahead` steps behind the step it is queueing, so the launch queue never
        drains and the group stops at most `lookahead` - 1 steps after its last row has produced an EOS.  Rows that finished
        earlier keep decoding (their columns are independent); callers cut every row after its first EOS (trim_at_eos).
        Without eos_token_id all max_new_tokens steps run."""
        l = self.cfg.llm
        M = len(ctxs)
        if M > g.n:
            raise V3DError("decode group too small")
        self._check_room(max(prompt_lens), max_new_tokens)
        toks = torch.empty((max_new_tokens, M), dtype=torch.int64, device=self.device)
        if not logits_ready:
            for m, c in enumerate(ctxs):
                ops.copy_rows(c.logits[:1], g.logits[m: m + 1])
        logits = g.logits[:M, : l.vocab]
        eos = () if eos_token_id is None else ((int(eos_token_id),) if isinstance(eos_token_id, int) else tuple(int(e) for e in eos_token_id))
        if eos:
            st = self._eos_state(g, eos, max_new_tokens)
            st.done[:M].zero_()
        n_steps = max_new_tokens
        for step in range(max_new_tokens):
            ops.argmax_rows(logits, toks[step], g.amax_ws)
            if eos:
                ops.eos_update(toks[step], st.eos, st.done[:M], st.n_done)
                st.host[step].copy_(st.n_done, non_blocking=True)
                st.events[step].record()
                back = step - (lookahead - 1)
                if back >= 0:
                    st.events[back].synchronize()
                    if int(st.host[back]) >= M:           # every row had finished by step `back`: the steps queued since are the overshoot
                        n_steps = step + 1
                        break
            if step + 1 == max_new_tokens:
                break
            ops.embed_gather(self.embed, toks[step], out=g.x[:M])
            logits = self.decode_forward_rows(g, ctxs, [S + step for S in prompt_lens], shared_prefix=shared_prefix, prefix_kv=prefix_kv)
        return toks[:n_steps].t()

    def _eos_state(self, g, eos, max_new_tokens):
        """Device / pinned buffers of the stop test, kept on the decode group."""
        st = getattr(g, "eos_state", None)
        if st is None or st.ids != eos or len(st.events) < max_new_tokens:
            st = g.eos_state = SceneContext()
            st.ids = eos
            st.eos = torch.tensor(eos, dtype=torch.int64, device=self.device)
            st.done = torch.zeros(getattr(g, "n", 1), dtype=torch.int32, device=self.device)
            st.n_done = torch.zeros(1, dtype=torch.int32, device=self.device)
            st.host = torch.zeros((max_new_tokens, 1), dtype=torch.int32).pin_memory()
            st.events = [torch.cuda.Event() for _ in range(max_new_tokens)]
        return st

    @staticmethod
    def trim_at_eos(rows, eos_token_id):
        """rows [M, n] (host or device) -> list of 1-D host LongTensors, each cut after its first EOS id (generate's contract)."""
        eos = () if eos_token_id is None else ((int(eos_token_id),) if isinstance(eos_token_id, int) else tuple(int(e) for e in eos_token_id))
        out = []
        for row in rows.cpu():
            if eos:
                hit = [i for i, t in enumerate(row.tolist()) if t in eos]
                if hit:
                    row = row[: hit[0] + 1]
            out.append(row)
        return out

    def decode_loop(self, logits, S, max_new_tokens, eos_token_id=None, stopping=None, lookahead=2):
        """Greedy loop of one scene; token ids stay on the device (argmax kernel -> embedding gather).  EOS: the stop test runs on
        the device and the host reads it `lookahead` - 1 steps late (as decode_group does), so the launch queue never drains; the
        returned ids end with the first EOS.  A stopping callback (HF stopping_criteria) forces one host synchronisation per step."""
        self._check_room(S, max_new_tokens)
        eos = () if eos_token_id is None else ((int(eos_token_id),) if isinstance(eos_token_id, int) else tuple(int(e) for e in eos_token_id))
        toks = torch.empty(max_new_tokens, dtype=torch.int64, device=self.device)
        st = None
        if eos and stopping is None:
            st = self._eos_state(self.ctx, eos, max_new_tokens)
            st.done[:1].zero_()
        n = 0
        for step in range(max_new_tokens):
            ops.argmax(logits, toks[step: step + 1])
            n = step + 1
            if st is not None:
                ops.eos_update(toks[step: step + 1], st.eos, st.done[:1], st.n_done)
                st.host[step].copy_(st.n_done, non_blocking=True)
                st.events[step].record()
                back = step - (lookahead - 1)
                if back >= 0:
                    st.events[back].synchronize()
                    if int(st.host[back]) >= 1:
                        break
            elif eos and int(toks[step]) in eos:
                break
            if stopping is not None and stopping(toks[:n]):
                break
            if step + 1 == max_new_tokens:
                break
            xe = ops.embed_gather(self.embed, toks[step: step + 1], out=self.l_x[S + step: S + step + 1])
            logits = self.decode_forward(xe, S + step)
        out = toks[:n]
        if st is not None:
            out = self.trim_at_eos(out[None], eos)[0].to(self.device)
        return out
