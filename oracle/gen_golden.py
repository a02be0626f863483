#!/usr/bin/env python3
"""Golden-vector generator: runs the REAL reference code (CPU) and stores inputs + outputs.

TEST INFRASTRUCTURE ONLY.  Runs in the build container (where /root/reference exists);
never on the GPU box, never from the product path.  Only data (inputs / expected
outputs) is written, as small .npz files under tests/golden/.  No reference source text
is copied: the reference modules are imported from where they lie, with the import
recipe of SURVEY.md section 8c (namespace packages so that the two __init__.py files are
skipped, an empty `cv2` module, a stub for the Q-Former resampler that no longer imports
on the installed transformers).

Usage:  python oracle/gen_golden.py [--only NAME ...]
"""
import sys
sys.dont_write_bytecode = True  # never write into the reference tree
import argparse
import io
import json
import os
import random
import re
import sys
import types

import numpy as np
import torch

REF = os.environ.get("V3D_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _import_reference():
    def ns(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
        return m

    ns("llava", REF + "/llava")
    ns("llava.model", REF + "/llava/model")
    sys.modules["cv2"] = types.ModuleType("cv2")
    q = types.ModuleType("llava.model.multimodal_resampler.qformer")

    class Qformer:  # never instantiated on this path
        pass

    q.Qformer = Qformer
    sys.modules["llava.model.multimodal_resampler.qformer"] = q


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}.npz  ({os.path.getsize(path) / 1024:.0f} KiB)")


def t2n(t):
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16).numpy().view(np.uint16)  # raw bits
    return t.numpy()


class _Tower:
    num_patches_per_side = 27


def make_arch(cfg_kw):
    """A minimal concrete LlavaMetaForCausalLM so its mixin methods can be called."""
    import llava.model.llava_arch as arch

    class M(arch.LlavaMetaForCausalLM):
        def __init__(self):
            self.config = types.SimpleNamespace(**cfg_kw)
            self.model = types.SimpleNamespace()

        def get_model(self):
            return self.model

        def get_vision_tower(self):
            return _Tower()

    return M()


DEFAULT_CFG = dict(mm_spatial_pool_mode="bilinear", mm_spatial_pool_stride=2, voxel_size=0.1,
                   min_xyz_range=[-15, -15, -5], max_xyz_range=[15, 15, 5],
                   world_position_embedding_type="avg-discrete-sin3d")


# --------------------------------------------------------------------------------------
def g_unproject():
    from llava.video_utils import unproject
    g = torch.Generator().manual_seed(1)
    V, H, W = 3, 48, 64
    depth = torch.randint(0, 6000, (V, H, W), generator=g, dtype=torch.int32)
    depth[0, :4, :4] = 0  # invalid-depth pixels
    K = torch.zeros(V, 4, 4)
    for v in range(V):
        K[v] = torch.tensor([[577.870605 + v, 0, 31.5 + 0.25 * v, 0], [0, 577.870605 - v, 23.5, 0],
                             [0, 0, 1, 0], [0, 0, 0, 1]])
    P = torch.zeros(V, 4, 4, dtype=torch.float64)
    for v in range(V):
        a, b = 0.3 + v, -0.2 * v
        Rz = torch.tensor([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
        Rx = torch.tensor([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
        P[v, :3, :3] = Rz @ Rx
        P[v, :3, 3] = torch.tensor([1.5 * v - 1, 0.7, 1.2 + 0.1 * v])
        P[v, 3, 3] = 1
    P = P.float()
    out = unproject(K, P, depth.float())
    save("unproject", depth=depth.numpy().astype(np.uint16), intrinsics=K.numpy(), poses=P.numpy(),
         world=out.numpy())


def _coords_frame(seed, V=1):
    """Plausible world coords whose values are all fp16-representable (so one file serves both dtypes)."""
    g = torch.Generator().manual_seed(seed)
    base = (torch.rand(V, 384, 384, 3, generator=g) - 0.5) * torch.tensor([12.0, 12.0, 4.0])
    base[:, :27, :27] = 20.0 + base[:, :27, :27]      # first patch beyond the +15 clamp
    base[:, 27:54, :27] = -20.0 + base[:, 27:54, :27]  # below the -15 clamp
    return base.half()


def g_coord_pool():
    m = make_arch(DEFAULT_CFG)
    x16 = _coords_frame(2)
    x32 = x16.float() * np.float32(1.001) + np.float32(0.0003)  # full-mantissa f32 variant (IEEE-deterministic)
    o32 = m.average_coordinate_in_patch(x32)
    o16 = m.average_coordinate_in_patch(x16)
    d32 = m.discrete_coords(o32, None)
    d16 = m.discrete_coords(o16, None)
    save("coord_pool", coords_f16=x16.numpy(), avg_f32=o32.numpy(), avg_f16=o16.numpy(),
         vox_f32=d32.numpy(), vox_f16=d16.numpy())


def g_discrete():
    m = make_arch(DEFAULT_CFG)
    # exhaustive fp16: every finite bit pattern, used as x, y and z
    bits = np.arange(65536, dtype=np.uint16)
    h = torch.from_numpy(bits.view(np.float16).copy())
    finite = torch.isfinite(h)
    hx = h[finite]
    xyz16 = torch.stack([hx, hx, hx], -1)
    out16 = m.discrete_coords(xyz16, None)
    # f32: exact ties, near ties, out of range, random
    g = torch.Generator().manual_seed(3)
    ks = torch.arange(0, 301, dtype=torch.float32)
    ties = (ks + 0.5) * 0.1 - 15.0
    f = torch.cat([ties, torch.nextafter(ties, torch.tensor(100.0)), torch.nextafter(ties, torch.tensor(-100.0)),
                   torch.tensor([-1e9, -15.0, -15.00001, 15.0, 15.00001, 1e9, 0.0, -0.0, 0.05, 0.15, 0.25]),
                   (torch.rand(20000, generator=g) - 0.5) * 40])
    xyz32 = torch.stack([f, f.flip(0), f * (1.0 / 3.0)], -1)
    out32 = m.discrete_coords(xyz32, None)
    save("discrete_coords", finite_mask=finite.numpy(), vox_f16=out16.numpy(), xyz_f32=xyz32.numpy(),
         vox_f32=out32.numpy())
    # VideoProcessor.discrete_point (host, int output)
    from llava.video_utils import VideoProcessor
    vp = object.__new__(VideoProcessor)
    vp.voxel_size = 0.1
    vp.min_xyz_range = None
    vp.max_xyz_range = None
    pts = ((torch.rand(500, 3, generator=g) - 0.5) * 20).tolist() + [[0, 0, 0], [0.05, 0.15, 0.25], [-0.05, -0.15, -0.25]]
    a = vp.discrete_point(pts)
    vp.min_xyz_range = torch.tensor([-15, -15, -5])
    vp.max_xyz_range = torch.tensor([15, 15, 5])
    b = vp.discrete_point(pts)
    save("discrete_point", pts=np.array(pts, dtype=np.float64), ids_norange=np.array(a, dtype=np.int32),
         ids_range=np.array(b, dtype=np.int32))


def g_sin3d():
    from llava.model.position_encoding import PositionEmbeddingSine3D
    pe = PositionEmbeddingSine3D(3584)
    ids = torch.arange(301, dtype=torch.float32)
    x = torch.stack([ids, ids.flip(0), ids.clamp(max=100)], -1)[None]     # [1,301,3]
    out = pe(x)[0]                                                          # [301,3584] f32
    nf = 3584 // 3
    d = torch.arange(nf, dtype=torch.float32)
    dim_t = 10000 ** (2 * (d // 2) / nf)   # same expression, same CPU pow as the module evaluates
    tab = out[:, :nf]
    assert torch.equal(out[:, nf:2 * nf], tab.flip(0))
    assert torch.equal(out[:101, 2 * nf:3 * nf], tab[:101])
    assert torch.equal(out[:, 3 * nf:], torch.zeros(301, 2))
    save("sin3d_table_3584", dim_t=dim_t.numpy(), table=tab.numpy())
    # fp16 / bf16 inputs -> output is cast to the input dtype
    g = torch.Generator().manual_seed(4)
    xi = torch.stack([torch.randint(0, 301, (64,), generator=g), torch.randint(0, 301, (64,), generator=g),
                      torch.randint(0, 101, (64,), generator=g)], -1)[None].float()
    o16 = pe(xi.half())
    ob16 = pe(xi.bfloat16())
    save("sin3d_tokens_3584", ids=xi.numpy().astype(np.int32), pe_f16=o16.numpy(), pe_bf16_bits=t2n(ob16))
    # odd num_feats branch (embedding 16 -> num_feats 5) and continuous (non-integer) coords, small width
    pe16 = PositionEmbeddingSine3D(16)
    xc = (torch.rand(1, 50, 3, generator=g) - 0.5) * 30
    o = pe16(xc)
    pe96 = PositionEmbeddingSine3D(96)
    o96 = pe96(xc)
    d5 = torch.arange(5, dtype=torch.float32)
    d32 = torch.arange(32, dtype=torch.float32)
    save("sin3d_small", xyz=xc.numpy(), pe16=o.numpy(), pe96=o96.numpy(),
         dim_t5=(10000 ** (2 * (d5 // 2) / 5)).numpy(), dim_t32=(10000 ** (2 * (d32 // 2) / 32)).numpy())


def g_pool2d():
    m = make_arch(DEFAULT_CFG)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 729, 64, generator=g)
    o32 = m.get_2dPool(x)
    o16 = m.get_2dPool(x.half())
    ob = m.get_2dPool(x.bfloat16())
    save("pool2d_bilinear", feat=x.numpy(), out_f32=o32.numpy(), out_f16=o16.numpy(), out_bf16_bits=t2n(ob))


def g_newline():
    m = make_arch(DEFAULT_CFG)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(3, 196, 8, generator=g)
    m.model.image_newline = torch.randn(8, generator=g)
    o = m.add_token_per_grid(x)
    save("add_token_per_grid", feat=x.numpy(), newline=m.model.image_newline.numpy(), out=o.contiguous().numpy())


def g_fused():
    """pool -> voxel PE -> add -> newline, composed exactly as prepare_inputs_labels_for_multimodal does
    (llava_arch.py:395-420, 469, 506-517, 536) but at a small channel width (C=96)."""
    from llava.model.position_encoding import PositionEmbeddingSine3D
    C = 96
    m = make_arch(DEFAULT_CFG)
    pe = PositionEmbeddingSine3D(C)
    g = torch.Generator().manual_seed(7)
    coords = _coords_frame(8, V=2)
    feats = torch.randn(2, 729, C, generator=g)
    m.model.image_newline = torch.randn(C, generator=g)
    out = {}
    for name, dt in (("f16", torch.float16), ("bf16", torch.bfloat16), ("f32", torch.float32)):
        wc = coords.to(dt)
        f = feats.to(dt)
        avg = m.average_coordinate_in_patch(wc)
        vox = m.discrete_coords(avg, None)
        pooled = m.get_2dPool(f)
        fused = pooled + pe(vox.flatten(1, 2).detach())
        old = m.model.image_newline
        m.model.image_newline = old.to(dt)
        seq = m.add_token_per_grid(fused).contiguous()
        m.model.image_newline = old
        out["vox_" + name] = vox.float().numpy()
        out["seq_" + name] = t2n(seq) if dt == torch.bfloat16 else seq.numpy()
    d = torch.arange(C // 3, dtype=torch.float32)
    save("fused_small", coords_f16=coords.numpy(), feat=feats.numpy(), newline=m.model.image_newline.numpy(),
         dim_t=(10000 ** (2 * (d // 2) / (C // 3))).numpy(), **out)


def g_frames():
    from llava.video_utils import VideoProcessor
    vp = object.__new__(VideoProcessor)
    vp.video_folder = "data"
    res = {}
    for n in (8, 31, 32, 33, 100, 517, 1):
        vp.scene = {"s": {"images": [{"img_path": f"posed_images/s/{i * 10:05d}.jpg"} for i in range(n)]}}
        for F in (8, 32):
            files = vp.sample_frame_files("s", force_sample=True, frames_upbound=F)
            res[f"n{n}_F{F}"] = [int(os.path.basename(f).split(".")[0]) // 10 for f in files]
        files = vp.sample_frame_files("s", force_sample=False, frames_upbound=F)
        res[f"n{n}_default"] = [int(os.path.basename(f).split(".")[0]) // 10 for f in files]
    # max-coverage prefix / ratio cuts
    rnd = random.Random(9)
    order = list(range(60))
    rnd.shuffle(order)
    voxel_nums = sorted([rnd.randint(1, 900) for _ in range(40)], reverse=True)
    entry = {"video_id": "s", "frame_files": [f"data/posed_images/s/{i * 20:05d}.jpg" for i in order[:40]],
             "voxel_nums": voxel_nums, "num_all_voxels": int(sum(voxel_nums) * 1.04)}
    mc = {}
    for strat in ("mc", "mc-ratio90", "mc-ratio95"):
        for F in (8, 16, 32):
            vp.frame_sampling_strategy = strat
            vp.mc_sampling_files = {"s": json.loads(json.dumps(entry))}
            mc[f"{strat}_F{F}"] = vp.sample_frame_files_mc("s", frames_upbound=F)
    with open(os.path.join(OUT, "frame_sampling.json"), "w") as f:
        json.dump({"uniform": res, "mc_entry": entry, "mc": mc}, f)
    print("  wrote frame_sampling.json")


def g_greedy():
    """Greedy max-coverage selection: executes the reference's own loop
    (scripts/3d/preprocessing/max_coverage_sampling.py:44-94) on synthetic voxel sets, with the
    unseeded random.choice tie-break replaced by "first candidate" (= lowest position in frame order)."""
    path = REF + "/scripts/3d/preprocessing/max_coverage_sampling.py"
    with open(path) as f:
        lines = f.read().split("\n")
    # lines 43..94 (1-based) of main(): from `world_coords = world_coords / voxel_size` to the break
    start = next(i for i, l in enumerate(lines) if "world_coords = world_coords / video_processor.voxel_size" in l)
    end = next(i for i, l in enumerate(lines) if "if len(select_frame_files) >= 32:" in l) + 2
    body = "\n".join(l[8:] if l.startswith("        ") else l for l in lines[start:end])
    body = body.replace("world_coords.to('cuda')", "world_coords")
    from llava.video_utils import VideoProcessor
    cases = {}
    for case, (n_frames, seed) in {"a": (40, 11), "b": (12, 12), "c": (70, 13)}.items():
        g = torch.Generator().manual_seed(seed)
        H, W = 24, 32
        centers = (torch.rand(n_frames, 1, 1, 3, generator=g) - 0.5) * torch.tensor([6.0, 6.0, 1.0])
        wc = centers + (torch.rand(n_frames, H, W, 3, generator=g) - 0.5) * torch.tensor([2.5, 2.5, 1.5])
        if case == "b":
            wc[5] = wc[2]  # exact duplicate frames -> forced ties
            wc[7] = wc[2]
        frame_files = [f"f{i:03d}.jpg" for i in range(n_frames)]
        vp = object.__new__(VideoProcessor)
        vp.voxel_size = 0.1
        vp.min_xyz_range = None
        vp.max_xyz_range = None
        scene_pts = (wc.reshape(-1, 3)[torch.randperm(wc.numel() // 3, generator=g)[: wc.numel() // 4]]).tolist()
        pc = list(set(tuple(x) for x in vp.discrete_point(scene_pts)))

        class FirstChoice:
            @staticmethod
            def choice(seq):
                return seq[0]

        ns = dict(world_coords=wc.clone(), video_processor=vp, frame_files=frame_files, torch=torch, np=np,
                  random=FirstChoice, pc_data={"s": pc}, scene_id="s", world_coords_discrete={})
        exec(compile(body, "<reference greedy loop>", "exec"), ns)
        cases[case] = dict(
            world=wc.numpy(), pc=np.array(pc, dtype=np.int32),
            select=np.array([frame_files.index(f) for f in ns["select_frame_files"]], dtype=np.int32),
            voxel_nums=np.array(ns["voxel_nums"], dtype=np.int64),
            num_all=np.int64(len(ns["all_voxel"] & ns["pc_voxel"])),
            num_sel=np.int64(len(ns["used_voxel"] & ns["pc_voxel"])))
    flat = {f"{c}_{k}": v for c, d in cases.items() for k, v in d.items()}
    save("greedy_cover", **flat)


def g_box():
    from llava.utils_3d import convert_pc_to_box
    g = np.random.default_rng(14)
    pc = g.normal(size=(200, 6)).astype(np.float32)
    c, s = convert_pc_to_box(pc)
    save("convert_pc_to_box", pc=pc, center=np.array(c), size=np.array(s))


def g_llm():
    """Qwen2 building blocks run through the reference's vendored modeling_qwen2 (eager attention spec)."""
    from llava.model.language_model.qwen2 import modeling_qwen2 as mq
    from transformers.models.qwen2.configuration_qwen2 import Qwen2Config
    cfg = Qwen2Config(vocab_size=320, hidden_size=256, intermediate_size=384, num_hidden_layers=2,
                      num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=4096, rms_norm_eps=1e-6,
                      rope_theta=1000000.0, use_sliding_window=False, attention_dropout=0.0)
    cfg.rope_theta = 1000000.0
    cfg._attn_implementation = "eager"
    torch.manual_seed(31)
    layer = mq.Qwen2DecoderLayer(cfg, 0).eval()
    for n, p_ in layer.named_parameters():
        with torch.no_grad():
            if "norm" in n:
                p_.copy_(1.0 + 0.1 * torch.randn_like(p_))
            elif "bias" in n:
                p_.copy_(0.1 * torch.randn_like(p_))
            else:
                p_.copy_(0.05 * torch.randn_like(p_))
    S = 150
    x = torch.randn(1, S, 256) * 0.7
    pos = torch.arange(S)[None, :, None].repeat(1, 1, 3)
    from transformers.modeling_attn_mask_utils import _prepare_4d_causal_attention_mask
    out = {}
    weights = {k: v.detach().clone() for k, v in layer.state_dict().items()}
    for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        l2 = mq.Qwen2DecoderLayer(cfg, 0).eval().to(dt)
        l2.load_state_dict({k: v.to(dt) for k, v in weights.items()})
        # from_pretrained(torch_dtype=...) (llava/model/builder.py:35-38, 206-228) leaves the explicitly-f32
        # rotary inv_freq buffer in f32; a bare module.to(dt) would round it - keep the eval-path behaviour.
        l2.self_attn.rotary_emb.inv_freq = layer.self_attn.rotary_emb.inv_freq.float().clone()
        xd = x.to(dt)
        mask = _prepare_4d_causal_attention_mask(None, (1, S), xd, 0)
        y = l2(xd, attention_mask=mask, position_ids=pos)[0]
        h = l2.input_layernorm(xd)
        cos, sin = l2.self_attn.rotary_emb(xd, pos)
        out["y_" + name] = t2n(y) if dt == torch.bfloat16 else y.numpy()
        out["norm_" + name] = t2n(h) if dt == torch.bfloat16 else h.numpy()
        out["cos_" + name] = t2n(cos[0, 0]) if dt == torch.bfloat16 else cos[0, 0].numpy()
        out["sin_" + name] = t2n(sin[0, 0]) if dt == torch.bfloat16 else sin[0, 0].numpy()
        m = l2.mlp(h)
        out["mlp_" + name] = t2n(m) if dt == torch.bfloat16 else m.numpy()
    save("qwen2_layer", x=x.numpy(), **{"w." + k: v.numpy() for k, v in weights.items()}, **out)


def g_vit():
    """One SigLIP encoder layer + embeddings + the mlp2x_gelu projector, through the reference modules."""
    from llava.model.multimodal_encoder import siglip_encoder as se
    from llava.model.multimodal_projector.builder import build_vision_projector
    cfg = se.SigLipVisionConfig(hidden_size=144, intermediate_size=272, num_hidden_layers=1, num_attention_heads=2,
                                image_size=56, patch_size=14)
    torch.manual_seed(41)
    layer = se.SigLipEncoderLayer(cfg).eval()
    emb = se.SigLipVisionEmbeddings(cfg).eval()
    proj = build_vision_projector(types.SimpleNamespace(mm_projector_type="mlp2x_gelu", mm_hidden_size=144, hidden_size=256)).eval()
    with torch.no_grad():
        for mod in (layer, emb, proj):
            for n, p_ in mod.named_parameters():
                if "layer_norm" in n and "weight" in n:
                    p_.copy_(1.0 + 0.1 * torch.randn_like(p_))
                elif "bias" in n:
                    p_.copy_(0.1 * torch.randn_like(p_))
                else:
                    p_.copy_(0.06 * torch.randn_like(p_))
    pix = torch.randn(3, 3, 56, 56)
    out = {}
    wl = {"layer." + k: v.detach().clone() for k, v in layer.state_dict().items()}
    we = {"emb." + k: v.detach().clone() for k, v in emb.state_dict().items() if "position_ids" not in k}
    wp = {"proj." + k: v.detach().clone() for k, v in proj.state_dict().items()}
    for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        e = emb.to(dt)(pix.to(dt))
        y = layer.to(dt)(e, None)[0]
        z = proj.to(dt)(y)
        for k, v in (("emb_", e), ("y_", y), ("proj_", z)):
            out[k + name] = t2n(v) if dt == torch.bfloat16 else v.numpy()
        emb.float(); layer.float(); proj.float()
    save("siglip_layer", pixels=pix.numpy(), **{k: v.numpy() for k, v in {**wl, **we, **wp}.items()}, **out)


def g_objects():
    """Object-proposal patch masks + masked-mean features + box-centre PE (llava_arch.py:351-376, 416-420,
    479-501, object_feature_type 'patch14-pe') and the infonce grounding head (llava_qwen.py:294-300).
    The two llava_arch blocks sit inside prepare_inputs_labels_for_multimodal, so their own source lines are
    executed here (read at run time, nothing copied) in a namespace holding the tensors they expect."""
    import textwrap
    import torch.nn.functional as F
    from llava.model.position_encoding import PositionEmbeddingSine3D
    with open(REF + "/llava/model/llava_arch.py") as f:
        lines = f.read().split("\n")
    a0 = next(i for i, l in enumerate(lines) if 'object_boxes = video_dict["objects"][0]' in l)
    a1 = next(i for i, l in enumerate(lines) if "use_mrope_position_embedding = False" in l)
    b0 = next(i for i, l in enumerate(lines) if "object_features = []" in l and i > a1)
    b1 = next(i for i, l in enumerate(lines) if "object_features += box_center_features" in l) + 1
    blk_a = textwrap.dedent("\n".join(lines[a0:a1]))
    blk_b = textwrap.dedent("\n".join(lines[b0:b1]))
    C, Fr = 96, 2
    g = torch.Generator().manual_seed(51)
    m = make_arch(dict(DEFAULT_CFG, object_feature_type="patch14-pe"))
    pe = PositionEmbeddingSine3D(C)
    m.model.world_position_embedding = pe
    out = {}
    coords = (torch.rand(Fr, 384, 384, 3, generator=g) - 0.5) * torch.tensor([8.0, 8.0, 3.0])
    # piecewise-constant regions so that some 14x14 patches fall wholly inside boxes
    coords = coords.view(Fr, 48, 8, 48, 8, 3)[:, :, :1, :, :1, :].expand(Fr, 48, 8, 48, 8, 3).reshape(Fr, 384, 384, 3).contiguous()
    coords = coords.half().float()          # fp16-representable so one stored array serves both dtypes
    boxes = torch.cat([(torch.rand(9, 3, generator=g) - 0.5) * torch.tensor([6.0, 6.0, 2.0]), torch.rand(9, 3, generator=g) * 4 + 0.5], 1)
    boxes[8] = torch.tensor([50.0, 50.0, 50.0, 0.1, 0.1, 0.1])        # selects nothing -> zero feature
    feats = torch.randn(Fr, 729, C, generator=g)
    for name, dt in (("f32", torch.float32), ("f16", torch.float16)):
        ns = dict(torch=torch, self=m, video_dict={"world_coords": coords.to(dt)[None], "objects": boxes.to(dt)[None]},
                  int=int)
        exec(compile(blk_a, "<llava_arch object masks>", "exec"), ns)
        centers = m.discrete_coords(ns["object_boxes_center"], None)           # :418-420
        enc = feats.to(dt)
        ns.update(image_features=[m.get_2dPool(enc)], encoded_image_features=[enc.flatten(0, 1)[None].squeeze(0).view(Fr, 729, C)],
                  use_mlp_pe=False, use_sin3d_pe=True, object_boxes_center=centers)
        exec(compile(blk_b, "<llava_arch object features>", "exec"), ns)
        out["mask_" + name] = torch.stack(ns["object_patch"]).numpy()
        out["objfeat_" + name] = ns["object_features"].float().numpy()
        out["centers_" + name] = centers.float().numpy()
    # infonce head (llava_qwen.py:87-104, 294-300)
    torch.manual_seed(52)
    H = 96
    head_obj = torch.nn.Sequential(torch.nn.Linear(H, H), torch.nn.ReLU(), torch.nn.LayerNorm(H), torch.nn.Linear(H, H))
    head_q = torch.nn.Sequential(torch.nn.Linear(H, H), torch.nn.ReLU(), torch.nn.LayerNorm(H), torch.nn.Linear(H, H))
    zero_t = torch.randn(H)
    objf = torch.from_numpy(out["objfeat_f32"])
    q = torch.randn(1, H)
    with torch.no_grad():
        of = torch.cat([objf, zero_t[None]], 0)
        scores = (F.normalize(head_obj(of)) * F.normalize(head_q(q))).sum(-1)
    hw = {"ho." + k: v.detach().numpy() for k, v in head_obj.state_dict().items()}
    hw.update({"hq." + k: v.detach().numpy() for k, v in head_q.state_dict().items()})
    d = torch.arange(C // 3, dtype=torch.float32)
    save("objects", coords=coords.half().numpy(), boxes=boxes.numpy(), feats=feats.numpy(), zero_target=zero_t.numpy(),
         query=q.numpy(), scores=scores.numpy(), dim_t=(10000 ** (2 * (d // 2) / (C // 3))).numpy(), **hw, **out)


def g_imgproc():
    """a7: SigLipImageProcessor.preprocess (siglip_encoder.py:47-67) on 384x384 RGB frames, the size VideoProcessor
    hands over (video_utils.py:292-308), so the bicubic resize is the identity; plus one 40x56 frame that really resizes."""
    from PIL import Image
    from llava.model.multimodal_encoder import siglip_encoder as se
    rng = np.random.default_rng(61)
    frames = rng.integers(0, 256, size=(2, 384, 384, 3), dtype=np.uint8)
    frames[0, :4, :4] = [[[0, 255, 128]]]
    proc = se.SigLipImageProcessor()
    pv = proc.preprocess([Image.fromarray(f) for f in frames], return_tensors="pt")["pixel_values"]
    small = rng.integers(0, 256, size=(40, 56, 3), dtype=np.uint8)
    pv_small = proc.preprocess([Image.fromarray(small)], return_tensors="pt")["pixel_values"]
    assert pv.dtype == torch.float32 and tuple(pv.shape) == (2, 3, 384, 384)
    save("imgproc", frames=frames[:, :96, :96].copy(), pixel_values=pv[:, :, :96, :96].numpy().copy(),
         full_sum=np.array(pv.double().sum().item()), small=small, small_pixel_values=pv_small[0, :, ::16, ::16].numpy().copy())


def g_rgb_resize():
    """a6 / a7, RGB half: the frame loop of VideoProcessor.preprocess, strategy "center_crop" (video_utils.py:285-306) - executed from
    the reference's own source text (lines read at run time, nothing stored) on PIL images: PIL `frame.resize((new_w, crop))` with
    the default filter + `frame.crop(...)`.  Frames: noise and smooth content, 640 x 480 (the ScanNet size) -> 512 x 384 -> 384 x 384,
    plus a small 50 x 70 frame -> crop 24 (another ratio, clamped taps at every border).  Stored: inputs as PNG-free raw uint8,
    outputs as uint8 (the 640 x 480 outputs as a strided sample + an exact checksum to stay small)."""
    import inspect
    import textwrap
    from PIL import Image
    from llava import video_utils as vu
    src = inspect.getsource(vu.VideoProcessor.preprocess).split("\n")
    a = next(i for i, l in enumerate(src) if l.strip() == "images = []")
    b = next(i for i, l in enumerate(src) if "images = [frame.crop(" in l and i > a)
    body = textwrap.dedent("\n".join(src[a: b + 1]))
    assert "images = [frame.resize((new_width, new_height)) for frame in images]" in body and 'strategy == "center_crop"' in body, body

    class _Proc:
        def __init__(self, crop):
            self.crop_size = {"height": crop, "width": crop}

    cv2stub = type("cv2stub", (), {"INTER_NEAREST": 0, "resize": staticmethod(
        lambda x, dsize, interpolation=None: np.zeros((dsize[1], dsize[0], 3), np.float32))})     # (the coordinate half: a6, elsewhere)

    def run(frames, crop):
        files = []
        for fr in frames:
            buf = io.BytesIO()
            Image.fromarray(fr).save(buf, format="PNG")          # lossless: the loop's Image.open sees exactly `fr`
            files.append(io.BytesIO(buf.getvalue()))
        H, W = frames.shape[1:3]
        env = {"Image": Image, "frame_files": files, "strategy": "center_crop", "image_processor": _Proc(crop), "cv2": cv2stub,
               "world_coords": [torch.zeros(H, W, 3) for _ in frames], "H": H, "W": W}
        exec(body, env)
        return np.stack([np.asarray(im) for im in env["images"]])

    rng = np.random.default_rng(77)
    noise = rng.integers(0, 256, size=(1, 480, 640, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:480, 0:640]
    smooth = np.stack([(127 + 120 * np.sin(xx / 37.0 + c) * np.cos(yy / 23.0 - c)).astype(np.uint8) for c in range(3)], -1)[None]
    edges = np.zeros((1, 480, 640, 3), np.uint8)
    edges[0, ::7] = 255
    edges[0, :, ::5, 1] = 255
    big = np.concatenate([noise, smooth, edges])
    big_out = run(big, 384)
    assert big_out.shape == (3, 384, 384, 3) and big_out.dtype == np.uint8
    small = rng.integers(0, 256, size=(2, 50, 70, 3), dtype=np.uint8)
    small_out = run(small, 24)
    save("rgb_resize", noise_seed=np.array(77), smooth=smooth[:, ::8, ::8].copy(), big_out_sample=big_out[:, ::4, ::4].copy(),
         big_out_sum=np.array([int(big_out[i].astype(np.int64).sum()) for i in range(3)]),
         big_out_rowsum=big_out.astype(np.int64).sum((2, 3)), small=small, small_out=small_out)


def g_chatml():
    """The eval drivers' prompt builder `preprocess_qwen` - model_scanqa.py:29-80 (ids only) and model_scanrefer.py:28-80 (ids + training
    targets) - executed from the drivers' own source text (read at run time; the modules themselves import ray / fasteners, absent here)
    on a stand-in word-level tokenizer with the ChatML specials (no Qwen2 tokenizer ships with the reference): pins the STRUCTURE of the
    id / label sequences the runners build (v3d.eval_scanqa.chatml_ids, v3d.eval_3d.chatml_ids_labels)."""
    import ast
    import re as _re
    from typing import Dict
    import transformers
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    names = {300: "system", 301: "user", 302: "assistant", 303: "\n", 304: "You", 305: "are", 306: "a", 307: "helpful",
             308: "assistant.", 309: "<|im_start|>", 310: "<|im_end|>"}
    vocab = {names.get(i, f"t{i}"): i for i in range(320)}
    tk = Tokenizer(models.WordLevel(vocab, unk_token="t0"))
    tk.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.Split("\n", "isolated"), pre_tokenizers.Split(" ", "removed")])
    fast = PreTrainedTokenizerFast(tokenizer_object=tk, eos_token="<|im_end|>", pad_token="t0", unk_token="t0",
                                   additional_special_tokens=["<|im_start|>", "<|im_end|>"])

    class Tok:                                   # the attribute the drivers read (:32), dropped by newer transformers
        additional_special_tokens_ids = [309, 310]

        def __call__(self, s):
            return fast(s)

    def load(path):
        src = open(os.path.join(REF, path)).read()
        fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "preprocess_qwen")
        env = {"transformers": transformers, "Dict": Dict, "torch": torch, "re": _re, "IGNORE_INDEX": -100, "IMAGE_TOKEN_INDEX": -200,
               "DEFAULT_IMAGE_TOKEN": "<image>"}
        exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), env)
        return env["preprocess_qwen"]

    qa, vg = load("llava/eval/model_scanqa.py"), load("llava/eval/model_scanrefer.py")
    cases = [
        [{"from": "human", "value": "<image>\nt1 t2 t3"}, {"from": "gpt", "value": None}],
        [{"from": "human", "value": "<image>\nt5 t317 t6 t7"}, {"from": "gpt", "value": None}],
        [{"from": "gpt", "value": "t9"}, {"from": "human", "value": "t4"}, {"from": "gpt", "value": "t5 t6"}],
    ]
    vg_cases = [
        [{"from": "human", "value": "<image>\nt11 t12 t13 t14"}, {"from": "gpt", "value": "t318"}],
        [{"from": "human", "value": "<image>\nt21"}, {"from": "gpt", "value": "t30 t318 t31"}],
    ]
    out = {"qa": [{"turns": c, "ids": qa(c, Tok(), has_image=True)[0].tolist()} for c in cases], "vg": []}
    for c in vg_cases:
        ids, labels = vg(c, Tok(), has_image=True)
        out["vg"].append({"turns": c, "ids": ids[0].tolist(), "labels": labels[0].tolist()})
    with open(os.path.join(OUT, "chatml.json"), "w") as f:
        json.dump(out, f)
    print("  wrote chatml.json")


def _bits(t):
    """bf16-representable f32 tensor -> raw bf16 bits (uint16), the compact storage form of the tiny-model weights."""
    b = t.detach().to(torch.bfloat16)
    assert torch.equal(b.float(), t.detach().float())
    return b.view(torch.int16).numpy().view(np.uint16)


def g_tiny_model():
    """Whole-model golden: a random-init tiny LlavaQwenForCausalLM (the reference's own class: llava_qwen.py:43-119)
    with a 2-layer SigLIP tower (siglip_encoder.py:538-589, true 384/14 geometry and head dim 72), the mlp2x_gelu projector,
    PositionEmbeddingSine3D and a 2-layer Qwen2 (head dim 128), run through the reference's own
    prepare_inputs_labels_for_multimodal (llava_arch.py:336-836) + Qwen2ForCausalLM.forward (modeling_qwen2.py:1132-1217):
    inputs_embeds, last-row logits, greedy continuation (each step = the reference forward over the sequence extended by the
    chosen token, use_cache=False: HF's cache classes drifted under the installed transformers, the arithmetic of the
    eager path is the same) and the ScanRefer-style grounding forward (use_object_proposals=True -> predict_box, infonce).
    Cases: F = 2 frames in f32 / bf16 / f16, F = 8 frames (BASELINE configs[0]) in f16 (the eval dtype, builder.py:27)."""
    from llava.model.language_model import llava_qwen as lq
    from llava.model.multimodal_encoder import siglip_encoder as se
    from llava.model.multimodal_projector.builder import build_vision_projector
    H, VH = 256, 144
    cfg = lq.LlavaQwenConfig(vocab_size=320, hidden_size=H, intermediate_size=384, num_hidden_layers=2, num_attention_heads=2,
                             num_key_value_heads=1, max_position_embeddings=4096, rms_norm_eps=1e-6, rope_theta=1000000.0,
                             use_sliding_window=False, attention_dropout=0.0)
    cfg.rope_theta = 1000000.0
    cfg._attn_implementation = "eager"
    for k, v in dict(world_position_embedding_type="avg-discrete-sin3d", mm_spatial_pool_mode="bilinear", mm_spatial_pool_stride=2,
                     voxel_size=0.1, min_xyz_range=[-15, -15, -5], max_xyz_range=[15, 15, 5], mm_patch_merge_type="spatial_unpad",
                     mm_newline_position="grid", ground_head_type="infonce", ground_head_temperature=0.07,
                     object_feature_type="patch14-pe", ground_token_ids=[318], coord_token_ids=[317]).items():
        setattr(cfg, k, v)
    torch.manual_seed(71)
    m = lq.LlavaQwenForCausalLM(cfg)
    vcfg = se.SigLipVisionConfig(hidden_size=VH, intermediate_size=272, num_hidden_layers=3, num_attention_heads=2, image_size=384, patch_size=14)
    tower = object.__new__(se.SigLipVisionTower)          # the class's forward/properties, without its hub download
    torch.nn.Module.__init__(tower)
    tower.is_loaded, tower.config, tower.vision_tower_name = True, vcfg, "tiny-random"
    tower.vision_tower = se.SigLipVisionModel(vcfg)
    del tower.vision_tower.vision_model.encoder.layers[-1:]          # load_model, siglip_encoder.py:570-571
    tower.vision_tower.vision_model.head = torch.nn.Identity()
    m.model.vision_tower = tower
    m.model.mm_projector = build_vision_projector(types.SimpleNamespace(mm_projector_type="mlp2x_gelu", mm_hidden_size=VH, hidden_size=H))
    m.model.image_newline = torch.nn.Parameter(torch.zeros(H))
    m.eval()
    g = torch.Generator().manual_seed(72)
    with torch.no_grad():
        for n, p_ in m.named_parameters():
            r = torch.randn(p_.shape, generator=g)
            if ("norm" in n and n.endswith("weight")) or re.search(r"ground_head_(obj|query)\.2\.weight", n):
                v = 1.0 + 0.1 * r
            elif n.endswith("bias"):
                v = 0.1 * r
            elif "ground_head_zero_target" in n:
                v = r
            else:
                v = 0.05 * r
            p_.copy_(v.to(torch.bfloat16).float())                 # bf16-representable: one stored array serves every dtype
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()
          if "post_layernorm" not in k and "position_ids" not in k and "inv_freq" not in k}
    inv = {k: v.detach().clone() for k, v in m.named_buffers() if "inv_freq" in k}

    def up8(x, dims):                                              # low-res -> 384 by 8x repetition (the stored form)
        for d in dims:
            x = x.repeat_interleave(8, dim=d)
        return x

    out = {}
    cases = {"F2": (2, ("f32", "bf16", "f16")), "F8": (8, ("f16",))}
    DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}
    for case, (Fr, kinds) in cases.items():
        img_lo = torch.randn(Fr, 3, 48, 48, generator=g).half().float()
        wc_lo = ((torch.rand(Fr, 48, 48, 3, generator=g) - 0.5) * torch.tensor([14.0, 14.0, 5.0])).half().float()
        boxes = torch.cat([(torch.rand(5, 3, generator=g) - 0.5) * torch.tensor([8.0, 8.0, 2.0]), torch.rand(5, 3, generator=g) * 5 + 1.0], 1).half().float()
        boxes[4] = torch.tensor([40.0, 40.0, 40.0, 0.5, 0.5, 0.5])            # selects nothing
        text = torch.randint(0, 300, (11,), generator=g)
        ids = torch.cat([text[:5], torch.tensor([-200]), text[5:]])[None]
        gids = torch.cat([text[:5], torch.tensor([-200]), text[5:9], torch.tensor([318]), text[9:]])[None]      # one <ground> label
        glabels = torch.full_like(gids, -100)
        glabels[0, 10] = 318
        cids = torch.cat([text[:5], torch.tensor([-200]), text[5:7], torch.tensor([317]), text[7:9], torch.tensor([317]), text[9:]])[None]   # two <coord> tokens
        box_in = torch.tensor([[1.23, -2.5, 0.71]]).half().float()
        images = up8(img_lo, (2, 3))[None]
        wc = up8(wc_lo, (1, 2))[None]
        out.update({f"{case}_cids": cids[0].numpy(), f"{case}_box_in": box_in.numpy(),
                    f"{case}_img_lo": img_lo.numpy(), f"{case}_wc_lo": wc_lo.numpy(), f"{case}_boxes": boxes.numpy(),
                    f"{case}_ids": ids[0].numpy(), f"{case}_gids": gids[0].numpy(), f"{case}_glabels": glabels[0].numpy()})
        for kind in kinds:
            dt = DT[kind]
            m.to(dt)
            for mod_name, b in inv.items():                       # from_pretrained(torch_dtype=...) keeps inv_freq f32 (see g_llm)
                mod = m.get_submodule(mod_name.rsplit(".", 1)[0])
                mod.inv_freq = b.float().clone()
            raw = (lambda t: t2n(t)) if dt == torch.bfloat16 else (lambda t: t.numpy())
            vd = {"world_coords": wc.to(dt), "box_input": [], "objects": boxes.to(dt)[None]}
            _, _, _, _, emb, _, _, _ = m.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, images.to(dt), ["video"], None, vd)
            logits = m(input_ids=ids, images=images.to(dt), modalities=["video"], video_dict=vd, use_cache=False).logits
            seq, toks, steps = emb, [], []
            lg = logits[0, -1]
            for _ in range(4):
                toks.append(int(torch.argmax(lg)))
                if len(toks) == 4:
                    break
                seq = torch.cat([seq, m.get_model().embed_tokens(torch.tensor([[toks[-1]]]))], 1)
                lg = m(inputs_embeds=seq, use_cache=False).logits[0, -1]
                steps.append(lg)
            _, scores = m(input_ids=gids, images=images.to(dt), modalities=["video"], video_dict=vd, labels=glabels,
                          use_object_proposals=True, use_cache=False)
            if case == "F2":     # Scan2Cap-style prompt: a <coord> token whose row gets the PE of the (discretised) input box centre
                vdc = dict(vd, box_input=box_in.to(dt))                      # llava_arch.py:416-417, 697-700; model_scan2cap.py:137-167
                _, _, _, _, emb_c, _, _, _ = m.prepare_inputs_labels_for_multimodal(cids, None, None, None, None, images.to(dt), ["video"], None, vdc)
                out[f"{case}_coord_rows_{kind}"] = raw(emb_c[0, [8 + Fr * 210 - 1, 11 + Fr * 210 - 1]])
                out[f"{case}_coord_logits_{kind}"] = m(input_ids=cids, images=images.to(dt), modalities=["video"], video_dict=vdc, use_cache=False).logits[0, -1].numpy()
            out[f"{case}_embeds_{kind}"] = raw(emb[0])
            out[f"{case}_logits_{kind}"] = logits[0, -1].numpy()          # lm_head(...).float(): f32 in every dtype
            out[f"{case}_tokens_{kind}"] = np.array(toks, np.int64)
            out[f"{case}_step_logits_{kind}"] = torch.stack(steps).numpy()
            out[f"{case}_scores_{kind}"] = raw(scores)
            m.float()
    save("tiny_model", **{"w." + k: _bits(v) for k, v in sd.items()}, **out)


def g_world_coords():
    """a4: VideoProcessor.calculate_world_coords (video_utils.py:196-238) on files written here the way the dataset holds
    them (16-bit depth PNGs, 4x4 pose text files, an EmbodiedScan-style scene record): PIL u16 -> int32 -> f32, np.loadtxt,
    axis_align @ pose in f64 then .float(), unproject; plus the `boundry` expression of preprocess (:268-273) on that output.
    The file CONTENTS are stored (depth array, pose values as f64: np.savetxt's default %.18e round-trips them), the test
    writes them back to disk and runs the mirror's loader."""
    import tempfile
    from PIL import Image
    from llava.video_utils import VideoProcessor
    rng = np.random.default_rng(81)
    V, H, W = 3, 48, 64
    depth = rng.integers(300, 6000, size=(V, H, W)).astype(np.uint16)
    depth[1, :5, :7] = 0                                            # holes in the depth map
    depth[2, 10, 10] = 40000                                        # above int16 range: the u16 -> int32 conversion must not wrap
    poses = np.zeros((V, 4, 4))
    for v in range(V):
        a, b = 0.4 * v + 0.1, 0.3 - 0.25 * v
        Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
        Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
        poses[v, :3, :3] = Rz @ Ry
        poses[v, :3, 3] = rng.normal(size=3) * 1.5
        poses[v, 3, 3] = 1
    t = 0.7
    align = np.array([[np.cos(t), np.sin(t), 0, -1.25], [-np.sin(t), np.cos(t), 0, 2.5], [0, 0, 1, -0.0625], [0, 0, 0, 1]])
    K = np.array([[57.787, 0, 31.5, 0], [0, 57.9, 23.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    with tempfile.TemporaryDirectory() as d:
        files = []
        for v in range(V):
            base = os.path.join(d, f"{v * 10:05d}")
            Image.fromarray(depth[v]).save(base + ".png")
            np.savetxt(base + ".txt", poses[v])
            files.append(base + ".jpg")
        vp = object.__new__(VideoProcessor)
        vp.scene = {"scannet/scene0000_00": {"axis_align_matrix": align.tolist(), "depth_cam2img": K.tolist()}}
        wc = vp.calculate_world_coords("scannet/scene0000_00", files)["world_coords"]
    flat = wc.reshape(-1, 3)
    boundry = torch.tensor([flat[:, 0].min().item(), flat[:, 0].max().item(), flat[:, 1].min().item(), flat[:, 1].max().item(),
                            flat[:, 2].min().item(), flat[:, 2].max().item()])
    save("world_coords", depth=depth, poses=poses, axis_align=align, cam2img=K, world=wc.numpy(), boundry=boundry.numpy(),
         aligned_f32=torch.stack([torch.from_numpy(align) @ torch.from_numpy(p_) for p_ in poses]).float().numpy())


def g_ground_variants():
    """The grounding variants the shipped scripts never select: object_feature_type 'patch27' (llava_arch.py:367-371, 485-486:
    the reference's own lines, executed as in g_objects) and ground_head_type 'mlp' / 'score' (llava_qwen.py:57-91, 283-292: the
    reference's own LlavaQwenForCausalLM.predict_box, its decoder replaced by a stub that returns a stored hidden state).  The
    heads' weights come from seeds (oracle/llm_oracle.py seeded_ground_head); the fixture keeps a checksum of them."""
    import textwrap
    from llava.model.language_model import llava_qwen as lq
    from llava.model.position_encoding import PositionEmbeddingSine3D
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from llm_oracle import seeded_ground_head
    with open(REF + "/llava/model/llava_arch.py") as f:
        lines = f.read().split("\n")
    a0 = next(i for i, l in enumerate(lines) if 'object_boxes = video_dict["objects"][0]' in l)
    a1 = next(i for i, l in enumerate(lines) if "use_mrope_position_embedding = False" in l)
    b0 = next(i for i, l in enumerate(lines) if "object_features = []" in l and i > a1)
    b1 = next(i for i, l in enumerate(lines) if "object_features += box_center_features" in l) + 1
    blk_a = textwrap.dedent("\n".join(lines[a0:a1]))
    blk_b = textwrap.dedent("\n".join(lines[b0:b1]))
    C, Fr = 96, 2
    g = torch.Generator().manual_seed(61)
    m = make_arch(dict(DEFAULT_CFG, object_feature_type="patch27-pe"))
    m.model.world_position_embedding = PositionEmbeddingSine3D(C)
    out = {}
    lo = (torch.rand(Fr, 48, 48, 3, generator=g) - 0.5) * torch.tensor([8.0, 8.0, 3.0])
    coords = lo.repeat_interleave(8, 1).repeat_interleave(8, 2).half().float()            # 8 x 8 constant regions (stored low-res)
    boxes = torch.cat([(torch.rand(7, 3, generator=g) - 0.5) * torch.tensor([6.0, 6.0, 2.0]), torch.rand(7, 3, generator=g) * 5 + 1.5], 1)
    boxes[6] = torch.tensor([50.0, 50.0, 50.0, 0.1, 0.1, 0.1])        # selects nothing -> zero feature
    feats = torch.randn(Fr, 729, C, generator=g).half().float()      # fp16-representable: one stored array serves both dtypes
    for name, dt in (("f32", torch.float32), ("f16", torch.float16)):
        ns = dict(torch=torch, self=m, video_dict={"world_coords": coords.to(dt)[None], "objects": boxes.to(dt)[None]}, int=int)
        exec(compile(blk_a, "<llava_arch object masks>", "exec"), ns)
        centers = m.discrete_coords(ns["object_boxes_center"], None)
        enc = feats.to(dt)
        ns.update(image_features=[m.get_2dPool(enc)], encoded_image_features=[enc], use_mlp_pe=False, use_sin3d_pe=True,
                  object_boxes_center=centers)
        exec(compile(blk_b, "<llava_arch object features>", "exec"), ns)
        out["mask27_" + name] = torch.stack(ns["object_patch"]).numpy()
        out["objfeat27_" + name] = ns["object_features"].float().numpy()
    assert out["mask27_f32"].any() and not out["mask27_f32"][6].any()
    # the heads
    H, T, n_obj = 128, 12, 9
    hidden = torch.randn(1, T, H, generator=g).to(torch.bfloat16).float()
    objf = torch.randn(n_obj, H, generator=g).to(torch.bfloat16).float()
    labels = torch.full((1, T), -100)
    labels[0, 7] = 318

    class Trunk(torch.nn.Module):                  # stands in for LlavaQwenModel: predict_box only takes outputs[0] from it
        def forward(self, **kw):
            return (self.h,)

    for kind, seed in (("mlp", 811), ("score", 812)):
        cfg = lq.LlavaQwenConfig(vocab_size=320, hidden_size=H, intermediate_size=128, num_hidden_layers=1, num_attention_heads=1,
                                 num_key_value_heads=1, max_position_embeddings=64, rms_norm_eps=1e-6, rope_theta=1000000.0,
                                 use_sliding_window=False, attention_dropout=0.0)
        cfg.rope_theta = 1000000.0
        cfg._attn_implementation = "eager"
        cfg.ground_head_type, cfg.ground_head_temperature, cfg.ground_token_ids = kind, 0.07, [318]
        model = lq.LlavaQwenForCausalLM(cfg).eval()
        w = seeded_ground_head(kind, H, seed)
        missing, unexpected = model.load_state_dict(w, strict=False)
        assert not unexpected and not [k for k in missing if k.startswith("ground_head")], (missing, unexpected)
        model.model = Trunk()
        out[kind + "_seed"] = np.int64(seed)
        out[kind + "_checksum"] = np.float64(sum(float(v.double().abs().sum()) for v in w.values()))
        for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
            model.to(dt)
            model.model.h = hidden.to(dt)
            _, scores = model.predict_box(labels=labels, object_features=objf.to(dt), box_labels=None)
            out[f"{kind}_scores_{name}"] = scores.float().numpy()
        del model
    save("ground_variants", coords_lo=lo.half().numpy(), boxes=boxes.numpy(), feats=feats.half().numpy(), hidden=hidden.numpy(),
         objf=objf.numpy(), ground_row=np.int64(7), **out)


GENS = {k[2:]: v for k, v in list(globals().items()) if k.startswith("g_")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    _import_reference()
    torch.set_num_threads(4)
    for name, fn in GENS.items():
        if a.only and name not in a.only:
            continue
        print(f"[{name}]")
        with torch.no_grad():
            fn()


if __name__ == "__main__":
    main()
