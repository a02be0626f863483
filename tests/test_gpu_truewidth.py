"""True-width layers through the Engine (its padded / interleaved weight layouts and its tile selection) against the CPU
oracle: ONE SigLIP layer at hidden 1152 / 16 heads x 72 / MLP 4304 (the 1280-column residual stride, the 4304 -> 4352 pad,
the 72 -> 96 head layout), the 1152 -> 3584 projector, and ONE Qwen2 layer at hidden 3584 / 28 q + 4 kv heads x 128 / MLP
18944 (the 4608-row fused QKV, tile-interleaved gate/up, GQA 28/4), in bf16 and f16 at F = 2 frames (S = 431) and in
bf16 at the full BASELINE configs[1] shape (F = 32: 23 328 ViT tokens, S = 6794), where the 256-wide GEMM tiles and the
long-sequence attention paths are the ones selected.  The oracle runs the reference's own torch ops in the same dtype
(oracle/llm_oracle.py, pinned bit-exact to the reference modules by tests/test_oracle_llm_golden.py).
Tolerances: relative L2 per stage 2e-2 (bf16) / 3e-3 (f16) as in tests/test_gpu_engine.py, and per ROW 4x that, so a
single wrong tile / row / head cannot hide in the global norm."""
import numpy as np
import pytest
import torch

from oracle import llm_oracle as L
from oracle import pipeline_oracle as PO

pytestmark = pytest.mark.gpu

OCFG = dict(layers=1, heads=28, kv_heads=4, rope_theta=1e6, eps=1e-6)


def true_cfg(max_pos):
    from v3d.engine import EngineConfig, LlmConfig, VitConfig
    return EngineConfig(vit=VitConfig(layers=1), llm=LlmConfig(layers=1, vocab=1024, max_pos=max_pos))


def rel_err(got, want):
    got, want = got.float().cpu(), want.float()
    return ((got - want).norm() / want.norm().clamp_min(1e-9)).item()


def row_err(got, want):
    got, want = got.float().cpu(), want.float()
    got, want = got.reshape(-1, got.shape[-1]), want.reshape(-1, want.shape[-1])
    return ((got - want).norm(dim=1) / want.norm(dim=1).clamp_min(1e-6)).max().item()


def _run(dt, tol, F_, n_pre, n_post, max_pos):
    from v3d.engine import Engine, random_state_dict
    cfg = true_cfg(max_pos)
    assert (cfg.vit.hidden, cfg.vit.heads, cfg.vit.inter) == (1152, 16, 4304)
    assert (cfg.llm.hidden, cfg.llm.heads, cfg.llm.kv_heads, cfg.llm.inter) == (3584, 28, 4, 18944)
    sd = random_state_dict(cfg, torch.float32, "cpu", seed=21, std=0.02)
    eng = Engine(cfg, sd, dtype=dt, device="cuda", max_frames=F_)
    assert eng.v_Hx == 1280 and eng.v_Ip == 4352 and eng.l_nqkv == 4608           # the layouts this test is about
    g = torch.Generator().manual_seed(22)
    images = torch.randn(F_, 3, 384, 384, generator=g)
    coords = (torch.rand(F_, 384, 384, 3, generator=g) - 0.5) * torch.tensor([30.0, 30.0, 10.0])
    t = torch.randint(0, 1024, (n_pre + n_post,), generator=g)
    input_ids = torch.cat([t[:n_pre], torch.tensor([PO.IMAGE_TOKEN_INDEX]), t[n_pre:]])

    w = {k: v.to(dt) for k, v in sd.items()}
    tower = L.siglip_tower(images.to(dt), w, 1, 16)
    feats_ref = L.projector(tower, w)
    ids_ref, vis = PO.visual_sequence(w, coords.to(dt), feats_ref, dt)
    x_ref = PO.inputs_embeds(w, input_ids, vis, dt)
    S = x_ref.shape[0]
    y_ref, _ = L.qwen2_layer(x_ref[None], w, "model.layers.0.", 28, 4, torch.arange(S), 1e6, 1e-6)
    last = L.rmsnorm(y_ref[:, -1:], w["model.norm.weight"], 1e-6)
    logits_ref = torch.nn.functional.linear(last, w["lm_head.weight"]).float()[0, 0]

    feats = eng.encode_images(images.cuda())
    assert rel_err(eng.vit_hidden(F_), tower) < tol and row_err(eng.vit_hidden(F_), tower) < 4 * tol
    assert rel_err(feats, feats_ref) < tol and row_err(feats, feats_ref) < 4 * tol
    vox = eng.voxel_ids(coords.to(dt).cuda())
    assert np.array_equal(vox.cpu().numpy(), ids_ref)
    x = eng.build_inputs_embeds(input_ids, feats, vox)
    assert x.shape[0] == S
    assert rel_err(x, x_ref) < tol and row_err(x, x_ref) < 4 * tol
    logits = eng.llm_forward(x, 0)
    y = eng.l_x[:S]                                   # the residual stream after the layer, every row
    assert rel_err(y, y_ref[0]) < 2 * tol
    assert row_err(y, y_ref[0]) < 6 * tol
    assert rel_err(logits, logits_ref) < 3 * tol
    # K/V rows appended to the cache: rotated keys and values of the layer (GQA: 4 kv heads x 128)
    h = L.rmsnorm(x_ref[None], w["model.layers.0.input_layernorm.weight"], 1e-6)
    k = torch.nn.functional.linear(h, w["model.layers.0.self_attn.k_proj.weight"], w["model.layers.0.self_attn.k_proj.bias"])
    v = torch.nn.functional.linear(h, w["model.layers.0.self_attn.v_proj.weight"], w["model.layers.0.self_attn.v_proj.bias"])
    cos, sin = L.rotary_cos_sin(torch.arange(S), 128, 1e6, dt)
    kh = k.view(1, S, 4, 128).transpose(1, 2)
    kr = (kh * cos[None, None]) + (L.rotate_half(kh) * sin[None, None])
    kv = eng.kv[0][:S]
    assert rel_err(kv[:, :512], kr.transpose(1, 2).reshape(S, 512)) < tol
    assert rel_err(kv[:, 512:], v[0]) < tol
    return eng


@pytest.mark.parametrize("dt,tol", [(torch.bfloat16, 2e-2), (torch.float16, 3e-3)])
def test_true_width_layers_two_frames(dt, tol):
    _run(dt, tol, F_=2, n_pre=5, n_post=6, max_pos=1024)


def test_true_width_layers_full_sequence_bf16():
    """BASELINE configs[1] shape: 32 frames, S = 14 + 6720 + 60 = 6794."""
    _run(torch.bfloat16, 2e-2, F_=32, n_pre=14, n_post=60, max_pos=8192)


def test_true_width_grounding_full_size_bf16():
    """BASELINE configs[2] at its full size on one GPU (r03): Engine.ground_scores with 32 frames and 50 object
    proposals through ONE true-width SigLIP layer + ONE true-width Qwen2 layer against oracle/pipeline_oracle.scene_ground
    (llava_arch.py:351-376, 479-501; llava_qwen.py:280-300): patch masks bit for bit, object features per row, cosine scores."""
    from v3d import ops
    from v3d.engine import Engine, random_state_dict
    dt, tol, F_ = torch.bfloat16, 2e-2, 32
    cfg = true_cfg(8192)
    sd = random_state_dict(cfg, torch.float32, "cpu", seed=31, std=0.02, ground_head=True)
    eng = Engine(cfg, sd, dtype=dt, device="cuda", max_frames=F_)
    g = torch.Generator().manual_seed(32)
    images = torch.randn(F_, 3, 384, 384, generator=g)
    coords = ((torch.rand(F_, 48, 1, 48, 1, 3, generator=g) - 0.5).expand(F_, 48, 8, 48, 8, 3).reshape(F_, 384, 384, 3) * torch.tensor([8.0, 8.0, 3.0])).contiguous()
    boxes = torch.cat([(torch.rand(50, 3, generator=g) - 0.5) * torch.tensor([6.0, 6.0, 2.0]), torch.rand(50, 3, generator=g) * 4 + 0.5], 1)
    t = torch.randint(0, 1024, (74,), generator=g)
    input_ids = torch.cat([t[:14], torch.tensor([PO.IMAGE_TOKEN_INDEX]), t[14:]])
    gidx = 14 + 1 + 40                                     # the <ground> label position, after the image token
    ocfg = dict(OCFG, vit_layers=1, vit_heads=16)
    want = PO.scene_ground(sd, ocfg, input_ids, gidx, images, coords, boxes, dt)
    got = eng.ground_scores(input_ids, gidx, images.cuda(), coords.cuda(), boxes)
    m = ops.object_patch_mask(coords.to(dt).cuda(), boxes.to(dt).cuda())
    assert np.array_equal(m.cpu().numpy().astype(bool), want["masks"].numpy())
    assert int(want["masks"].reshape(50, -1).any(1).sum()) >= 25            # the proposals do select patches
    feats = eng.feat[: F_ * 729].view(F_, 729, -1)
    objf = eng.object_features(feats, coords.to(dt).cuda(), boxes.to(dt).cuda())
    assert rel_err(objf, want["objf"]) < tol and row_err(objf, want["objf"]) < 4 * tol
    assert got.shape == (51,)
    assert (got.float().cpu() - want["scores"].float()).abs().max().item() < tol      # cosine scores in [-1, 1]: absolute
