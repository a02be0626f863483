"""Pins oracle/llm_oracle.py (torch restatement of the Qwen2 / SigLIP / projector modules) against
golden outputs of the reference's own modules (oracle/gen_golden.py g_llm, g_vit).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import llm_oracle as L
from oracle import v3d_oracle as O


def load_w(g, prefix, dt):
    return {k[len(prefix):]: torch.from_numpy(g[k]).to(dt) for k in g.files if k.startswith(prefix)}


def as_t(arr, name):
    if name == "bf16":
        return torch.from_numpy(O.bf16_bits_to_f32(arr))
    return torch.from_numpy(arr)


@pytest.mark.parametrize("name,dt", [("f32", torch.float32), ("bf16", torch.bfloat16)])
def test_qwen2_layer(golden, name, dt):
    g = golden("qwen2_layer")
    w = load_w(g, "w.", dt)
    x = torch.from_numpy(g["x"]).to(dt)
    S = x.shape[1]
    pos = torch.arange(S)
    # the same torch ops in the same order: bit-exact on the same machine
    assert torch.equal(L.rmsnorm(x, w["input_layernorm.weight"]).float(), as_t(g["norm_" + name], name))
    cos, sin = L.rotary_cos_sin(pos, 128, 1000000.0, dt)
    assert torch.equal(cos.float(), as_t(g["cos_" + name], name))
    assert torch.equal(sin.float(), as_t(g["sin_" + name], name))
    h = L.rmsnorm(x, w["input_layernorm.weight"])
    assert torch.equal(L.qwen2_mlp(h, w, "mlp.").float(), as_t(g["mlp_" + name], name))
    y, _ = L.qwen2_layer(x, w, "", 2, 1, pos, 1000000.0, 1e-6)
    assert torch.equal(y.float(), as_t(g["y_" + name], name))


@pytest.mark.parametrize("name,dt", [("f32", torch.float32), ("bf16", torch.bfloat16)])
def test_siglip_layer_and_projector(golden, name, dt):
    g = golden("siglip_layer")
    w = {}
    w.update({"L." + k: v for k, v in load_w(g, "layer.", dt).items()})
    w.update({"E." + k: v for k, v in load_w(g, "emb.", dt).items()})
    w.update({"P." + k: v for k, v in load_w(g, "proj.", dt).items()})
    pix = torch.from_numpy(g["pixels"]).to(dt)
    e = L.siglip_embeddings(pix, w, "E.")
    assert torch.equal(e.float(), as_t(g["emb_" + name], name))
    y = L.siglip_layer(e, w, "L.", 2)
    assert torch.equal(y.float(), as_t(g["y_" + name], name))
    z = L.projector(y, w, "P.")
    assert torch.equal(z.float(), as_t(g["proj_" + name], name))


def test_kv_cache_decode_matches_full_prefill(golden):
    """The oracle's incremental path (past_kv) equals recomputing the whole prefix (f32)."""
    g = golden("qwen2_layer")
    w = load_w(g, "w.", torch.float32)
    x = torch.from_numpy(g["x"])[:, :40]
    full, _ = L.qwen2_layer(x, w, "", 2, 1, torch.arange(40), 1e6, 1e-6)
    pre, kv = L.qwen2_layer(x[:, :39], w, "", 2, 1, torch.arange(39), 1e6, 1e-6)
    last, _ = L.qwen2_layer(x[:, 39:], w, "", 2, 1, torch.arange(39, 40), 1e6, 1e-6, past_kv=kv)
    torch.testing.assert_close(last, full[:, 39:], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name,dt", [("f32", torch.float32), ("f16", torch.float16)])
def test_object_masks_and_features(golden, name, dt):
    """K19 against the reference's own lines (executed by the generator): masks bit-exact, features to 1 ulp16."""
    g = golden("objects")
    coords = torch.from_numpy(g["coords"]).to(dt)
    boxes = torch.from_numpy(g["boxes"]).to(dt)
    masks = L.object_patch_mask(coords, boxes)
    assert np.array_equal(masks.numpy(), g["mask_" + name])
    centres = torch.from_numpy(O.discrete_coords(boxes[:, :3].float().numpy(), name)).to(dt)
    assert np.array_equal(centres.float().numpy(), g["centers_" + name])
    pe = torch.from_numpy(O.sin3d_pe(centres.float().numpy()[None], 96, name, dim_t=g["dim_t"])[0]).to(dt)
    f = L.object_features(torch.from_numpy(g["feats"]).to(dt), masks, pe)
    np.testing.assert_allclose(f.float().numpy(), g["objfeat_" + name], rtol=0, atol=2e-6 if name == "f32" else 2e-3)


def test_infonce_scores(golden):
    g = golden("objects")
    w = {k: torch.from_numpy(g[k]) for k in g.files if k.startswith(("ho.", "hq."))}
    s = L.infonce_scores(torch.from_numpy(g["objfeat_f32"]), torch.from_numpy(g["zero_target"]), torch.from_numpy(g["query"]),
                         w, "ho.", "hq.")
    np.testing.assert_allclose(s.numpy(), g["scores"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("name,dt", [("f32", torch.float32), ("f16", torch.float16)])
def test_object_features_patch27(golden, name, dt):
    """object_feature_type 'patch27' (llava_arch.py:367-371, 485-486) against the reference's own lines: the masks over the 14 x 14
    grid of 27-pixel cells bit-exact, the means of the POOLED rows + centre PE to 1 ulp16."""
    g = golden("ground_variants")
    coords = torch.from_numpy(g["coords_lo"]).float().repeat_interleave(8, 1).repeat_interleave(8, 2).to(dt)
    boxes = torch.from_numpy(g["boxes"]).to(dt)
    masks = L.object_patch_mask(coords, boxes, cell=27, thresh_frac=0.25)
    assert masks.shape[1:] == (2, 14, 14) and np.array_equal(masks.numpy(), g["mask27_" + name])
    centres = torch.from_numpy(O.discrete_coords(boxes[:, :3].float().numpy(), name)).to(dt)
    pe = torch.from_numpy(O.sin3d_pe(centres.float().numpy()[None], 96, name)[0]).to(dt)
    pooled = torch.from_numpy(O.get_2dpool_bilinear(g["feats"].astype(np.float32), name)).to(dt)
    f = L.object_features(pooled.view(2, 196, 96), masks, pe)
    np.testing.assert_allclose(f.float().numpy(), g["objfeat27_" + name], rtol=0, atol=2e-6 if name == "f32" else 2e-3)


@pytest.mark.parametrize("kind", ["mlp", "score"])
def test_ground_head_variants(golden, kind):
    """ground_head_type 'mlp' / 'score' (llava_qwen.py:57-91, 283-292) against the reference's own predict_box on seeded weights."""
    g = golden("ground_variants")
    w = L.seeded_ground_head(kind, 128, int(g[kind + "_seed"]))
    assert abs(sum(float(v.double().abs().sum()) for v in w.values()) - float(g[kind + "_checksum"])) < 1e-6, \
        "torch's CPU generator no longer reproduces the weights the fixture's scores were made with"
    query = torch.from_numpy(g["hidden"])[0, int(g["ground_row"])][None]
    objf = torch.from_numpy(g["objf"])
    s = L.mlp_scores(objf, query, w) if kind == "mlp" else L.score_scores(objf, query, w)
    np.testing.assert_allclose(s.numpy(), g[kind + "_scores_f32"], rtol=1e-5, atol=1e-5)
