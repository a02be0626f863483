"""The engine against the REFERENCE's own whole model (tests/golden/tiny_model.npz: a random-init tiny
LlavaQwenForCausalLM run through prepare_inputs_labels_for_multimodal + Qwen2ForCausalLM.forward on CPU by
oracle/gen_golden.py g_tiny_model): inputs_embeds, last-row logits, greedy continuation and infonce grounding scores.
F = 2 in bf16 / f16 and F = 8 in f16 (BASELINE configs[0]: uniform 8 frames; f16 is the eval dtype, builder.py:27).
Tolerances (relative L2 per stage, as tests/test_gpu_engine.py): 2e-2 bf16 / 3e-3 f16, x3 on the logits."""
import numpy as np
import pytest
import torch

import tiny_model_fixture as TM

pytestmark = pytest.mark.gpu


def rel_err(got, want):
    got, want = got.float().cpu(), want.float()
    return ((got - want).norm() / want.norm().clamp_min(1e-9)).item()


@pytest.fixture(scope="module")
def g():
    return TM.load()


def _engine(g, dt, frames):
    from v3d.engine import Engine
    return Engine(TM.engine_config(), TM.state_dict(g), dtype=dt, device="cuda", max_frames=frames)


@pytest.mark.parametrize("case,kind,tol", [("F2", "bf16", 2e-2), ("F2", "f16", 3e-3), ("F8", "f16", 3e-3)])
def test_engine_prefill_and_greedy_tokens_match_reference_model(g, case, kind, tol):
    dt = TM.DT[kind]
    inp, want = TM.case_inputs(g, case), TM.expected(g, case, kind)
    F_ = inp["images"].shape[0]
    eng = _engine(g, dt, F_)
    feats = eng.encode_images(inp["images"].cuda())
    vox = eng.voxel_ids(inp["world_coords"].to(dt).cuda())
    x = eng.build_inputs_embeds(inp["ids"], feats, vox)
    assert tuple(x.shape) == tuple(want["embeds"].shape) == (11 + F_ * 210, 256)
    assert rel_err(x, want["embeds"]) < tol
    n_pre = int((inp["ids"] == -200).nonzero()[0])
    assert torch.equal(x[:n_pre].float().cpu(), want["embeds"][:n_pre])            # text rows: pure gathers, bit-exact
    assert torch.equal(x[-(len(inp["ids"]) - n_pre - 1):].float().cpu(), want["embeds"][-(len(inp["ids"]) - n_pre - 1):])
    logits = eng.llm_forward(x, 0)
    assert rel_err(logits, want["logits"]) < 3 * tol
    # greedy continuation through the KV cache vs the reference's re-forward of the extended sequence
    toks = eng.generate(inp["ids"], inp["images"].cuda(), inp["world_coords"].cuda(), max_new_tokens=4).tolist()
    ref_steps = [want["logits"]] + list(want["step_logits"])
    for i, (a, b) in enumerate(zip(toks, want["tokens"])):
        if a != b:
            top2 = torch.topk(ref_steps[i], 2).values
            assert (top2[0] - top2[1]).item() < 0.05 * ref_steps[i].abs().max().item(), f"token {i}: {a} != {b}"
            break
    else:
        assert toks == want["tokens"]


@pytest.mark.parametrize("kind,tol", [("bf16", 4e-2), ("f16", 6e-3)])
def test_engine_grounding_scores_match_reference_model(g, kind, tol):
    """llava_qwen.forward(use_object_proposals=True) -> predict_box (infonce) of the reference on the same weights."""
    dt = TM.DT[kind]
    inp, want = TM.case_inputs(g, "F2"), TM.expected(g, "F2", kind)
    eng = _engine(g, dt, 2)
    gi = int((inp["glabels"] == TM.GROUND_TOKEN).nonzero()[0])
    got = eng.ground_scores(inp["gids"], gi, inp["images"].cuda(), inp["world_coords"].cuda(), inp["boxes"])
    assert got.shape == (6,)
    assert float((got.float().cpu() - want["scores"]).abs().max()) < tol         # cosine scores: absolute tolerance


@pytest.mark.parametrize("kind,tol", [("bf16", 2e-2), ("f16", 3e-3)])
def test_engine_coord_token_rows_match_reference_model(g, kind, tol):
    """Scan2Cap prompt: <coord> token rows = embedding + PE(discretised box_input centre) (llava_arch.py:416-417, 697-700)."""
    dt = TM.DT[kind]
    inp = TM.case_inputs(g, "F2")
    cids, box = torch.from_numpy(g["F2_cids"]), torch.from_numpy(g["F2_box_in"])
    eng = _engine(g, dt, 2)
    feats = eng.encode_images(inp["images"].cuda())
    vox = eng.voxel_ids(inp["world_coords"].to(dt).cuda())
    x = eng.build_inputs_embeds(cids, feats, vox, box_input=box, coord_token_id=317)
    want = TM.bits_to_f32(g["F2_coord_rows_bf16"]) if kind == "bf16" else torch.from_numpy(g[f"F2_coord_rows_{kind}"]).float()
    got = x[[8 + 420 - 1, 11 + 420 - 1]].float().cpu()
    ulp = 2.0 ** (-7 if kind == "bf16" else -10)
    assert bool(((got - want).abs() <= ulp * (want.abs() + 1.0)).all())          # table PE within 1 ulp16, one rounded add
    logits = eng.llm_forward(x, 0)
    assert rel_err(logits, torch.from_numpy(g[f"F2_coord_logits_{kind}"])) < 3 * tol
