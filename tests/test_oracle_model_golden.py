"""Pins oracle/pipeline_oracle.py (the composition: SigLIP tower -> projector -> 3-D fusion -> splice -> multi-layer
Qwen2 -> final norm -> LM head -> greedy continuation, and the grounding forward) against the reference's own
LlavaQwenForCausalLM run end to end (tests/golden/tiny_model.npz, generator: oracle/gen_golden.py g_tiny_model).
CPU only.  The dense stages are the same torch ops in the same order, so 16-bit results are compared bit for bit
where the op order is identical (inputs_embeds) and to rounding noise where the oracle uses its KV-cache form."""
import numpy as np
import pytest
import torch

import tiny_model_fixture as TM
from oracle import pipeline_oracle as PO


@pytest.fixture(scope="module")
def g():
    return TM.load()


@pytest.mark.parametrize("case,kind", [("F2", "f32"), ("F2", "bf16"), ("F2", "f16"), ("F8", "f16")])
def test_scene_forward_matches_reference_model(g, case, kind):
    sd, inp, want = TM.state_dict(g), TM.case_inputs(g, case), TM.expected(g, case, kind)
    dt = TM.DT[kind]
    got = PO.scene_forward(sd, TM.ORACLE_CFG, inp["ids"], inp["images"], inp["world_coords"], dt, max_new_tokens=4)
    assert got["embeds"].shape == want["embeds"].shape
    emb = got["embeds"].float()
    if kind == "f32":
        # f32: the numpy restatement of pool + PE and torch's ATen kernels order a few f32 sums differently
        assert torch.allclose(emb, want["embeds"], rtol=2e-5, atol=2e-5)
    else:
        # 16-bit: every element within 1 ulp16 of the add's operands (|PE| <= 1; the pooled value rounds once, the sum once),
        # < 0.2 % of them different at all
        ulp = 2.0 ** (-7 if kind == "bf16" else -10)
        diff = (emb - want["embeds"]).abs()
        assert bool((diff <= ulp * (want["embeds"].abs() + 1.0)).all())
        assert float((diff > 0).float().mean()) < 2e-3
    n_pre = int((inp["ids"] == -200).nonzero()[0])
    assert torch.equal(emb[:n_pre], want["embeds"][:n_pre])                 # text rows: pure gathers
    tol = dict(f32=2e-4, bf16=3e-2, f16=4e-3)[kind]
    rel = ((got["logits_last"] - want["logits"]).norm() / want["logits"].norm()).item()
    assert rel < tol, f"last-row logits rel-L2 {rel}"
    # greedy continuation: equal tokens unless the reference's own top-2 margin is inside the noise
    ref_steps = [want["logits"]] + list(want["step_logits"])
    for i, (a, b) in enumerate(zip(got["tokens"], want["tokens"])):
        if a != b:
            top2 = torch.topk(ref_steps[i], 2).values
            assert (top2[0] - top2[1]).item() < 3 * tol * ref_steps[i].abs().max().item(), f"token {i}: {a} != {b}"
            break
        if i:
            srel = ((got["step_logits"][i - 1] - ref_steps[i]).norm() / ref_steps[i].norm()).item()
            assert srel < tol, f"decode step {i} logits rel-L2 {srel} (KV-cache form vs the reference's full re-forward)"


@pytest.mark.parametrize("kind", ["f32", "bf16", "f16"])
def test_scene_ground_matches_reference_model(g, kind):
    sd, inp, want = TM.state_dict(g), TM.case_inputs(g, "F2"), TM.expected(g, "F2", kind)
    dt = TM.DT[kind]
    gi = int((inp["glabels"] == TM.GROUND_TOKEN).nonzero()[0])
    got = PO.scene_ground(sd, TM.ORACLE_CFG, inp["gids"], gi, inp["images"], inp["world_coords"], inp["boxes"], dt)
    assert got["scores"].shape == want["scores"].shape == (6,)
    tol = dict(f32=1e-4, bf16=4e-2, f16=5e-3)[kind]           # cosine scores in [-1, 1]: absolute tolerance
    assert float((got["scores"].float() - want["scores"]).abs().max()) < tol
    assert not bool(got["masks"][4].any())                       # the far-away box selects no patch


@pytest.mark.parametrize("kind", ["f32", "bf16", "f16"])
def test_coord_token_rows_match_reference_model(g, kind):
    """Scan2Cap-style prompt (model_scan2cap.py:137-167): the rows of the two <coord> tokens carry embed + PE(discretised box
    centre) (llava_arch.py:416-417, 697-700); pinned to the reference's prepare_inputs_labels_for_multimodal."""
    sd, inp = TM.state_dict(g), TM.case_inputs(g, "F2")
    dt = TM.DT[kind]
    cids, box = torch.from_numpy(g["F2_cids"]), torch.from_numpy(g["F2_box_in"])
    w = {k: v.to(dt) for k, v in sd.items()}
    from oracle import llm_oracle as L
    feats = L.projector(L.siglip_tower(inp["images"].to(dt), w, 2, 2), w)
    _, vis = PO.visual_sequence(w, inp["world_coords"].to(dt), feats, dt)
    x = PO.inputs_embeds(w, cids, vis, dt, box_input=box, coord_token_id=317)
    want = TM.bits_to_f32(g["F2_coord_rows_bf16"]) if kind == "bf16" else torch.from_numpy(g[f"F2_coord_rows_{kind}"]).float()
    got = x[[8 + 420 - 1, 11 + 420 - 1]].float()
    if kind == "f32":
        assert torch.allclose(got, want, rtol=0, atol=5e-7)         # numpy's and torch's f32 sin / cos differ in the last bit
    else:
        assert torch.equal(got, want)                               # gather + PE rounded to 16 bits + one rounded add: bit-exact
    plain = PO.inputs_embeds(w, cids, vis, dt)
    assert not torch.equal(plain[8 + 420 - 1].float(), want[0])     # the PE really was added
