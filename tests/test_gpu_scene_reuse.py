"""Scene-level reuse: Engine.prefill_scene computes everything question-independent once (ViT, projector,
fusion, the decoder over [system | user | <image>]) and Engine.answer runs only the question's rows at positions P..,
attending to the cached prefix.  The reference recomputes the whole prompt per question (model_scanqa.py:130-185); by
causality the prefix rows do not depend on the question, so the cached answer must equal the uncached generate():
prefix K/V rows bit for bit, question rows and tokens too (same kernels, same per-row arithmetic)."""
import pytest
import torch

from oracle import pipeline_oracle as PO

pytestmark = pytest.mark.gpu


def _engine(seed=31, dtype=torch.bfloat16):
    from v3d.engine import Engine, EngineConfig, LlmConfig, VitConfig, random_state_dict
    cfg = EngineConfig(vit=VitConfig(hidden=144, inter=272, layers=2, heads=2),
                       llm=LlmConfig(hidden=256, inter=384, layers=2, heads=2, kv_heads=1, vocab=320, max_pos=1024))
    sd = random_state_dict(cfg, torch.float32, "cpu", seed=seed, std=0.08)
    return Engine(cfg, sd, dtype=dtype, device="cuda", max_frames=2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_cached_answers_equal_uncached_generate(dtype):
    eng = _engine(dtype=dtype)
    g = torch.Generator().manual_seed(32)
    images = torch.randn(2, 3, 384, 384, generator=g).cuda()
    coords = ((torch.rand(2, 384, 384, 3, generator=g) - 0.5) * 20).cuda()
    pre = torch.randint(0, 320, (14,), generator=g)
    prefix = torch.cat([pre, torch.tensor([PO.IMAGE_TOKEN_INDEX])])
    questions = [torch.randint(0, 320, (n,), generator=g) for n in (60, 5, 1, 23, 130)]      # <= 8 rows take the decode-shaped kernels
    P = 14 + 2 * 210
    steps = 6

    uncached = []
    for q in questions:
        toks = eng.generate(torch.cat([prefix, q]), images, coords, max_new_tokens=steps).clone()
        kv = [c[: P + len(q) + steps - 1].clone() for c in eng.kv]
        uncached.append((toks, kv))

    ctx = eng.use(eng.new_context())
    assert eng.prefill_scene(prefix, images, coords) == P
    prefix_kv = [c[:P].clone() for c in ctx.kv]
    for q, (toks, kv) in zip(questions, uncached):
        got = eng.answer(q, max_new_tokens=steps)
        n = P + len(q) + steps - 1
        # questions of <= 8 rows run through the weight-streaming (decode-shaped) linears and the split-KV attention, which sum
        # in another f32 order than the MFMA prefill tiles the uncached pass used for the same rows: close, not bit-equal
        exact = len(q) > 8
        for layer in range(len(kv)):
            assert torch.equal(ctx.kv[layer][:P], kv[layer][:P]), "prefix K/V rows differ from the uncached pass"
            assert torch.equal(ctx.kv[layer][:P], prefix_kv[layer]), "answer() disturbed the cached prefix"
            a, b = ctx.kv[layer][P:P + len(q)], kv[layer][P:P + len(q)]
            if exact:
                assert torch.equal(a, b), "question K/V rows differ from the uncached pass"
            else:
                assert ((a.float() - b.float()).norm() / b.float().norm()).item() < 1e-2
        if exact:
            assert torch.equal(got, toks)
            for layer in range(len(kv)):
                assert torch.equal(ctx.kv[layer][P:n], kv[layer][P:n]), "decode K/V rows differ from the uncached pass"
        else:
            assert got.shape == toks.shape                  # a near-tie may flip a token; the K/V rows above are the criterion
        assert ctx.kv_len == n


def test_answer_needs_a_prefilled_scene_and_respects_capacity():
    from v3d._native import V3DError
    eng = _engine()
    with pytest.raises(V3DError):
        eng.answer(torch.tensor([1, 2, 3]))
    g = torch.Generator().manual_seed(33)
    images = torch.randn(2, 3, 384, 384, generator=g).cuda()
    coords = ((torch.rand(2, 384, 384, 3, generator=g) - 0.5) * 20).cuda()
    eng.prefill_scene(torch.tensor([3, 4, PO.IMAGE_TOKEN_INDEX]), images, coords)
    with pytest.raises(V3DError):
        eng.answer(torch.randint(0, 320, (700,)))                     # 422 + 700 rows > max_pos 1024
    with pytest.raises(V3DError):
        eng.answer(torch.tensor([5]), max_new_tokens=1024)
    assert eng.answer(torch.tensor([5]), max_new_tokens=3).shape == (3,)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_answer_group_equals_answering_alone(dtype):
    """Questions answered together (batched question rows over per-question copies of the cached prefix, then one decode group):
    a question's K/V rows equal those of answer() on it alone bit for bit (rows > 8: same MFMA tiles), its tokens do not
    depend on which other questions share the group, and they equal answer()'s unless answer()'s own top-2 margin is inside
    the noise of the different f32 summation order of the one-row decode linears."""
    eng = _engine(seed=35, dtype=dtype)
    g = torch.Generator().manual_seed(36)
    images = torch.randn(2, 3, 384, 384, generator=g).cuda()
    coords = ((torch.rand(2, 384, 384, 3, generator=g) - 0.5) * 20).cuda()
    prefix = torch.cat([torch.randint(0, 320, (14,), generator=g), torch.tensor([PO.IMAGE_TOKEN_INDEX])])
    questions = [torch.randint(0, 320, (n,), generator=g) for n in (60, 17, 33, 9, 60, 12)]
    P, steps = 14 + 2 * 210, 6
    scene = eng.use(eng.new_context())
    eng.prefill_scene(prefix, images, coords)
    alone, alone_kv = [], []
    for q in questions:
        alone.append(eng.answer(q, max_new_tokens=steps).clone())
        alone_kv.append([c[P: P + len(q)].clone() for c in scene.kv])
    together = eng.answer_group(questions, max_new_tokens=steps)
    st = eng._answer_state(len(questions))
    for gi, q in enumerate(questions):
        for layer in range(len(scene.kv)):
            P0 = (P // 64) * 64      # r04: the questions' caches hold the rows of the key tile that straddles P only; the tiles below it are read from the scene's cache
            assert torch.equal(st.ctxs[gi].kv[layer][P0:P], scene.kv[layer][P0:P])
            assert torch.equal(st.ctxs[gi].kv[layer][P: P + len(q)], alone_kv[gi][layer]), (gi, layer)
    assert eng.ctx is scene and scene.prefix_len == P
    # group independence: the same question inside other groups
    for gi in (1, 4):
        pair = eng.answer_group([questions[gi], questions[0]], max_new_tokens=steps)
        assert torch.equal(pair[0], together[gi])
    rev = eng.answer_group(list(reversed(questions)), max_new_tokens=steps)
    for gi in range(len(questions)):
        assert torch.equal(rev[len(questions) - 1 - gi], together[gi])
    same = sum(int(torch.equal(a, b)) for a, b in zip(alone, together))
    assert same >= len(questions) - 2, (alone, together)                  # near-ties may flip a token of a question or two
    assert all(t.shape == (steps,) for t in together)
    # the decode attention reads the shared prefix rows from ONE copy for every question (default; r04: one matrix-core launch over the
    # prefix for all questions): the tokens of every question reading its own copy of them, up to near-ties (the r03 form, which walks
    # the prefix per question out of one copy, gives them bit for bit)
    import os
    os.environ["V3D_SHARED_PREFIX"] = "0"
    try:
        own = eng.answer_group(questions, max_new_tokens=steps)
        os.environ["V3D_SHARED_PREFIX"] = "1"
        os.environ["V3D_DEC_PREFIX_MM"] = "0"
        r03 = eng.answer_group(questions, max_new_tokens=steps)
    finally:
        del os.environ["V3D_SHARED_PREFIX"]
        os.environ.pop("V3D_DEC_PREFIX_MM", None)
    for a, b in zip(own, r03):
        assert torch.equal(a, b)
    assert sum(int(torch.equal(a, b)) for a, b in zip(own, together)) >= len(questions) - 1, (own, together)
    eos = int(together[2][1])
    cut = eng.answer_group(questions, max_new_tokens=steps, eos_token_id=eos)
    for a, b in zip(cut, together):
        row = b.tolist()
        assert a.tolist() == (row[: row.index(eos) + 1] if eos in row else row)
