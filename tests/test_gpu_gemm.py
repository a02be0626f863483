"""GPU parity tests for v3d_gemm (every nn.Linear on the path) against a plain fp32 torch
reference on the CPU with the reference's rounding points (the oracle for a floating-point kernel)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DT = {"bf16": torch.bfloat16, "f16": torch.float16}
# tolerance: K-long f32 accumulation in a different order + one 16-bit rounding per stage
RTOL = {"bf16": 1.6e-2, "f16": 2e-3}


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from v3d import ops as _ops
    return _ops


def ref_linear(a, w, bias, dt):
    y = a.float() @ w.float().t()
    if bias is not None:
        y = y + bias.float()
    return y.to(dt)


def close(got, want, kind, scale=None):
    got, want = got.float().cpu(), want.float()
    tol = RTOL[kind]
    s = want.abs().mean().item() if scale is None else scale
    err = (got - want).abs()
    assert torch.all(err <= tol * want.abs() + tol * s), f"max err {err.max().item()} (scale {s})"


@pytest.mark.parametrize("kind", ["bf16", "f16"])
@pytest.mark.parametrize("M,N,K", [(300, 256, 192), (128, 128, 64), (1, 128, 64), (7, 384, 1152), (729, 1152, 640), (1000, 3584, 1152)])
def test_gemm_plain_and_bias(ops, kind, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N)
    dt = DT[kind]
    a = (torch.randn(M, K, generator=g) * 0.5).to(dt)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dt)
    b = torch.randn(N, generator=g).to(dt)
    close(ops.gemm(a.cuda(), w.cuda()), ref_linear(a, w, None, dt), kind)
    close(ops.gemm(a.cuda(), w.cuda(), bias=b.cuda(), epilogue=ops.EPI_BIAS), ref_linear(a, w, b, dt), kind)


@pytest.mark.parametrize("kind", ["bf16", "f16"])
@pytest.mark.parametrize("M", [5, 333])
def test_gemm_epilogues(ops, kind, M):
    g = torch.Generator().manual_seed(M)
    dt = DT[kind]
    N, K = 256, 320
    a = (torch.randn(M, K, generator=g) * 0.5).to(dt)
    w = (torch.randn(N, K, generator=g) * 0.08).to(dt)
    b = torch.randn(N, generator=g).to(dt)
    r = torch.randn(M, N, generator=g).to(dt)
    lin = ref_linear(a, w, b, dt)
    ac, wc, bc, rc = a.cuda(), w.cuda(), b.cuda(), r.cuda()
    close(ops.gemm(ac, wc, bias=bc, epilogue=ops.EPI_BIAS_GELU_ERF), torch.nn.functional.gelu(lin.float()).to(dt), kind, scale=1.0)
    close(ops.gemm(ac, wc, bias=bc, epilogue=ops.EPI_BIAS_GELU_TANH),
          torch.nn.functional.gelu(lin.float(), approximate="tanh").to(dt), kind, scale=1.0)
    close(ops.gemm(ac, wc, bias=bc, res=rc, epilogue=ops.EPI_BIAS_RES), (r.float() + lin.float()).to(dt), kind, scale=1.0)
    close(ops.gemm(ac, wc, res=rc, epilogue=ops.EPI_RES), (r.float() + ref_linear(a, w, None, dt).float()).to(dt), kind, scale=1.0)
    # residual broadcast by row modulo (ViT position embedding): res has 3 rows
    r3 = torch.randn(3, N, generator=g).to(dt)
    want = (r3.float()[torch.arange(M) % 3] + lin.float()).to(dt)
    close(ops.gemm(ac, wc, bias=bc, res=r3.cuda(), epilogue=ops.EPI_BIAS_RES, res_mod=3), want, kind, scale=1.0)


@pytest.mark.parametrize("kind", ["bf16", "f16"])
@pytest.mark.parametrize("M", [3, 200])
def test_gemm_swiglu(ops, kind, M):
    """Qwen2MLP gate/up (modeling_qwen2.py:188-189): act(gate(x)) * up(x), each stage rounded to dtype."""
    g = torch.Generator().manual_seed(M + 1)
    dt = DT[kind]
    I, K = 192, 256
    a = (torch.randn(M, K, generator=g) * 0.5).to(dt)
    wg = (torch.randn(I, K, generator=g) * 0.1).to(dt)
    wu = (torch.randn(I, K, generator=g) * 0.1).to(dt)
    gate = ref_linear(a, wg, None, dt)
    up = ref_linear(a, wu, None, dt)
    want = (torch.nn.functional.silu(gate.float()).to(dt).float() * up.float()).to(dt)
    wgu = ops.interleave_gate_up(wg, wu)
    # N must be a multiple of 128: 2*192 = 384 ok
    got = ops.gemm(a.cuda(), wgu.cuda(), epilogue=ops.EPI_SWIGLU)
    assert got.shape == (M, I)
    close(got, want, kind, scale=0.5)


def test_gemm_strided_operands_and_out(ops):
    """A with a padded row stride (K-padding of SigLIP's 4304-wide MLP) and out into a wider buffer."""
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(9)
    M, N, K = 150, 128, 128
    abuf = torch.zeros(M, K + 64, dtype=dt)
    abuf[:, :K] = (torch.randn(M, K, generator=g) * 0.5).to(dt)
    w = (torch.randn(N, K, generator=g) * 0.1).to(dt)
    ac = abuf.cuda()
    obuf = torch.full((M, N + 128), 7.0, dtype=dt, device="cuda")
    ops.gemm(ac[:, :K], w.cuda(), out=obuf[:, :N])
    close(obuf[:, :N], ref_linear(abuf[:, :K], w, None, dt), "bf16")
    assert torch.all(obuf[:, N:] == 7.0)


def test_gemm_rejects_bad_shapes(ops):
    from v3d import V3DError
    a = torch.zeros(4, 100, dtype=torch.bfloat16, device="cuda")
    w = torch.zeros(128, 100, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(V3DError, match="multiple"):
        ops.gemm(a, w)


@pytest.mark.parametrize("kind", ["bf16", "f16"])
@pytest.mark.parametrize("M,N,K,epi", [(256, 256, 128, "none"), (300, 512, 192, "bias"), (1000, 768, 1152, "res"), (513, 256, 640, "gelu"),
                                       (6794, 4608, 3584, "bias"), (2000, 1024, 256, "swiglu"), (6794, 3584, 3584, "res")])
def test_gemm_pingpong_tile_equals_v3_tile_bitwise(ops, kind, M, N, K, epi, monkeypatch):
    """The ping-pong schedule of the 256 x 256 tile (gemm256pp_kernel: half-tile ring, counted vmcnt, staggered wave rows)
    accumulates every output element in the same k order as the v3 kernel, so the two must agree BIT FOR BIT on every shape
    (a stale or early LDS read shows up as a difference); the v3 kernel itself is checked against the f32 reference above.
    K from 2 to 56 K-steps covers the prologue / second-last / last K-step code paths."""
    g = torch.Generator().manual_seed(M + N + K)
    dt = DT[kind]
    a = (torch.randn(M, K, generator=g) * 0.5).to(dt).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).to(dt).cuda()
    b = torch.randn(N, generator=g).to(dt).cuda()
    r = torch.randn(M, N, generator=g).to(dt).cuda()
    kw = {"none": dict(), "bias": dict(bias=b, epilogue=ops.EPI_BIAS), "res": dict(res=r, epilogue=ops.EPI_RES),
          "gelu": dict(bias=b, epilogue=ops.EPI_BIAS_GELU_TANH), "swiglu": dict(epilogue=ops.EPI_SWIGLU)}[epi]
    outs = {}
    monkeypatch.setenv("V3D_GEMM_STREAMK", "0")           # (a tile cut in two sums k in two runs: its own test below)
    for var in ("3", "4"):                                # force the 256 x 256 / 192 x 256 tile whatever the cost model would pick
        monkeypatch.setenv("V3D_GEMM_VARIANT", var)
        for pp in ("0", "1"):
            monkeypatch.setenv("V3D_GEMM_PP", pp)
            for rep in range(3):                          # repeated launches: a race would not be stable
                out = ops.gemm(a, w, **kw)
                torch.cuda.synchronize()
                if (var, pp) in outs:
                    assert torch.equal(out, outs[var, pp]), f"variant {var} pp={pp}: launch {rep} differs from launch 0"
                outs[var, pp] = out
    # the persistent tile loop: a grid of 3 workgroups makes every workgroup walk several tiles (next-tile prefetch under the
    # epilogue, ring / C-staging reuse, counted waits restarted per tile)
    monkeypatch.setenv("V3D_GEMM_PP_GRID", "3")
    for var in ("3", "4"):
        monkeypatch.setenv("V3D_GEMM_VARIANT", var)
        for rep in range(2):
            assert torch.equal(ops.gemm(a, w, **kw), outs[var, "1"]), f"variant {var}: persistent grid of 3 differs (launch {rep})"
    monkeypatch.delenv("V3D_GEMM_PP_GRID")
    assert torch.equal(outs["3", "0"], outs["3", "1"]) and torch.equal(outs["4", "0"], outs["4", "1"])
    assert torch.equal(outs["3", "1"], outs["4", "1"])    # every tile shape sums k in the same order
    monkeypatch.setenv("V3D_GEMM_VARIANT", "1")            # ... the 128 x 128 kernel included
    assert torch.equal(ops.gemm(a, w, **kw), outs["3", "1"])
    if epi in ("none", "bias"):
        close(outs["4", "1"], ref_linear(a.cpu(), w.cpu(), b.cpu() if epi == "bias" else None, dt), kind)


@pytest.mark.parametrize("kind", ["bf16", "f16"])
@pytest.mark.parametrize("M,N,K,epi,grid", [(6794, 3584, 3584, "res", None), (6794, 3584, 2048, "bias", None), (3000, 2560, 512, "none", "16"),
                                            (1024, 2560, 512, "gelu", "16"), (2000, 4096, 768, "swiglu", "24"), (1500, 3072, 1024, "res", "32")])
def test_gemm_stream_k_tail_matches_whole_tile_walk(ops, kind, M, N, K, epi, grid, monkeypatch):
    """Split-K tail of the ping-pong kernel (V3D_GEMM_STREAMK=2: wherever legal): the tiles left after the whole-tile rounds are cut
    into 2..4 K-chunks run side by side; chunk 0 adds the others' f32 accumulators.  A cut tile is the f32 sum of 2..4 runs of k
    instead of one: an output may differ from the whole-tile walk by the f32 rounding of those partial sums, i.e. by one rounding
    of the 16-bit result - never more - and repeated launches agree bit for bit (fixed cuts, fixed order).  Small grids
    (V3D_GEMM_PP_GRID) put whole-tile rounds, 2- / 3- / 4-way cuts (first, middle, last chunks) and the flag exchange into small
    shapes; the first two cases are the decoder's o_proj shape on the full chip."""
    g = torch.Generator().manual_seed(M + N + K)
    dt = DT[kind]
    a = (torch.randn(M, K, generator=g) * 0.5).to(dt).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).to(dt).cuda()
    b = torch.randn(N, generator=g).to(dt).cuda()
    r = torch.randn(M, N, generator=g).to(dt).cuda()
    kw = {"none": dict(), "bias": dict(bias=b, epilogue=ops.EPI_BIAS), "res": dict(res=r, epilogue=ops.EPI_RES),
          "gelu": dict(bias=b, epilogue=ops.EPI_BIAS_GELU_TANH), "swiglu": dict(epilogue=ops.EPI_SWIGLU)}[epi]
    monkeypatch.setenv("V3D_GEMM_VARIANT", "3")
    if grid:
        monkeypatch.setenv("V3D_GEMM_PP_GRID", grid)
    monkeypatch.setenv("V3D_GEMM_STREAMK", "0")
    whole = ops.gemm(a, w, **kw)
    monkeypatch.setenv("V3D_GEMM_STREAMK", "2")
    cut = [ops.gemm(a, w, **kw) for _ in range(3)]
    torch.cuda.synchronize()
    assert torch.equal(cut[0], cut[1]) and torch.equal(cut[0], cut[2])
    assert not torch.equal(cut[0], whole) or K <= 128, "stream-K did not engage (outputs identical to the whole-tile walk)"
    ulp = 2.0 ** (-7 if kind == "bf16" else -10)
    wf = whole.float()
    d = (cut[0].float() - wf).abs()
    tiny = 4e-6 * (a.float().abs().mean() * w.float().abs().mean() * K).item() ** 0.5 * K ** 0.25   # f32 rounding of the partial sums (~|sum| 2^-22)
    tiny = max(tiny, 1e-5)
    if epi in ("none", "bias"):
        bound = 1.01 * ulp * wf.abs() + tiny                                 # one rounding of the 16-bit result
    elif epi == "res":
        bound = 1.01 * ulp * ((wf - r.float()).abs() + wf.abs()) + tiny      # the product is rounded, then product + residual
    else:
        bound = 4 * ulp * (wf.abs() + wf.abs().mean())                       # the activation sees a pre-activation one ulp away
    assert bool((d <= bound + 1e-30).all()), f"max diff {d.max().item()}"
    assert float((d > 0).float().mean()) < 0.2            # and only a minority of outputs moves at all
    if epi in ("none", "bias"):
        close(cut[0], ref_linear(a.cpu(), w.cpu(), b.cpu() if epi == "bias" else None, dt), kind)
