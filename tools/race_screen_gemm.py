"""Race screen for the ping-pong GEMM's synchronisation structure (split-K exchange, wave-private epilogue, persistent tile loop):
many repeated launches of shapes that exercise them, interleaved with other work on the chip; every result must equal the first
bit for bit (fixed cut points, fixed summation order), and the uncut result wherever no tile is cut."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-3d-llm_amd"))
from v3d import ops
dt = torch.bfloat16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cases = [("down_proj split 2", 6794, 3584, 18944, ops.EPI_RES, "2", None), ("o_proj split 2 (forced)", 6794, 3584, 3584, ops.EPI_RES, "2", None),
         ("batch down split 4", 960, 3584, 18944, ops.EPI_RES, "2", None), ("gate/up whole tiles", 6794, 37888, 3584, ops.EPI_SWIGLU, "0", None),
         ("qkv bias grid 24 split 3", 2000, 4096, 768, ops.EPI_BIAS, "2", "24"), ("fc1 gelu grid 32 split 4", 1500, 3072, 1024, ops.EPI_BIAS_GELU_TANH, "2", "32")]
noise = torch.randn(8192, 8192, device="cuda", dtype=dt)
bad = 0
for name, M, N, K, epi, sk, grid in cases:
    g = torch.Generator().manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g) * 0.5).to(dt).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).to(dt).cuda()
    b = torch.randn(N, generator=g).to(dt).cuda()
    r = torch.randn(M, N, generator=g).to(dt).cuda()
    kw = dict(epilogue=epi)
    if epi in (ops.EPI_BIAS, ops.EPI_BIAS_GELU_TANH): kw["bias"] = b
    if epi == ops.EPI_RES: kw["res"] = r
    os.environ["V3D_GEMM_VARIANT"] = "3"
    os.environ["V3D_GEMM_STREAMK"] = sk
    if grid: os.environ["V3D_GEMM_PP_GRID"] = grid
    else: os.environ.pop("V3D_GEMM_PP_GRID", None)
    first = ops.gemm(a, w, **kw).clone()
    diff = 0
    for i in range(reps):
        if i % 7 == 3: noise.mul_(1.0001)                      # other traffic through L2 / HBM between launches
        out = ops.gemm(a, w, **kw)
        if not torch.equal(out, first): diff += 1
    torch.cuda.synchronize()
    print(f"{name:28s} {M}x{N}x{K}: {reps} launches, {diff} differ from the first", flush=True)
    bad += diff
print("RACE SCREEN", "FAILED" if bad else "clean")
sys.exit(1 if bad else 0)
