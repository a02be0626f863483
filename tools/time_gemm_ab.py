"""A/B timing of the 256-wide GEMM schedules (V3D_GEMM_PP=0 v3 one-barrier pipeline / 1 ping-pong) on the path's shapes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-3d-llm_amd"))
from v3d import ops
dt = torch.bfloat16
def run(M, N, K, epi, iters=30):
    a = torch.randn(M, K, device="cuda", dtype=dt) * 0.5
    w = torch.randn(N, K, device="cuda", dtype=dt) * 0.05
    out = torch.empty(M, N // 2 if epi == 6 else N, device="cuda", dtype=dt)
    bias = torch.zeros(N, device="cuda", dtype=dt) if epi in (1, 2, 3) else None
    res = torch.zeros(M, N, device="cuda", dtype=dt) if epi == 5 else None
    kw = {}
    if bias is not None: kw["bias"] = bias
    if res is not None: kw["res"] = res
    for _ in range(3): ops.gemm(a, w, epilogue=epi, out=out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.gemm(a, w, epilogue=epi, out=out, **kw)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
S = 6794
shapes = [("llm qkv", S, 4608, 3584, 1), ("llm o_proj", S, 3584, 3584, 5), ("llm gate_up swiglu", S, 37888, 3584, 6), ("llm down", S, 3584, 18944, 5),
          ("vit qkv", 23328, 4608, 1152, 1), ("vit fc1 gelu", 23328, 4352, 1152, 3), ("vit fc2", 23328, 1280, 4352, 4 if False else 1), ("proj1 gelu", 23328, 3584, 1152, 2),
          ("proj2", 23328, 3584, 3584, 1), ("vit out 1152", 23328, 1152, 1152, 1), ("vit out 1280", 23328, 1280, 1152, 1), ("patch embed", 23328, 1152, 640, 1),
          ("answer grp qkv", 960, 4608, 3584, 1), ("answer grp gate_up", 960, 37888, 3584, 6), ("answer grp down", 960, 3584, 18944, 5),
          ("square 4096", 4096, 4096, 4096, 0), ("square 8192", 8192, 8192, 8192, 0)]
for name, M, N, K, epi in shapes:
    row = []
    for var, skew in (("3", "0"), ("3", "8"), ("3", "16"), ("3", "32"), ("3", "64"), ("4", "0"), ("4", "16"), ("4", "32")):
        os.environ["V3D_GEMM_VARIANT"], os.environ["V3D_GEMM_PP"], os.environ["V3D_GEMM_SKEW"] = var, "1", skew
        ms = run(M, N, K, epi)
        row.append(f"{ms*1e3:7.1f}")
    print(f"{name:18s} {M:5d}x{N:5d}x{K:5d} | pp-256 skew 0/8/16/32/64: {' '.join(row[:5])} | pp-192 skew 0/16/32: {' '.join(row[5:])}", flush=True)
