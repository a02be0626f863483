"""A/B timing of the ping-pong GEMM on the path's shapes: tile choice and the stream-K tail (V3D_GEMM_STREAMK 0 / 1 / 2)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-3d-llm_amd"))
from v3d import ops, _native
if len(sys.argv) > 1: _native.LIB_PATH = os.path.abspath(sys.argv[1])      # A/B against another build of the library
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
    for _ in range(10): ops.gemm(a, w, epilogue=epi, out=out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.gemm(a, w, epilogue=epi, out=out, **kw)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
S = 6794
shapes = [("llm qkv", S, 4608, 3584, 1), ("llm o_proj", S, 3584, 3584, 5), ("llm gate_up swiglu", S, 37888, 3584, 6), ("llm down", S, 3584, 18944, 5),
          ("vit qkv", 23328, 4608, 1152, 1), ("vit qkv packed", 23328, 3584, 1152, 1), ("vit fc1 gelu", 23328, 4352, 1152, 3), ("vit fc2", 23328, 1280, 4352, 4 if False else 1), ("proj1 gelu", 23328, 3584, 1152, 2),
          ("proj2", 23328, 3584, 3584, 1), ("vit out 1152", 23328, 1152, 1152, 1), ("vit out 1280", 23328, 1280, 1152, 1), ("patch embed", 23328, 1152, 640, 1),
          ("answer grp o", 960, 3584, 3584, 5), ("answer grp qkv", 960, 4608, 3584, 1), ("answer grp gate_up", 960, 37888, 3584, 6), ("answer grp down", 960, 3584, 18944, 5),
          ("square 4096", 4096, 4096, 4096, 0), ("square 8192", 8192, 8192, 8192, 0)]
os.environ["V3D_GEMM_PP"] = "1"
os.environ.pop("V3D_GEMM_SKEW", None)
for name, M, N, K, epi in shapes:
    row = []
    # automatic tile choice without / with the stream-K tail, then the 256 x 256 tile forced without / with it
    for var, sk in ((None, "0"), (None, "1"), ("3", "0"), ("3", "2"), ("4", "0"), ("1", "0")):
        if var is None: os.environ.pop("V3D_GEMM_VARIANT", None)
        else: os.environ["V3D_GEMM_VARIANT"] = var
        os.environ["V3D_GEMM_STREAMK"] = sk
        ms = run(M, N, K, epi)
        row.append(f"{ms*1e3:7.1f}")
    tf = 2.0 * M * N * K / (float(row[1]) * 1e-6) / 1e12
    print(f"{name:18s} {M:5d}x{N:5d}x{K:5d} | auto sk=0 / sk=1: {row[0]} {row[1]} ({tf:6.0f} TF/s) | pp-256 sk=0 / sk=2: {row[2]} {row[3]} | pp-192: {row[4]} | 128x128: {row[5]}", flush=True)
