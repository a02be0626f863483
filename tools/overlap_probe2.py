import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-3d-llm_amd"))
from v3d import ops
dt = torch.bfloat16
# small-grid GEMM: 64 tiles of 256x256 -> at most 64 CUs busy
a = torch.randn(256, 16384, device="cuda", dtype=dt) * 0.1
w = torch.randn(16384, 16384, device="cuda", dtype=dt) * 0.02
o1 = torch.empty(256, 16384, device="cuda", dtype=dt); o2 = torch.empty_like(o1)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
def run(s, out, n):
    with torch.cuda.stream(s):
        for _ in range(n): ops.gemm(a, w, out=out)
def timed(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
run(sA, o1, 2); run(sB, o2, 2)
t1 = timed(lambda: run(sA, o1, 20)); t2 = timed(lambda: (run(sA, o1, 20), run(sB, o2, 20)))
print(f"64-CU GEMM x20 on one stream {t1:.1f} ms; same on two streams concurrently {t2:.1f} ms (serial would be {2*t1:.1f})")
print("GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"))
