"""Does a weight-streaming decode linear (64 VGPRs, no LDS) co-run with the 256-wide GEMM (218 VGPRs, 128 KiB LDS)?"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-3d-llm_amd"))
from v3d import ops
dt = torch.bfloat16
S = 6794
a = torch.randn(S, 3584, device="cuda", dtype=dt) * 0.5
wgu = torch.randn(37888, 3584, device="cuda", dtype=dt) * 0.02
act = torch.empty(S, 18944, device="cuda", dtype=dt)
x = torch.randn(3584, device="cuda", dtype=dt)
wd = [torch.randn(37888, 3584, device="cuda", dtype=dt) * 0.02 for _ in range(3)]
o = torch.empty(18944, device="cuda", dtype=dt)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream(priority=int(os.environ.get("PRIO_B", "0")))
def runA(n):
    with torch.cuda.stream(sA):
        for _ in range(n): ops.gemm(a, wgu, epilogue=ops.EPI_SWIGLU, out=act)
def runB(n):
    with torch.cuda.stream(sB):
        for i in range(n): ops.linear_decode(x, wd[i % 3], o, epilogue=ops.DEC_SWIGLU)
def timed(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
runA(2); runB(10)
nA, nB = 20, 600
tA = timed(lambda: runA(nA)); tB = timed(lambda: runB(nB))
tAB = timed(lambda: (runA(nA), runB(nB)))
tBA = timed(lambda: (runB(nB), runA(nA)))
print(f"GEMM alone {tA:.1f} ms, decode-linear alone {tB:.1f} ms, both (A queued first) {tAB:.1f} ms, both (B first) {tBA:.1f} ms, sum {tA+tB:.1f}")
