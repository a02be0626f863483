"""SigLIP out_proj (bias + residual epilogue, residual stream stride 1280): N = 1152 on the 128 x 128 tile vs N = 1280 (zero pad rows) on the 256-wide tile."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-3d-llm_amd"))
from v3d import ops
dt = torch.bfloat16
M, K = 23328, 1152
a = torch.randn(M, K, device="cuda", dtype=dt) * 0.5
x = torch.randn(M, 1280, device="cuda", dtype=dt)
for N in (1152, 1280):
    w = torch.randn(N, K, device="cuda", dtype=dt) * 0.05
    b = torch.zeros(N, device="cuda", dtype=dt)
    for _ in range(10): ops.gemm(a, w, bias=b, res=x, epilogue=ops.EPI_BIAS_RES, out=x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.gemm(a, w, bias=b, res=x, epilogue=ops.EPI_BIAS_RES, out=x)
    e1.record(); torch.cuda.synchronize()
    print(f"N={N}: {e0.elapsed_time(e1) * 20:.1f} us")
