"""One launch each of the gate/up GEMM, the down_proj GEMM and the causal prefill attention at the path's shapes (for rocprofv3 --pmc passes)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-3d-llm_amd"))
from v3d import ops
dt = torch.bfloat16
S = 6794
a = torch.randn(S, 3584, device="cuda", dtype=dt) * 0.5
w = torch.randn(37888, 3584, device="cuda", dtype=dt) * 0.05
q = torch.randn(1, S, 28, 128, device="cuda", dtype=dt)
k = torch.randn(1, S, 4, 128, device="cuda", dtype=dt)
v = torch.randn(1, S, 4, 128, device="cuda", dtype=dt)
for _ in range(3):
    ops.gemm(a, w, epilogue=ops.EPI_SWIGLU)
    ops.attention_bshd(q, k, v, causal=True)
torch.cuda.synchronize()
