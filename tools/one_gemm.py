import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-3d-llm_amd"))
from v3d import ops
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (6794, 3584, 3584)
a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16) * 0.5
w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.05
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(5): ops.gemm(a, w, out=out)
torch.cuda.synchronize()
