import os, sys, math, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-3d-llm_amd"))
from v3d import ops
S, H, KV, D = 6794, 28, 4, 128
q = torch.randn(1, S, H, D, device="cuda", dtype=torch.bfloat16)
k = torch.randn(1, S, KV, D, device="cuda", dtype=torch.bfloat16)
v = torch.randn(1, S, KV, D, device="cuda", dtype=torch.bfloat16)
for _ in range(5): ops.attention_bshd(q, k, v, causal=True)
torch.cuda.synchronize()
