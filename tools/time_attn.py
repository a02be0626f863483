import os, sys, math, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-3d-llm_amd"))
from v3d import ops
S, H, KV, D = 6794, 28, 4, 128
q = torch.randn(1, S, H, D, device="cuda", dtype=torch.bfloat16)
k = torch.randn(1, S, KV, D, device="cuda", dtype=torch.bfloat16)
v = torch.randn(1, S, KV, D, device="cuda", dtype=torch.bfloat16)
for _ in range(3): ops.attention_bshd(q, k, v, causal=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.attention_bshd(q, k, v, causal=True)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
print(f"attn prefill S={S}: {us:.1f} us  {2.0*S*S*D*H/us/1e6:.1f} TF/s  (debug={os.environ.get('V3D_ATTN_DEBUG','0')})")
