import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-3d-llm_amd"))
from v3d import ops
dt = torch.bfloat16
H, KV, D, Sk = 28, 4, 128, 6800
q = torch.randn(H * D, device="cuda", dtype=dt)
caches = [torch.randn(8192, 2 * KV * D, device="cuda", dtype=dt) for _ in range(28)]
out = torch.empty(H * D, device="cuda", dtype=dt); ws = ops.decode_workspace(H, KV, "cuda")
for i in range(84): ops.attention_decode(q, caches[i % 28], caches[i % 28][:, KV * D:], out, Sk, H, KV, 0.088, ws)
torch.cuda.synchronize()
