"""A few launches of the r04 decode kernels at the cached-question shapes (for rocprofv3 --pmc passes): the gate/up decode linear at M = 32 rows
(two 16-row blocks, 32 outputs per workgroup), the LM head at M = 16, and the shared-prefix decode attention of 32 questions over a 6734-row prefix."""
import math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-3d-llm_amd"))
from v3d import ops
dt = torch.bfloat16
torch.manual_seed(0)
x32 = torch.randn(32, 3584, device="cuda", dtype=dt)
wgu = torch.randn(37888, 3584, device="cuda", dtype=dt) * 0.02
act = torch.empty(32, 18944, device="cuda", dtype=dt)
whead = torch.randn(152064, 3584, device="cuda", dtype=dt) * 0.02
logits = torch.empty(16, 152064, device="cuda", dtype=dt)
H, KV, D, P, M = 28, 4, 128, 6734, 32
q = torch.randn(M, H * D, device="cuda", dtype=dt)
shared = torch.randn(P, 2 * KV * D, device="cuda", dtype=dt)
caches = []
for m in range(M):
    c = torch.randn(P + 80, 2 * KV * D, device="cuda", dtype=dt)
    c[:P] = shared
    caches.append(c)
lens = [P + 61 + (m % 7) for m in range(M)]
one = ops.decode_workspace(H, KV, "cuda")
ws = torch.empty(one.numel() * M, dtype=torch.float32, device="cuda")
out = torch.empty(M, H * D, dtype=dt, device="cuda")
# r04, second session: down_proj through the K split (resident chunks + combine) and the question rows of an answer batch over one copy of the prefix
wd = torch.randn(3584, 18944, device="cuda", dtype=dt) * 0.02
a32 = torch.randn(32, 18944, device="cuda", dtype=dt)
r32 = torch.randn(32, 3584, device="cuda", dtype=dt)
o32 = torch.empty(32, 3584, device="cuda", dtype=dt)
Sq, kvw = 60, KV * D
own = torch.randn(M, P + Sq + 4, 2 * kvw, device="cuda", dtype=dt)
qq = torch.randn(M * Sq, H * D, device="cuda", dtype=dt)
oo = torch.empty(M * Sq, H * D, dtype=dt, device="cuda")
o2 = own.view(-1, 2 * kvw)
for _ in range(3):
    ops.linear_decode_rows(a32, wd, o32, res=r32, epilogue=ops.DEC_RES)
    ops.attention_shared_prefix(qq, o2, o2[:, kvw:], shared, shared[:, kvw:], (P // 64) * 64, oo, M, Sq, P + Sq, H, KV, qq.stride(0), o2.stride(0), o2.stride(0),
                                oo.stride(0), Sq * qq.stride(0), own.stride(0), Sq * oo.stride(0), D, D, D, P, 1 / math.sqrt(D))
    ops.linear_decode_rows(x32, wgu, act, epilogue=ops.DEC_SWIGLU)
    ops.linear_decode_rows(x32[:16], whead, logits)
    ops.attention_decode_rows(q, caches, [c[:, KV * D:] for c in caches], out, lens, H, KV, 1 / math.sqrt(D), ws, prefix=P)
torch.cuda.synchronize()
