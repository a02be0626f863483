"""Debug aid: tiled attention backward against the materialised form, error per row / head / column block."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-3d-llm_amd"))
from v3d import ops, train
S = int(sys.argv[1]) if len(sys.argv) > 1 else 150
n_q, n_kv, hd = 4, 2, 128
width = (n_q + 2 * n_kv) * hd
torch.manual_seed(S)
Sp = (S + 127) // 128 * 128
dev = torch.zeros(Sp, width, dtype=torch.bfloat16, device="cuda")
dev[:S] = torch.randn(S, width, device="cuda").to(torch.bfloat16)
do = torch.randn(S, n_q * hd, device="cuda").to(torch.bfloat16)
a = torch.zeros(S, width, dtype=torch.bfloat16, device="cuda")
train.attention_backward_materialised(dev, do, a, S, n_q, n_kv, hd, hd ** -0.5)
o = torch.empty(S, n_q * hd, dtype=torch.bfloat16, device="cuda")
lse = ops.attention_train(dev, o, S, n_q, n_kv, hd ** -0.5)
b = torch.zeros(S, width, dtype=torch.bfloat16, device="cuda")
ops.attention_backward(dev, o, do, lse, b, S, n_q, n_kv, hd ** -0.5)
a, b = a.float().cpu(), b.float().cpu()
names = [f"dq{h}" for h in range(n_q)] + [f"dk{g}" for g in range(n_kv)] + [f"dv{g}" for g in range(n_kv)]
for i, nm in enumerate(names):
    x, y = a[:, i * hd:(i + 1) * hd], b[:, i * hd:(i + 1) * hd]
    err_rows = (x - y).norm(dim=1) / x.norm(dim=1).clamp_min(1e-6)
    bad = (err_rows > 0.03).nonzero().flatten().tolist()
    err_cols = (x - y).norm(dim=0) / x.norm(dim=0).clamp_min(1e-6)
    print(f"{nm}: total {float((x - y).norm() / x.norm()):.4f}; bad rows {len(bad)}: {bad[:24]}; col-block err {[round(float(err_cols[c:c + 32].mean()), 3) for c in range(0, 128, 32)]}")
