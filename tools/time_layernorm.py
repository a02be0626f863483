"""LayerNorm at the SigLIP shape (32 x 729 rows of 1152, row stride 1280): time per launch and bit-equality of the three-rows-per-wave
form (rows >= 4096) with the one-row form (the same rows sent in chunks below 4096)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-3d-llm_amd"))
from v3d import ops
rows, cols, ld = 32 * 729, 1152, 1280
x = torch.randn(rows, ld, device="cuda", dtype=torch.bfloat16)[:, :cols]
w = (1 + 0.1 * torch.randn(cols, device="cuda")).bfloat16(); b = (0.1 * torch.randn(cols, device="cuda")).bfloat16()
out = torch.empty(rows, ld, device="cuda", dtype=torch.bfloat16)[:, :cols]
ops.layernorm(x, w, b, 1e-6, out=out)
ref = torch.cat([ops.layernorm(x[i: i + 2048], w, b, 1e-6) for i in range(0, rows, 2048)])
print("bit-equal to the one-row form:", torch.equal(out, ref))
want = torch.nn.functional.layer_norm(x.float(), (cols,), w.float(), b.float(), 1e-6)
print("max abs err vs f32:", (out.float() - want).abs().max().item())
for name, fn in (("3 rows/wave (full)", lambda: ops.layernorm(x, w, b, 1e-6, out=out)),
                 ("1 row/wave (chunks of 2048)", lambda: [ops.layernorm(x[i: i + 2048], w, b, 1e-6, out=out[i: i + 2048]) for i in range(0, rows, 2048)])):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) * 1e3 / 50:.1f} us")
