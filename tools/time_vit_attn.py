"""SigLIP attention (32 frames x 16 heads x 729 tokens, head dim 72 packed at stride 72, 96-wide tile): time per layer."""
import os, sys, math, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-3d-llm_amd"))
from v3d import ops
dt = torch.bfloat16
F_, n, nh, hd, DP = 32, 729, 16, 72, 96
H = nh * hd
qkv = torch.randn(F_ * n, 3584, device="cuda", dtype=dt)
att = torch.empty(F_ * n, 1280, device="cuda", dtype=dt)
def run():
    ops.attention(qkv, qkv[:, H:], qkv[:, 2 * H:], att, F_, n, n, nh, nh, DP, hd, qkv.stride(0), qkv.stride(0), qkv.stride(0), att.stride(0),
                  n * qkv.stride(0), n * qkv.stride(0), n * att.stride(0), hd, hd, hd, False, 0, 1 / math.sqrt(hd))
outs = {}
for mode in ("6", "5", "6", "5"):
    if mode == "6": os.environ["V3D_ATTN_KS6"] = "1"
    else: os.environ.pop("V3D_ATTN_KS6", None)
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    outs[mode] = att[:, :H].clone()
    print(f"SigLIP attention, {mode} k-steps of QK^T: {e0.elapsed_time(e1) * 20:.1f} us per layer")
print("equal bits:", torch.equal(outs["5"], outs["6"]))
