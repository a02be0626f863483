"""SigLIP attention (32 frames x 16 heads x 729 tokens, 72-wide heads packed at stride 72, 96-wide tile): the persistent kernel (r04, default)
against one workgroup per item (V3D_ATTN_VIT_PERSIST=0, the r03 kernel); interleaved, bitwise comparison.  Also between the layer's GEMMs."""
import os, sys, math, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-3d-llm_amd"))
from v3d import ops
dt = torch.bfloat16
F_, n, nh, hd, DP = 32, 729, 16, 72, 96
H = nh * hd
torch.manual_seed(0)
qkv = torch.randn(F_ * n, 3584, device="cuda", dtype=dt)
att = torch.empty(F_ * n, 1280, device="cuda", dtype=dt)
def run():
    ops.attention(qkv, qkv[:, H:], qkv[:, 2 * H:], att, F_, n, n, nh, nh, DP, hd, qkv.stride(0), qkv.stride(0), qkv.stride(0), att.stride(0),
                  n * qkv.stride(0), n * qkv.stride(0), n * att.stride(0), hd, hd, hd, False, 0, 1 / math.sqrt(hd))
outs = {}
flops = 4.0 * n * n * hd * nh * F_
for rep in range(3):
    for mode in ("0", "1"):
        os.environ["V3D_ATTN_VIT_PERSIST"] = mode
        att.zero_()
        for _ in range(5): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): run()
        e1.record(); torch.cuda.synchronize()
        outs[mode] = att[:, :H].clone()
        us = e0.elapsed_time(e1) * 20
        print(f"rep {rep} persist={mode}: {us:.1f} us per layer = {flops / us / 1e6:.0f} TF/s = {flops / us / 1e6 / 2500:.3f} of 2.5 PF", flush=True)
print("equal bits:", torch.equal(outs["0"], outs["1"]), " finite:", bool(torch.isfinite(outs["1"].float()).all()))
