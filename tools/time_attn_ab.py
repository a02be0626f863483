"""A/B of the two prefill attention kernels (V3D_ATTN64=0 attn_prefill_kernel / 1 attn_prefill64_kernel) in one process."""
import os, sys, math, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-3d-llm_amd"))
from v3d import ops
H, KV, D = 28, 4, 128
for S in (6794, 8192, 2048):
    q = torch.randn(1, S, H, D, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(1, S, KV, D, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(1, S, KV, D, device="cuda", dtype=torch.bfloat16)
    outs = {}
    for mode in ("0", "1", "0", "1"):
        os.environ["V3D_ATTN64"] = mode
        for _ in range(3): o = ops.attention_bshd(q, k, v, causal=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): o = ops.attention_bshd(q, k, v, causal=True)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        outs[mode] = o
        print(f"S={S} V3D_ATTN64={mode}: {us:.1f} us  {2.0*S*S*D*H/us/1e6:.1f} TF/s", flush=True)
    d = (outs["0"].float() - outs["1"].float()).abs()
    print(f"   max |diff| between the kernels {d.max().item():.3e}  (mean |o| {outs['0'].float().abs().mean().item():.3e}), equal bits: {torch.equal(outs['0'], outs['1'])}")
