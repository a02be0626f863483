"""r04 A/B of the prefill attention's MFMA shape (MI355X_MICROARCH.md 'DVFS give-back' item 7, rule 28): attn_prefill_kernel
(v_mfma_f32_32x32x16) against attn_prefill16_kernel (v_mfma_f32_16x16x32), same per-wave tile, inside ONE process on random data, by wall:
  (a) back-to-back launches (the chip settles at its sustained clock for this kernel alone),
  (b) the kernel as it runs in the scene pipeline: one launch between the layer's qkv and o_proj GEMMs (HIP events around it).
Modes interleaved, three repetitions.   python tools/attn_mfma_ab.py [S]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-3d-llm_amd"))
from v3d import ops  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 6794
H, KV, D = 28, 4, 128
torch.manual_seed(0)
dt = torch.bfloat16
q = torch.randn(1, S, H, D, device="cuda", dtype=dt)
k = torch.randn(1, S, KV, D, device="cuda", dtype=dt)
v = torch.randn(1, S, KV, D, device="cuda", dtype=dt)
x = torch.randn(S, 3584, device="cuda", dtype=dt)
wqkv = torch.randn(4608, 3584, device="cuda", dtype=dt) * 0.02
wo = torch.randn(3584, 3584, device="cuda", dtype=dt) * 0.02
wgu = torch.randn(37888, 3584, device="cuda", dtype=dt) * 0.02
qkv_out = torch.empty(S, 4608, device="cuda", dtype=dt)
o_out = torch.empty(S, 3584, device="cuda", dtype=dt)
act = torch.empty(S, 18944, device="cuda", dtype=dt)
CAUSAL = os.environ.get("V3D_AB_CAUSAL", "1") != "0"          # 0: every key for every query (the kernels' steady state alone)
flops = (2.0 if CAUSAL else 4.0) * S * S * D * H


def attn():
    return ops.attention_bshd(q, k, v, causal=CAUSAL)


def back_to_back(n=40):
    for _ in range(5):
        attn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        attn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def in_layer(n=28):
    """qkv GEMM -> attention -> o_proj -> gate/up GEMM, as a decoder layer issues them; events around the attention launch only."""
    pairs = []
    for i in range(n + 3):
        ops.gemm(x, wqkv, out=qkv_out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        attn()
        e1.record()
        ops.gemm(x, wo, out=o_out)
        ops.gemm(x, wgu, epilogue=ops.EPI_SWIGLU, out=act)
        if i >= 3:
            pairs.append((e0, e1))
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in pairs) * 1e3 / len(pairs)


MODES = {"32": {"V3D_ATTN_MFMA": "32"}, "16": {"V3D_ATTN_MFMA": "16"}}
SEL = os.environ.get("V3D_AB_MODES", "32,16").split(",")
outs = {}
for mode in SEL:
    os.environ.update(MODES[mode])
    outs[mode] = attn().float()
torch.cuda.synchronize()
d = (outs[SEL[-1]] - outs[SEL[0]]).abs()
print(f"S={S}: max |out_{SEL[-1]} - out_{SEL[0]}| = {d.max().item():.3e} (mean {d.mean().item():.3e}; |v| ~ 1, bf16 ulp at 1 = 7.8e-3)", flush=True)
for rep in range(3):
    row = []
    for mode in SEL:
        os.environ.update(MODES[mode])
        a, b = back_to_back(), in_layer()
        row.append(f"{mode}: back-to-back {a:.1f} us = {flops / a / 1e6:.0f} TF/s = {flops / a / 1e6 / 2500:.3f} | in layer {b:.1f} us = "
                   f"{flops / b / 1e6:.0f} TF/s = {flops / b / 1e6 / 2500:.3f}")
    print(f"rep {rep}:  " + "   ||   ".join(row), flush=True)
