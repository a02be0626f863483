"""Decode linears at M rows (weights rotated through 4 copies > L2/MALL): python tools/time_decode_rows.py   (V3D_DEC_OG=1|2|3 selects the
outputs-per-workgroup rule of linear_decode_mfma_kernel)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-3d-llm_amd"))
from v3d import ops
dt = torch.bfloat16
def tm(name, N, K, epi, M, iters=40):
    x = torch.randn(M, K, device="cuda", dtype=dt)
    ws = [torch.randn(N, K, device="cuda", dtype=dt) * 0.02 for _ in range(4)]
    b = torch.zeros(N, device="cuda", dtype=dt); r = torch.zeros(M, N, device="cuda", dtype=dt)
    out = torch.empty(M, N // 2 if epi == 3 else N, device="cuda", dtype=dt)
    f = lambda w: ops.linear_decode_rows(x, w, out, bias=b if epi == 1 else None, res=r if epi == 2 else None, epilogue=epi)
    for w in ws: f(w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): f(ws[i % 4])
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"V2={os.environ.get('V3D_DEC_V2', '1')} OG={os.environ.get('V3D_DEC_OG', '2')} {name:8s} M={M:2d} N={N:6d} K={K:6d} {us:8.1f} us  {N*K*2/us/1e6:6.2f} TB/s", flush=True)
for M in (16, 32):
  for v2 in os.environ.get("V3D_DEC_V2_LIST", os.environ.get("V3D_DEC_V2", "1")).split(","):
    os.environ["V3D_DEC_V2"] = v2                           # read per call by the library
    tm("qkv", 4608, 3584, 1, M)
    tm("o_proj", 3584, 3584, 2, M)
    tm("gate_up", 37888, 3584, 3, M)
    tm("down", 3584, 18944, 2, M)
    tm("lm_head", 152064, 3584, 0, M, iters=12)
