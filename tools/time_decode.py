import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-3d-llm_amd"))
from v3d import ops
dt = torch.bfloat16
def t(name, N, K, epi, iters=50, norm=False):
    x = torch.randn(K, device="cuda", dtype=dt)
    ws = [torch.randn(N, K, device="cuda", dtype=dt) * 0.02 for _ in range(4)]   # rotate > L3
    nw = torch.ones(K, device="cuda", dtype=dt) if norm else None
    b = torch.zeros(N, device="cuda", dtype=dt); r = torch.zeros(N, device="cuda", dtype=dt)
    out = torch.empty(N, device="cuda", dtype=dt)
    f = lambda w: ops.linear_decode(x, w, out, norm_weight=nw, bias=b if epi == 1 else None, res=r if epi == 2 else None, epilogue=epi)
    for w in ws: f(w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): f(ws[i % 4])
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"{name:14s} N={N:6d} K={K:6d} {us:8.1f} us  {N*K*2/us/1e6:6.2f} TB/s")
t("qkv", 4608, 3584, 1, norm=True)
t("o_proj", 3584, 3584, 2)
t("gate_up", 37888, 3584, 3, norm=True)
t("down", 3584, 18944, 2)
t("lm_head", 152064, 3584, 0, norm=False, iters=12)
def tm(name, N, K, epi, M, iters=40, norm=False):
    x = torch.randn(M, K, device="cuda", dtype=dt)
    ws = [torch.randn(N, K, device="cuda", dtype=dt) * 0.02 for _ in range(4)]
    nw = torch.ones(K, device="cuda", dtype=dt) if norm else None
    b = torch.zeros(N, device="cuda", dtype=dt); r = torch.zeros(M, N, device="cuda", dtype=dt)
    out = torch.empty(M, N // 2 if epi == 3 else N, device="cuda", dtype=dt)
    f = lambda w: ops.linear_decode_rows(x, w, out, norm_weight=nw, bias=b if epi == 1 else None, res=r if epi == 2 else None, epilogue=epi)
    for w in ws: f(w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): f(ws[i % 4])
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"{name:10s} M={M} N={N:6d} K={K:6d} {us:8.1f} us  {N*K*2/us/1e6:6.2f} TB/s")
for M in (8, 16):
    tm("o_proj", 3584, 3584, 2, M)
    tm("down", 3584, 18944, 2, M)
    tm("lm_head", 152064, 3584, 0, M, iters=12)
    tm("qkv-nonorm", 4608, 3584, 1, M)
    tm("gu-nonorm", 37888, 3584, 3, M)
def tm8(name, N, K, epi, M, iters=40):
    x = torch.randn(M, K, device="cuda", dtype=dt)
    ws = [ops.quantize_fp8_rows(torch.randn(N, K, device="cuda", dtype=dt) * 0.02) for _ in range(4)]
    r = torch.zeros(M, N, device="cuda", dtype=dt)
    out = torch.empty(M, N // 2 if epi == 3 else N, device="cuda", dtype=dt)
    f = lambda w: ops.linear_decode_fp8_rows(x, w[0], w[1], out, res=r if epi == 2 else None, epilogue=epi)
    for w in ws: f(w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): f(ws[i % 4])
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"fp8 {name:8s} M={M} N={N:6d} K={K:6d} {us:8.1f} us  {N*K/us/1e6:6.2f} TB/s")
for M in (1, 4, 16):
    tm8("qkv", 4608, 3584, 0, M)
    tm8("o_proj", 3584, 3584, 2, M)
    tm8("gate_up", 37888, 3584, 3, M)
    tm8("down", 3584, 18944, 2, M)
    tm8("lm_head", 152064, 3584, 0, M, iters=12)
# decode attention
H, KV, D, Sk = 28, 4, 128, 6800
q = torch.randn(H * D, device="cuda", dtype=dt)
caches = [torch.randn(8192, 2 * KV * D, device="cuda", dtype=dt) for _ in range(28)]
out = torch.empty(H * D, device="cuda", dtype=dt); ws = ops.decode_workspace(H, KV, "cuda")
for c in caches: ops.attention_decode(q, c, c[:, KV * D:], out, Sk, H, KV, 0.088, ws)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(56): ops.attention_decode(q, caches[i % 28], caches[i % 28][:, KV * D:], out, Sk, H, KV, 0.088, ws)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 56
print(f"attn_decode Sk={Sk} {us:8.1f} us  {Sk*2*KV*D*2/us/1e6:6.2f} TB/s")
for M in (4, 8, 16):
    qs = torch.randn(M, H * D, device="cuda", dtype=dt)
    outs = torch.empty(M, H * D, device="cuda", dtype=dt)
    wsM = torch.empty(ws.numel() * M, dtype=torch.float32, device="cuda")
    cs = [caches[i % 28] for i in range(M)]
    f = lambda: ops.attention_decode_rows(qs, cs, [c[:, KV * D:] for c in cs], outs, [Sk] * M, H, KV, 0.088, wsM)
    for _ in range(3): f()
    torch.cuda.synchronize(); e0.record()
    for i in range(40): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 40
    print(f"attn_decode_rows M={M} Sk={Sk} {us:8.1f} us  ({us/M:.1f} us per scene, {M*Sk*2*KV*D*2/us/1e6:5.2f} TB/s)")
