import os, sys, torch
sys.path.insert(0, "/root/repo/video-3d-llm_amd")
from v3d import ops
dt = torch.bfloat16
def t(name, M, N, K, epi=0, iters=10):
    a = torch.randn(M, K, device="cuda", dtype=dt) * 0.5
    w = torch.randn(N, K, device="cuda", dtype=dt) * 0.05
    out = torch.empty(M, N // 2 if epi == 6 else N, device="cuda", dtype=dt)
    kw = {}
    if epi in (1, 2, 3): kw["bias"] = torch.zeros(N, device="cuda", dtype=dt)
    if epi == 5: kw["res"] = torch.zeros(M, N, device="cuda", dtype=dt)
    for _ in range(2): ops.gemm(a, w, epilogue=epi, out=out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.gemm(a, w, epilogue=epi, out=out, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:34s} M={M} N={N} K={K}: {ms*1e3:8.1f} us {2.0*M*N*K/ms/1e9:7.1f} TF/s")
S = 6794
t("N=37888 none", S, 37888, 3584, 0)
t("N=37888 bias", S, 37888, 3584, 1)
t("N=37888 res", S, 37888, 3584, 5)
t("N=37888 swiglu", S, 37888, 3584, 6)
t("N=4608 bias (qkv)", S, 4608, 3584, 1)
t("N=4608 none", S, 4608, 3584, 0)
t("N=9216 bias (2x qkv)", S, 9216, 3584, 1)
t("N=18432 bias (4x qkv)", S, 18432, 3584, 1)
t("N=3584 res (o)", S, 3584, 3584, 5)
t("N=7168 res", S, 7168, 3584, 5)
