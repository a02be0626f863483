"""A/B of a GEMM launch-time option in ONE process, alternating the variants (thermal state and clocks drift between
processes): sets an env var the library re-reads?  No - the library caches it, so the two variants run in two child
processes that alternate shape by shape is not possible either; instead this script times every shape N times and is run
once per variant back to back by the caller.  Prints median of 5 blocks of 10 launches."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "video-3d-llm_amd"))
from v3d import ops
dt = torch.bfloat16
def t(name, M, N, K, epi=0):
    a = torch.randn(M, K, device="cuda", dtype=dt) * 0.5
    w = torch.randn(N, K, device="cuda", dtype=dt) * 0.05
    out = torch.empty(M, N // 2 if epi == 6 else N, device="cuda", dtype=dt)
    kw = {}
    if epi in (1, 3): kw["bias"] = torch.zeros(N, device="cuda", dtype=dt)
    if epi == 5: kw["res"] = torch.zeros(M, N, device="cuda", dtype=dt)
    for _ in range(3): ops.gemm(a, w, epilogue=epi, out=out, **kw)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.gemm(a, w, epilogue=epi, out=out, **kw)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    ms = statistics.median(ts)
    print(f"{name:22s} {ms*1e3:8.1f} us {2.0*M*N*K/ms/1e9:7.1f} TF/s")
S = 6794
t("llm qkv", S, 4608, 3584, 1); t("llm o_proj", S, 3584, 3584, 5); t("llm gate_up", S, 37888, 3584, 6); t("llm down", S, 3584, 18944, 5)
t("vit qkv", 23328, 4608, 1152, 1); t("vit fc1", 23328, 4352, 1152, 3); t("vit fc2 (1280)", 23328, 1280, 4352, 0); t("proj2", 23328, 3584, 3584, 1)
