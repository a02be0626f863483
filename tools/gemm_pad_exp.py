"""Experiment: SigLIP N = 1152 GEMMs on the 128x128 kernel vs zero-padded to N = 1280 on the 256-wide kernel."""
import sys, torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0] + "/video-3d-llm_amd")
from v3d import ops
dt = torch.bfloat16
def t(name, M, N, K, epi=0, iters=20):
    a = torch.randn(M, K, device="cuda", dtype=dt) * 0.5
    w = torch.randn(N, K, device="cuda", dtype=dt) * 0.05
    out = torch.empty(M, N, device="cuda", dtype=dt)
    kw = {}
    if epi == 4:
        kw = dict(bias=torch.zeros(N, device="cuda", dtype=dt), res=torch.zeros(M, N, device="cuda", dtype=dt))
    for _ in range(3): ops.gemm(a, w, epilogue=epi, out=out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.gemm(a, w, epilogue=epi, out=out, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:28s} M={M} N={N} K={K}: {ms*1e3:8.1f} us")
for rep in range(2):
    t("out  N=1152 (128 kernel)", 23328, 1152, 1152, 4)
    t("out  N=1280 (256 kernel)", 23328, 1280, 1152, 4)
    t("fc2  N=1152 (128 kernel)", 23328, 1152, 4352, 4)
    t("fc2  N=1280 (256 kernel)", 23328, 1280, 4352, 4)
