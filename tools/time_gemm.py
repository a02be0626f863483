import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-3d-llm_amd"))
from v3d import ops
dt = torch.bfloat16
def t(name, M, N, K, epi=0, iters=20):
    a = torch.randn(M, K, device="cuda", dtype=dt) * 0.5
    w = torch.randn(N, K, device="cuda", dtype=dt) * 0.05
    out = torch.empty(M, N // 2 if epi == 6 else N, device="cuda", dtype=dt)
    bias = torch.zeros(N, device="cuda", dtype=dt) if epi in (1, 2, 3) else None
    kw = dict(bias=bias) if bias is not None else {}
    for _ in range(3): ops.gemm(a, w, epilogue=epi, out=out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.gemm(a, w, epilogue=epi, out=out, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * M * N * K
    # torch (hipBLASLt) cross-check timing
    for _ in range(3): torch.matmul(a, w.t())
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): torch.matmul(a, w.t())
    e1.record(); torch.cuda.synchronize()
    ms2 = e0.elapsed_time(e1) / iters
    print(f"{name:28s} M={M:6d} N={N:6d} K={K:6d}  {ms*1e3:9.1f} us  {fl/ms/1e9:8.1f} TF/s   (hipBLASLt {fl/ms2/1e9:8.1f} TF/s)")
S = 6794
t("llm qkv", S, 4608, 3584)
t("llm o_proj", S, 3584, 3584)
t("llm gate_up swiglu", S, 37888, 3584, epi=6)
t("llm down", S, 3584, 18944)
t("vit qkv", 23328, 4608, 1152)
t("vit out", 23328, 1152, 1152)
t("vit fc1", 23328, 4352, 1152)
t("vit fc1 +bias+gelu_tanh", 23328, 4352, 1152, epi=3)
t("proj1 +bias+gelu_erf", 23328, 3584, 1152, epi=2)
t("vit fc2", 23328, 1152, 4352)
t("proj1", 23328, 3584, 1152)
t("proj2", 23328, 3584, 3584)
t("square 4096", 4096, 4096, 4096)
t("square 8192", 8192, 8192, 8192)
t("decode qkv", 1, 4608, 3584)
t("decode gate_up", 1, 37888, 3584, epi=6)
t("decode down", 1, 3584, 18944)
t("decode lm_head", 1, 152064, 3584)
