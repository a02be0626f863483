"""The tile plans of v3d_gemm beside each other on the shapes where the library (hipBLASLt via torch.matmul) is ahead, one process, variants interleaved
(V3D_GEMM_VARIANT / V3D_GEMM_STREAMK are read per call): median of 5 blocks of 10 launches, us."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "video-3d-llm_amd"))
from v3d import ops
dt = torch.bfloat16
VARIANTS = [("auto", {"V3D_GEMM_VARIANT": "0", "V3D_GEMM_STREAMK": "1"}), ("256 whole rounds", {"V3D_GEMM_VARIANT": "3", "V3D_GEMM_STREAMK": "0"}),
            ("256 + split tail", {"V3D_GEMM_VARIANT": "3", "V3D_GEMM_STREAMK": "2"}), ("192", {"V3D_GEMM_VARIANT": "4", "V3D_GEMM_STREAMK": "0"}),
            ("128 x 128", {"V3D_GEMM_VARIANT": "1", "V3D_GEMM_STREAMK": "0"})]


def block(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 100


def t(name, M, N, K):
    a = torch.randn(M, K, device="cuda", dtype=dt) * 0.5
    w = torch.randn(N, K, device="cuda", dtype=dt) * 0.05
    out = torch.empty(M, N, device="cuda", dtype=dt)
    res = {v: [] for v, _ in VARIANTS}
    res["hipBLASLt"] = []
    for rep in range(6):
        for v, env in VARIANTS:
            os.environ.update(env)
            us = block(lambda: ops.gemm(a, w, out=out))
            if rep:
                res[v].append(us)
        us = block(lambda: torch.matmul(a, w.t()))
        if rep:
            res["hipBLASLt"].append(us)
    print(f"{name:10s} M={M} N={N} K={K}: " + "   ".join(f"{v} {statistics.median(x):.1f}" for v, x in res.items()), flush=True)


t("o_proj", 6794, 3584, 3584)
t("vit fc2", 23328, 1280, 4352)
t("vit out", 23328, 1280, 1152)
t("llm qkv", 6794, 4608, 3584)
t("sq 4096", 4096, 4096, 4096)
