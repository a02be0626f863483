"""Times v3d_gemm_fp8 against the bf16 v3d_gemm on the LLM linear shapes (S = 6794 rows).  GPU only."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "video-3d-llm_amd"))
import torch
from v3d import ops

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

S = 6794
for (M, N, K, epi) in [(S, 4608, 3584, "none"), (S, 3584, 3584, "none"), (S, 37888, 3584, "swiglu"), (S, 3584, 18944, "none"), (8192, 8192, 8192, "none")]:
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    qa, sa = ops.quantize_fp8_rows(a)
    qw, sw = ops.quantize_fp8_rows(w)
    e = ops.EPI_SWIGLU if epi == "swiglu" else ops.EPI_NONE
    out8 = torch.empty(M, N // 2 if epi == "swiglu" else N, dtype=torch.bfloat16, device="cuda")
    out16 = torch.empty_like(out8)
    t8 = timeit(lambda: ops.gemm_fp8(qa, sa, qw, sw, torch.bfloat16, epilogue=e, out=out8))
    t16 = timeit(lambda: ops.gemm(a, w, epilogue=e, out=out16))
    tq = timeit(lambda: ops.quantize_fp8_rows(a, qa, sa))
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K} {epi}: fp8 {t8*1e3:.0f} us ({fl/t8/1e9:.0f} TF)  bf16 {t16*1e3:.0f} us ({fl/t16/1e9:.0f} TF)  quant {tq*1e3:.0f} us ({M*K*3/tq/1e6:.0f} GB/s)", flush=True)
