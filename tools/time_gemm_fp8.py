"""e4m3 LLM linears of BASELINE configs[3] at S = 6794: time per launch (20 launches back to back: power-limited clocks, about 12 %
   slower than inside the pipeline) and max error against the same dequantised operands in f32.   python tools/time_gemm_fp8.py
   Round 3 used it to A/B an e4m3 instantiation of the bf16 ping-pong kernel (gemm.hip: same ring, staging and fragment reads, the two
   16-byte fragments of a row as one v_mfma_scale_f32_16x16x128_f8f6f4 operand, scales applied in front of the epilogue) against
   gemm_fp8.hip's kernel: identical results; down_proj (148 K-steps) 445.9 vs 444.0 us, gate/up 1134 vs 897, qkv 137 vs 98, o 143 vs 98
   (the port's tile prologue / epilogue spilled 135-240 registers; its K loop did not).  With the K loop no faster where it dominates,
   the e4m3 GEMM is not schedule-bound on this chip (2.07 PF = 0.41 of 5 PF in both kernels) and the port was dropped."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-3d-llm_amd"))
from v3d import ops
S = 6794
torch.manual_seed(0)
shapes = [("gate_up+swiglu", 37888, 3584, ops.EPI_SWIGLU), ("down+res", 3584, 18944, ops.EPI_RES), ("qkv+bias", 4608, 3584, ops.EPI_BIAS), ("o+res", 3584, 3584, ops.EPI_RES)]
for name, N, K, epi in shapes:
    x = torch.randn(S, K, device="cuda", dtype=torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * K ** -0.5)
    if epi == ops.EPI_SWIGLU:
        w = ops.interleave_gate_up(w[: N // 2].contiguous(), w[N // 2:].contiguous())
    qa, sa = ops.quantize_fp8_rows(x)
    qw, sw = ops.quantize_fp8_rows(w)
    bias = torch.randn(N, device="cuda", dtype=torch.bfloat16) if epi == ops.EPI_BIAS else None
    res = torch.randn(S, N, device="cuda", dtype=torch.bfloat16) if epi == ops.EPI_RES else None
    fn = lambda: ops.gemm_fp8(qa, sa, qw, sw, torch.bfloat16, bias=bias, res=res, epilogue=epi)
    out = fn()
    # reference on a slice of rows: dequantised operands in f32
    r = slice(0, 512)
    a32 = qa[r].view(torch.float8_e4m3fn).float() * sa[r, None]
    w32 = qw.view(torch.float8_e4m3fn).float() * sw[:, None]
    y = a32 @ w32.t()
    if epi == ops.EPI_BIAS: y = y + bias.float()
    if epi == ops.EPI_RES: y = y.bfloat16().float() + res[r].float()
    if epi == ops.EPI_SWIGLU:
        y = y.bfloat16().float().view(512, N // 128, 2, 64)
        y = (torch.nn.functional.silu(y[:, :, 0]).bfloat16().float() * y[:, :, 1]).reshape(512, N // 2)
    err = (out[r].float() - y).abs().max().item() / y.abs().max().item()
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"{name:16s} N={N:6d} K={K:6d}: {us:8.1f} us  {2.0 * S * N * K / us / 1e6:7.0f} TF/s   max rel err vs f32 of the same e4m3 operands {err:.2e}", flush=True)
