"""Norm backward kernels at the training step's shapes (us per launch, bytes = x + dy + add read, dx written):  python tools/time_norm_grad.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-3d-llm_amd"))
from v3d import ops
dt = torch.bfloat16
def tm(name, fn, rows, cols, iters=20):
    x = torch.randn(rows, cols, device="cuda", dtype=dt); dy = torch.randn(rows, cols, device="cuda", dtype=dt); add = torch.randn(rows, cols, device="cuda", dtype=dt)
    w = torch.randn(cols, device="cuda", dtype=dt)
    for _ in range(3): fn(x, w, dy, 1e-6, add)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn(x, w, dy, 1e-6, add)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"{name:22s} {rows:6d} x {cols:5d} {us:8.1f} us  {4 * rows * cols * 2 / us / 1e6:5.2f} TB/s", flush=True)
tm("layernorm_grad (SigLIP)", ops.layernorm_grad, 23328, 1152)
tm("rmsnorm_grad (Qwen2)", ops.rmsnorm_grad, 6794, 3584)
tm("layernorm_grad (wide)", ops.layernorm_grad, 6794, 3584)
tm("rmsnorm_grad (narrow)", ops.rmsnorm_grad, 23328, 1152)
