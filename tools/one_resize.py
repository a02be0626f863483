"""Three launches each of the r03 RGB resize kernel (32 x 1296 x 968 frames -> 512 x 384 -> crop 384, fused normalise) and of the 3-D position kernels
at the bench shapes (for rocprofv3 --pmc passes)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-3d-llm_amd"))
from v3d import ops
g = torch.Generator(device="cuda").manual_seed(0)
fr = torch.randint(0, 256, (32, 968, 1296, 3), generator=g, device="cuda", dtype=torch.int32).to(torch.uint8)
small = torch.randint(0, 256, (32, 480, 640, 3), generator=g, device="cuda", dtype=torch.int32).to(torch.uint8)
for _ in range(3):
    ops.resize_crop_rgb(fr, (384, 512), crop=(0, 64, 384, 384), dtype=torch.bfloat16)
    ops.resize_crop_rgb(small, (384, 512), crop=(0, 64, 384, 384), dtype=torch.bfloat16)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, t in (("1296x968", fr), ("640x480", small)):
    e0.record()
    for _ in range(10):
        ops.resize_crop_rgb(t, (384, 512), crop=(0, 64, 384, 384), dtype=torch.bfloat16)
    e1.record(); torch.cuda.synchronize()
    print(f"resize_bicubic {name}: {e0.elapsed_time(e1) * 100:.1f} us per launch (32 frames)")
