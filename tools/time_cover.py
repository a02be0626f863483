"""a3 at a ScanNet-like size: 300 frames x 480x640 points, ~200k scene voxels; device vs the host C++ version."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-3d-llm_amd"))
from v3d import ops
n_frames, pts, span = int(os.environ.get("FRAMES", 300)), 480 * 640, 45
g = torch.Generator(device="cuda").manual_seed(0)
scene = torch.unique(torch.randint(-span, span, (400000, 3), generator=g, device="cuda", dtype=torch.int32), dim=0)
centers = torch.randint(-span, span, (n_frames, 1, 3), generator=g, device="cuda", dtype=torch.int32)
keys = centers + torch.randint(-14, 15, (n_frames, pts, 3), generator=g, device="cuda", dtype=torch.int32)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dev = ops.greedy_cover_device(keys, scene, 32)
    torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"device: {1e3*(t1-t0):.1f} ms  ({n_frames} frames x {pts} points = {n_frames*pts*12/1e9:.2f} GB of keys, {scene.shape[0]} scene voxels)")
kh, sh = keys.cpu().numpy(), scene.cpu().numpy()
t0 = time.perf_counter(); host = ops.greedy_cover(kh, sh, 32); t1 = time.perf_counter()
print(f"host C++ (1 thread): {1e3*(t1-t0):.1f} ms")
assert dev[0].tolist() == host[0].tolist() and dev[1].tolist() == host[1].tolist() and dev[2:] == host[2:]
print("picks, gains and totals identical")
