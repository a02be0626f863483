"""Quick timing of the 3-D position kernels at BASELINE size (scratch tool; bench.py is the contract)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-3d-llm_amd"))
from v3d import ops

V, C = 32, 3584
dt = torch.float16
coords = ((torch.rand((V, 384, 384, 3), device="cuda") - 0.5) * 30).to(dt)
feat = torch.randn((V, 729, C), device="cuda", dtype=dt)
nl = torch.randn(C, device="cuda", dtype=dt)
table = ops.Sin3DTable(C, 301, dt, "cuda")
depth = torch.randint(400, 5000, (V, 480, 640), device="cuda", dtype=torch.int32)
d16 = depth.to(torch.int16)
K = torch.eye(4, device="cuda").repeat(V, 1, 1); K[:, 0, 0] = K[:, 1, 1] = 577.87; K[:, 0, 2] = 319.5; K[:, 1, 2] = 239.5
P = torch.eye(4, device="cuda").repeat(V, 1, 1)
out = torch.empty((V * 210, C), device="cuda", dtype=dt)

def timeit(name, fn, nbytes, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"{name:34s} {us:9.1f} us  {nbytes / us / 1e3:8.1f} GB/s (algorithmic {nbytes/1e6:.1f} MB)")

_, _, ids = ops.coord_pool_voxel(coords)
timeit("unproject_f32 [32,480,640]", lambda: ops.unproject(K, P, depth.float()), 0)
df = depth.float()
timeit("unproject_f32 (pre-cast)", lambda: ops.unproject(K, P, df), V*480*640*16)
timeit("unproject_sampled_u16 ->f16", lambda: ops.unproject_sampled(d16, K, P, 384, dt), V*384*384*(2+6))
timeit("coord_pool_voxel f16", lambda: ops.coord_pool_voxel(coords), coords.numel()*2)
c32 = coords.float()
timeit("coord_pool_voxel f32", lambda: ops.coord_pool_voxel(c32), c32.numel()*4)
timeit("visual_tokens pool+pe+nl f16", lambda: ops.visual_tokens(feat, ids, table, nl, out=out), feat.numel()*2 + out.numel()*2)
pooled = ops.visual_tokens(feat, pool=True).view(V, 196, C)
timeit("visual_tokens pe+nl (no pool)", lambda: ops.visual_tokens(pooled, ids, table, nl, pool=False, out=out), pooled.numel()*2 + out.numel()*2)
