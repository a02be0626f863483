"""Where do stream-K outputs differ from the whole-tile walk by more than one 16-bit rounding?  (debug aid)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-3d-llm_amd"))
from v3d import ops
dt = torch.bfloat16
for (M, N, K) in ((6794, 3584, 2048), (6794, 3584, 3584), (6794, 3584, 18944)):
    g = torch.Generator().manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g) * 0.5).to(dt).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).to(dt).cuda()
    os.environ["V3D_GEMM_VARIANT"] = "3"
    os.environ["V3D_GEMM_STREAMK"] = "0"
    whole = ops.gemm(a, w)
    os.environ["V3D_GEMM_STREAMK"] = "2"
    cut = ops.gemm(a, w)
    torch.cuda.synchronize()
    wf, cf = whole.float(), cut.float()
    d = (cf - wf).abs()
    ulp = 2.0 ** -7
    bad = d > 1.01 * ulp * wf.abs()
    print(f"{M}x{N}x{K}: differing {int((d > 0).sum())}  beyond one ulp {int(bad.sum())}  max diff {d.max().item():.4g}")
    if bad.any():
        idx = bad.nonzero()[:12]
        for r, c in idx.tolist():
            print(f"   row {r} (tile {r // 256}, +{r % 256}) col {c} (tile {c // 256}, +{c % 256}): whole {wf[r, c].item():.6f} cut {cf[r, c].item():.6f}")
        tiles = torch.unique((bad.nonzero()[:, 0] // 256) * 100 + bad.nonzero()[:, 1] // 256)
        print("   tiles (100 tm + tn):", tiles.tolist()[:40], "count", len(tiles))
    ref = (a.float() @ w.float().t())
    print("   whole vs f32 ref max err", (wf - ref).abs().max().item(), " cut vs ref", (cf - ref).abs().max().item())
