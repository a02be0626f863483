"""Times of the causal prefill attention (S = 6794, 28/4 heads x 128) through the given builds of libv3d_hip.so, results not compared:
   python tools/attn_time_one.py lib_a.so lib_b.so ...   (tools/attn_hp_ablate.sh builds the ablation variants)"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-3d-llm_amd"))
from v3d import _native, ops
libs = []
for path in sys.argv[1:]:
    l = ctypes.CDLL(os.path.join(ROOT, path))
    for name, (res, args) in _native.SIGNATURES.items():
        if hasattr(l, name):
            fn = getattr(l, name); fn.restype = res; fn.argtypes = args
    libs.append((os.path.basename(path), l))
H, KV, D, S = 28, 4, 128, 6794
torch.manual_seed(0)
q = torch.randn(1, S, H, D, device="cuda", dtype=torch.bfloat16)
k = torch.randn(1, S, KV, D, device="cuda", dtype=torch.bfloat16)
v = torch.randn(1, S, KV, D, device="cuda", dtype=torch.bfloat16)
def timeit(lib, n=30):
    _native._lib = lib
    for _ in range(5): ops.attention_bshd(q, k, v, causal=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): ops.attention_bshd(q, k, v, causal=True)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for rep in range(3):
    print("rep", rep, "  ".join(f"{name}: {timeit(l):.1f}" for name, l in libs), flush=True)
