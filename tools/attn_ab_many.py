"""Prefill attention (S = 6794 causal 28/4 x 128, and the ViT shape) across several builds of libv3d_hip.so in one process, interleaved:
   python tools/attn_ab_many.py lib_a.so lib_b.so lib_c.so ..."""
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
def timeit(lib, fn, n=20):
    _native._lib = lib
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): o = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n, o
H, KV, D, S = 28, 4, 128, 6794
torch.manual_seed(0)
q = torch.randn(1, S, H, D, device="cuda", dtype=torch.bfloat16)
k = torch.randn(1, S, KV, D, device="cuda", dtype=torch.bfloat16)
v = torch.randn(1, S, KV, D, device="cuda", dtype=torch.bfloat16)
F, N, Hh, d = 32, 729, 16, 72
qkv = torch.randn(F * N, 3584, device="cuda", dtype=torch.bfloat16)
att = torch.zeros(F * N, 1152, device="cuda", dtype=torch.bfloat16)
ld = qkv.stride(0)
vit = lambda: ops.attention(qkv, qkv[:, 1152:], qkv[:, 2304:], att, F, N, N, Hh, Hh, 96, d, ld, ld, ld, att.stride(0), N * ld, N * ld, N * att.stride(0), d, d, d, False, 0, d ** -0.5)
ref = None
for rep in range(3):
    for name, l in libs:
        us, o = timeit(l, lambda: ops.attention_bshd(q, k, v, causal=True))
        us2, _ = timeit(l, vit)
        if ref is None: ref = o.clone()
        print(f"rep {rep} {name}: causal S={S} {us:.1f} us ({2.0*S*S*D*H/us/1e6:.0f} TF/s)   ViT {us2:.1f} us   equal bits {torch.equal(o, ref)}", flush=True)
