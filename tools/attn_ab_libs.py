"""A/B of prefill attention between two builds of libv3d_hip.so in one process (same box, same data):
   python tools/attn_ab_libs.py tools/probes/_build/libv3d_hip_r02.so video-3d-llm_amd/libv3d_hip.so
Times 20 launches each (interleaved A B A B), checks that the outputs are bit-identical, at S = 6794 / 8192 / 2048 / 777 (causal, 28q/4kv
x 128) plus the ViT shape (non-causal, 32 x 16 heads x 729 x 72 on the 96-wide tile)."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-3d-llm_amd"))
from v3d import _native, ops
libs = []
for path in sys.argv[1:3]:
    l = ctypes.CDLL(os.path.join(ROOT, path))
    _native._declare_partial = None
    for name, (res, args) in _native.SIGNATURES.items():
        if hasattr(l, name):
            fn = getattr(l, name); fn.restype = res; fn.argtypes = args
    libs.append((path, l))

def run(lib, fn):
    _native._lib = lib
    return fn()

def timeit(lib, fn, n=20):
    for _ in range(3): run(lib, fn)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): o = run(lib, fn)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n, o

H, KV, D = 28, 4, 128
torch.manual_seed(0)
for S in (6794, 8192, 2048, 777):
    q = torch.randn(1, S, H, D, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(1, S, KV, D, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(1, S, KV, D, device="cuda", dtype=torch.bfloat16)
    outs = {}
    for rep in range(2):
        for name, l in libs:
            us, o = timeit(l, lambda: ops.attention_bshd(q, k, v, causal=True))
            outs[name] = o
            print(f"S={S} {name}: {us:.1f} us  {2.0*S*S*D*H/us/1e6:.1f} TF/s", flush=True)
    a, b = (outs[n] for n, _ in libs)
    print(f"   equal bits: {torch.equal(a, b)}  max|diff| {(a.float()-b.float()).abs().max().item():.3e}", flush=True)
# spiky scores: rows whose maximum keeps rising (exercises the raise path)
S = 4096
q = torch.randn(1, S, H, D, device="cuda", dtype=torch.bfloat16) * 3
k = torch.randn(1, S, KV, D, device="cuda", dtype=torch.bfloat16) * 3
k[0, :, :, :] *= torch.linspace(0.2, 3.0, S, device="cuda")[:, None, None].to(torch.bfloat16)
v = torch.randn(1, S, KV, D, device="cuda", dtype=torch.bfloat16)
outs = [run(l, lambda: ops.attention_bshd(q, k, v, causal=True)) for _, l in libs]
print(f"spiky S={S}: equal bits {torch.equal(outs[0], outs[1])}, finite {bool(torch.isfinite(outs[1].float()).all())}", flush=True)
# ViT shape
F, N, Hh, d = 32, 729, 16, 72
qkv = torch.randn(F * N, 3584, device="cuda", dtype=torch.bfloat16)
att = [torch.zeros(F * N, 1152, device="cuda", dtype=torch.bfloat16) for _ in libs]
def vit(out):
    ld = qkv.stride(0)
    return ops.attention(qkv, qkv[:, 1152:], qkv[:, 2304:], out, F, N, N, Hh, Hh, 96, d, ld, ld, ld, out.stride(0), N * ld, N * ld, N * out.stride(0), d, d, d, False, 0, d ** -0.5)
for rep in range(2):
    for i, (name, l) in enumerate(libs):
        us, _ = timeit(l, lambda: vit(att[i]))
        print(f"ViT {name}: {us:.1f} us", flush=True)
print(f"   ViT equal bits: {torch.equal(att[0], att[1])}", flush=True)
