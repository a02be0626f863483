"""Times one Qwen2-7B decoder layer of the training step (v3d/train.py) at the path's sequence length: forward, backward, and the
backward's parts (dense products / attention as tiled kernels and in its materialised first form)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-3d-llm_amd"))
from v3d import ops, train
S = int(sys.argv[1]) if len(sys.argv) > 1 else 6794
H, I, n_q, n_kv, hd = 3584, 18944, 28, 4, 128
dt, dev = torch.bfloat16, "cuda"
width = (n_q + 2 * n_kv) * hd
mk = lambda *shape, s=1.0: (torch.randn(*shape, device=dev) * s).to(dt)
p = {"ln1": torch.ones(H, device=dev, dtype=dt), "qkv": mk(width, H, s=H ** -0.5), "qkv_bias": mk(width, s=0.1), "o": mk(H, n_q * hd, s=H ** -0.5),
     "ln2": torch.ones(H, device=dev, dtype=dt), "gate_up": mk(2 * I, H, s=H ** -0.5), "down": mk(H, I, s=I ** -0.5)}
h, dout = mk(S, H), mk(S, H)
rope = train.RopeTables(hd, 8192, 1e6, dt, dev)
def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r
t_f, (out, saved) = timed(lambda: train.decoder_layer_forward(h, p, rope, n_q, n_kv, hd))
t_b, _ = timed(lambda: train.decoder_layer_backward(dout, saved, p, rope, n_q, n_kv, hd))
s1, s2 = saved
t_mlp, _ = timed(lambda: train.mlp_block_backward(dout, s2, p["ln2"], p["gate_up"], p["down"]))
qkv, o, lse = s1[2], s1[3], s1[4]
dqkv = torch.empty(S, width, device=dev, dtype=dt)
t_att, _ = timed(lambda: ops.attention_backward(qkv, o, o, lse, dqkv, S, n_q, n_kv, hd ** -0.5))
t_mat, _ = timed(lambda: train.attention_backward_materialised(qkv, o, dqkv, S, n_q, n_kv, hd, hd ** -0.5), n=1)
t_lin, _ = timed(lambda: (train.linear_backward(o, p["o"], dout), train.linear_backward(s1[1], p["qkv"], dqkv, need_db=True)))
fl_dense = 2.0 * S * H * (width + n_q * hd + 3 * I)
print(f"S={S}: forward {t_f:.2f} ms | backward {t_b:.2f} ms = MLP block {t_mlp:.2f} + attention (tiled kernels) {t_att:.2f} [materialised form: {t_mat:.2f}] + qkv / o products {t_lin:.2f} (+ norm, rotary)")
print(f"dense products of the backward: {2 * fl_dense / 1e12:.2f} TFLOP in {t_mlp + t_lin:.2f} ms = {2 * fl_dense / (t_mlp + t_lin) / 1e9:.0f} TF/s (with their transposes and row passes)")
fl_att = 2.0 * S * S * hd * n_q * 7 / 2        # seven products over the causal half
print(f"attention backward: {fl_att / 1e12:.2f} TFLOP (7 causal products) in {t_att:.2f} ms = {fl_att / t_att / 1e9:.0f} TF/s")
