"""Times the language model's training step (v3d/train.py: llm_forward_backward + AdamW) at Qwen2-7B size and the path's sequence
length, random weights: forward with labels, backward, optimizer, per step."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-3d-llm_amd"))
from v3d import ops, train


def build(L=28, H=3584, I=18944, n_q=28, n_kv=4, hd=128, V=152064, dt=torch.bfloat16, dev="cuda"):
    width = (n_q + 2 * n_kv) * hd
    def mk(*shape, s=1.0):
        t = torch.empty(*shape, device=dev, dtype=dt)
        t.normal_(0.0, s)
        return t
    ones = lambda: torch.ones(H, device=dev, dtype=dt)
    layers = [{"ln1": ones(), "qkv": mk(width, H, s=H ** -0.5), "qkv_bias": mk(width, s=0.1), "o": mk(H, n_q * hd, s=H ** -0.5),
               "ln2": ones(), "gate_up": mk(2 * I, H, s=H ** -0.5), "down": mk(H, I, s=I ** -0.5)} for _ in range(L)]
    return {"layers": layers, "norm": ones(), "lm_head": mk(V, H, s=H ** -0.5)}


def run(S=6794, steps=3, L=28, answer_tokens=64):
    n_q, n_kv, hd, V, H = 28, 4, 128, 152064, 3584
    params = build(L=L)
    rope = train.RopeTables(hd, 8192, 1e6, torch.bfloat16, "cuda")
    opt = train.AdamW(params, lr=1e-5)
    x = torch.empty(S, H, device="cuda", dtype=torch.bfloat16).normal_()
    labels = torch.full((S,), -100, dtype=torch.int64, device="cuda")
    labels[S - answer_tokens:] = torch.randint(0, V, (answer_tokens,), device="cuda")
    times = []
    for i in range(steps + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss, dx, grads = train.llm_forward_backward(params, x, labels, rope, n_q, n_kv, hd)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        opt.step(params, grads)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        del grads, dx
        if i: times.append((t1 - t0, t2 - t1))
    fb = sum(t[0] for t in times) / len(times) * 1e3
    ad = sum(t[1] for t in times) / len(times) * 1e3
    n_par = sum(p.numel() for l in params["layers"] for p in l.values()) + params["norm"].numel() + params["lm_head"].numel()
    flops = 6.0 * S * (n_par - 0) + 3.5 * 2.0 * S * S * hd * n_q * L          # 6 N S for the products + causal attention (2 fwd + 5 bwd products, halved)
    return {"ms_forward_backward": fb, "ms_adamw": ad, "ms_step": fb + ad, "tokens_per_s": S / ((fb + ad) * 1e-3), "loss": float(loss), "params": n_par,
            "model_tflops_per_s": flops / ((fb + ad) * 1e-3) / 1e12, "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30, "layers": L, "seq_len": S}


if __name__ == "__main__":
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 28
    print(run(L=L))
