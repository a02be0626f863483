"""Backward of the decoder's dense blocks (BASELINE configs[4]) against torch autograd on the CPU in f32:
  * nn.Linear's dx / dW / db through v3d_gemm on transposed operands (v3d/train.py: linear_backward);
  * Qwen2RMSNorm (modeling_qwen2.py:76-90), Qwen2MLP (:177-189) and the second half of Qwen2DecoderLayer.forward (:783-789);
  * causal GQA attention backward in its first, materialised form (eager Qwen2Attention, :236-327), the rotary's transpose, and one
    whole Qwen2DecoderLayer (:727-801) forward + every gradient, at a small width and at the 7B model's true width.
The reference side is the reference's own formulae restated with torch ops in f32 on the SAME 16-bit inputs, differentiated by
autograd; the device side rounds every tensor it stores to 16 bits (as bf16 training does), so the comparison is by tolerance,
stated per test: a norm-wise relative error and an element-wise bound relative to the largest reference magnitude."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from v3d import ops as o
    return o


@pytest.fixture(scope="module")
def train():
    from v3d import train as t
    return t


def _close(got, ref, rel, elem, what, floor=1e-30):
    got, ref = got.float().cpu(), ref.float()
    err = (got - ref).norm() / ref.norm().clamp_min(floor)
    worst = (got - ref).abs().max() / ref.abs().max().clamp_min(floor)
    assert err < rel and worst < elem, f"{what}: norm-wise {err:.3e} (bound {rel}), element-wise {worst:.3e} of max |ref| (bound {elem})"


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,cols,pad", [(5, 8, 8), (70, 136, 128), (6794, 3584, 6848), (129, 64, 192)])
def test_transpose_is_exact_and_zero_pads(ops, dt, rows, cols, pad):
    x = torch.randn(rows, cols, dtype=dt, device="cuda")
    out = torch.full((cols, pad), 7.0, dtype=dt, device="cuda")
    ops.transpose(x, out_cols=pad, out=out)
    assert torch.equal(out[:, :rows], x.t())
    assert not bool(out[:, rows:].any())


def test_transpose_strided_source_and_destination(ops):
    big = torch.randn(100, 256, dtype=torch.bfloat16, device="cuda")
    x = big[:, 64:192]                                   # row stride 256, 128 columns
    dst = torch.zeros(128, 160, dtype=torch.bfloat16, device="cuda")
    ops.transpose(x, out_cols=104, out=dst[:, :104])
    assert torch.equal(dst[:, :100], x.t()) and not bool(dst[:, 100:].any())


def test_transpose_rejects_unaligned_shapes(ops):
    from v3d._native import V3DError
    with pytest.raises(V3DError):
        ops.transpose(torch.zeros(8, 12, dtype=torch.bfloat16, device="cuda"))          # cols % 8
    with pytest.raises(V3DError):
        ops.transpose(torch.zeros(8, 16, dtype=torch.float32, device="cuda"))           # 16-bit only


@pytest.mark.parametrize("rows,cols", [(1, 8), (33, 4608), (6794, 3584)])
def test_colsum_matches_f64_and_is_deterministic(ops, rows, cols):
    x = torch.randn(rows, cols, dtype=torch.bfloat16, device="cuda")
    a = ops.colsum(x, dtype=torch.float32)
    b = ops.colsum(x, dtype=torch.float32)
    assert torch.equal(a, b)
    ref = x.double().sum(0).cpu()
    assert float((a.double().cpu() - ref).abs().max()) <= 1e-5 * float(x.double().abs().sum(0).max())       # f32 accumulation
    assert torch.equal(ops.colsum(x), a.to(torch.bfloat16))                                                  # 16-bit output = one rounding


def _rmsnorm_ref(x, w, eps):
    # Qwen2RMSNorm.forward, modeling_qwen2.py:85-90 (f32 inside; the reference's cast to the input dtype is the device side's rounding)
    var = x.pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(var + eps))


@pytest.mark.parametrize("rows,cols,with_add", [(70, 256, False), (33, 3584, True), (300, 3584, False)])
def test_rmsnorm_grad_matches_autograd(ops, rows, cols, with_add):
    g = torch.Generator().manual_seed(rows + cols)
    x = (torch.randn(rows, cols, generator=g) * 1.5).to(torch.bfloat16)
    w = (1 + 0.2 * torch.randn(cols, generator=g)).to(torch.bfloat16)
    dy = torch.randn(rows, cols, generator=g).to(torch.bfloat16)
    add = torch.randn(rows, cols, generator=g).to(torch.bfloat16) if with_add else None
    xr, wr = x.float().requires_grad_(), w.float().requires_grad_()
    _rmsnorm_ref(xr, wr, 1e-6).backward(dy.float())
    ref_dx = xr.grad + (add.float() if with_add else 0)
    dx, dw = ops.rmsnorm_grad(x.cuda(), w.cuda(), dy.cuda(), 1e-6, add=add.cuda() if with_add else None, dw_dtype=torch.float32)
    _close(dx, ref_dx, 6e-3, 1.5e-2, "dx")               # 16-bit output: 2^-9 per element
    _close(dw, wr.grad, 4e-3, 6e-3, "dweight")           # f32 sums of products with the 16-bit normalised row
    dx2, dw2 = ops.rmsnorm_grad(x.cuda(), w.cuda(), dy.cuda(), 1e-6, add=add.cuda() if with_add else None, dw_dtype=torch.float32)
    assert torch.equal(dx, dx2) and torch.equal(dw, dw2)                                # fixed summation order


def test_swiglu_forward_and_gradient(ops):
    g = torch.Generator().manual_seed(3)
    rows, inter = 77, 1192                                # inter % 8 == 0, not a multiple of 64
    gu = (torch.randn(rows, 2 * inter, generator=g) * 2).to(torch.bfloat16)
    dh = torch.randn(rows, inter, generator=g).to(torch.bfloat16)
    # Qwen2MLP.forward, modeling_qwen2.py:188: act_fn(gate) * up, each a bf16 tensor
    ref_h = (F.silu(gu[:, :inter].float()).to(torch.bfloat16).float() * gu[:, inter:].float()).to(torch.bfloat16)
    h = ops.swiglu(gu.cuda())
    ulp = (h.cpu().view(torch.int16).int() - ref_h.view(torch.int16).int()).abs().max()
    assert int(ulp) <= 1                                  # exp is the only inexact step
    gr = gu.float().requires_grad_()
    (F.silu(gr[:, :inter]) * gr[:, inter:]).backward(dh.float())
    dgu = ops.swiglu_grad(gu.cuda(), dh.cuda())
    _close(dgu, gr.grad, 6e-3, 1.5e-2, "dgu")


@pytest.mark.parametrize("M,K,N", [(300, 256, 384), (520, 3584, 4608), (64, 18944, 3584)])
def test_linear_backward_matches_f32_products(ops, train, M, K, N):
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16)
    dy = torch.randn(M, N, generator=g).to(torch.bfloat16)
    res = torch.randn(M, K, generator=g).to(torch.bfloat16)
    dx, dw, db = train.linear_backward(x.cuda(), w.cuda(), dy.cuda(), res=res.cuda(), need_db=True)
    _close(dx, dy.float() @ w.float() + res.float(), 5e-3, 1.5e-2, "dx")
    _close(dw, dy.float().t() @ x.float(), 5e-3, 1.5e-2, "dW")
    _close(db, dy.float().sum(0), 5e-3, 1e-2, "db")
    dx_only, none_w, none_b = train.linear_backward(x.cuda(), w.cuda(), dy.cuda(), need_dw=False)
    assert none_w is None and none_b is None
    _close(dx_only, dy.float() @ w.float(), 5e-3, 1.5e-2, "dx without residual")


def _mlp_block_ref(h, ln_w, w_gu, w_down, eps):
    # Qwen2DecoderLayer.forward, modeling_qwen2.py:783-789, with Qwen2MLP :188 and Qwen2RMSNorm :85-90
    inter = w_gu.shape[0] // 2
    n = _rmsnorm_ref(h, ln_w, eps)
    gu = n @ w_gu.t()
    a = F.silu(gu[:, :inter]) * gu[:, inter:]
    return h + a @ w_down.t()


@pytest.mark.parametrize("S,H,I", [(200, 256, 512), (300, 3584, 18944)])
def test_mlp_block_forward_and_backward_match_autograd(train, S, H, I):
    """The MLP half of a Qwen2 decoder layer, forward and backward, at a small width and at the 7B model's true width."""
    g = torch.Generator().manual_seed(S + H)
    h = torch.randn(S, H, generator=g).to(torch.bfloat16)
    ln_w = (1 + 0.1 * torch.randn(H, generator=g)).to(torch.bfloat16)
    w_gu = (torch.randn(2 * I, H, generator=g) * H ** -0.5).to(torch.bfloat16)
    w_down = (torch.randn(H, I, generator=g) * I ** -0.5).to(torch.bfloat16)
    dout = torch.randn(S, H, generator=g).to(torch.bfloat16)
    leaves = [t.float().requires_grad_() for t in (h, ln_w, w_gu, w_down)]
    ref_out = _mlp_block_ref(*leaves, 1e-6)
    ref_out.backward(dout.float())
    dev = [t.cuda() for t in (h, ln_w, w_gu, w_down)]
    out, saved = train.mlp_block_forward(*dev)
    _close(out, ref_out.detach(), 6e-3, 2e-2, "forward")
    dh, grads = train.mlp_block_backward(dout.cuda(), saved, dev[1], dev[2], dev[3])
    _close(dh, leaves[0].grad, 1e-2, 3e-2, "dh")
    _close(grads["ln"], leaves[1].grad, 1e-2, 3e-2, "d ln weight")
    _close(grads["gate_up"], leaves[2].grad, 1e-2, 3e-2, "d gate/up weight")
    _close(grads["down"], leaves[3].grad, 1e-2, 3e-2, "d down weight")


# ------------------------------------------------------------------------------ attention block, whole decoder layer


def _rope_ref(x, n_heads, hd, base=1e6):
    # Qwen2RotaryEmbedding + apply_rotary_pos_emb, modeling_qwen2.py:93-173 (positions 0..S-1), f32
    S = x.shape[0]
    inv = 1.0 / (base ** (torch.arange(0, hd, 2, dtype=torch.int64).float() / hd))
    ang = torch.arange(S, dtype=torch.float32)[:, None] * inv[None, :]
    cos, sin = torch.cat([ang, ang], -1).cos()[:, None, :], torch.cat([ang, ang], -1).sin()[:, None, :]
    xh = x.view(S, n_heads, hd)
    rot = torch.cat([-xh[..., hd // 2:], xh[..., :hd // 2]], -1)
    return (xh * cos + rot * sin).reshape(S, n_heads * hd)


def _attention_ref(q, k, v, n_q, n_kv, hd):
    # eager Qwen2Attention.forward, modeling_qwen2.py:248-327: repeat_kv (:236-245), causal mask, softmax in f32
    S = q.shape[0]
    qh = q.view(S, n_q, hd).transpose(0, 1)
    kh = k.view(S, n_kv, hd).transpose(0, 1).repeat_interleave(n_q // n_kv, 0)
    vh = v.view(S, n_kv, hd).transpose(0, 1).repeat_interleave(n_q // n_kv, 0)
    s = qh @ kh.transpose(1, 2) / hd ** 0.5
    s = s + torch.full((S, S), float("-inf")).triu(1)
    return (torch.softmax(s, -1) @ vh).transpose(0, 1).reshape(S, n_q * hd)


@pytest.mark.parametrize("S,dt", [(150, torch.bfloat16), (257, torch.bfloat16), (700, torch.bfloat16), (40, torch.bfloat16), (64, torch.bfloat16),
                                  (128, torch.bfloat16), (1, torch.bfloat16), (300, torch.float16)])
def test_attention_backward_matches_autograd(ops, train, S, dt):
    n_q, n_kv, hd = 4, 2, 128
    g = torch.Generator().manual_seed(S)
    width = (n_q + 2 * n_kv) * hd
    qkv = torch.randn(S, width, generator=g).to(dt)
    do = torch.randn(S, n_q * hd, generator=g).to(dt)
    ref = qkv.float().requires_grad_()
    _attention_ref(ref[:, :n_q * hd], ref[:, n_q * hd:(n_q + n_kv) * hd], ref[:, (n_q + n_kv) * hd:], n_q, n_kv, hd).backward(do.float())
    Sp = (S + 127) // 128 * 128
    dev = torch.zeros(Sp, width, dtype=dt, device="cuda")
    dev[:S] = qkv.cuda()
    dqkv = torch.full((S, width), float("nan"), dtype=dt, device="cuda")
    train.attention_backward_materialised(dev, do.cuda(), dqkv, S, n_q, n_kv, hd, hd ** -0.5)
    floor = 1.0 if S == 1 else 1e-30           # one key: dq = dk = 0 exactly in the reference, compare absolutely
    parts = (("dq", slice(0, n_q * hd)), ("dk", slice(n_q * hd, (n_q + n_kv) * hd)), ("dv", slice((n_q + n_kv) * hd, width)))
    for name, sl in parts:
        _close(dqkv[:, sl], ref.grad[:, sl], 1.2e-2, 3e-2, name + " (materialised)", floor)          # p, dp, ds are 16-bit tensors (2^-9 each)
    # the tiled kernels: forward with the row log-sum-exp, then the backward that recomputes the probabilities
    o = torch.empty(S, n_q * hd, dtype=dt, device="cuda")
    lse = ops.attention_train(dev, o, S, n_q, n_kv, hd ** -0.5)
    o_plain = torch.empty_like(o)
    ops.attention(dev, dev[:, n_q * hd:], dev[:, (n_q + n_kv) * hd:], o_plain, 1, S, S, n_q, n_kv, hd, hd, width, width, width, n_q * hd,
                  0, 0, 0, hd, hd, hd, True, 0, hd ** -0.5)
    if S > 8:                                                                            # (v3d_attention sends <= 8 rows to its decode kernel)
        assert torch.equal(o, o_plain)                                                   # same kernel, one more store
    s_ref = (qkv[:, :n_q * hd].float().view(S, n_q, hd).transpose(0, 1) @
             qkv[:, n_q * hd:(n_q + n_kv) * hd].float().view(S, n_kv, hd).transpose(0, 1).repeat_interleave(n_q // n_kv, 0).transpose(1, 2)) * hd ** -0.5
    lse_ref = torch.logsumexp(s_ref + torch.full((S, S), float("-inf")).triu(1), -1) * 1.4426950408889634
    assert float((lse.cpu() - lse_ref).abs().max()) < 2e-2                               # scaled log2 units; c q is rounded to 16 bit
    dq2 = torch.full((S, width), float("nan"), dtype=dt, device="cuda")
    ops.attention_backward(dev, o, do.cuda(), lse, dq2, S, n_q, n_kv, hd ** -0.5)
    for name, sl in parts:
        _close(dq2[:, sl], ref.grad[:, sl], 1.2e-2, 3e-2, name + " (tiled)", floor)
    dq3 = torch.empty_like(dq2)
    ops.attention_backward(dev, o, do.cuda(), lse, dq3, S, n_q, n_kv, hd ** -0.5)
    assert torch.equal(dq2, dq3)                                                         # no atomics: run-to-run identical


def _layer_ref(h, p, n_q, n_kv, hd, eps):
    # Qwen2DecoderLayer.forward, modeling_qwen2.py:727-801
    n = _rmsnorm_ref(h, p["ln1"], eps)
    qkv = n @ p["qkv"].t() + p["qkv_bias"]
    q = _rope_ref(qkv[:, :n_q * hd], n_q, hd)
    k = _rope_ref(qkv[:, n_q * hd:(n_q + n_kv) * hd], n_kv, hd)
    o = _attention_ref(q, k, qkv[:, (n_q + n_kv) * hd:], n_q, n_kv, hd)
    mid = h + o @ p["o"].t()
    return _mlp_block_ref(mid, p["ln2"], p["gate_up"], p["down"], eps)


@pytest.mark.parametrize("S,H,I,n_q,n_kv", [(200, 512, 1024, 4, 2), (333, 512, 1024, 4, 2), (300, 3584, 18944, 28, 4)])
def test_decoder_layer_forward_and_backward_match_autograd(train, S, H, I, n_q, n_kv):
    """One Qwen2 decoder layer, forward and every gradient, against autograd in f32: GQA 4/2 heads of 128, hidden 512, MLP 1024, and
    the 7B model's own layer (hidden 3584, 28/4 heads, MLP 18944)."""
    hd = 128
    g = torch.Generator().manual_seed(S)
    width = (n_q + 2 * n_kv) * hd
    mk = lambda *shape, s=1.0: (torch.randn(*shape, generator=g) * s).to(torch.bfloat16)
    p = {"ln1": (1 + 0.1 * torch.randn(H, generator=g)).to(torch.bfloat16), "qkv": mk(width, H, s=H ** -0.5), "qkv_bias": mk(width, s=0.5),
         "o": mk(H, n_q * hd, s=(n_q * hd) ** -0.5), "ln2": (1 + 0.1 * torch.randn(H, generator=g)).to(torch.bfloat16),
         "gate_up": mk(2 * I, H, s=H ** -0.5), "down": mk(H, I, s=I ** -0.5)}
    h, dout = mk(S, H), mk(S, H)
    leaves = {k: v.float().requires_grad_() for k, v in p.items()}
    hr = h.float().requires_grad_()
    ref_out = _layer_ref(hr, leaves, n_q, n_kv, hd, 1e-6)
    ref_out.backward(dout.float())
    rope = train.RopeTables(hd, 512, 1e6, torch.bfloat16, "cuda")
    dev = {k: v.cuda() for k, v in p.items()}
    out, saved = train.decoder_layer_forward(h.cuda(), dev, rope, n_q, n_kv, hd)
    _close(out, ref_out.detach(), 8e-3, 3e-2, "forward")
    for materialised in (False, True):           # attention backward as tiled kernels (default) and in its materialised first form
        dh, grads = train.decoder_layer_backward(dout.cuda(), saved, dev, rope, n_q, n_kv, hd, materialised=materialised)
        _close(dh, hr.grad, 1.5e-2, 4e-2, "dh")
        for k in p:
            _close(grads[k], leaves[k].grad, 1.5e-2, 4e-2, "d " + k)


# ------------------------------------------------------------------------------ optimizer, embedding gradient, the language model's step


def test_adamw_step_matches_torch_adamw(ops):
    g = torch.Generator().manual_seed(11)
    n = 5000
    p0 = torch.randn(n, generator=g)
    ref_p = p0.clone().requires_grad_()
    opt = torch.optim.AdamW([ref_p], lr=1e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05)
    p32, m, v = p0.clone().cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    p16 = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    for step in range(1, 4):
        grad = torch.randn(n, generator=g).to(torch.bfloat16)
        ref_p.grad = grad.float() * 0.5
        opt.step()
        ops.adamw_step(p32, m, v, grad.cuda(), p16=p16, lr=1e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05, step=step, grad_scale=0.5)
        assert torch.allclose(p32.cpu(), ref_p.detach(), rtol=2e-6, atol=2e-7), step          # f32 arithmetic in the same order
        assert torch.equal(p16, p32.to(torch.bfloat16))
    assert torch.allclose(m.cpu(), opt.state[ref_p]["exp_avg"], rtol=2e-6, atol=1e-7)
    assert torch.allclose(v.cpu(), opt.state[ref_p]["exp_avg_sq"], rtol=2e-6, atol=1e-9)


def test_sumsq_is_deterministic_and_matches_f64(ops):
    """v3d_sumsq (the global gradient norm's building block): f32 / bf16 / f16, odd lengths, chained accumulation; run-to-run identical."""
    g = torch.Generator().manual_seed(21)
    for dt, n in ((torch.float32, 100003), (torch.bfloat16, 3 * 4096 + 5), (torch.float16, 8), (torch.bfloat16, 37_000_000)):
        x = torch.randn(n, generator=g).to(dt).cuda()
        a, b = ops.sumsq(x).clone(), ops.sumsq(x).clone()
        assert torch.equal(a, b)
        want = x.double().pow(2).sum().item()
        assert abs(a.item() - want) <= 2e-6 * want
    parts = [torch.randn(k, generator=g).to(torch.bfloat16).cuda() for k in (16, 4096, 999)]
    acc = None
    for t in parts:
        acc = ops.sumsq(t, out=acc, accumulate=acc is not None)
    want = sum(t.double().pow(2).sum().item() for t in parts)
    assert abs(acc.item() - want) <= 2e-6 * want


def test_optimizer_groups_clipping_and_schedule_match_torch(ops, train):
    """r04, the trainer-side fidelity of f4 (VERDICT r03 missing #4): per-module learning rates (llava_trainer.py:446-523; train_multi.sh:45
    --mm_vision_tower_lr 2e-6 beside :65 --learning_rate 1e-5), no weight decay on biases / LayerNorm parameters (:459-460), global
    gradient-norm clipping (HF max_grad_norm 1.0 = zero2.json:36 "gradient_clipping": "auto"), the cosine schedule with 3 % warm-up
    (train_multi.sh:67-68) - train.AdamW against torch.optim.AdamW with the reference's parameter groups + torch.nn.utils.clip_grad_norm_
    + transformers' get_cosine_schedule_with_warmup on f32 copies of the same parameters and the same 16-bit gradients."""
    from transformers import get_cosine_schedule_with_warmup
    g = torch.Generator().manual_seed(31)
    mk = lambda *s_, sc=0.1: (torch.randn(*s_, generator=g) * sc).to(torch.bfloat16).cuda()      # noqa: E731
    tree = {"vision": {"patch_w": mk(32, 64), "patch_b": mk(32), "layers": [{"ln1_w": mk(32), "ln1_b": mk(32), "qkv": mk(96, 32), "qkv_b": mk(96)}]},
            "projector": {"w1": mk(48, 32), "b1": mk(48)}, "newline": mk(48),
            "llm": {"layers": [{"ln1": mk(48), "qkv": mk(144, 48), "qkv_bias": mk(144)}], "norm": mk(48), "lm_head": mk(64, 48)}}
    total, lr, wd = 40, 1e-3, 0.05
    by = {"vision_tower": 2e-4, "mm_projector": 5e-4}                       # the reference's module keywords
    opt = train.AdamW(tree, lr=lr, weight_decay=wd, lr_by_module=by, max_grad_norm=1.0, schedule=train.cosine_warmup_schedule(total, 0.03))
    paths = train._paths(tree)
    leaves = train._leaves(tree)
    assert [p_ for p_ in paths if train._no_decay(p_)] == ["vision.patch_b", "vision.layers.0.ln1_w", "vision.layers.0.ln1_b", "vision.layers.0.qkv_b",
                                                           "projector.b1", "llm.layers.0.qkv_bias"]       # (Qwen2's RMSNorm weights ARE decayed)
    ref = [t.detach().float().clone().requires_grad_() for t in leaves]
    groups = []
    for path, t in zip(paths, ref):
        top = path.split(".")[0]
        groups.append({"params": [t], "lr": {"vision": 2e-4, "projector": 5e-4}.get(top, lr), "weight_decay": 0.0 if train._no_decay(path) else wd})
    topt = torch.optim.AdamW(groups, lr=lr, betas=(0.9, 0.999), eps=1e-8)
    sched = get_cosine_schedule_with_warmup(topt, num_warmup_steps=2, num_training_steps=total)      # ceil(0.03 * 40) = 2
    for step in range(1, 7):
        grads = train._tree_map(lambda t: (torch.randn(t.shape, generator=g) * (3.0 if step % 2 else 0.01)).to(torch.bfloat16).cuda(), tree)
        for t, gr in zip(ref, train._leaves(grads)):
            t.grad = gr.float() * 0.5                                                     # grad_scale 0.5: two accumulated micro-batches
        norm = torch.nn.utils.clip_grad_norm_(ref, 1.0)
        topt.step()
        sched.step()
        opt.step(tree, grads, grad_scale=0.5)
        assert abs(opt.last_grad_norm - norm.item()) <= 1e-5 * norm.item()
        for path, (p32, _, _), t in zip(paths, train._leaves_of_state(opt.state), ref):
            assert torch.allclose(p32.cpu(), t.detach().cpu(), rtol=3e-6, atol=3e-7), (step, path)
    lam = train.cosine_warmup_schedule(total, 0.03)
    assert lam(1) == 0.0 and lam(2) == 0.5 and lam(3) == 1.0 and abs(lam(total + 1)) < 1e-12


def test_adamw_eight_wide_form_equals_the_scalar_form_bitwise(ops):
    """r03: 16-bit gradients with n % 8 == 0 take the kernel with 16-byte loads / stores; every element goes through the same
    arithmetic, so three steps leave p32 / m / v / p16 bit-identical to the scalar kernel's (taken by an odd-length prefix view)."""
    g = torch.Generator().manual_seed(12)
    n = 3 * 4096 + 8
    for gdt, pdt in ((torch.bfloat16, torch.bfloat16), (torch.float16, torch.float16)):
        p0 = torch.randn(n, generator=g)
        grads = [torch.randn(n, generator=g).to(gdt).cuda() for _ in range(3)]
        state = []
        for length in (n, n - 1):                           # n: eight-wide; n - 1: scalar
            p32, m, v = p0[:length].clone().cuda(), torch.zeros(length, device="cuda"), torch.zeros(length, device="cuda")
            p16 = torch.zeros(length, dtype=pdt, device="cuda")
            for t, gr in enumerate(grads):
                ops.adamw_step(p32, m, v, gr[:length].contiguous(), p16=p16, lr=1e-2, betas=(0.9, 0.95), weight_decay=0.05, step=t + 1, grad_scale=0.5)
            state.append((p32, m, v, p16))
        for a_, b_ in zip(*state):
            assert torch.equal(a_[: n - 1], b_)


def test_embed_grad_sums_repeated_tokens(ops):
    g = torch.Generator().manual_seed(5)
    S, H, V = 40, 264, 50
    dh = torch.randn(S, H, generator=g).to(torch.bfloat16)
    rows = torch.tensor([0, 1, 2, 30, 31, 32, 33, 39])
    ids = torch.tensor([7, 3, 7, 7, 49, 0, 3, 12])
    ref = torch.zeros(V, H).index_add_(0, ids, dh[rows].float()).to(torch.bfloat16)
    dE = torch.zeros(V, H, dtype=torch.bfloat16, device="cuda")
    ops.embed_grad(dh.cuda(), rows.cuda(), ids.cuda(), dE)
    assert torch.equal(dE.cpu(), ref)                      # f32 sums of at most three 16-bit rows, one rounding: exact against index_add_
    # ids outside the table (IMAGE_TOKEN_INDEX = -200 of a raw prompt, ids >= vocab) and rows outside dh are skipped, not dereferenced
    # (ADVICE r2): the rows of dE around the table stay as they were
    ids_raw = torch.tensor([7, -200, 7, 7, 49, V, 3, 12])
    rows_raw = torch.tensor([0, 1, 2, 30, 31, 32, 33, S + 5])
    keep = (ids_raw >= 0) & (ids_raw < V) & (rows_raw < S)
    ref2 = torch.zeros(V, H).index_add_(0, ids_raw[keep], dh[rows_raw[keep]].float()).to(torch.bfloat16)
    big = torch.full((V + 2, H), 7.0, dtype=torch.bfloat16, device="cuda")
    dE2 = big[1:V + 1]
    dE2.zero_()
    ops.embed_grad(dh.cuda(), rows_raw.cuda(), ids_raw.cuda(), dE2)
    assert torch.equal(dE2.cpu(), ref2) and bool((big[0] == 7).all()) and bool((big[V + 1] == 7).all())


def test_llm_step_loss_and_gradients_match_autograd(ops, train):
    """Qwen2ForCausalLM.forward with labels + backward (modeling_qwen2.py:1145-1217) on a 2-layer model, then one AdamW update."""
    H, I, n_q, n_kv, hd, V, S, L = 512, 1024, 4, 2, 128, 1024, 200, 2
    g = torch.Generator().manual_seed(21)
    width = (n_q + 2 * n_kv) * hd
    mk = lambda *shape, s=1.0: (torch.randn(*shape, generator=g) * s).to(torch.bfloat16)
    ln = lambda: (1 + 0.1 * torch.randn(H, generator=g)).to(torch.bfloat16)
    layers = [{"ln1": ln(), "qkv": mk(width, H, s=H ** -0.5), "qkv_bias": mk(width, s=0.3), "o": mk(H, n_q * hd, s=(n_q * hd) ** -0.5),
               "ln2": ln(), "gate_up": mk(2 * I, H, s=H ** -0.5), "down": mk(H, I, s=I ** -0.5)} for _ in range(L)]
    params = {"layers": layers, "norm": ln(), "lm_head": mk(V, H, s=H ** -0.5)}
    x = mk(S, H)
    labels = torch.full((S,), -100, dtype=torch.int64)
    labels[150:] = torch.randint(0, V, (50,), generator=g)
    ref = train._tree_map(lambda t: t.float().requires_grad_(), params)
    xr = x.float().requires_grad_()
    h = xr
    for p in ref["layers"]:
        h = _layer_ref(h, p, n_q, n_kv, hd, 1e-6)
    logits = _rmsnorm_ref(h, ref["norm"], 1e-6) @ ref["lm_head"].t()
    ref_loss = F.cross_entropy(logits[:-1], labels[1:], ignore_index=-100)       # modeling_qwen2.py:1195-1205
    ref_loss.backward()
    rope = train.RopeTables(hd, 512, 1e6, torch.bfloat16, "cuda")
    dev = train._tree_map(lambda t: t.cuda(), params)
    loss, dx, grads = train.llm_forward_backward(dev, x.cuda(), labels.cuda(), rope, n_q, n_kv, hd)
    assert abs(float(loss) - float(ref_loss.detach())) < 2e-2 * float(ref_loss.detach())
    _close(dx, xr.grad, 2.5e-2, 6e-2, "d inputs_embeds")
    _close(grads["norm"], ref["norm"].grad, 2.5e-2, 6e-2, "d norm")
    _close(grads["lm_head"], ref["lm_head"].grad, 2.5e-2, 6e-2, "d lm_head")
    for i in range(L):
        for k in layers[i]:
            _close(grads["layers"][i][k], ref["layers"][i][k].grad, 2.5e-2, 6e-2, f"layer {i} d {k}")
    opt = train.AdamW(dev, lr=1e-3)
    before = dev["layers"][0]["down"].clone()
    opt.step(dev, grads)
    moved = (dev["layers"][0]["down"].float() - before.float()).abs()
    assert float(moved.max()) > 0 and float(moved.max()) < 4e-3           # |update| <= lr on the first step (+ one 16-bit rounding)


# ------------------------------------------------------------------------------ projector, splice


@pytest.mark.parametrize("tanh_form", [False, True])
def test_gelu_forward_and_gradient(ops, tanh_form):
    g = torch.Generator().manual_seed(9)
    z = (torch.randn(90, 1160, generator=g) * 2).to(torch.bfloat16)
    dy = torch.randn(90, 1160, generator=g).to(torch.bfloat16)
    zr = z.float().requires_grad_()
    ref = F.gelu(zr, approximate="tanh" if tanh_form else "none")
    ref.backward(dy.float())
    out = ops.gelu(z.cuda(), tanh_form)
    diff = (out.float().cpu() - ref.detach()).abs()
    assert bool((diff <= 2.0 ** -8 * ref.detach().abs() + 1e-6).all())      # one 16-bit rounding; 1 + erf(x) cancels for x << 0 (values < 1e-5)
    _close(ops.gelu_grad(z.cuda(), dy.cuda(), tanh_form), zr.grad, 4e-3, 1e-2, "dz")


def test_projector_forward_and_backward_match_autograd(train):
    """mlp2x_gelu at its true widths (1152 -> 3584 -> 3584, multimodal_projector/builder.py:41-48) on two frames of patch features."""
    g = torch.Generator().manual_seed(13)
    rows, C, H = 2 * 729, 1152, 3584
    mk = lambda *shape, s=1.0: (torch.randn(*shape, generator=g) * s).to(torch.bfloat16)
    x, w1, b1, w2, b2, dy = mk(rows, C), mk(H, C, s=C ** -0.5), mk(H, s=0.2), mk(H, H, s=H ** -0.5), mk(H, s=0.2), mk(rows, H)
    leaves = [t.float().requires_grad_() for t in (x, w1, b1, w2, b2)]
    ref = F.gelu(leaves[0] @ leaves[1].t() + leaves[2]) @ leaves[3].t() + leaves[4]
    ref.backward(dy.float())
    dev = [t.cuda() for t in (x, w1, b1, w2, b2)]
    y, saved = train.projector_forward(*dev)
    _close(y, ref.detach(), 6e-3, 2e-2, "forward")
    dx, grads = train.projector_backward(dy.cuda(), saved, dev[1], dev[3])
    _close(dx, leaves[0].grad, 1e-2, 3e-2, "dx")
    for k, leaf in (("w1", leaves[1]), ("b1", leaves[2]), ("w2", leaves[3]), ("b2", leaves[4])):
        _close(grads[k], leaf.grad, 1e-2, 3e-2, "d " + k)


def test_inputs_embeds_backward_routes_rows(ops, train):
    """The splice's backward: visual rows -> v3d_visual_tokens_grad, text rows -> embedding rows summed per token id."""
    frames, H, n_pre, n_post, vocab = 2, 256, 5, 7, 64
    n_vis = frames * 14 * 15
    S = n_pre + n_vis + n_post
    g = torch.Generator().manual_seed(2)
    dx = torch.randn(S, H, generator=g).to(torch.bfloat16).cuda()
    rows = torch.cat([torch.arange(n_pre), torch.arange(n_pre + n_vis, S)]).cuda()
    ids = torch.tensor([3, 9, 3, 1, 0, 9, 9, 5, 63, 2, 3, 7]).cuda()
    dE = torch.zeros(vocab, H, dtype=torch.bfloat16, device="cuda")
    dfeat, dnl = train.inputs_embeds_backward(dx, n_pre, frames, rows, ids, dE)
    ref_feat, ref_nl = ops.visual_tokens_grad(dx[n_pre:n_pre + n_vis].contiguous(), frames)
    assert torch.equal(dfeat, ref_feat) and torch.equal(dnl, ref_nl)           # (that kernel's own parity: tests/test_gpu_train_kernels.py)
    ref_E = torch.zeros(vocab, H).index_add_(0, ids.cpu(), dx[rows].float().cpu()).to(torch.bfloat16)
    assert torch.equal(dE.cpu(), ref_E)


# ------------------------------------------------------------------------------ the SigLIP tower's layers


@pytest.mark.parametrize("rows,cols,with_add", [(70, 1152, True), (1458, 1152, False), (33, 256, False)])
def test_layernorm_grad_matches_autograd(ops, rows, cols, with_add):
    g = torch.Generator().manual_seed(rows)
    x = (torch.randn(rows, cols, generator=g) * 1.5 + 0.3).to(torch.bfloat16)
    w = (1 + 0.2 * torch.randn(cols, generator=g)).to(torch.bfloat16)
    b = (0.1 * torch.randn(cols, generator=g)).to(torch.bfloat16)
    dy = torch.randn(rows, cols, generator=g).to(torch.bfloat16)
    add = torch.randn(rows, cols, generator=g).to(torch.bfloat16) if with_add else None
    xr, wr, br = x.float().requires_grad_(), w.float().requires_grad_(), b.float().requires_grad_()
    F.layer_norm(xr, (cols,), wr, br, 1e-6).backward(dy.float())          # nn.LayerNorm, siglip_encoder.py:272-274
    dx, dw, db = ops.layernorm_grad(x.cuda(), w.cuda(), dy.cuda(), 1e-6, add=add.cuda() if with_add else None, dw_dtype=torch.float32)
    _close(dx, xr.grad + (add.float() if with_add else 0), 6e-3, 1.5e-2, "dx")
    _close(dw, wr.grad, 2e-3, 4e-3, "dweight")
    _close(db, br.grad, 1e-4, 1e-4, "dbias")                                # plain f32 column sums of 16-bit values


def _siglip_layer_ref(x, sd, frames, tokens, heads):
    # SigLipEncoderLayer.forward, siglip_encoder.py:264-305, with SigLipAttention :197-250 and SigLipMLP :253-262
    H = x.shape[1]
    hd = H // heads
    n1 = F.layer_norm(x, (H,), sd["ln1_w"], sd["ln1_b"], 1e-6)
    q, k, v = (F.linear(n1, sd[n + "_w"], sd[n + "_b"]).view(frames, tokens, heads, hd).transpose(1, 2) for n in ("q", "k", "v"))
    p = torch.softmax(q @ k.transpose(2, 3) * hd ** -0.5, -1)
    o = (p @ v).transpose(1, 2).reshape(frames * tokens, H)
    mid = x + F.linear(o, sd["o_w"], sd["o_b"])
    n2 = F.layer_norm(mid, (H,), sd["ln2_w"], sd["ln2_b"], 1e-6)
    return mid + F.linear(F.gelu(F.linear(n2, sd["fc1_w"], sd["fc1_b"]), approximate="tanh"), sd["fc2_w"], sd["fc2_b"])


def test_siglip_layer_forward_and_backward_match_autograd(train):
    """One SigLIP encoder layer at its true width (hidden 1152, 16 heads of 72, MLP 4304) on two frames of 729 patches: the training
    layout pads the heads to 128 and the MLP to 4352; forward and every gradient (un-padded) against autograd in f32."""
    frames, tokens, H, heads, inter = 2, 729, 1152, 16, 4304
    g = torch.Generator().manual_seed(31)
    mk = lambda *shape, s=1.0: (torch.randn(*shape, generator=g) * s).to(torch.bfloat16)
    sd = {"ln1_w": (1 + 0.1 * torch.randn(H, generator=g)).to(torch.bfloat16), "ln1_b": mk(H, s=0.1),
          "ln2_w": (1 + 0.1 * torch.randn(H, generator=g)).to(torch.bfloat16), "ln2_b": mk(H, s=0.1),
          "q_w": mk(H, H, s=H ** -0.5), "q_b": mk(H, s=0.2), "k_w": mk(H, H, s=H ** -0.5), "k_b": mk(H, s=0.2),
          "v_w": mk(H, H, s=H ** -0.5), "v_b": mk(H, s=0.2), "o_w": mk(H, H, s=H ** -0.5), "o_b": mk(H, s=0.2),
          "fc1_w": mk(inter, H, s=H ** -0.5), "fc1_b": mk(inter, s=0.2), "fc2_w": mk(H, inter, s=inter ** -0.5), "fc2_b": mk(H, s=0.2)}
    x, dout = mk(frames * tokens, H), mk(frames * tokens, H)
    leaves = {k: v.float().requires_grad_() for k, v in sd.items()}
    xr = x.float().requires_grad_()
    ref = _siglip_layer_ref(xr, leaves, frames, tokens, heads)
    ref.backward(dout.float())
    p = train.siglip_pad_layer({k: v.cuda() for k, v in sd.items()})
    out, saved = train.siglip_layer_forward(x.cuda(), p, frames)
    _close(out, ref.detach(), 8e-3, 3e-2, "forward")
    dx, grads = train.siglip_layer_backward(dout.cuda(), saved, p, frames)
    _close(dx, xr.grad, 1.5e-2, 4e-2, "dx")
    pad = grads["qkv"].view(3, heads, 128, H)[:, :, 72:]
    assert not bool(pad.any()) and not bool(grads["o"].view(H, heads, 128)[:, :, 72:].any()) and not bool(grads["fc1"][inter:].any())   # the padding learns nothing
    real = train.siglip_unpad_grads(grads)
    for k in sd:
        # the key bias has NO gradient (a constant added to every key shifts a query's scores alike: softmax does not see it): the
        # reference's is f32 noise, the device's 16-bit noise - held to the scale of the query bias' gradient instead
        if k == "k_b":
            assert float(real[k].float().abs().max()) <= 0.04 * float(leaves["q_b"].grad.abs().max())
            continue
        _close(real[k], leaves[k].grad, 1.5e-2, 4e-2, "d " + k)


@pytest.mark.parametrize("shape", ["narrow-2+2", "true-width-1+1"])
def test_whole_sample_step_matches_autograd(ops, train, shape, monkeypatch):
    """SigLIP tower (true width, 2 frames) -> mm_projector -> pool + 3-D PE + newline, spliced between text rows -> Qwen2 with labels:
    loss and the gradient of EVERY parameter group against autograd over the reference's composition in f32
    (llava_qwen.py:121-205, llava_arch.py:191-210, 307-328, 506-517, 650-836).  "narrow-2+2": 2 SigLIP layers + a 2-layer Qwen2 at
    hidden 768; "true-width-1+1" (r03): ONE SigLIP layer + ONE Qwen2-7B layer at 3584 / 28 + 4 heads x 128 / 18944 - the widths, tile
    shapes and GQA ratio of the measured configs[4] step."""
    frames, tokens, Hv, heads, inter, kpad = 2, 729, 1152, 16, 4304, 640
    H, I, n_q, n_kv, hd, V, L = (768, 1024, 4, 2, 128, 1024, 2) if shape == "narrow-2+2" else (3584, 18944, 28, 4, 128, 1024, 1)
    n_vit = 2 if shape == "narrow-2+2" else 1
    g = torch.Generator().manual_seed(77)
    mk = lambda *shape, s=1.0: (torch.randn(*shape, generator=g) * s).to(torch.bfloat16)
    ln = lambda n_: (1 + 0.1 * torch.randn(n_, generator=g)).to(torch.bfloat16)

    def vit_layer():
        return {"ln1_w": ln(Hv), "ln1_b": mk(Hv, s=0.1), "ln2_w": ln(Hv), "ln2_b": mk(Hv, s=0.1),
                "q_w": mk(Hv, Hv, s=Hv ** -0.5), "q_b": mk(Hv, s=0.2), "k_w": mk(Hv, Hv, s=Hv ** -0.5), "k_b": mk(Hv, s=0.2),
                "v_w": mk(Hv, Hv, s=Hv ** -0.5), "v_b": mk(Hv, s=0.2), "o_w": mk(Hv, Hv, s=Hv ** -0.5), "o_b": mk(Hv, s=0.2),
                "fc1_w": mk(inter, Hv, s=Hv ** -0.5), "fc1_b": mk(inter, s=0.2), "fc2_w": mk(Hv, inter, s=inter ** -0.5), "fc2_b": mk(Hv, s=0.2)}

    vit = [vit_layer() for _ in range(n_vit)]
    patch_w, patch_b, pos = mk(Hv, kpad, s=588 ** -0.5), mk(Hv, s=0.1), mk(tokens, Hv, s=0.5)
    patch_w[:, 588:] = 0                                                     # the k padding of the patch convolution's GEMM
    proj = {"w1": mk(H, Hv, s=Hv ** -0.5), "b1": mk(H, s=0.1), "w2": mk(H, H, s=H ** -0.5), "b2": mk(H, s=0.1)}
    newline, embed = mk(H, s=0.5), mk(V, H, s=0.5)
    width = (n_q + 2 * n_kv) * hd
    layers = [{"ln1": ln(H), "qkv": mk(width, H, s=H ** -0.5), "qkv_bias": mk(width, s=0.3), "o": mk(H, n_q * hd, s=(n_q * hd) ** -0.5),
               "ln2": ln(H), "gate_up": mk(2 * I, H, s=H ** -0.5), "down": mk(H, I, s=I ** -0.5)} for _ in range(L)]
    llm = {"layers": layers, "norm": ln(H), "lm_head": mk(V, H, s=H ** -0.5)}
    patches = mk(frames * tokens, kpad)
    patches[:, 588:] = 0
    ids = torch.randint(0, 64, (frames, 14, 14, 3), generator=g, dtype=torch.int32)
    pre_ids, post_ids = torch.randint(0, V, (9,), generator=g), torch.randint(0, V, (31,), generator=g)
    post_ids[3] = pre_ids[2]                                                 # a token that occurs on both sides of <image>
    n_vis = frames * 14 * 15
    S = 9 + n_vis + 31
    labels = torch.full((S,), -100, dtype=torch.int64)
    labels[S - 20:] = torch.randint(0, V, (20,), generator=g)

    # ---- device
    table = ops.Sin3DTable(H, 64, torch.bfloat16, "cuda")
    pe = ops.sin3d_pe(ids.view(frames, 196, 3).to(torch.bfloat16).cuda(), H).float().cpu()          # the PE rows as constants of the reference
    rope = train.RopeTables(hd, 1024, 1e6, torch.bfloat16, "cuda")
    cu = lambda t: train._tree_map(lambda a: a.cuda(), t)
    params = {"vision": {"patch_w": patch_w.cuda(), "patch_b": patch_b.cuda(), "pos": pos.cuda(), "layers": [train.siglip_pad_layer(cu(l)) for l in vit]},
              "projector": cu(proj), "newline": newline.cuda(), "embed": embed.cuda(), "llm": cu(llm)}
    coord_rows = torch.tensor([9 + frames * 14 * 15 + 4, 9 + frames * 14 * 15 + 11])   # two <coord> tokens of a Scan2Cap prompt (llava_arch.py:697-700)
    coord_pe = mk(H, s=0.5)
    loss, grads = train.sample_forward_backward(params, patches.cuda(), ids.cuda(), table, pre_ids.cuda(), post_ids.cuda(), labels.cuda(), rope,
                                                frames, n_q, n_kv, hd, coord_rows=coord_rows.cuda(), coord_pe=coord_pe.cuda())

    # activation re-computation (the reference's gradient checkpointing, train_multi.sh:72): the same loss and gradients, bit for bit
    torch.cuda.reset_peak_memory_stats()
    base_mem = torch.cuda.memory_allocated()
    loss_r, grads_r = train.sample_forward_backward(params, patches.cuda(), ids.cuda(), table, pre_ids.cuda(), post_ids.cuda(), labels.cuda(), rope,
                                                    frames, n_q, n_kv, hd, coord_rows=coord_rows.cuda(), coord_pe=coord_pe.cuda(), recompute=True)
    peak_r = torch.cuda.max_memory_allocated() - base_mem
    assert float(loss_r) == float(loss)
    for a_, b_ in zip(train._leaves(grads_r), train._leaves(grads)):
        assert torch.equal(a_, b_)
    del grads_r
    torch.cuda.reset_peak_memory_stats()
    base_mem = torch.cuda.memory_allocated()
    _l, _g = train.sample_forward_backward(params, patches.cuda(), ids.cuda(), table, pre_ids.cuda(), post_ids.cuda(), labels.cuda(), rope,
                                           frames, n_q, n_kv, hd, coord_rows=coord_rows.cuda(), coord_pe=coord_pe.cuda())
    peak_k = torch.cuda.max_memory_allocated() - base_mem
    del _l, _g
    assert peak_r < peak_k, (peak_r, peak_k)

    # weight gradients on the side stream (the default inside a step function) against the backward's own stream: the same kernels on
    # the same operands, only their order in time differs - loss and every gradient bit for bit
    monkeypatch.setenv("V3D_TRAIN_WGRAD_STREAM", "0")
    loss_1, grads_1 = train.sample_forward_backward(params, patches.cuda(), ids.cuda(), table, pre_ids.cuda(), post_ids.cuda(), labels.cuda(), rope,
                                                    frames, n_q, n_kv, hd, coord_rows=coord_rows.cuda(), coord_pe=coord_pe.cuda())
    monkeypatch.delenv("V3D_TRAIN_WGRAD_STREAM")
    assert train._WGRAD["stream"] is None and train._WGRAD["depth"] == 0
    assert float(loss_1) == float(loss)
    for a_, b_ in zip(train._leaves(grads_1), train._leaves(grads)):
        assert torch.equal(a_, b_)
    del grads_1

    # ---- reference (f32 autograd over the same 16-bit parameters)
    f32 = lambda t: train._tree_map(lambda a: a.float().requires_grad_(), t)
    r_vit, r_proj, r_llm = f32(vit), f32(proj), f32(llm)
    r_pw, r_pb, r_pos, r_nl, r_emb = (t.float().requires_grad_() for t in (patch_w, patch_b, pos, newline, embed))
    h = (patches.float() @ r_pw.t() + r_pb).view(frames, tokens, Hv) + r_pos
    h = h.view(frames * tokens, Hv)
    for sd in r_vit:
        h = _siglip_layer_ref(h, sd, frames, tokens, heads)
    y = F.gelu(h @ r_proj["w1"].t() + r_proj["b1"]) @ r_proj["w2"].t() + r_proj["b2"]
    pooled = F.interpolate(y.view(frames, 27, 27, H).permute(0, 3, 1, 2), size=[14, 14], mode="bilinear").permute(0, 2, 3, 1)
    tok = pooled + pe.view(frames, 14, 14, H)
    vis = torch.cat([tok, r_nl[None, None, None, :].expand(frames, 14, 1, H)], 2).reshape(-1, H)
    x = torch.cat([r_emb[pre_ids], vis, r_emb[post_ids]], 0)
    x = x.index_add(0, coord_rows, coord_pe.float()[None].expand(2, H))
    for p in r_llm["layers"]:
        x = _layer_ref(x, p, n_q, n_kv, hd, 1e-6)
    logits = _rmsnorm_ref(x, r_llm["norm"], 1e-6) @ r_llm["lm_head"].t()
    ref_loss = F.cross_entropy(logits[:-1], labels[1:], ignore_index=-100)
    ref_loss.backward()

    assert abs(float(loss) - float(ref_loss.detach())) < 3e-2 * float(ref_loss.detach())
    tol = (4e-2, 1e-1)                                   # four normalised blocks deep in 16-bit storage
    _close(grads["newline"], r_nl.grad, *tol, "d newline")
    _close(grads["embed"], r_emb.grad, *tol, "d embed")
    for k in proj:
        _close(grads["projector"][k], r_proj[k].grad, *tol, "d projector " + k)
    _close(grads["vision"]["pos"], r_pos.grad, *tol, "d position embedding")
    _close(grads["vision"]["patch_w"][:, :588], r_pw.grad[:, :588], *tol, "d patch weight")
    _close(grads["vision"]["patch_b"], r_pb.grad, *tol, "d patch bias")
    for i in range(n_vit):
        real = train.siglip_unpad_grads(grads["vision"]["layers"][i])
        for k in vit[i]:
            if k == "k_b":
                continue                                  # no gradient (see the layer test)
            _close(real[k], r_vit[i][k].grad, *tol, f"vit layer {i} d {k}")
    for i in range(L):
        for k in layers[i]:
            _close(grads["llm"]["layers"][i][k], r_llm["layers"][i][k].grad, *tol, f"llm layer {i} d {k}")
    _close(grads["llm"]["lm_head"], r_llm["lm_head"].grad, *tol, "d lm_head")


def test_attention_train_non_causal_batch_equals_plain_attention(ops):
    """v3d_attention_train on a batch of non-causal sequences (the SigLIP encoder's shape: 729 rows, heads padded to 128): the outputs are
    the plain kernel's bit for bit, the log-sum-exp matches the f32 definition."""
    B, S, heads, hd = 3, 729, 4, 128
    g = torch.Generator().manual_seed(8)
    qkv = torch.randn(B * S, 3 * heads * hd, generator=g).to(torch.bfloat16).cuda()
    o = torch.empty(B * S, heads * hd, dtype=torch.bfloat16, device="cuda")
    lse = ops.attention_train(qkv, o, S, heads, heads, 72 ** -0.5, B=B, causal=False)
    w = qkv.stride(0)
    plain = torch.empty_like(o)
    ops.attention(qkv, qkv[:, heads * hd:], qkv[:, 2 * heads * hd:], plain, B, S, S, heads, heads, hd, hd, w, w, w, heads * hd, S * w, S * w, S * heads * hd,
                  hd, hd, hd, False, 0, 72 ** -0.5)
    assert torch.equal(o, plain)
    q = qkv[:, :heads * hd].float().cpu().view(B, S, heads, hd).transpose(1, 2)
    k = qkv[:, heads * hd:2 * heads * hd].float().cpu().view(B, S, heads, hd).transpose(1, 2)
    ref = torch.logsumexp(q @ k.transpose(2, 3) * 72 ** -0.5, -1) * 1.4426950408889634          # [B, heads, S] in log2 units
    assert float((lse.cpu() - ref).abs().max()) < 3e-2


def test_axpy_and_gradient_accumulation(ops, train):
    g = torch.Generator().manual_seed(4)
    for n in (8, 1000003, 77):
        y = torch.randn(n, generator=g).to(torch.bfloat16)
        x = torch.randn(n, generator=g).to(torch.bfloat16)
        got = ops.axpy(y.clone().cuda(), x.cuda())
        assert torch.equal(got.cpu(), y + x)                                   # torch's own 16-bit add: f32 sum, one rounding
        got = ops.axpy(y.clone().cuda(), x.cuda(), alpha=0.5)
        assert torch.equal(got.cpu(), (y.float() + 0.5 * x.float()).to(torch.bfloat16))
    a = {"w": torch.randn(16, 24, generator=g).to(torch.bfloat16).cuda(), "l": [torch.randn(40, generator=g).cuda()]}
    b = {"w": torch.randn(16, 24, generator=g).to(torch.bfloat16).cuda(), "l": [torch.randn(40, generator=g).cuda()]}
    want_w, want_l = a["w"] + b["w"], a["l"][0] + b["l"][0]
    total = train.accumulate_grads(None, a)
    total = train.accumulate_grads(total, b)
    assert torch.equal(total["w"], want_w) and torch.equal(total["l"][0], want_l)


def test_a_few_optimizer_steps_fit_one_sample(train):
    """Functional check of the whole loop: a 2-layer language model memorises one sample's 40 answer tokens in a few AdamW steps."""
    H, I, n_q, n_kv, hd, V, S, L = 512, 1024, 4, 2, 128, 1024, 120, 2
    g = torch.Generator().manual_seed(99)
    width = (n_q + 2 * n_kv) * hd
    mk = lambda *shape, s=1.0: (torch.randn(*shape, generator=g) * s).to(torch.bfloat16).cuda()
    ones = lambda: torch.ones(H, dtype=torch.bfloat16, device="cuda")
    layers = [{"ln1": ones(), "qkv": mk(width, H, s=H ** -0.5), "qkv_bias": mk(width, s=0.02), "o": mk(H, n_q * hd, s=(n_q * hd) ** -0.5),
               "ln2": ones(), "gate_up": mk(2 * I, H, s=H ** -0.5), "down": mk(H, I, s=I ** -0.5)} for _ in range(L)]
    params = {"layers": layers, "norm": ones(), "lm_head": mk(V, H, s=H ** -0.5)}
    x = mk(S, H)
    labels = torch.full((S,), -100, dtype=torch.int64)
    labels[80:] = torch.randint(0, V, (40,), generator=g)
    labels = labels.cuda()
    rope = train.RopeTables(hd, 256, 1e6, torch.bfloat16, "cuda")
    opt = train.AdamW(params, lr=2e-3)
    losses = []
    for _ in range(12):
        loss, _, grads = train.llm_forward_backward(params, x, labels, rope, n_q, n_kv, hd)
        losses.append(float(loss))
        opt.step(params, grads)
    assert losses[0] > 6.0 and losses[-1] < 0.5 * losses[0], losses        # ln(1024) = 6.9 at the start
    assert all(torch.isfinite(p.float()).all() for l in layers for p in l.values())


def test_grounding_sample_step_matches_autograd(ops, train):
    """A ScanRefer / Multi3DRefer training sample: tower -> projector -> splice -> decoder, then predict_box's infonce loss between the
    <ground> row's hidden state and the object proposals (llava_qwen.py:239-310, llava_arch.py:479-501); loss, scores and the
    gradients of the heads, the zero-target, the decoder, the projector and the tower against autograd in f32."""
    frames, tokens, Hv, heads, inter, kpad = 2, 729, 1152, 16, 4304, 640
    H, I, n_q, n_kv, hd, V, L, n_obj = 768, 1024, 4, 2, 128, 1024, 2, 6
    g = torch.Generator().manual_seed(55)
    mk = lambda *shape, s=1.0: (torch.randn(*shape, generator=g) * s).to(torch.bfloat16)
    ln = lambda n_: (1 + 0.1 * torch.randn(n_, generator=g)).to(torch.bfloat16)
    vit = {"ln1_w": ln(Hv), "ln1_b": mk(Hv, s=0.1), "ln2_w": ln(Hv), "ln2_b": mk(Hv, s=0.1),
           "q_w": mk(Hv, Hv, s=Hv ** -0.5), "q_b": mk(Hv, s=0.2), "k_w": mk(Hv, Hv, s=Hv ** -0.5), "k_b": mk(Hv, s=0.2),
           "v_w": mk(Hv, Hv, s=Hv ** -0.5), "v_b": mk(Hv, s=0.2), "o_w": mk(Hv, Hv, s=Hv ** -0.5), "o_b": mk(Hv, s=0.2),
           "fc1_w": mk(inter, Hv, s=Hv ** -0.5), "fc1_b": mk(inter, s=0.2), "fc2_w": mk(Hv, inter, s=inter ** -0.5), "fc2_b": mk(Hv, s=0.2)}
    patch_w, patch_b, pos = mk(Hv, kpad, s=588 ** -0.5), mk(Hv, s=0.1), mk(tokens, Hv, s=0.5)
    patch_w[:, 588:] = 0
    proj = {"w1": mk(H, Hv, s=Hv ** -0.5), "b1": mk(H, s=0.1), "w2": mk(H, H, s=H ** -0.5), "b2": mk(H, s=0.1)}
    newline, embed = mk(H, s=0.5), mk(V, H, s=0.5)
    width = (n_q + 2 * n_kv) * hd
    layers = [{"ln1": ln(H), "qkv": mk(width, H, s=H ** -0.5), "qkv_bias": mk(width, s=0.3), "o": mk(H, n_q * hd, s=(n_q * hd) ** -0.5),
               "ln2": ln(H), "gate_up": mk(2 * I, H, s=H ** -0.5), "down": mk(H, I, s=I ** -0.5)} for _ in range(L)]
    llm = {"layers": layers, "norm": ln(H)}
    head = lambda: {"w0": mk(H, H, s=H ** -0.5), "b0": mk(H, s=0.1), "ln_w": ln(H), "ln_b": mk(H, s=0.1), "w3": mk(H, H, s=H ** -0.5), "b3": mk(H, s=0.1)}
    ground = {"obj": head(), "query": head(), "zero_target": mk(H)}
    patches = mk(frames * tokens, kpad)
    patches[:, 588:] = 0
    ids = torch.randint(0, 64, (frames, 14, 14, 3), generator=g, dtype=torch.int32)
    pre_ids, post_ids = torch.randint(0, V, (9,), generator=g), torch.randint(0, V, (21,), generator=g)
    n_vis = frames * 14 * 15
    ground_row = 9 + n_vis + 17                                              # the <ground> label token sits among the trailing text rows
    mask = (torch.rand(n_obj, frames * tokens, generator=g) < 0.02).to(torch.uint8)
    mask[4] = 0                                                               # a proposal that covers no patch: its feature is the PE alone
    box_pe = mk(n_obj, H, s=0.5)
    positive = torch.zeros(n_obj + 1, dtype=torch.uint8)
    positive[[1, 3]] = 1

    table = ops.Sin3DTable(H, 64, torch.bfloat16, "cuda")
    pe = ops.sin3d_pe(ids.view(frames, 196, 3).to(torch.bfloat16).cuda(), H).float().cpu()
    rope = train.RopeTables(hd, 1024, 1e6, torch.bfloat16, "cuda")
    cu = lambda t: train._tree_map(lambda a: a.cuda(), t)
    params = {"vision": {"patch_w": patch_w.cuda(), "patch_b": patch_b.cuda(), "pos": pos.cuda(), "layers": [train.siglip_pad_layer(cu(vit))]},
              "projector": cu(proj), "newline": newline.cuda(), "embed": embed.cuda(), "llm": cu(llm), "ground": cu(ground)}
    loss, scores, grads = train.ground_sample_forward_backward(params, patches.cuda(), ids.cuda(), table, pre_ids.cuda(), post_ids.cuda(), ground_row,
                                                               mask.cuda(), box_pe.cuda(), positive.cuda(), rope, frames, n_q, n_kv, hd)

    f32 = lambda t: train._tree_map(lambda a: a.float().requires_grad_(), t)
    r_vit, r_proj, r_llm, r_gr = f32(vit), f32(proj), f32(llm), f32(ground)
    r_pw, r_nl, r_emb = (t.float().requires_grad_() for t in (patch_w, newline, embed))
    h = ((patches.float() @ r_pw.t() + patch_b.float()).view(frames, tokens, Hv) + pos.float()).view(frames * tokens, Hv)
    h = _siglip_layer_ref(h, r_vit, frames, tokens, heads)
    y = F.gelu(h @ r_proj["w1"].t() + r_proj["b1"]) @ r_proj["w2"].t() + r_proj["b2"]
    pooled = F.interpolate(y.view(frames, 27, 27, H).permute(0, 3, 1, 2), size=[14, 14], mode="bilinear").permute(0, 2, 3, 1)
    vis = torch.cat([pooled + pe.view(frames, 14, 14, H), r_nl[None, None, None, :].expand(frames, 14, 1, H)], 2).reshape(-1, H)
    x = torch.cat([r_emb[pre_ids], vis, r_emb[post_ids]], 0)
    for p in r_llm["layers"]:
        x = _layer_ref(x, p, n_q, n_kv, hd, 1e-6)
    query = _rmsnorm_ref(x[ground_row:ground_row + 1], r_llm["norm"], 1e-6)
    objs = []
    for i in range(n_obj):                                                    # llava_arch.py:482-501
        rows = mask[i].bool()
        objs.append((y[rows].mean(0) if bool(rows.any()) else torch.zeros(H)) + box_pe[i].float())
    of = torch.cat([torch.stack(objs), r_gr["zero_target"][None]], 0)         # llava_qwen.py:297
    mlp = lambda t, hp: F.linear(F.layer_norm(F.relu(F.linear(t, hp["w0"], hp["b0"])), (H,), hp["ln_w"], hp["ln_b"], 1e-5), hp["w3"], hp["b3"])
    sc = (F.normalize(mlp(of, r_gr["obj"])) * F.normalize(mlp(query, r_gr["query"]))).sum(-1)
    lg = torch.exp(sc / 0.07)
    ref_loss = -torch.log(lg[positive.bool()].sum() / lg.sum())                # :306-307
    ref_loss.backward()

    assert float((scores.cpu() - sc.detach()).abs().max()) < 2e-2
    assert abs(float(loss) - float(ref_loss.detach())) < 0.05 * max(1.0, float(ref_loss.detach()))
    tol = (5e-2, 1.2e-1)
    for hname in ("obj", "query"):
        for k in ground[hname]:
            _close(grads["ground"][hname][k], r_gr[hname][k].grad, *tol, f"d {hname} head {k}")
    _close(grads["ground"]["zero_target"], r_gr["zero_target"].grad, *tol, "d zero target")
    _close(grads["llm"]["norm"], r_llm["norm"].grad, *tol, "d final norm")
    for k in layers[0]:
        _close(grads["llm"]["layers"][0][k], r_llm["layers"][0][k].grad, *tol, "llm layer 0 d " + k)
    for k in proj:
        _close(grads["projector"][k], r_proj[k].grad, *tol, "d projector " + k)
    real = train.siglip_unpad_grads(grads["vision"]["layers"][0])
    for k in ("q_w", "o_w", "fc1_w", "fc2_w", "ln1_w"):
        _close(real[k], r_vit[k].grad, *tol, "vit d " + k)
    _close(grads["embed"], r_emb.grad, *tol, "d embed")


def test_trainer_surface_loss_backward_and_checkpoint_round_trip(ops, train):
    """v3d.train_module.LlavaQwenTrainable (r03; VERDICT r2 missing #5): built from a state dict in the REFERENCE's keys, its forward
    returns a loss whose .backward() fills every parameter's .grad with the device-side backward's gradients (bit-identical to
    train.sample_forward_backward), a torch optimizer steps it, and reference_state_dict() gives the reference's keys / shapes back -
    exactly the input before the step."""
    from v3d.train_module import LlavaQwenTrainable, VIT
    Hv, heads, inter, H, I, n_q, n_kv, hd, V = 1152, 16, 4304, 512, 768, 4, 2, 128, 512
    g = torch.Generator().manual_seed(91)
    mk = lambda *shape, s=1.0: (torch.randn(*shape, generator=g) * s).to(torch.bfloat16)      # noqa: E731
    ln = lambda n_: (1 + 0.1 * torch.randn(n_, generator=g)).to(torch.bfloat16)                 # noqa: E731
    sd = {VIT + "embeddings.patch_embedding.weight": mk(Hv, 3, 14, 14, s=588 ** -0.5), VIT + "embeddings.patch_embedding.bias": mk(Hv, s=0.1),
          VIT + "embeddings.position_embedding.weight": mk(729, Hv, s=0.5)}
    p = VIT + "encoder.layers.0."
    for n_ in ("q_proj", "k_proj", "v_proj", "out_proj"):
        sd[p + f"self_attn.{n_}.weight"], sd[p + f"self_attn.{n_}.bias"] = mk(Hv, Hv, s=Hv ** -0.5), mk(Hv, s=0.2)
    for n_ in ("layer_norm1", "layer_norm2"):
        sd[p + n_ + ".weight"], sd[p + n_ + ".bias"] = ln(Hv), mk(Hv, s=0.1)
    sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"], sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"] = mk(inter, Hv, s=Hv ** -0.5), mk(inter, s=0.2), mk(Hv, inter, s=inter ** -0.5), mk(Hv, s=0.2)
    sd["model.mm_projector.0.weight"], sd["model.mm_projector.0.bias"] = mk(H, Hv, s=Hv ** -0.5), mk(H, s=0.1)
    sd["model.mm_projector.2.weight"], sd["model.mm_projector.2.bias"] = mk(H, H, s=H ** -0.5), mk(H, s=0.1)
    sd["model.image_newline"], sd["model.embed_tokens.weight"] = mk(H, s=0.5), mk(V, H, s=0.5)
    p = "model.layers.0."
    sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"] = mk(n_q * hd, H, s=H ** -0.5), mk(n_q * hd, s=0.3)
    for n_ in ("k_proj", "v_proj"):
        sd[p + f"self_attn.{n_}.weight"], sd[p + f"self_attn.{n_}.bias"] = mk(n_kv * hd, H, s=H ** -0.5), mk(n_kv * hd, s=0.3)
    sd[p + "self_attn.o_proj.weight"] = mk(H, n_q * hd, s=(n_q * hd) ** -0.5)
    sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"], sd[p + "mlp.down_proj.weight"] = mk(I, H, s=H ** -0.5), mk(I, H, s=H ** -0.5), mk(H, I, s=I ** -0.5)
    sd[p + "input_layernorm.weight"], sd[p + "post_attention_layernorm.weight"] = ln(H), ln(H)
    sd["model.norm.weight"], sd["lm_head.weight"] = ln(H), mk(V, H, s=H ** -0.5)
    # tensors a real LlavaQwen checkpoint holds and the training step does not model: the tower's post_layernorm (kept after
    # siglip_encoder.py:570-571 deletes the last layer; :416), the infonce grounding head (llava_qwen.py:98-111: Sequential indices 0, 2, 3)
    g2 = torch.Generator().manual_seed(92)               # (a generator of their own: the modelled tensors and inputs stay those of r03)
    mk2 = lambda *shape, s=1.0: (torch.randn(*shape, generator=g2) * s).to(torch.bfloat16)      # noqa: E731
    unmodelled = {VIT + "post_layernorm.weight": 1 + mk2(Hv, s=0.1), VIT + "post_layernorm.bias": mk2(Hv, s=0.1), "ground_head_zero_target": mk2(H)}
    for head in ("ground_head_obj", "ground_head_query"):
        unmodelled.update({f"{head}.0.weight": mk2(H, H, s=H ** -0.5), f"{head}.0.bias": mk2(H, s=0.1), f"{head}.2.weight": 1 + mk2(H, s=0.1),
                           f"{head}.2.bias": mk2(H, s=0.1), f"{head}.3.weight": mk2(H, H, s=H ** -0.5), f"{head}.3.bias": mk2(H, s=0.1)})
    sd.update(unmodelled)

    model = LlavaQwenTrainable.from_reference_state_dict(sd, n_q, n_kv, max_pos=1024)
    back = model.reference_state_dict()
    assert set(back) == set(sd) and all(torch.equal(back[k].cpu(), sd[k]) for k in sd)          # exact round trip
    frames = 2
    images = torch.randn(frames, 3, 384, 384, generator=g)
    coords = (torch.rand(frames, 384, 384, 3, generator=g) - 0.5) * torch.tensor([20.0, 20.0, 8.0])
    t_ = torch.randint(0, V, (30,), generator=g)
    input_ids = torch.cat([t_[:9], torch.tensor([-200]), t_[9:]])
    labels = torch.full((31,), -100, dtype=torch.int64)
    labels[20:] = input_ids[20:]
    loss = model(input_ids, labels, images, coords)
    assert loss.requires_grad and loss.dtype == torch.float32
    loss.backward()
    # the same gradients as the explicit step on the same tensors
    tr = model.param_tree()
    patches = ops.patchify(images.to(torch.bfloat16).cuda(), 14, 640)
    _, _, vox = ops.coord_pool_voxel(coords.to(torch.bfloat16).cuda(), want_avg=False, want_vox=False)
    n_vis = frames * 210
    full = torch.cat([labels[:9], torch.full((n_vis,), -100, dtype=torch.int64), labels[10:]]).cuda()
    loss2, grads = train.sample_forward_backward(tr, patches, vox, model.pe_table, input_ids[:9].cuda(), input_ids[10:].cuda(), full, model.rope,
                                                 frames, n_q, n_kv, hd)
    assert float(loss.detach()) == float(loss2)
    from v3d.train_module import _flatten
    by_name = dict(_flatten(grads))
    for name, prm in zip(model.names, model._params):
        assert prm.grad is not None and torch.equal(prm.grad, by_name[name].to(prm.dtype)), name      # (image_newline's gradient is formed in f32)
    # loss scaling (gradient accumulation divides the loss): gradients scale with it
    model.zero_grad()
    (model(input_ids, labels, images, coords) * 0.5).backward()
    g_half = model._params[model.names.index("projector.w2")].grad
    ref = (by_name["projector.w2"].float() * 0.5).to(torch.bfloat16)
    assert torch.equal(g_half, ref)
    model.zero_grad()
    model(input_ids, labels, images, coords).backward()
    before = model.reference_state_dict()
    torch.optim.SGD(model.parameters(), lr=0.5).step()
    after = model.reference_state_dict()
    changed = [k for k in sd if not torch.equal(before[k], after[k])]
    assert not set(changed) & set(unmodelled) and all(torch.equal(after[k].cpu(), sd[k]) for k in unmodelled)      # carried, frozen
    # (the tower's k_proj bias has no gradient - softmax shift invariance - and a 16-bit weight near 1 does not move by a small step)
    assert len(changed) >= len(sd) - len(unmodelled) - 6 and VIT + "encoder.layers.0.self_attn.k_proj.bias" not in changed, sorted(set(sd) - set(changed))
