"""GPU parity tests for RMSNorm / LayerNorm / rotary / attention kernels (C ABI) against the torch
oracle (oracle/llm_oracle.py, itself pinned to the reference modules) run on the CPU in the same
dtype, so both sides share the reference's rounding points."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import llm_oracle as L

pytestmark = pytest.mark.gpu
DT = {"bf16": torch.bfloat16, "f16": torch.float16}
EPS = {"bf16": 2.0 ** -7, "f16": 2.0 ** -10}


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from v3d import ops as _ops
    return _ops


def close(got, want, kind, ulps=2.0, floor=None):
    got, want = got.float().cpu(), want.float()
    f = want.abs().mean().item() if floor is None else floor
    err = (got - want).abs()
    bound = ulps * EPS[kind] * (want.abs() + f)
    assert torch.all(err <= bound), f"max err {err.max().item():.4g}, worst ratio {(err / bound).max().item():.3g}"


@pytest.mark.parametrize("kind", ["bf16", "f16"])
@pytest.mark.parametrize("cols", [256, 3584, 1152])
def test_rmsnorm(ops, kind, cols):
    g = torch.Generator().manual_seed(cols)
    x = (torch.randn(301, cols, generator=g) * 2).to(DT[kind])
    w = (1 + 0.1 * torch.randn(cols, generator=g)).to(DT[kind])
    got = ops.rmsnorm(x.cuda(), w.cuda(), 1e-6)
    want = L.rmsnorm(x, w, 1e-6)
    close(got, want, kind, ulps=1.01)
    assert (got.float().cpu() != want.float()).float().mean() < 5e-3   # almost everywhere bit-equal


@pytest.mark.parametrize("kind", ["bf16", "f16"])
def test_layernorm(ops, kind):
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(733, 1152, generator=g) * 3 + 0.5).to(DT[kind])
    w = (1 + 0.1 * torch.randn(1152, generator=g)).to(DT[kind])
    b = (0.1 * torch.randn(1152, generator=g)).to(DT[kind])
    got = ops.layernorm(x.cuda(), w.cuda(), b.cuda(), 1e-6)
    want = F.layer_norm(x.float(), (1152,), w.float(), b.float(), 1e-6).to(DT[kind])
    close(got, want, kind, ulps=1.01)


@pytest.mark.parametrize("kind", ["bf16", "f16"])
def test_layernorm_three_rows_per_wave_form_equals_one_row_form_bitwise(ops, kind):
    """Rows of at most 1536 elements in launches of >= 4096 rows take layernorm_kernel<3, 3> (three rows in flight per wave):
    the same rows sent in chunks below 4096 take the one-row form - the outputs must agree bit for bit (ragged last wave,
    padded row stride as the ViT's 1280)."""
    g = torch.Generator().manual_seed(15)
    rows, cols, ld = 4096 + 7, 1152, 1280
    x = (torch.randn(rows, ld, generator=g) * 2 - 0.3).to(DT[kind]).cuda()[:, :cols]
    w = (1 + 0.1 * torch.randn(cols, generator=g)).to(DT[kind]).cuda()
    b = (0.1 * torch.randn(cols, generator=g)).to(DT[kind]).cuda()
    full = torch.full((rows + 1, ld), 7.0, dtype=DT[kind], device="cuda")
    ops.layernorm(x, w, b, 1e-6, out=full[:rows, :cols])
    parts = torch.cat([ops.layernorm(x[i: i + 1024], w, b, 1e-6) for i in range(0, rows, 1024)])
    assert torch.equal(full[:rows, :cols], parts)
    assert bool((full[rows] == 7.0).all()) and bool((full[:rows, cols:] == 7.0).all())      # nothing written past the rows / columns
    want = F.layer_norm(x.float().cpu(), (cols,), w.float().cpu(), b.float().cpu(), 1e-6).to(DT[kind])
    close(full[:rows, :cols], want, kind, ulps=1.01)


@pytest.mark.parametrize("kind", ["bf16", "f16"])
def test_rope_table_and_apply(ops, kind):
    dt = DT[kind]
    S, H, KV, D = 700, 4, 2, 128
    table = ops.RopeTable(D, 8192, 1000000.0, dt, "cuda")
    pos = torch.arange(S)
    cos, sin = L.rotary_cos_sin(pos, D, 1000000.0, dt)
    assert (table.cos[:S].cpu().float() != cos[:, :64].float()).float().mean() < 2e-3
    close(table.cos[:S], cos[:, :64], kind, ulps=1.01, floor=0.0)
    close(table.sin[:S], sin[:, :64], kind, ulps=1.01, floor=1e-3)
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn(S, (H + 2 * KV) * D, generator=g).to(dt)
    q = qkv[:, :H * D].view(1, S, H, D).transpose(1, 2)
    k = qkv[:, H * D:(H + KV) * D].view(1, S, KV, D).transpose(1, 2)
    qe, ke = L.apply_rope(q, k, cos, sin)
    dev = qkv.cuda()
    ops.rope_apply(dev, H + KV, D, table, pos0=0)
    got_q = dev[:, :H * D].view(S, H, D)
    got_k = dev[:, H * D:(H + KV) * D].view(S, KV, D)
    # two rounded products are summed: the bound is in ulps of the operands (|x| up to ~4), not of the sum
    close(got_q, qe[0].transpose(0, 1), kind, ulps=2.0, floor=2.0)
    close(got_k, ke[0].transpose(0, 1), kind, ulps=2.0, floor=2.0)
    assert torch.equal(dev[:, (H + KV) * D:].cpu(), qkv[:, (H + KV) * D:])          # v untouched
    # decode position offset
    one = qkv[5:6].clone().cuda()
    ops.rope_apply(one, H + KV, D, table, pos0=5)
    assert torch.equal(one, dev[5:6])


def ref_attention(q, k, v, causal, scale, q_pos0=0):
    """f32 reference: q [B,Sq,H,D], k/v [B,Sk,Hkv,D]."""
    B, Sq, H, D = q.shape
    Sk, Hkv = k.shape[1], k.shape[2]
    qf, kf, vf = q.float().transpose(1, 2), k.float().transpose(1, 2), v.float().transpose(1, 2)
    kf = L.repeat_kv(kf, H // Hkv)
    vf = L.repeat_kv(vf, H // Hkv)
    s = qf @ kf.transpose(2, 3) * scale
    if causal:
        i = torch.arange(Sq)[:, None] + q_pos0
        j = torch.arange(Sk)[None, :]
        s = s.masked_fill(j > i, float("-inf"))
    return (torch.softmax(s, -1) @ vf).transpose(1, 2)


@pytest.mark.parametrize("kind", ["bf16", "f16"])
@pytest.mark.parametrize("S", [1, 77, 128, 300, 1000])
def test_attention_prefill_causal_gqa(ops, kind, S):
    dt = DT[kind]
    H, KV, D = 28, 4, 128
    g = torch.Generator().manual_seed(S)
    q = torch.randn(1, S, H, D, generator=g).to(dt)
    k = torch.randn(1, S, KV, D, generator=g).to(dt)
    v = torch.randn(1, S, KV, D, generator=g).to(dt)
    got = ops.attention_bshd(q.cuda(), k.cuda(), v.cuda(), causal=True)
    want = ref_attention(q, k, v, True, 1 / math.sqrt(D))
    # P is rounded to 16 bit before P.V (as the reference's eager path does); error relative to |v|~1
    close(got, want, kind, ulps=3.0, floor=0.3)


@pytest.mark.parametrize("kind", ["bf16", "f16"])
def test_attention_matches_reference_eager_rounding(ops, kind):
    """Against the oracle's eager attention run in the model dtype (the reference's spec, :289-311)."""
    dt = DT[kind]
    S, H, KV, D = 300, 28, 4, 128
    g = torch.Generator().manual_seed(8)
    q = (torch.randn(1, H, S, D, generator=g) * 0.7).to(dt)
    k = (torch.randn(1, KV, S, D, generator=g) * 0.7).to(dt)
    v = torch.randn(1, KV, S, D, generator=g).to(dt)
    want = L.eager_attention(q, k, v, H // KV, L.causal_mask(S, S, dt)).transpose(1, 2)
    got = ops.attention_bshd(q.transpose(1, 2).contiguous().cuda(), k.transpose(1, 2).contiguous().cuda(),
                             v.transpose(1, 2).contiguous().cuda(), causal=True)
    close(got, want, kind, ulps=4.0, floor=0.3)


@pytest.mark.parametrize("kind", ["bf16", "f16"])
@pytest.mark.parametrize("Sq,Sk", [(64, 64), (300, 300), (1000, 1000), (2100, 2100), (130, 900), (517, 2048)])
def test_attention_prefill64_optin_kernel_equals_default_kernel_bitwise(ops, kind, Sq, Sk, monkeypatch):
    """V3D_ATTN64=1 selects attn_prefill64_kernel (64 queries per wave, 3-deep K/V ring, softmax pieces in the MFMA gaps): same
    arithmetic in the same order as attn_prefill_kernel, so the outputs are equal bit for bit - causal, GQA, ragged tails,
    question rows at q_pos0 > 0 over a longer K/V prefix (scene reuse), and the rare max-raise branch (spiked key)."""
    dt = DT[kind]
    H, KV, D = 28, 4, 128
    g = torch.Generator().manual_seed(Sq * 7 + Sk)
    q = torch.randn(1, Sq, H, D, generator=g) * 0.5
    k = torch.randn(1, Sk, KV, D, generator=g) * 0.5
    v = torch.randn(1, Sk, KV, D, generator=g)
    if Sk >= 300:
        k[0, Sk - 90, 1] = q[0, Sq - 10, 9] * 30          # a late spike: the running maximum of one query jumps
    q, k, v = q.to(dt).cuda(), k.to(dt).cuda(), v.to(dt).cuda()
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("V3D_ATTN64", mode)
        outs[mode] = ops.attention_bshd(q, k, v, causal=True, q_pos0=Sk - Sq)
    assert torch.equal(outs["0"], outs["1"])
    want = ref_attention(q.cpu(), k.cpu(), v.cpu(), True, 1 / math.sqrt(D), q_pos0=Sk - Sq)
    close(outs["1"], want, kind, ulps=4.0, floor=0.3)      # (the spiked row's P is one 16-bit rounding away from one-hot)


@pytest.mark.parametrize("kind", ["bf16", "f16"])
@pytest.mark.parametrize("Sq,Sk", [(1, 1), (64, 64), (77, 77), (300, 300), (1000, 1000), (2100, 2100), (130, 900), (517, 2048)])
def test_attention_mfma16_variant_of_the_prefill_kernel(ops, kind, Sq, Sk, monkeypatch):
    """r04 A/B (V3D_ATTN_MFMA=16): attn_prefill16_kernel = attn_prefill_kernel<128> on v_mfma_f32_16x16x32 (32-wide k chunks, a query's
    keys over 4 lanes): against the f32 reference at the default kernel's bound, against the default kernel (summation order only:
    2 ulp16 of |v|), causal GQA, ragged tails, question rows at q_pos0 > 0 over a longer prefix, the rare max-raise branch (spiked
    key), and the scene-reuse property: a query row's bits do not depend on the rows that share its launch."""
    dt = DT[kind]
    H, KV, D = 28, 4, 128
    g = torch.Generator().manual_seed(Sq * 7 + Sk)
    q = torch.randn(1, Sq, H, D, generator=g) * 0.5
    k = torch.randn(1, Sk, KV, D, generator=g) * 0.5
    v = torch.randn(1, Sk, KV, D, generator=g)
    if Sk >= 300:
        k[0, Sk - 90, 1] = q[0, Sq - 10, 9] * 30          # a late spike: the running maximum of one query jumps
    q, k, v = q.to(dt).cuda(), k.to(dt).cuda(), v.to(dt).cuda()
    outs = {}
    for mode in ("32", "16"):
        monkeypatch.setenv("V3D_ATTN_MFMA", mode)
        outs[mode] = ops.attention_bshd(q, k, v, causal=True, q_pos0=Sk - Sq)
    want = ref_attention(q.cpu(), k.cpu(), v.cpu(), True, 1 / math.sqrt(D), q_pos0=Sk - Sq)
    close(outs["16"], want, kind, ulps=4.0, floor=0.3)
    close(outs["16"], outs["32"].float().cpu(), kind, ulps=2.0, floor=0.3)
    if Sq >= 64:       # rows [a, b) alone (other wave / lane positions, other neighbours): the same bits
        a, b = Sq // 3 + 5, Sq - 7
        part = ops.attention_bshd(q[:, a:b].contiguous(), k, v, causal=True, q_pos0=Sk - Sq + a)
        assert torch.equal(part, outs["16"][:, a:b])
    # non-causal (the training forward of the padded tower heads takes this kernel at head dim 128)
    nc = ops.attention_bshd(q, k, v, causal=False)
    close(nc, ref_attention(q.cpu(), k.cpu(), v.cpu(), False, 1 / math.sqrt(D)), kind, ulps=4.0, floor=0.3)


def test_attention_online_softmax_rescale_branch(ops):
    """Force the running max to jump at a late KV tile (rule: rare branches need their own test)."""
    dt = torch.bfloat16
    S, H, D = 256, 1, 128
    g = torch.Generator().manual_seed(4)
    q = torch.randn(1, S, H, D, generator=g) * 0.3
    k = torch.randn(1, S, H, D, generator=g) * 0.3
    v = torch.randn(1, S, H, D, generator=g)
    k[0, 200, 0] = q[0, 230, 0] * 40          # one key spikes against one query, in the 4th tile
    q, k, v = q.to(dt), k.to(dt), v.to(dt)
    got = ops.attention_bshd(q.cuda(), k.cuda(), v.cuda(), causal=True)
    want = ref_attention(q, k, v, True, 1 / math.sqrt(D))
    close(got, want, "bf16", ulps=3.0, floor=0.3)


@pytest.mark.parametrize("kind", ["bf16", "f16"])
def test_attention_vit_noncausal_padded_heads(ops, kind):
    """SigLIP: 16 heads of 72, stored zero-padded to 96 (QKV weight layout), batch of frames, 729 keys."""
    dt = DT[kind]
    B, S, H, D, DP = 3, 729, 16, 72, 96
    g = torch.Generator().manual_seed(6)
    qkv = torch.zeros(B, S, 3, H, DP)
    qkv[..., :D] = torch.randn(B, S, 3, H, D, generator=g)
    qkv = qkv.to(dt)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    dev = qkv.cuda()
    got = ops.attention_bshd(dev[:, :, 0], dev[:, :, 1], dev[:, :, 2], causal=False, scale=D ** -0.5, d_out=D)
    want = ref_attention(q[..., :D], k[..., :D], v[..., :D], False, D ** -0.5)
    assert got.shape == (B, S, H, D)
    close(got, want, kind, ulps=3.0, floor=0.1)


@pytest.mark.parametrize("kind", ["bf16", "f16"])
def test_attention_vit_heads_packed_at_true_stride(ops, kind):
    """SigLIP as the engine lays it out: q | k | v blocks of 16 heads x 72 packed with head stride 72 in one row (the QKV GEMM
    computes no padding columns); the kernel's 96-wide tile reaches 24 columns into the NEXT head (or the row padding), which
    d_out = 72 makes it ignore - the result must equal attention over the 72 true dims, and the layout with zero-padded heads."""
    import math
    dt = DT[kind]
    B, S, H, D, DP = 3, 729, 16, 72, 96
    g = torch.Generator().manual_seed(16)
    ld = 3584                                                     # 3 * 1152 + 24 readable columns, rounded up to 256
    row = torch.randn(B * S, ld, generator=g).to(dt)              # the pad columns hold (finite) garbage on purpose
    q = row[:, : H * D].view(B, S, H, D)
    k = row[:, H * D: 2 * H * D].view(B, S, H, D)
    v = row[:, 2 * H * D: 3 * H * D].view(B, S, H, D)
    dev = row.cuda()
    out = torch.empty(B * S, H * D, dtype=dt, device="cuda")
    ops.attention(dev, dev[:, H * D:], dev[:, 2 * H * D:], out, B, S, S, H, H, DP, D, ld, ld, ld, out.stride(0), S * ld, S * ld,
                  S * out.stride(0), D, D, D, False, 0, 1 / math.sqrt(D))
    want = ref_attention(q, k, v, False, D ** -0.5)
    close(out.view(B, S, H, D), want, kind, ulps=3.0, floor=0.1)
    padded = torch.zeros(B, S, 3, H, DP)
    padded[:, :, 0, :, :D], padded[:, :, 1, :, :D], padded[:, :, 2, :, :D] = q.float(), k.float(), v.float()
    pd = padded.to(dt).cuda()
    ref2 = ops.attention_bshd(pd[:, :, 0], pd[:, :, 1], pd[:, :, 2], causal=False, scale=D ** -0.5, d_out=D)
    assert torch.equal(ref2.reshape(B * S, H * D), out)           # same arithmetic on the same values: bit for bit


@pytest.mark.parametrize("kind", ["bf16", "f16"])
@pytest.mark.parametrize("B,S", [(32, 729), (3, 729), (1, 729), (5, 256), (2, 128), (40, 729), (7, 200)])
def test_attention_vit_persistent_kernel_equals_one_workgroup_per_item_bitwise(ops, kind, B, S, monkeypatch):
    """r04: attn_vit_persistent_kernel (workgroups walk their items; the last step of an item stages the next item's first tiles and
    loads its Q) = attn_prefill_kernel<96, non-causal, 5 k-steps> (V3D_ATTN_VIT_PERSIST=0), bit for bit: the true shape (32 frames: one
    (frame, head) pair per workgroup), fewer / more pairs than workgroup slots (3, 1, 40 frames), 4 and 2 key tiles, a short last tile
    (200 keys), the output columns beyond the 16 x 72 head dims untouched; and against the f32 reference."""
    dt = DT[kind]
    H, D, DP = 16, 72, 96
    g = torch.Generator().manual_seed(B * 1000 + S)
    qkv = torch.randn(B * S, 3584, generator=g).to(dt).cuda()
    W = H * D

    def run(mode):
        monkeypatch.setenv("V3D_ATTN_VIT_PERSIST", mode)
        att = torch.full((B * S, 1280), 3.0, dtype=dt, device="cuda")
        ld = qkv.stride(0)
        ops.attention(qkv, qkv[:, W:], qkv[:, 2 * W:], att, B, S, S, H, H, DP, D, ld, ld, ld, att.stride(0), S * ld, S * ld, S * att.stride(0),
                      D, D, D, False, 0, D ** -0.5)
        return att
    a, b = run("0"), run("1")
    assert torch.equal(a, b)
    assert bool((b[:, W:] == 3.0).all())
    q = qkv[:, :W].view(B, S, H, D).cpu()
    k = qkv[:, W: 2 * W].view(B, S, H, D).cpu()
    v = qkv[:, 2 * W: 3 * W].view(B, S, H, D).cpu()
    want = ref_attention(q, k, v, False, D ** -0.5)
    close(b[:, :W].view(B, S, H, D), want, kind, ulps=3.0, floor=0.3)


@pytest.mark.parametrize("kind", ["bf16", "f16"])
@pytest.mark.parametrize("past", [0, 5, 700])
def test_attention_decode(ops, kind, past):
    dt = DT[kind]
    H, KV, D = 28, 4, 128
    Sk = past + 1
    g = torch.Generator().manual_seed(past)
    q = torch.randn(1, 1, H, D, generator=g).to(dt)
    kc = torch.randn(1, Sk + 9, KV, D, generator=g).to(dt)   # cache longer than the valid prefix
    vc = torch.randn(1, Sk + 9, KV, D, generator=g).to(dt)
    out = torch.empty(1, 1, H, D, dtype=dt, device="cuda")
    kd, vd = kc.cuda(), vc.cuda()
    ops.attention(q.cuda(), kd, vd, out, 1, 1, Sk, H, KV, D, D, q.stride(1), kd.stride(1), vd.stride(1), out.stride(1),
                  0, 0, 0, D, D, D, True, past, 1 / math.sqrt(D))
    want = ref_attention(q, kc[:, :Sk], vc[:, :Sk], True, 1 / math.sqrt(D), q_pos0=past)
    close(out, want, kind, ulps=2.0, floor=0.3)


def test_patchify_and_copy_rows(ops):
    img = torch.randn(2, 3, 56, 56).to(torch.bfloat16)
    got = ops.patchify(img.cuda(), 14, 640).cpu()
    want = F.unfold(img.float(), 14, stride=14).transpose(1, 2).reshape(2 * 16, 588).to(torch.bfloat16)
    assert torch.equal(got[:, :588], want) and torch.all(got[:, 588:] == 0)
    src = torch.randn(10, 64).to(torch.bfloat16).cuda()
    dst = torch.zeros(20, 128, dtype=torch.bfloat16, device="cuda")
    ops.copy_rows(src, dst[5:15, 32:96])
    assert torch.equal(dst[5:15, 32:96], src) and int((dst != 0).sum()) == int((src != 0).sum())   # nothing else written


@pytest.mark.parametrize("kind", ["bf16", "f16"])
@pytest.mark.parametrize("Sk,H,KV", [(1, 28, 4), (130, 28, 4), (6800, 28, 4), (999, 2, 1), (4097, 8, 1)])
def test_attention_decode_split_kv(ops, kind, Sk, H, KV):
    """Split-KV decode kernel == softmax(q K^T / sqrt(d)) V over the first Sk cache rows."""
    dt = DT[kind]
    D = 128
    g = torch.Generator().manual_seed(Sk)
    q = torch.randn(1, 1, H, D, generator=g).to(dt)
    cache = torch.randn(Sk + 3, 2 * KV * D, generator=g).to(dt)       # [k | v] rows, as the engine lays it out
    cd = cache.cuda()
    out = torch.empty(H * D, dtype=dt, device="cuda")
    ws = ops.decode_workspace(H, KV, "cuda")
    ops.attention_decode(q.cuda().view(-1), cd, cd[:, KV * D:], out, Sk, H, KV, 1 / math.sqrt(D), ws)
    k = cache[:Sk, :KV * D].view(1, Sk, KV, D)
    v = cache[:Sk, KV * D:].view(1, Sk, KV, D)
    want = ref_attention(q, k, v, True, 1 / math.sqrt(D), q_pos0=Sk - 1)
    close(out.view(1, 1, H, D), want, kind, ulps=2.0, floor=0.3)


@pytest.mark.parametrize("lens", [[6800], [130, 130, 130], [1, 999, 6800, 64, 65], list(range(40, 40 + 16 * 37, 37))])
def test_attention_decode_rows_ragged(ops, lens):
    """Scenes decoding together (own cache, own length) in one launch pair: each row equals softmax(q K^T / sqrt(d)) V
    over its own keys, and reproduces the single-scene launch bit for bit whatever the other scenes' lengths (every scene
    partitions its keys by its own length)."""
    dt = torch.bfloat16
    H, KV, D = 28, 4, 128
    M = len(lens)
    g = torch.Generator().manual_seed(sum(lens))
    q = torch.randn(M, H * D, generator=g).to(dt)
    caches = [torch.randn(n + 2, 2 * KV * D, generator=g).to(dt) for n in lens]
    cds = [c.cuda() for c in caches]
    out = torch.empty(M, H * D, dtype=dt, device="cuda")
    one = ops.decode_workspace(H, KV, "cuda")
    ws = torch.empty(one.numel() * M, dtype=torch.float32, device="cuda")
    ops.attention_decode_rows(q.cuda(), cds, [c[:, KV * D:] for c in cds], out, lens, H, KV, 1 / math.sqrt(D), ws)
    for m, n in enumerate(lens):
        k = caches[m][:n, :KV * D].view(1, n, KV, D)
        v = caches[m][:n, KV * D:].view(1, n, KV, D)
        want = ref_attention(q[m].view(1, 1, H, D), k, v, True, 1 / math.sqrt(D), q_pos0=n - 1)
        close(out[m].view(1, 1, H, D), want, "bf16", ulps=2.0, floor=0.3)
    for m, n in enumerate(lens):
        alone = torch.empty(H * D, dtype=dt, device="cuda")
        ops.attention_decode(q[m].cuda(), cds[m], cds[m][:, KV * D:], alone, n, H, KV, 1 / math.sqrt(D), one)
        assert torch.equal(alone, out[m]), f"scene {m} (length {n}) depends on its group"


@pytest.mark.parametrize("kind", ["bf16", "f16"])
@pytest.mark.parametrize("mm", [0, 1])
@pytest.mark.parametrize("P,tails", [(6734, [60, 61, 77, 3, 64]), (500, [1] * 16), (256, [0, 5, 300]), (37, [19, 40]), (1500, list(range(1, 28)))])
def test_attention_decode_rows_shared_prefix(ops, kind, P, tails, mm, monkeypatch):
    """Questions about one scene: the caches start with the same P rows, and with prefix=P nobody reads the other copies
    (poisoned with NaN here).
    mm = 0 (V3D_DEC_PREFIX_MM=0, the r03 form): the split-KV kernels read those keys from the first cache for every question - the
    outputs equal the launch in which every question reads its own copy bit for bit (prefix ends inside a 16-key group / a 64-key chunk /
    a split, tails of 0 rows).
    mm = 1 (r04, default): the prefix keys are ONE matrix-core launch for all rows (attn_prefill16_kernel<PART>: a chunk of the prefix
    through LDS once for all M x 7 query heads of a kv head), the rows' own keys stay with the split kernels, the merge folds both:
    within 2 ulp16 of |v| of the own-copy launch (other f32 summation order), equal to the f32 reference at the decode bound, and a row's
    bits depend neither on M nor on the other rows (up to 27 rows = two query tiles; 1500 = a short last chunk)."""
    dt = DT[kind]
    H, KV, D = 28, 4, 128
    M = len(tails)
    monkeypatch.setenv("V3D_DEC_PREFIX_MM", str(mm))
    g = torch.Generator().manual_seed(P + sum(tails))
    q = torch.randn(M, H * D, generator=g).to(dt).cuda()
    shared = torch.randn(P, 2 * KV * D, generator=g).to(dt)
    caches = []
    for n in tails:
        c = torch.randn(P + n + 3, 2 * KV * D, generator=g).to(dt)
        c[:P] = shared
        caches.append(c.cuda())
    lens = [P + n for n in tails]
    one = ops.decode_workspace(H, KV, "cuda")
    ws = torch.empty(one.numel() * M, dtype=torch.float32, device="cuda")
    own = torch.empty(M, H * D, dtype=dt, device="cuda")
    ops.attention_decode_rows(q, caches, [c[:, KV * D:] for c in caches], own, lens, H, KV, 1 / math.sqrt(D), ws)
    # poison every copy of the prefix but the first: with prefix=P nobody may read them
    for c in caches[1:]:
        c[:P] = float("nan")
    got = torch.empty(M, H * D, dtype=dt, device="cuda")
    ops.attention_decode_rows(q, caches, [c[:, KV * D:] for c in caches], got, lens, H, KV, 1 / math.sqrt(D), ws, prefix=P)
    if mm == 0:
        assert torch.equal(got, own)
    else:
        assert bool(torch.isfinite(got.float()).all())
        close(got, own.float().cpu(), kind, ulps=2.0, floor=0.3)
        for m in (0, M - 1):                                   # against the f32 reference over the row's own keys
            cm = caches[m].clone()
            cm[:P] = shared.cuda()
            k = cm[:lens[m], :KV * D].view(1, lens[m], KV, D).cpu()
            v = cm[:lens[m], KV * D:].view(1, lens[m], KV, D).cpu()
            want = ref_attention(q[m].cpu().view(1, 1, H, D), k, v, True, 1 / math.sqrt(D), q_pos0=lens[m] - 1)
            close(got[m].view(1, 1, H, D), want, kind, ulps=2.0, floor=0.3)
        if M >= 3:                                             # group independence: the last two rows as their own launch
            sub = torch.empty(2, H * D, dtype=dt, device="cuda")
            c2 = caches[M - 2].clone()
            c2[:P] = shared.cuda()                             # (the shared prefix is read from the FIRST cache handed in)
            ops.attention_decode_rows(q[M - 2:].contiguous(), [c2, caches[M - 1]], [c2[:, KV * D:], caches[M - 1][:, KV * D:]], sub, lens[M - 2:],
                                      H, KV, 1 / math.sqrt(D), ws, prefix=P)
            assert torch.equal(sub, got[M - 2:])
    with pytest.raises(Exception):
        ops.attention_decode_rows(q, caches, [c[:, KV * D:] for c in caches], got, [P - 1] + lens[1:], H, KV, 1 / math.sqrt(D), ws, prefix=P)


def test_rope_kv_store_equals_rope_apply_plus_copy(ops):
    """The fused prefill rotary + cache append (contiguous and indexed destination rows, given positions) against
    v3d_rope_apply + v3d_copy_rows, bit for bit."""
    dt = torch.bfloat16
    H, KV, D, S = 6, 2, 128, 333
    table = ops.RopeTable(D, 1024, 1e6, dt, "cuda")
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(S, (H + 2 * KV) * D + 128, generator=g).to(dt).cuda()          # padded row stride
    ref = qkv.clone()
    ops.rope_apply(ref, H + KV, D, table, pos0=100)
    cache = torch.zeros(700, 2 * KV * D + 64, dtype=dt, device="cuda")
    got = qkv.clone()
    ops.rope_kv_store(got, H, KV, D, table, cache, pos0=100)
    assert torch.equal(got[:, : H * D], ref[:, : H * D])
    assert torch.equal(got[:, H * D:], qkv[:, H * D:])                                # k / v columns of the buffer untouched
    assert torch.equal(cache[100:100 + S, : 2 * KV * D], ref[:, H * D: (H + 2 * KV) * D])
    assert not bool(cache[:100].any()) and not bool(cache[100 + S:].any()) and not bool(cache[:, 2 * KV * D:].any())
    # indexed form: two "questions" of one scene appending at rows 50.. of their own caches inside one allocation
    pos = torch.cat([torch.arange(50, 50 + 200), torch.arange(50, 50 + 133)]).to(torch.int32).cuda()
    rows = torch.cat([torch.arange(50, 250), 350 + torch.arange(50, 183)]).to(torch.int64).cuda()
    ref2 = qkv.clone()
    ops.rope_apply(ref2, H + KV, D, table, positions=pos)
    cache2 = torch.zeros(700, 2 * KV * D, dtype=dt, device="cuda")
    got2 = qkv.clone()
    ops.rope_kv_store(got2, H, KV, D, table, cache2, positions=pos, dst_rows=rows)
    assert torch.equal(got2[:, : H * D], ref2[:, : H * D])
    assert torch.equal(cache2[rows], ref2[:, H * D: (H + 2 * KV) * D])
    from v3d._native import V3DError
    with pytest.raises(V3DError):
        ops.rope_kv_store(got, H, KV, D, table, cache, pos0=100, row0=500)            # 500 + 333 rows > 700


def test_rope_rows_and_argmax_rows(ops):
    """Batched rotary + cache append (own position and cache row per scene) and batched argmax == the single-row calls."""
    dt = torch.bfloat16
    H, KV, D, M = 4, 2, 128, 5
    table = ops.RopeTable(D, 512, 1e6, dt, "cuda")
    g = torch.Generator().manual_seed(3)
    rows = torch.randn(M, (H + 2 * KV) * D + 64, generator=g).to(dt).cuda()       # padded row stride
    pos = [0, 77, 511, 3, 200]
    want_rows, want_cache = [], []
    for m in range(M):
        r = rows[m, : (H + 2 * KV) * D].clone()
        c = torch.zeros(2 * KV * D, dtype=dt, device="cuda")
        ops.rope_kv_append(r, H, KV, D, table, pos[m], c)
        want_rows.append(r)
        want_cache.append(c)
    got = rows.clone()
    crow = [torch.zeros(2 * KV * D, dtype=dt, device="cuda") for _ in range(M)]
    ops.rope_kv_append_rows(got[:, : (H + 2 * KV) * D], H, KV, D, table, pos, crow)
    for m in range(M):
        assert torch.equal(got[m, : (H + 2 * KV) * D], want_rows[m]) and torch.equal(crow[m], want_cache[m])
    assert torch.equal(got[:, (H + 2 * KV) * D:], rows[:, (H + 2 * KV) * D:])       # the padding is untouched
    with pytest.raises(Exception, match="outside the table"):
        ops.rope_kv_append_rows(got[:, : (H + 2 * KV) * D], H, KV, D, table, [0, 1, 2, 3, 512], crow)
    x = torch.randn(M, 152064 + 64, generator=g).to(dt)
    x[1, 777] = x[1, 12] = 60.0            # tie -> lowest index
    x[3, 152063] = 70.0                    # last valid column; the padding columns behind it hold larger values
    x[:, 152064:] = 100.0
    xd = x.cuda()
    idx = torch.zeros(M, dtype=torch.int64, device="cuda")
    ops.argmax_rows(xd[:, :152064], idx, torch.empty(256 * M, dtype=torch.float32, device="cuda"))
    assert idx.tolist() == x[:, :152064].float().argmax(1).tolist() and idx[1].item() == 12 and idx[3].item() == 152063


@pytest.mark.parametrize("kind", ["bf16", "f16"])
def test_linear_decode_fused_variants(ops, kind):
    """Decode-step linears: fused RMSNorm prologue, bias / residual / SwiGLU epilogues vs the oracle ops."""
    dt = DT[kind]
    g = torch.Generator().manual_seed(12)
    K, N, I = 512, 384, 256
    x = (torch.randn(K, generator=g)).to(dt)
    lnw = (1 + 0.1 * torch.randn(K, generator=g)).to(dt)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dt)
    b = torch.randn(N, generator=g).to(dt)
    r = torch.randn(N, generator=g).to(dt)
    xn = L.rmsnorm(x[None], lnw, 1e-6)[0]
    out = torch.empty(N, dtype=dt, device="cuda")
    ops.linear_decode(x.cuda(), w.cuda(), out, norm_weight=lnw.cuda(), eps=1e-6, bias=b.cuda(), epilogue=ops.DEC_BIAS)
    close(out, (xn.float() @ w.float().t() + b.float()).to(dt), kind, ulps=2.5)
    ops.linear_decode(x.cuda(), w.cuda(), out, res=r.cuda(), epilogue=ops.DEC_RES)
    close(out, (r.float() + (x.float() @ w.float().t()).to(dt).float()).to(dt), kind, ulps=2.5, floor=1.0)
    ops.linear_decode(x.cuda(), w.cuda(), out)
    close(out, (x.float() @ w.float().t()).to(dt), kind, ulps=2.5)
    wg = (torch.randn(I, K, generator=g) * 0.05).to(dt)
    wu = (torch.randn(I, K, generator=g) * 0.05).to(dt)
    want = (F.silu((xn.float() @ wg.float().t()).to(dt).float()).to(dt).float() * (xn.float() @ wu.float().t()).to(dt).float()).to(dt)
    act = torch.empty(I, dtype=dt, device="cuda")
    ops.linear_decode(x.cuda(), ops.interleave_gate_up(wg, wu).cuda(), act, norm_weight=lnw.cuda(), eps=1e-6, epilogue=ops.DEC_SWIGLU)
    close(act, want, kind, ulps=3.0, floor=0.3)


@pytest.mark.parametrize("M", [1, 2, 3, 4, 7, 16, 17, 32])
@pytest.mark.parametrize("K,N", [(512, 384), (3584, 512), (4736, 256), (18944, 128), (520, 36)])
def test_linear_decode_rows(ops, M, K, N):
    """Scenes decoding together share one pass over the weights.  Matrix-core shapes (K % 128 == 0, N % 16 == 0), M >= 2:
    a row's result depends neither on the other rows nor on M (bit for bit), and equals the one-row kernel up to the f32
    summation order (checked against an f64 reference: 2^-7 |ref| + 2^-8 rms).  Other shapes / fused norm (VALU form,
    M <= 4): bit-identical to the one-row kernel."""
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(100 + M)
    x = torch.randn(M, K, generator=g).to(dt).cuda()
    lnw = (1 + 0.1 * torch.randn(K, generator=g)).to(dt).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).to(dt).cuda()
    b = torch.randn(N, generator=g).to(dt).cuda()
    r = torch.randn(M, N, generator=g).to(dt).cuda()
    cases = [dict(), dict(bias=b, epilogue=ops.DEC_BIAS), dict(res=r, epilogue=ops.DEC_RES), dict(epilogue=ops.DEC_SWIGLU)]
    mfma_shape = K % 128 == 0 and N % 16 == 0
    if M > 4 and not mfma_shape:
        with pytest.raises(Exception, match="activation rows"):
            ops.linear_decode_rows(x, w, torch.empty((M, N), dtype=dt, device="cuda"))
        return
    if K <= 4096 and M <= 4:
        cases += [dict(norm_weight=lnw, eps=1e-6, bias=b, epilogue=ops.DEC_BIAS), dict(norm_weight=lnw, eps=1e-6, epilogue=ops.DEC_SWIGLU)]
    for kw in cases:
        swiglu = kw.get("epilogue") == ops.DEC_SWIGLU
        n_out = N // 2 if swiglu else N
        if swiglu and N % 128:
            continue
        got = torch.full((M, n_out), 7.0, dtype=dt, device="cuda")
        ops.linear_decode_rows(x, w, got, **kw)
        valu_form = M == 1 or not mfma_shape or "norm_weight" in kw
        if valu_form:
            for m in range(M):
                one = torch.empty(n_out, dtype=dt, device="cuda")
                kw1 = dict(kw)
                if "res" in kw1:
                    kw1["res"] = r[m]
                ops.linear_decode(x[m], w, one, **kw1)
                assert torch.equal(got[m], one), (kw.get("epilogue"), m)
        else:       # independence of the other rows and of M: the last two rows as their own batch
            sub = torch.empty((2, n_out), dtype=dt, device="cuda")
            kw2 = dict(kw)
            if "res" in kw2:
                kw2["res"] = r[M - 2:]
            ops.linear_decode_rows(x[M - 2:], w, sub, **kw2)
            assert torch.equal(sub, got[M - 2:])
            if M > 16:      # r04, two 16-row blocks per pass: the first rows as a group of 16 (the one-block kernel): the same bits
                sub16 = torch.empty((16, n_out), dtype=dt, device="cuda")
                kw3 = dict(kw)
                if "res" in kw3:
                    kw3["res"] = r[:16]
                ops.linear_decode_rows(x[:16], w, sub16, **kw3)
                assert torch.equal(sub16, got[:16])
            want = x.double() @ w.double().t()                   # f64 reference of the plain product (the epilogues are checked above / below)
            if not kw:
                err = (got.double() - want).abs()
                assert bool((err <= 2.0 ** -7 * want.abs() + 2.0 ** -8 * want.pow(2).mean().sqrt()).all())
        if "norm_weight" not in kw and not swiglu:       # and the values are right (f64 reference)
            ref = x.double() @ w.double().t()
            if "bias" in kw:
                ref = ref + b.double()
            if "res" in kw:
                ref = ref.to(dt).double() + r.double()
            err = (got.double() - ref).abs()
            assert bool((err <= 2.0 ** -7 * ref.abs() + 2.0 ** -8 * ref.pow(2).mean().sqrt()).all())
        if swiglu and "norm_weight" not in kw:
            rr = (x.double() @ w.double().t()).to(dt).float().view(M, N // 128, 2, 64)
            want = (F.silu(rr[:, :, 0]).to(dt).float() * rr[:, :, 1]).reshape(M, N // 2).double()
            err = (got.double() - want).abs()
            assert bool((err <= 2.0 ** -6 * want.abs() + 2.0 ** -6 * want.pow(2).mean().sqrt()).all())


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [2, 16, 17, 32])
@pytest.mark.parametrize("K,N", [(1152, 256), (2048, 4608), (3584, 512), (3584, 8192), (4096, 384), (1024, 128), (4224, 256), (18944, 256)])
def test_linear_decode_rows_pipelined_forms_equal_the_r04_kernel_bitwise(ops, monkeypatch, dt, M, K, N):
    """r04, second session: linear_decode_mfma2_kernel (persistent workgroups, activation fragments resident in registers, weight tiles
    double-buffered in registers; 9..32 K tiles) and linear_decode_mfma_stream_kernel (any K) sum an output exactly as
    linear_decode_mfma_kernel does - slice w of K = tiles w, w + 8, ... in ascending order, the eight slices added in order - so every bit
    agrees.  V3D_DEC_V2 = 0: the r04 kernel; 2: the persistent form wherever the shape allows; 3: the streaming form (grids below, at and
    above the chip's CU count, 2..4 tiles per wave, waves with one tile fewer than their neighbours, odd and even numbers of groups per
    workgroup).  K > 4096 without SwiGLU is the K-split form at V3D_DEC_V2 = 1 / 2 (another f32 summation order: next test)."""
    g = torch.Generator().manual_seed(7 * M + K % 97)
    x = torch.randn(M, K, generator=g).to(dt).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).to(dt).cuda()
    b = torch.randn(N, generator=g).to(dt).cuda()
    r = torch.randn(M, N, generator=g).to(dt).cuda()
    cases = [dict(), dict(bias=b, epilogue=ops.DEC_BIAS), dict(res=r, epilogue=ops.DEC_RES)]
    if N % 128 == 0:
        cases.append(dict(epilogue=ops.DEC_SWIGLU))
    for kw in cases:
        swiglu = kw.get("epilogue") == ops.DEC_SWIGLU
        n_out = N // 2 if swiglu else N
        outs = []
        for mode in ("0", "3") + (("2",) if K <= 4096 or swiglu else ()):
            monkeypatch.setenv("V3D_DEC_V2", mode)
            o = torch.full((M, n_out), 5.0, dtype=dt, device="cuda")
            ops.linear_decode_rows(x, w, o, **kw)
            outs.append(o)
        for o in outs[1:]:
            assert torch.equal(outs[0], o), (kw.get("epilogue"), M, K, N)
    monkeypatch.setenv("V3D_DEC_V2", "2")                   # rows do not depend on their neighbours or on M (absent rows' columns are never stored)
    o_all = torch.empty((M, N), dtype=dt, device="cuda")
    ops.linear_decode_rows(x, w, o_all)
    o_two = torch.empty((2, N), dtype=dt, device="cuda")
    ops.linear_decode_rows(x[M - 2:], w, o_two)
    assert torch.equal(o_two, o_all[M - 2:])


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("K,N", [(4224, 256), (4736, 512), (18944, 3584), (18944, 128), (9216, 4112)])
def test_linear_decode_rows_k_split(ops, monkeypatch, dt, K, N):
    """K > 4096 (down_proj): K is cut into chunks whose activation fragments stay in registers; the chunks' f32 sums meet in a second
    launch, added in ascending order.  The cut depends on K, N and the chip only: a row's bits depend neither on M nor on its neighbours
    (2, 16, 17 and 32 rows), two launches agree bit for bit, the values meet the f64 reference as the r04 kernel does (2^-7 |ref| + 2^-8
    rms) and stay within one 16-bit rounding (of |value| + rms) of the r04 kernel's."""
    g = torch.Generator().manual_seed(K % 1013 + N)
    x = torch.randn(32, K, generator=g).to(dt).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).to(dt).cuda()
    b = torch.randn(N, generator=g).to(dt).cuda()
    r = torch.randn(32, N, generator=g).to(dt).cuda()
    monkeypatch.setenv("V3D_DEC_V2", "1")
    for kw in (dict(), dict(bias=b, epilogue=ops.DEC_BIAS), dict(res=r, epilogue=ops.DEC_RES)):
        full = torch.full((32, N), 3.0, dtype=dt, device="cuda")
        ops.linear_decode_rows(x, w, full, **kw)
        again = torch.full((32, N), 4.0, dtype=dt, device="cuda")
        ops.linear_decode_rows(x, w, again, **kw)
        assert torch.equal(full, again)
        for lo, hi in ((30, 32), (0, 16), (3, 20), (9, 11)):
            kw2 = dict(kw)
            if "res" in kw2:
                kw2["res"] = r[lo:hi]
            sub = torch.full((hi - lo, N), 9.0, dtype=dt, device="cuda")
            ops.linear_decode_rows(x[lo:hi], w, sub, **kw2)
            assert torch.equal(sub, full[lo:hi]), (kw.get("epilogue"), lo, hi)
        ref = x.double() @ w.double().t()
        if "bias" in kw:
            ref = ref + b.double()
        if "res" in kw:
            ref = ref.to(dt).double() + r.double()
        err = (full.double() - ref).abs()
        assert bool((err <= 2.0 ** -7 * ref.abs() + 2.0 ** -8 * ref.pow(2).mean().sqrt()).all())
        monkeypatch.setenv("V3D_DEC_V2", "0")
        old = torch.empty((32, N), dtype=dt, device="cuda")
        ops.linear_decode_rows(x, w, old, **kw)
        monkeypatch.setenv("V3D_DEC_V2", "1")
        ulp = 2.0 ** (-7 if dt == torch.bfloat16 else -10)
        rms = old.double().pow(2).mean().sqrt()
        assert bool(((full.double() - old.double()).abs() <= ulp * (old.double().abs() + rms)).all())


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,Sq,P", [(3, 60, 6734), (2, 61, 200), (4, 1, 64), (2, 130, 130), (1, 17, 63), (5, 64, 1024), (2, 33, 6720)])
def test_attention_shared_prefix_equals_attention_on_full_copies_bitwise(ops, dt, B, Sq, P):
    """r04: the question rows of an answer batch (v3d_attention_shared_prefix).  Key tiles below 64 floor(P / 64) come from ONE cache, the
    rest from each question's own cache - whose rows below that are POISONED here (NaN) to prove nobody reads them - and a workgroup's
    query slots hold the rows of all seven query heads of a kv head.  Same tiles in the same order, the raise decided per lane: every
    output bit equals v3d_attention over B full copies (prefix lengths on / off a tile boundary, below one tile, rows filling less and
    more than one 128-slot tile, a single row)."""
    Hq, Hkv, hd = 28, 4, 128
    g = torch.Generator().manual_seed(B * 1000 + Sq + P)
    kvw = Hkv * hd
    Sk = P + Sq
    cap = Sk + 7
    scene = torch.randn(cap, 2 * kvw, generator=g).to(dt).cuda()
    own = torch.randn(B, cap, 2 * kvw, generator=g).to(dt).cuda()
    full = own.clone()
    full[:, :P] = scene[:P]
    q = (torch.randn(B * Sq, Hq * hd, generator=g) * 2).to(dt).cuda()
    P0 = (P // 64) * 64
    own[:, P0:P] = scene[P0:P]
    own[:, :P0] = float("nan")
    scale = 1.0 / math.sqrt(hd)
    want = torch.empty(B * Sq, Hq * hd, dtype=dt, device="cuda")
    c2 = full.view(-1, 2 * kvw)
    ops.attention(q, c2, c2[:, kvw:], want, B, Sq, Sk, Hq, Hkv, hd, hd, q.stride(0), c2.stride(0), c2.stride(0), want.stride(0), Sq * q.stride(0),
                  full.stride(0), Sq * want.stride(0), hd, hd, hd, True, P, scale)
    got = torch.full((B * Sq, Hq * hd), 3.0, dtype=dt, device="cuda")
    o2 = own.view(-1, 2 * kvw)
    ops.attention_shared_prefix(q, o2, o2[:, kvw:], scene, scene[:, kvw:], P0, got, B, Sq, Sk, Hq, Hkv, q.stride(0), o2.stride(0), o2.stride(0),
                                got.stride(0), Sq * q.stride(0), own.stride(0), Sq * got.stride(0), hd, hd, hd, P, scale)
    assert not bool(torch.isnan(got.float()).any())
    assert torch.equal(got, want)
    with pytest.raises(Exception, match="shared_len"):
        ops.attention_shared_prefix(q, o2, o2[:, kvw:], scene, scene[:, kvw:], P0 + 8, got, B, Sq, Sk, Hq, Hkv, q.stride(0), o2.stride(0),
                                    o2.stride(0), got.stride(0), Sq * q.stride(0), own.stride(0), Sq * got.stride(0), hd, hd, hd, P, scale)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [5, 16, 17, 32])
@pytest.mark.parametrize("K,N", [(1152, 256), (2048, 4608), (3584, 512), (3584, 37888 // 4)])
def test_linear_decode_rows_fused_norm_equals_rmsnorm_then_linear_bitwise(ops, monkeypatch, dt, M, K, N):
    """r04: with more than four rows the persistent decode linear forms Qwen2RMSNorm of its input rows itself (1 / rms in v3d_rmsnorm's
    summation order for that many rows, weight * round(x * r) element by element): the bits of rmsnorm() followed by the unfused call,
    for the bias, SwiGLU and plain epilogues (QKV, gate/up, LM head).  Not offered (the engine then normalises first): up to four rows
    (VALU form, bit-identical to the one-row kernel), K > 4096, a residual epilogue, V3D_DEC_V2=0."""
    g = torch.Generator().manual_seed(3 * M + K)
    x = (torch.randn(M, K, generator=g) * 3).to(dt).cuda()
    lnw = (1 + 0.1 * torch.randn(K, generator=g)).to(dt).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).to(dt).cuda()
    b = torch.randn(N, generator=g).to(dt).cuda()
    monkeypatch.setenv("V3D_DEC_V2", "1")
    for kw in (dict(), dict(bias=b, epilogue=ops.DEC_BIAS), dict(epilogue=ops.DEC_SWIGLU)):
        swiglu = kw.get("epilogue") == ops.DEC_SWIGLU
        if swiglu and N % 128:
            continue
        n_out = N // 2 if swiglu else N
        assert ops.linear_decode_rows_fuses_norm(M, N, K, kw.get("epilogue", ops.DEC_NONE))
        h = ops.rmsnorm(x, lnw, 1e-6)
        want = torch.empty((M, n_out), dtype=dt, device="cuda")
        ops.linear_decode_rows(h, w, want, **kw)
        got = torch.full((M, n_out), 2.0, dtype=dt, device="cuda")
        ops.linear_decode_rows(x, w, got, norm_weight=lnw, eps=1e-6, **kw)
        assert torch.equal(got, want), (kw.get("epilogue"), M, K, N)
    assert not ops.linear_decode_rows_fuses_norm(4, N, K)
    assert not ops.linear_decode_rows_fuses_norm(M, N, 18944)
    assert not ops.linear_decode_rows_fuses_norm(M, N, K, ops.DEC_RES)
    monkeypatch.setenv("V3D_DEC_V2", "0")
    assert not ops.linear_decode_rows_fuses_norm(M, N, K)
    with pytest.raises(Exception, match="activation rows"):
        ops.linear_decode_rows(x, w, torch.empty((M, N), dtype=dt, device="cuda"), norm_weight=lnw, eps=1e-6)


def test_rope_kv_append_and_argmax(ops):
    dt = torch.bfloat16
    H, KV, D = 4, 2, 128
    table = ops.RopeTable(D, 512, 1e6, dt, "cuda")
    row = torch.randn((H + 2 * KV) * D).to(dt)
    ref = row.clone().cuda().view(1, -1)
    ops.rope_apply(ref, H + KV, D, table, pos0=77)
    got = row.clone().cuda()
    cache_row = torch.zeros(2 * KV * D, dtype=dt, device="cuda")
    ops.rope_kv_append(got, H, KV, D, table, 77, cache_row)
    assert torch.equal(got, ref[0])
    assert torch.equal(cache_row, ref[0, H * D:])
    x = torch.randn(152064).to(dt)
    x[77777] = x[1234] = 50.0           # tie -> lowest index
    idx = torch.zeros(1, dtype=torch.int64, device="cuda")
    ops.argmax(x.cuda(), idx)
    assert int(idx) == 1234 == int(torch.argmax(x.float()))
