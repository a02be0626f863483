"""GPU parity tests for the 3-D position path (K1-K9): HIP kernels, called through the C ABI,
against (a) the golden vectors produced by the reference itself and (b) the CPU oracle on seeded
inputs, up to BASELINE's full size (32 frames, 384x384 coords, 729x3584 features)."""
import numpy as np
import pytest
import torch

from oracle import v3d_oracle as O

pytestmark = pytest.mark.gpu

KINDS = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}
EPS16 = {"f16": 2.0 ** -10, "bf16": 2.0 ** -7}


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from v3d import ops as _ops
    return _ops


def dev(a, kind=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if kind is not None:
        t = t.to(KINDS[kind])
    return t.cuda()


def host32(t):
    return t.float().cpu().numpy()


def from_bits(bits, kind):
    if kind == "bf16":
        return O.bf16_bits_to_f32(bits)
    return np.asarray(bits).astype(np.float32)


def ulp16(a, b, kind):
    def key(x):
        if kind == "f16":
            i = np.asarray(x, np.float32).astype(np.float16).view(np.int16).astype(np.int32)
        else:
            i = O.f32_to_bf16_bits(x).view(np.int16).astype(np.int32)
        return np.where(i < 0, -(i & 0x7FFF), i)
    return np.abs(key(a) - key(b))


# ------------------------------------------------------------------------------ K1 / K2


def test_unproject_golden(ops, golden):
    g = golden("unproject")
    out = ops.unproject(dev(g["intrinsics"]), dev(g["poses"]), dev(g["depth"].astype(np.float32)))
    # fp tolerance: the reference's 4-term dot runs in BLAS (order / FMA unspecified)
    np.testing.assert_allclose(host32(out), g["world"], rtol=2e-6, atol=2e-6)


def synth_scene(V, H=480, W=640, seed=0):
    g = np.random.default_rng(seed)
    depth = g.integers(400, 5000, size=(V, H, W), dtype=np.uint16)
    depth[:, :3, :5] = 0
    K = np.zeros((V, 4, 4), np.float32)
    K[:, 0, 0] = K[:, 1, 1] = 577.87
    K[:, 0, 2], K[:, 1, 2] = 319.5, 239.5
    K[:, 2, 2] = K[:, 3, 3] = 1
    P = np.zeros((V, 4, 4), np.float32)
    for v in range(V):
        a = g.uniform(0, 2 * np.pi)
        P[v, :3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]
        P[v, :3, 3] = g.normal(0, 1.5, 3)
        P[v, 3, 3] = 1
    return depth, K, P


def test_unproject_full_size_vs_oracle(ops):
    depth, K, P = synth_scene(4)
    got = host32(ops.unproject(dev(K), dev(P), dev(depth.astype(np.float32))))
    want = O.unproject(K, P, depth.astype(np.float32))
    # both sides use the same sequential IEEE order -> identical except for numpy's einsum blocking
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6)


def test_unproject_odd_width_scalar_path(ops):
    depth, K, P = synth_scene(2, H=11, W=13, seed=3)
    got = host32(ops.unproject(dev(K), dev(P), dev(depth.astype(np.float32))))
    np.testing.assert_allclose(got, O.unproject(K, P, depth.astype(np.float32)), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("kind", ["f32", "f16", "bf16"])
def test_unproject_sampled_vs_oracle(ops, kind):
    """K1+K2 fused == unproject -> nearest resize + centre crop -> dtype cast (eval driver's .half())."""
    depth, K, P = synth_scene(3, seed=5)
    d16 = torch.from_numpy(depth.view(np.int16)).cuda()
    got = host32(ops.unproject_sampled(d16, dev(K), dev(P), 384, KINDS[kind]))
    full = O.unproject(K, P, depth.astype(np.float32))
    want = O.round_to(O.resize_crop_coords(full, 384), kind)
    assert got.shape == (3, 384, 384, 3)
    if kind == "f32":
        np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6)
    else:
        d = ulp16(got, want, kind)
        assert d.max() <= 1 and (d != 0).mean() < 1e-3


# ------------------------------------------------------------------------------ K3 / K4


def test_coord_pool_voxel_golden_bitexact(ops, golden):
    g = golden("coord_pool")
    x16 = g["coords_f16"]
    x32 = x16.astype(np.float32) * np.float32(1.001) + np.float32(0.0003)
    avg, vox, ids = ops.coord_pool_voxel(dev(x32))
    assert np.array_equal(host32(avg), g["avg_f32"])
    assert np.array_equal(host32(vox), g["vox_f32"])
    assert np.array_equal(ids.cpu().numpy(), g["vox_f32"].astype(np.int32))
    avg, vox, ids = ops.coord_pool_voxel(dev(x16))
    assert np.array_equal(host32(avg), g["avg_f16"].astype(np.float32))
    assert np.array_equal(host32(vox), g["vox_f16"].astype(np.float32))
    assert np.array_equal(ids.cpu().numpy(), g["vox_f16"].astype(np.int32))


@pytest.mark.parametrize("kind", ["f32", "f16", "bf16"])
def test_coord_pool_voxel_full_size_bitexact(ops, kind):
    """BASELINE size: 32 frames of 384x384x3.  Voxel ids and patch means bit-exact vs the oracle."""
    g = np.random.default_rng(11)
    x = ((g.random((32, 384, 384, 3), dtype=np.float32) - 0.5) * np.array([34, 34, 12], np.float32))
    x = O.round_to(x, kind)
    avg, vox, ids = ops.coord_pool_voxel(dev(x, kind))
    want_avg = O.average_coordinate_in_patch(x, kind)
    want_vox = O.discrete_coords(want_avg, kind)
    assert np.array_equal(host32(avg), want_avg)
    assert np.array_equal(host32(vox), want_vox)
    assert np.array_equal(ids.cpu().numpy(), want_vox.astype(np.int32))
    assert ids.min().item() >= 0 and ids.max().item() <= 300


def test_discrete_coords_exhaustive_f16(ops, golden):
    g = golden("discrete_coords")
    h = np.arange(65536, dtype=np.uint16).view(np.float16)[g["finite_mask"]]
    xyz = np.stack([h, h, h], -1)
    vox, ids = ops.discrete_coords(dev(xyz), want_ids=True)
    assert np.array_equal(vox.cpu().numpy(), g["vox_f16"])
    assert np.array_equal(ids.cpu().numpy(), g["vox_f16"].astype(np.int32))


def test_discrete_coords_exhaustive_bf16_vs_oracle(ops):
    bits = np.arange(65536, dtype=np.uint16)
    x = O.bf16_bits_to_f32(bits)
    x = x[np.isfinite(x)]
    xyz = np.stack([x, x, x], -1)
    vox = ops.discrete_coords(dev(xyz, "bf16"))
    assert np.array_equal(host32(vox), O.discrete_coords(xyz, "bf16"))


def test_discrete_coords_f32_ties_golden(ops, golden):
    g = golden("discrete_coords")
    vox = ops.discrete_coords(dev(g["xyz_f32"]))
    assert np.array_equal(host32(vox), g["vox_f32"])


def test_discrete_coords_nan_and_empty(ops):
    x = torch.tensor([[float("nan"), 1.0, -1.0]], device="cuda")
    v = ops.discrete_coords(x)
    assert torch.isnan(v[0, 0]) and v[0, 1].item() == 160 and v[0, 2].item() == 40
    e = ops.discrete_coords(torch.zeros((0, 3), device="cuda"))
    assert e.shape == (0, 3)


# ------------------------------------------------------------------------------ K5


def test_sin3d_table_golden(ops, golden):
    g = golden("sin3d_table_3584")
    t = ops.Sin3DTable(3584, 301, torch.float16, "cuda", dim_t=torch.from_numpy(g["dim_t"]), keep_f32=True)
    # correctly-rounded f32 trig vs torch's vectorised trig: both within 1 ulp (<= 1.2e-7 for |v|<=1)
    np.testing.assert_allclose(t.table_f32.cpu().numpy(), g["table"], rtol=0, atol=1.5e-7)
    # the same dim_t expression evaluated by the host mirror reproduces the golden dim_t bit for bit
    assert np.array_equal(ops.reference_dim_t(1194).numpy(), g["dim_t"])
    # shifted 16-bit layout: axis a's row holds T(value) at element shift_a + j, zero elsewhere
    tab = t.table.float().cpu().numpy()
    ref16 = g["table"].astype(np.float16).astype(np.float32)
    for a, shift in enumerate((0, 2, 4)):
        d = ulp16(tab[a][:, shift:shift + 1194], ref16, "f16")
        assert d.max() <= 1 and (d != 0).mean() < 1e-3
        assert np.all(tab[a][:, :shift] == 0) and np.all(tab[a][:, shift + 1194:] == 0)


@pytest.mark.parametrize("kind", ["f16", "bf16"])
def test_sin3d_pe_tokens_golden(ops, golden, kind):
    g = golden("sin3d_tokens_3584")
    t = golden("sin3d_table_3584")
    ids = O.round_to(g["ids"].astype(np.float32), kind)
    out = host32(ops.sin3d_pe(dev(ids, kind), 3584, dim_t=torch.from_numpy(t["dim_t"])))
    ref = from_bits(g["pe_f16"] if kind == "f16" else g["pe_bf16_bits"], kind)
    d = ulp16(out, ref, kind)
    assert d.max() <= 1 and (d != 0).mean() < 2e-3


def test_sin3d_pe_small_odd_and_continuous(ops, golden):
    g = golden("sin3d_small")
    o16 = host32(ops.sin3d_pe(dev(g["xyz"]), 16, dim_t=torch.from_numpy(g["dim_t5"])))
    np.testing.assert_allclose(o16, g["pe16"], rtol=0, atol=5e-7)
    o96 = host32(ops.sin3d_pe(dev(g["xyz"]), 96, dim_t=torch.from_numpy(g["dim_t32"])))
    np.testing.assert_allclose(o96, g["pe96"], rtol=0, atol=5e-7)


# ------------------------------------------------------------------------------ K7 / K8 / fused


@pytest.mark.parametrize("kind", ["f32", "f16", "bf16"])
def test_pool_only_golden(ops, golden, kind):
    g = golden("pool2d_bilinear")
    out = host32(ops.visual_tokens(dev(g["feat"], kind), side=27, n=14, pool=True)).reshape(2, 196, 64)
    if kind == "f32":
        np.testing.assert_allclose(out, g["out_f32"], rtol=1e-6, atol=1e-6)
        # and bit-exact against the oracle, which rounds in the same order
        assert np.array_equal(out, O.get_2dpool_bilinear(g["feat"], "f32"))
    else:
        ref = from_bits(g["out_f16"] if kind == "f16" else g["out_bf16_bits"], kind)
        d = ulp16(out, ref, kind)
        assert d.max() <= 1 and (d != 0).mean() < 1e-3
        assert np.array_equal(out, O.get_2dpool_bilinear(O.round_to(g["feat"], kind), kind))


def test_newline_only_golden(ops, golden):
    g = golden("add_token_per_grid")
    out = ops.visual_tokens(dev(g["feat"]), newline=dev(g["newline"]), n=14, pool=False)
    assert np.array_equal(host32(out), g["out"])


def fused_bound(feat, kind):
    """|got - ref| <= eps16 * (|pooled| + |pe|<=1): each operand may be one 16-bit ulp off."""
    pooled = O.get_2dpool_bilinear(feat, kind)
    return O.add_token_per_grid(EPS16[kind] * (np.abs(pooled) + 1.0), np.zeros(feat.shape[-1], np.float32))


@pytest.mark.parametrize("kind", ["f32", "f16", "bf16"])
def test_fused_small_golden(ops, golden, kind):
    """coords -> ids -> (pool + PE + add + newline) against the reference's own composition."""
    g = golden("fused_small")
    coords = O.round_to(g["coords_f16"].astype(np.float32), kind)
    feat = O.round_to(g["feat"], kind)
    C = feat.shape[-1]
    _, _, ids = ops.coord_pool_voxel(dev(coords, kind))
    assert np.array_equal(ids.cpu().numpy().astype(np.float32), g["vox_" + kind])          # bit-exact ids
    table = ops.Sin3DTable(C, 301, KINDS[kind], "cuda", dim_t=torch.from_numpy(g["dim_t"]))
    out = host32(ops.visual_tokens(dev(feat, kind), ids, table, dev(g["newline"], kind)))
    ref = g["seq_" + kind]
    if kind == "f32":
        np.testing.assert_allclose(out, ref, rtol=2e-6, atol=2e-6)
    else:
        ref = from_bits(ref, kind)
        assert np.all(np.abs(out - ref) <= fused_bound(feat, kind))
        assert (out != ref).mean() < 2e-3


@pytest.mark.parametrize("kind", ["f16", "bf16"])
def test_fused_full_size_vs_oracle(ops, kind):
    """BASELINE shape: 32 frames x 729 x 3584 -> [6720, 3584].  The oracle is evaluated on three
    frames (frames are independent); the rest is covered by size-independent properties."""
    V, C = 32, 3584
    g = np.random.default_rng(21)
    coords = O.round_to((g.random((V, 384, 384, 3), dtype=np.float32) - 0.5) * np.array([30, 30, 10], np.float32), kind)
    newline = O.round_to(g.standard_normal(C, dtype=np.float32), kind)
    feat_t = torch.randn((V, 729, C), generator=torch.Generator().manual_seed(22)).to(KINDS[kind])
    dim_t = ops.reference_dim_t(C // 3)
    _, _, ids = ops.coord_pool_voxel(dev(coords, kind))
    table = ops.Sin3DTable(C, 301, KINDS[kind], "cuda", dim_t=dim_t)
    out_t = ops.visual_tokens(feat_t.cuda(), ids, table, dev(newline, kind))
    assert out_t.shape == (V * 14 * 15, C)
    out = out_t.float().cpu().numpy().reshape(V, 14, 15, C)
    # property: every 15th row is the newline token
    assert np.array_equal(out[:, :, 14, :], np.broadcast_to(newline, (V, 14, C)))
    for v in (0, 13, 31):
        f = feat_t[v:v + 1].float().numpy()
        want_ids, want = O.fused_visual_tokens(coords[v:v + 1], f, newline, kind, dim_t=dim_t.numpy())
        assert np.array_equal(ids[v].cpu().numpy(), want_ids[0])
        got = out[v].reshape(14 * 15, C)
        assert np.all(np.abs(got - want) <= fused_bound(f, kind))
        assert (got != want).mean() < 2e-3
    # property: PE-only and pool-only compose to the fused result (same rounding points)
    pooled = ops.visual_tokens(feat_t[:2].cuda(), pool=True)
    composed = ops.visual_tokens(pooled.view(2, 196, C), ids[:2], table, dev(newline, kind), pool=False)
    assert torch.equal(composed, out_t[:2 * 210])


def test_fused_writes_into_a_slice_of_inputs_embeds(ops):
    """K9: the fused kernel can write straight into the [S, C] sequence buffer at a row offset."""
    V, C, pre = 2, 64, 5
    feat = torch.randn((V, 729, C), device="cuda", dtype=torch.float16)
    seq = torch.zeros((pre + V * 210 + 3, C), device="cuda", dtype=torch.float16)
    nl = torch.randn(C, device="cuda", dtype=torch.float16)
    ops.visual_tokens(feat, newline=nl, out=seq[pre:pre + V * 210])
    ref = ops.visual_tokens(feat, newline=nl)
    assert torch.equal(seq[pre:pre + V * 210], ref)
    assert seq[:pre].abs().sum().item() == 0 and seq[pre + V * 210:].abs().sum().item() == 0


def test_embed_gather(ops):
    w = torch.randn((1000, 128), device="cuda", dtype=torch.bfloat16)
    ids = torch.randint(0, 1000, (77,), device="cuda")
    assert torch.equal(ops.embed_gather(w, ids), w[ids])


def test_bad_shapes_raise(ops):
    from v3d import V3DError
    with pytest.raises(V3DError):
        ops.visual_tokens(torch.zeros((1, 100, 64), device="cuda", dtype=torch.float16))
    with pytest.raises(V3DError):
        ops.coord_pool_voxel(torch.zeros((1, 384, 300, 3), device="cuda"))


def test_preprocess_rgb_bit_exact(golden):
    """a7 on the device vs the reference's own output (golden) and vs the oracle at the full 32 x 384 x 384 shape."""
    import torch
    from v3d import ops
    from oracle import v3d_oracle as O
    g = golden("imgproc")
    got = ops.preprocess_rgb(torch.from_numpy(g["frames"]).cuda())
    assert np.array_equal(got.cpu().numpy(), g["pixel_values"])
    rng = np.random.default_rng(3)
    frames = rng.integers(0, 256, size=(32, 384, 384, 3), dtype=np.uint8)
    want = O.image_preprocess(frames)
    got = ops.preprocess_rgb(torch.from_numpy(frames).cuda())
    assert np.array_equal(got.cpu().numpy(), want)
    for dt in (torch.bfloat16, torch.float16):            # 16-bit output = one rounding of the f32 value
        got16 = ops.preprocess_rgb(torch.from_numpy(frames[:2]).cuda(), dtype=dt)
        assert torch.equal(got16.cpu(), torch.from_numpy(want[:2]).to(dt))


# ---- a6 / a7 RGB half on the device (r03): Pillow's bicubic resize + centre crop, bit for bit


def test_resize_crop_rgb_equals_reference_loop_golden_and_live_pil(ops, golden):
    """v3d_resize_bicubic_u8 against (1) the fixture produced by the reference's own frame loop (video_utils.py:285-306) and (2) PIL
    itself run here on the same bytes, at the ScanNet frame size 640 x 480 -> 512 x 384 -> crop 384: every byte equal."""
    from PIL import Image
    from test_oracle_golden import _rgb_resize_inputs
    g = golden("rgb_resize")
    big = _rgb_resize_inputs(g)
    got = ops.resize_crop_rgb(torch.from_numpy(big).cuda(), (384, 512), crop=(0, 64, 384, 384)).cpu().numpy()
    assert np.array_equal(got[:, ::4, ::4], g["big_out_sample"])
    assert np.array_equal(got.astype(np.int64).sum((2, 3)), g["big_out_rowsum"])
    for i in range(big.shape[0]):
        want = np.asarray(Image.fromarray(big[i]).resize((512, 384)).crop((64, 0, 448, 384)))
        assert np.array_equal(got[i], want)
    small = g["small"]
    new_w = int(70 * (24 / 50))
    left = (new_w - 24) // 2
    got_s = ops.resize_crop_rgb(torch.from_numpy(small).cuda(), (24, new_w), crop=(0, left, 24, 24)).cpu().numpy()
    assert np.array_equal(got_s, g["small_out"])


@pytest.mark.parametrize("shape", [(48, 64, 38, 51), (30, 40, 45, 60), (97, 33, 20, 31), (16, 16, 16, 24), (480, 640, 384, 384), (968, 1296, 384, 512), (968, 1296, 384, 514)])
def test_resize_rgb_other_sizes_equal_live_pil(ops, shape):
    """up- and down-scaling, odd sizes, the ScanNet colour-stream size (1296 x 968): whole resized image vs PIL."""
    from PIL import Image
    H, W, OH, OW = shape
    rng = np.random.default_rng(H * 7 + OW)
    fr = rng.integers(0, 256, size=(2, H, W, 3), dtype=np.uint8)
    got = ops.resize_crop_rgb(torch.from_numpy(fr).cuda(), (OH, OW)).cpu().numpy()
    for i in range(2):
        assert np.array_equal(got[i], np.asarray(Image.fromarray(fr[i]).resize((OW, OH)))), shape


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_resize_crop_fused_with_image_processor_equals_the_two_steps(ops, dtype):
    """fused output (resize + crop + SigLipImageProcessor's rescale / normalise / CHW) == preprocess_rgb of the 8-bit crops, bit for bit,
    at the eval shape (32 frames of 640 x 480)."""
    g = torch.Generator().manual_seed(3)
    fr = torch.randint(0, 256, (32, 480, 640, 3), generator=g, dtype=torch.uint8).cuda()
    crops = ops.resize_crop_rgb(fr, (384, 512), crop=(0, 64, 384, 384))
    want = ops.preprocess_rgb(crops, dtype)
    got = ops.resize_crop_rgb(fr, (384, 512), crop=(0, 64, 384, 384), dtype=dtype)
    assert got.shape == (32, 3, 384, 384) and torch.equal(got, want)


def test_resize_rejects_bad_windows(ops):
    fr = torch.zeros((1, 48, 64, 3), dtype=torch.uint8).cuda()
    with pytest.raises(ops.V3DError):
        ops.resize_crop_rgb(fr, (38, 51), crop=(0, 40, 38, 24))          # window leaves the resized image
    with pytest.raises(ops.V3DError):
        ops.resize_crop_rgb(torch.zeros((1, 4800, 64, 3), dtype=torch.uint8).cuda(), (38, 51))      # 126x vertical reduction: more than 32 taps
