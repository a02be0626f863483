"""Pins the CPU oracle (oracle/v3d_oracle.py) against golden vectors produced by running the
reference itself (oracle/gen_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import v3d_oracle as O

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def ulp16_diff(a_f32, b_f32, kind):
    """max distance, in units of the 16-bit type's ulp, between two arrays of 16-bit-representable values."""
    if kind == "f16":
        ia = np.asarray(a_f32, np.float32).astype(np.float16).view(np.int16).astype(np.int32)
        ib = np.asarray(b_f32, np.float32).astype(np.float16).view(np.int16).astype(np.int32)
    else:
        ia = O.f32_to_bf16_bits(a_f32).view(np.int16).astype(np.int32)
        ib = O.f32_to_bf16_bits(b_f32).view(np.int16).astype(np.int32)
    # sign-magnitude -> monotone integer
    ia = np.where(ia < 0, -(ia & 0x7FFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFF), ib)
    return np.abs(ia - ib)


def test_unproject(golden):
    g = golden("unproject")
    out = O.unproject(g["intrinsics"], g["poses"], g["depth"].astype(np.float32))
    # fp tolerance: the reference's 4-term dot runs in BLAS with unspecified order / FMA use
    np.testing.assert_allclose(out, g["world"], rtol=2e-6, atol=2e-6)


def test_coord_pool_bitexact(golden):
    g = golden("coord_pool")
    x16 = g["coords_f16"].astype(np.float32)
    x32 = x16 * np.float32(1.001) + np.float32(0.0003)
    a32 = O.average_coordinate_in_patch(x32, "f32")
    a16 = O.average_coordinate_in_patch(x16, "f16")
    assert np.array_equal(a32, g["avg_f32"])                      # sequential f32 sum: bit-exact
    assert np.array_equal(a16, g["avg_f16"].astype(np.float32))
    assert np.array_equal(O.discrete_coords(a32, "f32"), g["vox_f32"])
    assert np.array_equal(O.discrete_coords(a16, "f16"), g["vox_f16"].astype(np.float32))
    # clamp cases present
    assert g["vox_f32"].max() == 300 and g["vox_f32"].min() == 0


def test_discrete_coords_exhaustive_f16(golden):
    g = golden("discrete_coords")
    bits = np.arange(65536, dtype=np.uint16)
    h = bits.view(np.float16)[g["finite_mask"]].astype(np.float32)
    xyz = np.stack([h, h, h], -1)
    got = O.discrete_coords(xyz, "f16")
    assert np.array_equal(got, g["vox_f16"].astype(np.float32))


def test_discrete_coords_f32_ties(golden):
    g = golden("discrete_coords")
    got = O.discrete_coords(g["xyz_f32"], "f32")
    assert np.array_equal(got, g["vox_f32"])


def test_discrete_point(golden):
    g = golden("discrete_point")
    assert np.array_equal(O.discrete_point(g["pts"]), g["ids_norange"])
    assert np.array_equal(O.discrete_point(g["pts"], 0.1, [-15, -15, -5], [15, 15, 5]), g["ids_range"])


def test_sin3d_table(golden):
    g = golden("sin3d_table_3584")
    ids = np.arange(301, dtype=np.float32)
    xyz = np.stack([ids, ids[::-1], np.minimum(ids, 100)], -1)[None]
    pe = O.sin3d_pe(xyz, 3584, "f32", dim_t=g["dim_t"])[0]
    tab = g["table"]
    # libm sin/cos vs torch's vectorised sin/cos: both within ~1 ulp of the true value
    np.testing.assert_allclose(pe[:, :1194], tab, rtol=0, atol=2.5e-7)
    np.testing.assert_allclose(pe[:, 1194:2388], tab[::-1], rtol=0, atol=2.5e-7)
    np.testing.assert_allclose(pe[:101, 2388:3582], tab[:101], rtol=0, atol=2.5e-7)
    assert np.all(pe[:, 3582:] == 0)
    # default dim_t: correctly rounded pow; differs from torch's by <= 1 ulp in a few entries
    d = O.sin3d_dim_t(1194)
    assert np.max(np.abs(d.view(np.int32) - g["dim_t"].view(np.int32))) <= 1


def test_sin3d_tokens_16bit(golden):
    g = golden("sin3d_tokens_3584")
    t = golden("sin3d_table_3584")
    ids = g["ids"].astype(np.float32)
    p16 = O.sin3d_pe(ids, 3584, "f16", dim_t=t["dim_t"])
    d = ulp16_diff(p16, g["pe_f16"].astype(np.float32), "f16")
    assert d.max() <= 1 and (d != 0).mean() < 2e-3
    # bf16 cannot hold every integer above 256: the reference sees the rounded ids
    pb = O.sin3d_pe(O.round_to(ids, "bf16"), 3584, "bf16", dim_t=t["dim_t"])
    d = ulp16_diff(pb, O.bf16_bits_to_f32(g["pe_bf16_bits"]), "bf16")
    assert d.max() <= 1 and (d != 0).mean() < 2e-3


def test_sin3d_small_odd_and_continuous(golden):
    g = golden("sin3d_small")
    np.testing.assert_allclose(O.sin3d_pe(g["xyz"], 16, "f32", dim_t=g["dim_t5"]), g["pe16"], rtol=0, atol=5e-7)
    np.testing.assert_allclose(O.sin3d_pe(g["xyz"], 96, "f32", dim_t=g["dim_t32"]), g["pe96"], rtol=0, atol=5e-7)


def test_pool2d_bilinear(golden):
    g = golden("pool2d_bilinear")
    o32 = O.get_2dpool_bilinear(g["feat"], "f32")
    np.testing.assert_allclose(o32, g["out_f32"], rtol=1e-6, atol=1e-6)
    x16 = g["feat"].astype(np.float16).astype(np.float32)
    d = ulp16_diff(O.get_2dpool_bilinear(x16, "f16"), g["out_f16"].astype(np.float32), "f16")
    assert d.max() <= 1 and (d != 0).mean() < 1e-3
    xb = O.round_to(g["feat"], "bf16")
    d = ulp16_diff(O.get_2dpool_bilinear(xb, "bf16"), O.bf16_bits_to_f32(g["out_bf16_bits"]), "bf16")
    assert d.max() <= 1 and (d != 0).mean() < 1e-3


def test_add_token_per_grid(golden):
    g = golden("add_token_per_grid")
    assert np.array_equal(O.add_token_per_grid(g["feat"], g["newline"]), g["out"])


@pytest.mark.parametrize("kind", ["f32", "f16", "bf16"])
def test_fused_small(golden, kind):
    g = golden("fused_small")
    coords = O.round_to(g["coords_f16"].astype(np.float32), kind)
    feat = O.round_to(g["feat"], kind)
    ids, seq = O.fused_visual_tokens(coords, feat, g["newline"], kind, dim_t=g["dim_t"])
    assert np.array_equal(ids.astype(np.float32), g["vox_" + kind])          # voxel ids: bit-exact
    ref = g["seq_" + kind]
    if kind == "f32":
        np.testing.assert_allclose(seq, ref, rtol=2e-6, atol=2e-6)
    else:
        ref32 = ref.astype(np.float32) if kind == "f16" else O.bf16_bits_to_f32(ref)
        # each operand (pooled feature, PE) may be 1 ulp16 off; after cancellation that is many ulps
        # of the SUM, so the bound is in ulps of the operands:  eps16 * (|pooled| + |pe|)
        eps = 2.0 ** -10 if kind == "f16" else 2.0 ** -7
        pooled = O.get_2dpool_bilinear(feat, kind)
        bound = O.add_token_per_grid(eps * (np.abs(pooled) + 1.0), np.zeros(feat.shape[-1], np.float32))
        assert np.all(np.abs(seq - ref32) <= bound)
        assert (seq != ref32).mean() < 2e-3


def test_image_preprocess(golden):
    """a7 against SigLipImageProcessor.preprocess itself (96 x 96 windows of two 384 x 384 frames): bit-exact."""
    g = golden("imgproc")
    got = O.image_preprocess(g["frames"])
    assert got.dtype == np.float32 and np.array_equal(got, g["pixel_values"])


def test_frame_sampling():
    with open(os.path.join(GOLDEN, "frame_sampling.json")) as f:
        g = json.load(f)
    for key, want in g["uniform"].items():
        n = int(key.split("_")[0][1:])
        F = 10 if key.endswith("default") else int(key.split("_F")[1])
        assert O.uniform_frame_indices(n, F).tolist() == want, key
    for key, want in g["mc"].items():
        strat, F = key.rsplit("_F", 1)
        assert O.sample_frame_files_mc(g["mc_entry"], strat, int(F)) == want, key


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_greedy_cover(golden, case):
    g = golden("greedy_cover")
    sel, gains, n_all, n_sel = O.greedy_max_coverage(g[case + "_world"], g[case + "_pc"])
    assert sel.tolist() == g[case + "_select"].tolist()
    assert gains.tolist() == g[case + "_voxel_nums"].tolist()
    assert n_all == int(g[case + "_num_all"]) and n_sel == int(g[case + "_num_sel"])


def test_convert_pc_to_box(golden):
    g = golden("convert_pc_to_box")
    c, s = O.convert_pc_to_box(g["pc"])
    assert np.array_equal(np.array(c), g["center"]) and np.array_equal(np.array(s), g["size"])


# ---- a6 / a7 RGB half: Pillow's bicubic resize (the arithmetic behind video_utils.py:303), restated in oracle/v3d_oracle.py


def _rgb_resize_inputs(g):
    rng = np.random.default_rng(int(g["noise_seed"]))
    noise = rng.integers(0, 256, size=(1, 480, 640, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:480, 0:640]
    smooth = np.stack([(127 + 120 * np.sin(xx / 37.0 + c) * np.cos(yy / 23.0 - c)).astype(np.uint8) for c in range(3)], -1)[None]
    assert np.array_equal(smooth[:, ::8, ::8], g["smooth"])          # the regenerated input is the generator's
    edges = np.zeros((1, 480, 640, 3), np.uint8)
    edges[0, ::7] = 255
    edges[0, :, ::5, 1] = 255
    rng.integers(0, 256, size=(2, 50, 70, 3), dtype=np.uint8)         # (the generator drew `small` next; it is stored explicitly)
    return np.concatenate([noise, smooth, edges])


def test_pil_resize_oracle_matches_reference_loop_golden(golden):
    g = golden("rgb_resize")
    big = _rgb_resize_inputs(g)
    out = O.resize_crop_rgb(big, 384)
    assert out.shape == (3, 384, 384, 3)
    assert np.array_equal(out[:, ::4, ::4], g["big_out_sample"])
    assert np.array_equal(out.astype(np.int64).sum((2, 3)), g["big_out_rowsum"])
    assert np.array_equal(O.resize_crop_rgb(g["small"], 24), g["small_out"])


def test_pil_resize_oracle_matches_live_pil_on_other_sizes():
    from PIL import Image
    rng = np.random.default_rng(5)
    for (H, W, OH, OW) in ((48, 64, 38, 51), (30, 40, 45, 60), (97, 33, 20, 31), (16, 16, 16, 24)):
        fr = rng.integers(0, 256, size=(1, H, W, 3), dtype=np.uint8)
        want = np.asarray(Image.fromarray(fr[0]).resize((OW, OH)))
        assert np.array_equal(O.pil_resize_bicubic(fr, (OH, OW))[0], want), (H, W, OH, OW)


def test_product_resample_tables_equal_the_oracles():
    from v3d import ops
    for (n_in, n_out) in ((640, 512), (480, 384), (70, 33), (50, 24), (33, 70), (16, 16)):
        b, k, ks = ops.pil_resample_tables(n_in, n_out)
        ob, ok = O.pil_resample_coeffs(n_in, n_out)
        assert ks == ok.shape[1] and np.array_equal(b.numpy(), ob) and np.array_equal(k.numpy(), ok)
