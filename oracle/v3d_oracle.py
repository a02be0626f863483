"""CPU oracle (numpy) for the position-aware video->LLM hot path of Video-3D-LLM.

TEST INFRASTRUCTURE ONLY.  This module is the *checker*: it may be imported by tests/,
by __graft_entry__.smoke() and by bench.py's cpu_baseline leg, and by nothing else.  The
product path (video-3d-llm_amd/) never imports it and has no CPU fallback.

Every function restates one reference function and cites the file:line it follows
(paths relative to the reference checkout).  Pinned against golden vectors generated
by running the reference itself (oracle/gen_golden.py -> tests/golden/), see
tests/test_oracle_golden.py.  Items that could not be pinned say "parity unpinned".

Conventions: "f16"/"bf16"/"f32" name the tensor dtype the reference function would see.
bf16 values travel as uint16 bit patterns (numpy has no bf16); helpers below convert.
"""
import math

import numpy as np

# ----------------------------------------------------------------------------- bf16 helpers


def bf16_bits_to_f32(b):
    b = np.asarray(b, dtype=np.uint16)
    return (b.astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16_bits(x):
    """Round-to-nearest-even f32 -> bf16 bit pattern (NaN kept NaN), as torch's c10::BFloat16."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    r = ((u + (np.uint32(0x7FFF) + ((u >> 16) & 1))) >> 16).astype(np.uint16)
    nan = np.isnan(x)
    if np.any(nan):
        r = np.where(nan, np.uint16(0x7FC0), r)
    return r


def round_to(x32, kind):
    """Round an f32 array to the storage dtype `kind` and return it widened back to f32."""
    if kind == "f32":
        return np.asarray(x32, dtype=np.float32)
    if kind == "f16":
        return np.asarray(x32, dtype=np.float32).astype(np.float16).astype(np.float32)
    if kind == "bf16":
        return bf16_bits_to_f32(f32_to_bf16_bits(x32))
    raise ValueError(kind)


# ----------------------------------------------------------------------------- K1  unproject


def unproject(intrinsics, poses, depths):
    """llava/video_utils.py:38-68.  intrinsics, poses [V,4,4] f32; depths [V,H,W] f32 (millimetres).

    f32 throughout: z = d/1000; x = (u-cx)*z/fx; y = (v-cy)*z/fy; world = pose @ [x,y,z,1]; / w.
    The reference's 4-term dot product runs inside a BLAS batched matmul whose summation order is
    not specified, so fp tolerance (not bit-exactness) applies to this function's output.
    """
    K = np.asarray(intrinsics, dtype=np.float32)
    P = np.asarray(poses, dtype=np.float32)
    d = np.asarray(depths, dtype=np.float32)
    V, H, W = d.shape
    u = np.arange(W, dtype=np.float32)[None, None, :]
    v = np.arange(H, dtype=np.float32)[None, :, None]
    fx, fy = K[:, 0, 0][:, None, None], K[:, 1, 1][:, None, None]
    cx, cy = K[:, 0, 2][:, None, None], K[:, 1, 2][:, None, None]
    z = d / np.float32(1000)
    x = (u - cx) * z / fx
    y = (v - cy) * z / fy
    cam = np.stack([x, y, z, np.ones_like(z)], -1)                 # [V,H,W,4]
    w = np.einsum("vij,vhwj->vhwi", P, cam).astype(np.float32)     # f32 accumulate
    return (w[..., :3] / w[..., 3:4]).astype(np.float32)


def nearest_resize_crop_index(src_h=480, src_w=640, crop=384):
    """llava/video_utils.py:296-308 ("center_crop"): cv2.resize(INTER_NEAREST) to
    (new_w=int(W*crop/H), new_h=crop) then crop columns [left, left+crop).

    PARITY UNPINNED: cv2 is not installable here, so this restates OpenCV's published
    INTER_NEAREST rule (src = min(floor(dst * src_size/dst_size), src_size-1)) and could not be
    checked against cv2 itself.  Returns (row_idx[crop], col_idx[crop]) into the source image.
    """
    new_h = crop
    new_w = int(src_w * (crop / src_h))
    left = (new_w - crop) // 2
    top = (new_h - crop) // 2
    fy = src_h / new_h
    fx = src_w / new_w
    rows = np.minimum(np.floor(np.arange(new_h) * fy).astype(np.int64), src_h - 1)[top:top + crop]
    cols = np.minimum(np.floor(np.arange(new_w) * fx).astype(np.int64), src_w - 1)[left:left + crop]
    return rows, cols


def resize_crop_coords(world, crop=384):
    """Apply nearest_resize_crop_index to [V,H,W,3] coords -> [V,crop,crop,3]."""
    r, c = nearest_resize_crop_index(world.shape[1], world.shape[2], crop)
    return world[:, r][:, :, c]


# ----------------------------------------------------------------------------- K3  patch mean


def average_coordinate_in_patch(world_coords, kind="f32", patch=27):
    """llava/model/llava_arch.py:213-223.  [V,384,384,3] -> [V,14,14,3]: drop the last 6 rows/cols,
    27x27 mean.  torch avg_pool2d (CPU, channels-last kernel; the CUDA kernel has the same order)
    accumulates in f32 SEQUENTIALLY over (ih, iw), divides by 729 in f32 and rounds to the dtype.
    `world_coords` is f32 holding values representable in `kind`."""
    x = np.asarray(world_coords, dtype=np.float32)
    V, H, W, D = x.shape
    n = (H - 6) // patch
    x = x[:, :n * patch, :n * patch].reshape(V, n, patch, n, patch, D)
    acc = np.zeros((V, n, n, D), dtype=np.float32)
    for ih in range(patch):
        for iw in range(patch):
            acc = acc + x[:, :, ih, :, iw, :]                      # one f32 rounding per add
    return round_to(acc / np.float32(patch * patch), kind)


# ----------------------------------------------------------------------------- K4  voxelise


def discrete_coords(xyz, kind="f32", min_xyz=(-15, -15, -5), max_xyz=(15, 15, 5), voxel_size=0.1):
    """llava/model/llava_arch.py:259-272.  clamp -> (x-min)/voxel -> round half-to-even; the result
    stays floating (integer-valued).  dtype rules of torch on CPU, pinned exhaustively for fp16:
      f32:  (x - min) in f32, / float32(voxel_size) true division, rint.
      f16/bf16: clamp exact; (x - min) rounded to the 16-bit type; then
                float(x) / float32(voxel_size) (the python scalar enters in opmath precision,
                not rounded to 16 bit) rounded to the 16-bit type; rint (exact in 16 bit)."""
    x = np.asarray(xyz, dtype=np.float32)
    lo = np.asarray(min_xyz, dtype=np.float32)
    hi = np.asarray(max_xyz, dtype=np.float32)
    # torch.maximum/minimum propagate NaN; np.maximum does too.
    x = np.minimum(np.maximum(x, lo), hi)
    t = round_to(x - lo, kind)
    q = round_to(t / np.float32(voxel_size), kind)
    return np.rint(q).astype(np.float32)


def voxel_ids(world_coords, kind="f32", **kw):
    """K3+K4 composed: [V,384,384,3] -> int32 ids [V,14,14,3]."""
    return discrete_coords(average_coordinate_in_patch(world_coords, kind), kind, **kw).astype(np.int32)


# ----------------------------------------------------------------------------- K5  3-D sinusoid


def sin3d_dim_t(num_feats, temperature=10000.0):
    """llava/model/position_encoding.py:24-25.  f32 pow as correctly-rounded libm would give it.
    NOTE: torch's vectorised CPU pow differs from this by 1 ulp in a few entries (16 of 1194 at
    width 3584), so parity tests pass the golden `dim_t` explicitly; this default is for benches."""
    i = np.arange(num_feats, dtype=np.float32)
    e = (np.float32(2) * np.floor(i / np.float32(2))) / np.float32(num_feats)
    return np.power(np.float64(temperature), e.astype(np.float64)).astype(np.float32)


def sin3d_pe(xyz, embedding_size, out_kind="f32", dim_t=None, temperature=10000.0):
    """llava/model/position_encoding.py:17-49 (n_points == 1).  xyz [B,N,3] (any float dtype, passed
    as f32 values) -> [B,N,embedding_size].  f32 math; per axis num_feats = E//3 features,
    p/dim_t, even index -> sin, odd -> cos (for odd num_feats the padded column is dropped, :30-36);
    x|y|z concatenated, remaining columns zero, result cast to the input dtype (:46-47)."""
    x = np.asarray(xyz, dtype=np.float32)
    nf = embedding_size // 3
    if dim_t is None:
        dim_t = sin3d_dim_t(nf, temperature)
    dim_t = np.asarray(dim_t, dtype=np.float32)
    B, N, _ = x.shape
    out = np.zeros((B, N, embedding_size), dtype=np.float32)
    even = (np.arange(nf) % 2) == 0
    for a in range(3):
        arg = (x[:, :, a][..., None] / dim_t).astype(np.float32)
        pe = np.where(even, np.sin(arg, dtype=np.float32), np.cos(arg, dtype=np.float32))
        out[:, :, a * nf:(a + 1) * nf] = pe
    return round_to(out, out_kind)


# ----------------------------------------------------------------------------- K7  bilinear 2x pool


def image_preprocess(frames_u8, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5), rescale=1 / 255):
    """a7, SigLipImageProcessor.preprocess (siglip_encoder.py:47-67) for frames already at the tower size (the bicubic
    resize is then the identity): rescale in f64 -> f32 (transformers.image_transforms.rescale), (x - mean) / std in
    f32 (normalize), HWC -> CHW.  frames_u8 [F,H,W,3] -> [F,3,H,W] f32."""
    x = (frames_u8.astype(np.float64) * rescale).astype(np.float32)
    x = (x - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2)).astype(np.float32)


def _pil_bicubic_filter(x):
    """Pillow libImaging/Resample.c bicubic_filter (a = -0.5), support 2.0."""
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_resample_coeffs(in_size, out_size, support=2.0):
    """Pillow 12.2.0 libImaging/Resample.c precompute_coeffs + normalize_coeffs_8bpc for box (0, in_size): per output index the
    first source index, the tap count and the 22-bit fixed-point taps.  (Third-party arithmetic behind the reference's
    `frame.resize(...)`, video_utils.py:303; Pillow is not under /root/reference - it is importable here and on the GPU box, so this
    restatement is pinned against PIL itself: tests/test_oracle_golden.py::test_pil_resize_*.)"""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    sup = support * filterscale
    ksize = int(math.ceil(sup)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = max(int(center - sup + 0.5), 0)
        xmax = min(int(center + sup + 0.5), in_size) - xmin
        w = [_pil_bicubic_filter((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        bounds[xx] = (xmin, xmax)
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << 22)) if v < 0 else int(0.5 + v * (1 << 22))
    return bounds, kk


def pil_resize_bicubic(frames_u8, out_hw):
    """`Image.resize((OW, OH))` of 8-bit RGB frames with Pillow's default filter (BICUBIC), as VideoProcessor.preprocess calls it
    (video_utils.py:303): ImagingResampleHorizontal_8bpc, then ImagingResampleVertical_8bpc on the 8-bit intermediate;
    accumulators start at 1 << 21, clip8 = clamp(ss >> 22, 0, 255).  frames_u8 [F,H,W,3] -> [F,OH,OW,3] uint8."""
    F_, H, W, _ = frames_u8.shape
    OH, OW = out_hw
    bh, kh = pil_resample_coeffs(W, OW)
    bv, kv = pil_resample_coeffs(H, OH)
    src = frames_u8.astype(np.int64)
    tmp = np.zeros((F_, H, OW, 3), np.int64)
    for xx in range(OW):
        x0, n = bh[xx]
        acc = (src[:, :, x0:x0 + n, :] * kh[xx, :n].astype(np.int64)[None, None, :, None]).sum(2) + (1 << 21)
        tmp[:, :, xx] = np.clip(acc >> 22, 0, 255)
    out = np.zeros((F_, OH, OW, 3), np.int64)
    for yy in range(OH):
        y0, n = bv[yy]
        acc = (tmp[:, y0:y0 + n] * kv[yy, :n].astype(np.int64)[None, :, None, None]).sum(1) + (1 << 21)
        out[:, yy] = np.clip(acc >> 22, 0, 255)
    return out.astype(np.uint8)


def resize_crop_rgb(frames_u8, crop=384):
    """The RGB half of VideoProcessor.preprocess, strategy "center_crop" (video_utils.py:297-306): new_height = crop,
    new_width = int(W * (crop / H)), PIL bicubic resize, centre crop of the columns."""
    H, W = frames_u8.shape[1:3]
    new_w = int(W * (crop / H))
    left = (new_w - crop) // 2
    return pil_resize_bicubic(frames_u8, (crop, new_w))[:, :, left:left + crop]


def bilinear_taps(n_in=27, n_out=14):
    """Source taps of F.interpolate(mode='bilinear', align_corners=False) (ATen
    area_pixel_compute_source_index): src = scale*(o+0.5)-0.5 clamped at 0, scale = n_in/n_out in f32.
    Pinned against the reference run: ATen's builds contract that expression into ONE fused
    multiply-add (single rounding) - emulated here in f64, which holds the f32 product exactly."""
    scale = np.float32(n_in) / np.float32(n_out)
    o = np.arange(n_out, dtype=np.float32)
    src = (np.float64(scale) * (o + np.float32(0.5)).astype(np.float64) - 0.5).astype(np.float32)
    src = np.maximum(src, np.float32(0)).astype(np.float32)
    i0 = src.astype(np.int64)
    i1 = np.minimum(i0 + 1, n_in - 1)
    l1 = (src - i0.astype(np.float32)).astype(np.float32)
    l0 = (np.float32(1) - l1).astype(np.float32)
    return i0, i1, l0, l1


def get_2dpool_bilinear(feat, kind="f32", side=27, stride=2):
    """llava/model/llava_arch.py:191-210 with mm_spatial_pool_mode == 'bilinear'.
    feat [V, side*side, C] (f32 values of dtype `kind`) -> [V, ceil(side/stride)^2, C].
    f32 arithmetic  h0*(w0*x00 + w1*x01) + h1*(w0*x10 + w1*x11), rounded once to `kind`."""
    x = np.asarray(feat, dtype=np.float32)
    V, T, C = x.shape
    n_out = math.ceil(side / stride)
    x = x.reshape(V, side, side, C)
    i0, i1, l0, l1 = bilinear_taps(side, n_out)
    h0 = l0[None, :, None, None]
    h1 = l1[None, :, None, None]
    w0 = l0[None, None, :, None]
    w1 = l1[None, None, :, None]
    top = w0 * x[:, i0][:, :, i0] + w1 * x[:, i0][:, :, i1]
    bot = w0 * x[:, i1][:, :, i0] + w1 * x[:, i1][:, :, i1]
    out = (h0 * top + h1 * bot).astype(np.float32)
    return round_to(out.reshape(V, n_out * n_out, C), kind)


# ----------------------------------------------------------------------------- K6 / K8


def add_pe(feat, pe, kind):
    """llava/model/llava_arch.py:515: feat + PE in the model dtype (f32 add, one rounding)."""
    return round_to(np.asarray(feat, np.float32) + np.asarray(pe, np.float32), kind)


def add_token_per_grid(feat, newline):
    """llava/model/llava_arch.py:307-328 (no faster-video): [V, h*h, C] -> [V*h*(h+1), C]; after each
    row of h tokens one `image_newline` token."""
    x = np.asarray(feat)
    V, T, C = x.shape
    h = int(math.isqrt(T))
    x = x.reshape(V, h, h, C)
    nl = np.broadcast_to(np.asarray(newline, dtype=x.dtype)[None, None, None, :], (V, h, 1, C))
    return np.concatenate([x, nl], axis=2).reshape(V * h * (h + 1), C)


def fused_visual_tokens(world_coords, feat, newline, kind, dim_t=None):
    """The composition the reference performs for `avg-discrete-sin3d` + bilinear pool + grid newline
    (llava_arch.py:395-420 -> :469 -> :506-517 -> :536).  Returns (ids int32 [V,14,14,3],
    tokens [V*14*15, C] as f32 values of dtype `kind`)."""
    C = feat.shape[-1]
    avg = average_coordinate_in_patch(world_coords, kind)
    vox = discrete_coords(avg, kind)
    pooled = get_2dpool_bilinear(feat, kind)
    pe = sin3d_pe(vox.reshape(vox.shape[0], -1, 3), C, kind, dim_t=dim_t)
    fused = add_pe(pooled, pe, kind)
    return vox.astype(np.int32), add_token_per_grid(fused, round_to(newline, kind))


# ----------------------------------------------------------------------------- a1/a2 frame sampling


def uniform_frame_indices(total_frames, num_frames):
    """llava/video_utils.py:187  np.linspace(0, total-1, n, dtype=int) (truncation toward zero)."""
    return np.linspace(0, total_frames - 1, num_frames, dtype=int)


def sample_frame_files_mc(entry, strategy, frames_upbound=32):
    """llava/video_utils.py:131-159.  entry = {'frame_files', 'voxel_nums', 'num_all_voxels'}."""
    files = list(entry["frame_files"][:frames_upbound])
    nums = list(entry["voxel_nums"][:frames_upbound])
    ratio = 0.95 if "ratio95" in strategy else (0.9 if "ratio90" in strategy else 1.0)
    if ratio != 1.0:
        out, cc = [], 0
        for f, n in zip(files, nums):
            out.append(f)
            cc += n
            if cc >= entry["num_all_voxels"] * ratio:
                break
        files = out
    files.sort(key=lambda p: int(p.split("/")[-1].split(".")[0]))
    return files


# ----------------------------------------------------------------------------- a3 max coverage


def discrete_point(xyz, voxel_size=0.1, min_xyz=None, max_xyz=None):
    """llava/video_utils.py:348-358.  torch.tensor(list of python floats) is f32; clamp, shift by min,
    / voxel_size (f32), round half-to-even, int."""
    x = np.asarray(xyz, dtype=np.float32)
    if min_xyz is not None:
        x = np.maximum(x, np.asarray(min_xyz, dtype=np.float32))
    if max_xyz is not None:
        x = np.minimum(x, np.asarray(max_xyz, dtype=np.float32))
    if min_xyz is not None:
        x = x - np.asarray(min_xyz, dtype=np.float32)
    return np.rint(x / np.float32(voxel_size)).astype(np.int32)


def voxel_keys(world, voxel_size=0.1):
    """scripts/3d/preprocessing/max_coverage_sampling.py:44-45: round(xyz / voxel) as int32 triples."""
    return np.rint(np.asarray(world, dtype=np.float32) / np.float32(voxel_size)).astype(np.int32)


def greedy_max_coverage(world, scene_voxels, voxel_size=0.1, max_frames=32):
    """scripts/3d/preprocessing/max_coverage_sampling.py:44-94 with the declared tie rule
    "lowest frame position wins" in place of the unseeded random.choice (:84).

    world [n,H,W,3] f32; scene_voxels [m,3] int32 (the scene point cloud's voxel set, pc_voxel).
    Per step the gain of a frame is |frame_voxels & pc_voxel| - |used & frame_voxels & pc_voxel|,
    but `used` accumulates ALL of the chosen frame's voxels (:86), in-scene or not.
    Returns (selected indices, gains, num_all_voxels, num_select_voxels)."""
    keys = voxel_keys(world, voxel_size)
    n = keys.shape[0]
    frame_sets = [set(map(tuple, np.unique(keys[i].reshape(-1, 3), axis=0).tolist())) for i in range(n)]
    pc = set(map(tuple, np.asarray(scene_voxels).tolist()))
    all_voxel = set().union(*frame_sets)
    remaining = list(range(n))
    used = set()
    sel, gains = [], []
    for _ in range(n):
        best, best_i = -1, None
        for i in remaining:
            cur = frame_sets[i] & pc
            gain = len(cur) - len(used & cur)
            if gain > best:
                best, best_i = gain, i
        used |= frame_sets[best_i]
        sel.append(best_i)
        gains.append(best)
        remaining.remove(best_i)
        if len(sel) >= max_frames:
            break
    return (np.array(sel, np.int32), np.array(gains, np.int64), len(all_voxel & pc), len(used & pc))


# ----------------------------------------------------------------------------- a25


def convert_pc_to_box(obj_pc):
    """llava/utils_3d.py:3-13."""
    p = np.asarray(obj_pc)
    lo = p[:, :3].min(0)
    hi = p[:, :3].max(0)
    return [((lo[i] + hi[i]) / 2) for i in range(3)], [(hi[i] - lo[i]) for i in range(3)]
