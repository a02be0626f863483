"""Tensor-level wrappers over the C ABI.  Every op takes CUDA(=HIP) tensors, launches on torch's
current stream and returns CUDA tensors.  PyTorch only provides the memory and the stream."""
import ctypes
import math

import numpy as np
import torch

from ._native import V3DError, check, lib

F32, F16, BF16 = 0, 1, 2
_DT = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}

VT_POOL, VT_PE, VT_NEWLINE = 1, 2, 4


def _code(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise V3DError(f"unsupported dtype {t.dtype}") from None


def _dev(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise V3DError(f"{name} must be a CUDA/HIP tensor resident in HBM (got {type(t).__name__}"
                       f"{'' if not isinstance(t, torch.Tensor) else ' on ' + str(t.device)}); there is no CPU path")
    return t.contiguous()


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _f3(v):
    return (ctypes.c_float * 3)(*[float(x) for x in v])


def _rows_fit(out, rows, cols, dtype, what):
    """`out` must hold `rows` x `cols` elements of `dtype` behind its data pointer: the kernels trust the sizes they are
    given, so an empty or short slice (e.g. a row view taken past the end of its buffer) is refused here."""
    if not isinstance(out, torch.Tensor) or not out.is_cuda:
        raise V3DError(f"{what}: out must be a CUDA/HIP tensor")
    if out.dtype != dtype:
        raise V3DError(f"{what}: out is {out.dtype}, expected {dtype}")
    if out.dim() == 1:
        ok = rows <= 1 and out.shape[0] >= cols
    else:
        ok = out.dim() == 2 and out.shape[0] >= rows and out.shape[1] >= cols and out.stride(1) == 1
    if not ok:
        raise V3DError(f"{what}: out {tuple(out.shape)} cannot hold [{rows},{cols}]")


# ------------------------------------------------------------------------------ geometry


def unproject(intrinsics, poses, depths):
    """K1.  intrinsics, poses [V,4,4]; depths [V,H,W] (millimetres) -> [V,H,W,3] f32."""
    K = _dev(intrinsics, "intrinsics").float()
    P = _dev(poses, "poses").float()
    d = _dev(depths, "depths").float()
    V, H, W = d.shape
    out = torch.empty((V, H, W, 3), dtype=torch.float32, device=d.device)
    check(lib().v3d_unproject_f32(_p(d), _p(K), _p(P), _p(out), V, H, W, _stream()), "v3d_unproject_f32")
    return out


def unproject_sampled(depth_u16, intrinsics, poses, crop=384, dtype=torch.float32):
    """K1+K2.  depth [V,H,W] uint16/int16 raw PNG values -> [V,crop,crop,3] in `dtype`."""
    d = _dev(depth_u16, "depth")
    if d.dtype not in (torch.uint16, torch.int16):
        raise V3DError("depth must be a 16-bit integer tensor (raw PNG millimetres)")
    K = _dev(intrinsics, "intrinsics").float()
    P = _dev(poses, "poses").float()
    V, H, W = d.shape
    out = torch.empty((V, crop, crop, 3), dtype=dtype, device=d.device)
    check(lib().v3d_unproject_sampled_u16(_p(d), _p(K), _p(P), _p(out), _DT[dtype], V, H, W, crop, _stream()),
          "v3d_unproject_sampled_u16")
    return out


def unproject_resized(depth_u16, intrinsics, poses, size=384, dtype=torch.float32):
    """VideoProcessor.preprocess(strategy="resize") (video_utils.py:293-296): [V,H,W] raw depth -> [V,size,size,3], nearest resize of
    both axes, no crop."""
    d = _dev(depth_u16, "depth")
    if d.dtype not in (torch.uint16, torch.int16):
        raise V3DError("depth must be a 16-bit integer tensor (raw PNG millimetres)")
    K = _dev(intrinsics, "intrinsics").float()
    P = _dev(poses, "poses").float()
    V, H, W = d.shape
    out = torch.empty((V, size, size, 3), dtype=dtype, device=d.device)
    check(lib().v3d_unproject_resized_u16(_p(d), _p(K), _p(P), _p(out), _DT[dtype], V, H, W, size, _stream()), "v3d_unproject_resized_u16")
    return out


def clamp_xyz(xyz, lo, hi):
    """xyz [..., 3] (contiguous, device) clamped in place to [lo, hi] per axis (calculate_world_coords(do_normalize=True))."""
    x = _dev(xyz, "xyz")
    if x.shape[-1] != 3 or not x.is_contiguous():
        raise V3DError("clamp_xyz wants a contiguous [..., 3] tensor")
    lo_c, hi_c = (ctypes.c_float * 3)(*[float(v) for v in lo]), (ctypes.c_float * 3)(*[float(v) for v in hi])
    check(lib().v3d_clamp_xyz(_p(x), x.numel() // 3, lo_c, hi_c, _code(x), _stream()), "v3d_clamp_xyz")
    return x


def unproject_bounds(depth_u16, intrinsics, poses):
    """[x_min, x_max, y_min, y_max, z_min, z_max] of the full-resolution back-projection (video_utils.py:268-273) -> f32 [6] (device)."""
    d = _dev(depth_u16, "depth")
    if d.dtype not in (torch.uint16, torch.int16):
        raise V3DError("depth must be a 16-bit integer tensor (raw PNG millimetres)")
    K = _dev(intrinsics, "intrinsics").float()
    P = _dev(poses, "poses").float()
    V, H, W = d.shape
    ws = torch.empty(lib().v3d_unproject_bounds_workspace_bytes(V) // 4, dtype=torch.float32, device=d.device)
    out = torch.empty(6, dtype=torch.float32, device=d.device)
    check(lib().v3d_unproject_bounds_u16(_p(d), _p(K), _p(P), V, H, W, _p(out), _p(ws), ws.numel() * 4, _stream()), "v3d_unproject_bounds_u16")
    return out


def coord_pool_voxel(coords, patch=27, min_xyz=(-15, -15, -5), max_xyz=(15, 15, 5), voxel_size=0.1,
                     want_avg=True, want_vox=True, want_ids=True):
    """K3+K4.  coords [V,S,S,3] -> (avg [V,n,n,3], vox [V,n,n,3] same dtype, ids int32)."""
    x = _dev(coords, "coords")
    V, S, S2, D = x.shape
    if S != S2 or D != 3:
        raise V3DError(f"coords must be [V,S,S,3], got {tuple(x.shape)}")
    n = (S - 6) // patch
    avg = torch.empty((V, n, n, 3), dtype=x.dtype, device=x.device) if want_avg else None
    vox = torch.empty((V, n, n, 3), dtype=x.dtype, device=x.device) if want_vox else None
    ids = torch.empty((V, n, n, 3), dtype=torch.int32, device=x.device) if want_ids else None
    check(lib().v3d_coord_pool_voxel(_p(x), _code(x), V, S, patch, _f3(min_xyz), _f3(max_xyz), float(np.float32(voxel_size)),
                                     _p(avg), _p(vox), _p(ids), _stream()), "v3d_coord_pool_voxel")
    return avg, vox, ids


def discrete_coords(xyz, min_xyz=(-15, -15, -5), max_xyz=(15, 15, 5), voxel_size=0.1, want_ids=False):
    """K4 on arbitrary [...,3] points."""
    x = _dev(xyz, "xyz")
    if x.shape[-1] != 3:
        raise V3DError("xyz must end in a dimension of 3")
    N = x.numel() // 3
    vox = torch.empty_like(x)
    ids = torch.empty(x.shape, dtype=torch.int32, device=x.device) if want_ids else None
    if N == 0:          # empty `box_input` (merge_video_dict yields torch.Tensor([]) when there are no boxes)
        return (vox, ids) if want_ids else vox
    check(lib().v3d_discrete_coords(_p(x), _code(x), N, _f3(min_xyz), _f3(max_xyz), float(np.float32(voxel_size)),
                                    _p(vox), _p(ids), _stream()), "v3d_discrete_coords")
    return (vox, ids) if want_ids else vox


# ------------------------------------------------------------------------------ 3-D sinusoid


def reference_dim_t(num_feats, temperature=10000):
    """position_encoding.py:24-25 evaluated exactly as the reference's CPU path does (torch pow)."""
    d = torch.arange(num_feats, dtype=torch.float32)
    return temperature ** (2 * (d // 2) / num_feats)


class Sin3DTable:
    """PE values for every integer voxel id, in the shifted layout v3d_visual_tokens reads."""

    def __init__(self, embedding_size, n_ids, dtype, device, dim_t=None, temperature=10000, keep_f32=False):
        self.embedding_size = embedding_size
        self.n_ids = n_ids
        self.dtype = dtype
        nf = embedding_size // 3
        if dim_t is None:
            dim_t = reference_dim_t(nf, temperature)
        self.dim_t = dim_t.to(device=device, dtype=torch.float32).contiguous()
        row = lib().v3d_sin3d_table_row_elems(embedding_size, _DT[dtype])
        self.row_elems = row
        self.table = torch.empty((3, n_ids, row), dtype=dtype, device=device)
        self.table_f32 = torch.empty((n_ids, nf), dtype=torch.float32, device=device) if keep_f32 else None
        check(lib().v3d_sin3d_table_build(_p(self.dim_t), embedding_size, n_ids, _DT[dtype], _p(self.table),
                                          _p(self.table_f32), _stream()), "v3d_sin3d_table_build")


def sin3d_pe(xyz, embedding_size, dim_t=None, temperature=10000):
    """PositionEmbeddingSine3D.forward (n_points=1): xyz [B,N,3] -> [B,N,E] in xyz.dtype."""
    x = _dev(xyz, "xyz")
    if dim_t is None:
        dim_t = reference_dim_t(embedding_size // 3, temperature)
    dim_t = dim_t.to(device=x.device, dtype=torch.float32).contiguous()
    N = x.numel() // 3
    out = torch.empty(x.shape[:-1] + (embedding_size,), dtype=x.dtype, device=x.device)
    check(lib().v3d_sin3d_pe(_p(x), _code(x), N, _p(dim_t), embedding_size, _p(out), _stream()), "v3d_sin3d_pe")
    return out


# ------------------------------------------------------------------------------ fusion


def visual_tokens(feat, ids=None, table=None, newline=None, side=27, n=14, pool=True, out=None):
    """K7 (+K5+K6) (+K8).  feat [V, side*side | n*n, C]; ids [V,n,n,3] int32; table Sin3DTable;
    newline [C].  Returns / fills `out` [rows, C] (rows = V*n*(n+1) with newline else V*n*n);
    `out` may be a row-slice view of a larger [S, C] buffer (the LLM's inputs_embeds)."""
    f = _dev(feat, "feat")
    V, T, C = f.shape
    flags = (VT_POOL if pool else 0) | (VT_PE if table is not None else 0) | (VT_NEWLINE if newline is not None else 0)
    expect = side * side if pool else n * n
    if T != expect:
        raise V3DError(f"feat has {T} tokens per frame, expected {expect}")
    rows = V * n * ((n + 1) if newline is not None else n)
    if out is None:
        out = torch.empty((rows, C), dtype=f.dtype, device=f.device)
    if out.dtype != f.dtype or out.shape[0] != rows or out.shape[1] != C or out.stride(1) != 1:
        raise V3DError(f"out must be [{rows},{C}] {f.dtype} with unit column stride")
    idt = tbl = nl = None
    n_ids = 0
    if table is not None:
        if ids is None:
            raise V3DError("PE requested without voxel ids")
        idt = _dev(ids, "ids")
        if idt.dtype != torch.int32 or idt.numel() != V * n * n * 3:
            raise V3DError("ids must be int32 [V,n,n,3]")
        if table.dtype != f.dtype or table.embedding_size != C:
            raise V3DError("PE table was built for another dtype / width")
        tbl, n_ids = table.table, table.n_ids
    if newline is not None:
        nl = _dev(newline, "newline").to(f.dtype)
    check(lib().v3d_visual_tokens(_p(f), _p(idt), _p(tbl), n_ids, _p(nl), _p(out), out.stride(0), _code(f), V, side, n, C,
                                  flags, _stream()), "v3d_visual_tokens")
    return out


def embed_gather(weight, ids, out=None):
    w = _dev(weight, "weight")
    i = _dev(ids, "ids").to(torch.int64).reshape(-1)
    vocab, C = w.shape
    if out is None:
        out = torch.empty((i.numel(), C), dtype=w.dtype, device=w.device)
    _rows_fit(out, i.numel(), C, w.dtype, "embed_gather")
    check(lib().v3d_embed_gather(_p(w), vocab, C, _p(i), i.numel(), _p(out), out.stride(0), _code(w), _stream()),
          "v3d_embed_gather")
    return out


# ------------------------------------------------------------------------------ host helpers


def uniform_frame_indices(total_frames, n):
    out = (ctypes.c_int32 * n)()
    check(lib().v3d_uniform_frame_indices_host(total_frames, n, out), "v3d_uniform_frame_indices_host")
    return list(out)


def voxel_keys(xyz, voxel_size=0.1):
    """round(xyz / voxel_size) as int32 (max_coverage_sampling.py:44-45) for an f32 device tensor [..., 3]."""
    x = _dev(xyz, "xyz")
    if x.dtype != torch.float32:
        raise V3DError("voxel_keys wants f32 coordinates")
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=torch.int32, device=x.device)
    check(lib().v3d_voxel_keys_f32(_p(x), x.numel(), float(voxel_size), _p(out), _stream()), "v3d_voxel_keys_f32")
    return out


def greedy_cover_device(keys, scene_voxels, max_frames=32):
    """a3 on the device: keys [n_frames, pts, 3] int32, scene_voxels [m, 3] int32 (both in HBM).
    Returns (sel, gains, num_all_voxels, num_select_voxels) like greedy_cover (host); synchronises to read them."""
    k = _dev(keys, "keys")
    sc = _dev(scene_voxels, "scene_voxels")
    if k.dtype != torch.int32 or sc.dtype != torch.int32 or k.dim() != 3 or k.shape[-1] != 3:
        raise V3DError("greedy_cover_device wants int32 keys [n_frames, pts, 3] and scene [m, 3]")
    k, sc = k.contiguous(), sc.contiguous().reshape(-1, 3)
    n_frames, pts, m = k.shape[0], k.shape[1], sc.shape[0]
    need = lib().v3d_greedy_cover_workspace_bytes(n_frames, m)
    ws = torch.empty(need, dtype=torch.uint8, device=k.device)
    sel = torch.zeros(max_frames, dtype=torch.int32, device=k.device)
    gain = torch.zeros(max_frames, dtype=torch.int64, device=k.device)
    totals = torch.zeros(2, dtype=torch.int64, device=k.device)
    n_sel = torch.zeros(1, dtype=torch.int32, device=k.device)
    check(lib().v3d_greedy_cover(_p(k), n_frames, pts, _p(sc) if m else None, m, max_frames, _p(sel), _p(gain), _p(totals), _p(n_sel),
                                 _p(ws), need, _stream()), "v3d_greedy_cover")
    if int(ws[need - 256: need - 252].view(torch.int32)[0]):
        raise V3DError("v3d_greedy_cover: scene voxel coordinate outside [-2^20, 2^20)")
    n = int(n_sel[0])
    t = totals.tolist()
    return sel[:n].cpu().numpy(), gain[:n].cpu().numpy(), t[0], t[1]


def greedy_cover(keys, scene_voxels, max_frames=32):
    """a3.  keys [n_frames, pts, 3] int32 (numpy), scene_voxels [m,3] int32 ->
    (sel, gains, num_all_voxels, num_select_voxels)."""
    k = np.ascontiguousarray(keys, dtype=np.int32)
    s = np.ascontiguousarray(scene_voxels, dtype=np.int32).reshape(-1, 3)
    n_frames = k.shape[0]
    pts = int(np.prod(k.shape[1:-1]))
    sel = np.zeros(max_frames, np.int32)
    gain = np.zeros(max_frames, np.int64)
    na = ctypes.c_int64()
    nsel = ctypes.c_int64()
    picks = check(lib().v3d_greedy_cover_host(k.ctypes.data_as(ctypes.c_void_p), n_frames, pts,
                                              s.ctypes.data_as(ctypes.c_void_p), s.shape[0], max_frames,
                                              sel.ctypes.data_as(ctypes.c_void_p), gain.ctypes.data_as(ctypes.c_void_p),
                                              ctypes.byref(na), ctypes.byref(nsel)), "v3d_greedy_cover_host")
    return sel[:picks], gain[:picks], na.value, nsel.value


# ------------------------------------------------------------------------------ dense linears

EPI_NONE, EPI_BIAS, EPI_BIAS_GELU_ERF, EPI_BIAS_GELU_TANH, EPI_BIAS_RES, EPI_RES, EPI_SWIGLU, EPI_BIAS_RELU = range(8)


def gemm(a, w, bias=None, res=None, epilogue=EPI_NONE, out=None, res_mod=0):
    """out = epilogue(a @ w.T).  a [M,K] (row stride may exceed K), w [N,K]; N%128==0, K%64==0."""
    if not a.is_cuda or not w.is_cuda:
        raise V3DError("gemm operands must live in HBM")
    if a.stride(-1) != 1 or w.stride(-1) != 1:
        raise V3DError("gemm operands must be K-contiguous")
    M, K = a.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise V3DError(f"gemm: a is [*,{K}] but w is {tuple(w.shape)}")
    n_out = N // 2 if epilogue == EPI_SWIGLU else N
    if out is None:
        out = torch.empty((M, n_out), dtype=a.dtype, device=a.device)
    _rows_fit(out, M, n_out, a.dtype, "gemm")
    if res is not None and (res.shape[0] < (res_mod or M) or res.shape[1] < n_out):
        raise V3DError(f"gemm: res {tuple(res.shape)} is smaller than the output [{res_mod or M},{n_out}]")
    check(lib().v3d_gemm(_p(a), a.stride(0), _p(w), w.stride(0), _p(bias), _p(res),
                         res.stride(0) if res is not None else 0, res_mod, _p(out), out.stride(0), M, N, K, _code(a),
                         epilogue, _stream()), "v3d_gemm")
    return out


def interleave_gate_up(wg, wu):
    """Row order v3d_gemm's SWIGLU epilogue expects: per 128-row tile, 64 gate rows then the 64 up rows."""
    I, K = wg.shape
    assert I % 64 == 0
    return torch.stack([wg.view(I // 64, 64, K), wu.view(I // 64, 64, K)], 1).reshape(2 * I, K).contiguous()


# ------------------------------------------------------------------------------ fp8 (configs[3])


def quantize_fp8_rows(x, q=None, scale=None):
    """x [rows, cols] f16/bf16 -> (q uint8 e4m3 [rows, cols], scale f32 [rows]); per-row amax/448 scales."""
    rows, cols = x.shape
    if q is None:
        q = torch.empty((rows, cols), dtype=torch.uint8, device=x.device)
    if scale is None:
        scale = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(lib().v3d_quantize_fp8_rows(_p(x), x.stride(0), rows, cols, _code(x), _p(q), q.stride(0), _p(scale), _stream()),
          "v3d_quantize_fp8_rows")
    return q, scale


def rmsnorm_quantize_fp8(x, weight, eps, q, scale):
    """RMSNorm + row-wise e4m3 quantisation in one pass: q [rows, cols] uint8 and scale [rows] f32 are written, the 16-bit
    normalised rows are not."""
    rows, cols = x.shape
    check(lib().v3d_rmsnorm_quantize_fp8(_p(x), x.stride(0), _p(weight), eps, rows, cols, _code(x), _p(q), q.stride(0), _p(scale),
                                         _stream()), "v3d_rmsnorm_quantize_fp8")
    return q, scale


def gemm_fp8(qa, sa, qw, sw, out_dtype, bias=None, res=None, epilogue=EPI_NONE, out=None):
    """out = epilogue( (qa @ qw.T) * sa[:,None] * sw[None,:] ); qa [M,K], qw [N,K] e4m3 bytes."""
    M, K = qa.shape
    N = qw.shape[0]
    n_out = N // 2 if epilogue == EPI_SWIGLU else N
    if out is None:
        out = torch.empty((M, n_out), dtype=out_dtype, device=qa.device)
    check(lib().v3d_gemm_fp8(_p(qa), qa.stride(0), _p(sa), _p(qw), qw.stride(0), _p(sw), _p(bias), _p(res),
                             res.stride(0) if res is not None else 0, _p(out), out.stride(0), M, N, K, _DT[out_dtype],
                             epilogue, _stream()), "v3d_gemm_fp8")
    return out


# ------------------------------------------------------------------------------ norms / rotary


def rmsnorm(x, weight, eps=1e-6, out=None):
    rows, cols = x.shape
    if out is None:
        out = torch.empty((rows, cols), dtype=x.dtype, device=x.device)
    _rows_fit(out, rows, cols, x.dtype, "rmsnorm")
    check(lib().v3d_rmsnorm(_p(x), x.stride(0), _p(weight), _p(out), out.stride(0), rows, cols, eps, _code(x), _stream()),
          "v3d_rmsnorm")
    return out


def layernorm(x, weight, bias, eps=1e-6, out=None):
    rows, cols = x.shape
    if out is None:
        out = torch.empty((rows, cols), dtype=x.dtype, device=x.device)
    _rows_fit(out, rows, cols, x.dtype, "layernorm")
    check(lib().v3d_layernorm(_p(x), x.stride(0), _p(weight), _p(bias), _p(out), out.stride(0), rows, cols, eps, _code(x),
                              _stream()), "v3d_layernorm")
    return out


def reference_inv_freq(head_dim, base):
    """modeling_qwen2.py:100 evaluated as the reference does (torch on the host)."""
    return 1.0 / (base ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))


class RopeTable:
    def __init__(self, head_dim, n_pos, base, dtype, device, inv_freq=None):
        self.head_dim, self.n_pos, self.dtype = head_dim, n_pos, dtype
        if inv_freq is None:
            inv_freq = reference_inv_freq(head_dim, base)
        self.inv_freq = inv_freq.to(device=device, dtype=torch.float32).contiguous()
        self.cos = torch.empty((n_pos, head_dim // 2), dtype=dtype, device=device)
        self.sin = torch.empty((n_pos, head_dim // 2), dtype=dtype, device=device)
        check(lib().v3d_rope_table_build(_p(self.inv_freq), head_dim, n_pos, _DT[dtype], _p(self.cos), _p(self.sin),
                                         _stream()), "v3d_rope_table_build")


def rope_apply(x, n_heads, head_dim, table, pos0=0, positions=None):
    """In place on the first n_heads*head_dim columns of x [tokens, ld]."""
    tokens = x.shape[0]
    check(lib().v3d_rope_apply(_p(x), x.stride(0), tokens, n_heads, head_dim, _p(table.cos), _p(table.sin), table.n_pos,
                               _p(positions), pos0, _code(x), _stream()), "v3d_rope_apply")
    return x


def rope_kv_store(qkv, n_q, n_kv, head_dim, table, cache, pos0=0, row0=None, positions=None, dst_rows=None):
    """Prefill rotary + KV-cache append in one launch (include/v3d.h): q heads rotated in place in qkv [tokens, ld], rotated k and
    v written to cache rows row0 + t (default row0 = pos0) or dst_rows[t]; token t at position pos0 + t or positions[t]."""
    tokens = qkv.shape[0]
    if cache.dim() != 2 or cache.stride(1) != 1 or cache.shape[1] < 2 * n_kv * head_dim:
        raise V3DError("rope_kv_store: cache must be [rows, >= 2*n_kv*head_dim]")
    if dst_rows is None:
        row0 = pos0 if row0 is None else row0
        if row0 < 0 or row0 + tokens > cache.shape[0]:
            raise V3DError(f"rope_kv_store: rows {row0}..{row0 + tokens} exceed the cache ({cache.shape[0]} rows)")
    elif dst_rows.dtype != torch.int64 or dst_rows.numel() != tokens or not dst_rows.is_cuda:
        raise V3DError("rope_kv_store: dst_rows must be a device int64 tensor with one entry per token")
    if positions is not None and (positions.dtype != torch.int32 or positions.numel() != tokens or not positions.is_cuda):
        raise V3DError("rope_kv_store: positions must be a device int32 tensor with one entry per token")
    check(lib().v3d_rope_kv_store(_p(qkv), qkv.stride(0), tokens, n_q, n_kv, head_dim, _p(table.cos), _p(table.sin), table.n_pos,
                                  _p(positions), pos0, _p(cache), cache.stride(0), _p(dst_rows), row0 or 0, _code(qkv), _stream()),
          "v3d_rope_kv_store")
    return qkv


def add_row(x, rows, add):
    """x[rows[i]] += add for a device int64 index tensor (the box-centre PE on <coord> token rows, llava_arch.py:697-700)."""
    if rows.dtype != torch.int64 or not rows.is_cuda or rows.numel() == 0:
        raise V3DError("add_row: rows must be a non-empty device int64 tensor")
    if add.dtype != x.dtype or add.numel() != x.shape[1]:
        raise V3DError("add_row: `add` must be one row of x's width and dtype")
    check(lib().v3d_add_row(_p(x), x.stride(0), _p(rows), rows.numel(), x.shape[1], _p(add.contiguous()), _code(x), _stream()), "v3d_add_row")
    return x


# ------------------------------------------------------------------------------ attention


def attention(q, k, v, out, B, Sq, Sk, Hq, Hkv, D, d_out, ldq, ldk, ldv, ldo, bsq, bsk, bso, hsq, hsk, hso, causal,
              q_pos0, scale):
    """Raw strided form (see include/v3d.h); q/k/v/out are tensors whose data_ptr is element (0,0,0,0)."""
    check(lib().v3d_attention(_p(q), _p(k), _p(v), _p(out), _code(q), B, Sq, Sk, Hq, Hkv, D, d_out, ldq, ldk, ldv, ldo,
                              bsq, bsk, bso, hsq, hsk, hso, 1 if causal else 0, q_pos0, scale, _stream()), "v3d_attention")
    return out


def attention_shared_prefix(q, k, v, k_shared, v_shared, shared_len, out, B, Sq, Sk, Hq, Hkv, ldq, ldk, ldv, ldo, bsq, bsk, bso, hsq, hsk, hso,
                            q_pos0, scale):
    """Causal attention (D = 128) of B x Sq question rows at positions q_pos0.. : key tiles below shared_len (a multiple of 64) come from
    the scene's cache k_shared / v_shared, the rest from each question's own cache k / v + b bsk (v3d_attention_shared_prefix; every bit
    equals attention() on B full copies)."""
    check(lib().v3d_attention_shared_prefix(_p(q), _p(k), _p(v), _p(k_shared), _p(v_shared), int(shared_len), _p(out), _code(q), B, Sq, Sk, Hq, Hkv,
                                            ldq, ldk, ldv, ldo, bsq, bsk, bso, hsq, hsk, hso, q_pos0, scale, _stream()),
          "v3d_attention_shared_prefix")
    return out


def attention_bshd(q, k, v, causal, scale=None, q_pos0=0, d_out=None):
    """Convenience: q [B,Sq,Hq,D], k/v [B,Sk,Hkv,D] (last dim contiguous) -> [B,Sq,Hq,d_out]."""
    B, Sq, Hq, D = q.shape
    Sk, Hkv = k.shape[1], k.shape[2]
    d_out = d_out or D
    scale = scale if scale is not None else 1.0 / math.sqrt(d_out)
    out = torch.empty((B, Sq, Hq, d_out), dtype=q.dtype, device=q.device)
    return attention(q, k, v, out, B, Sq, Sk, Hq, Hkv, D, d_out, q.stride(1), k.stride(1), v.stride(1), out.stride(1),
                     q.stride(0), k.stride(0), out.stride(0), q.stride(2), k.stride(2), out.stride(2), causal, q_pos0, scale)


def decode_workspace(Hq, Hkv, device):
    n = lib().v3d_attention_decode_workspace_bytes(Hq, max(1, 1024 // Hkv))
    return torch.empty(n // 4, dtype=torch.float32, device=device)


def attention_decode(q, k_cache, v_cache, out, Sk, Hq, Hkv, scale, workspace, hd=128):
    """One query row [Hq*hd] against caches [>=Sk, Hkv*hd (+...)] (row strides from the tensors)."""
    check(lib().v3d_attention_decode(_p(q), _p(k_cache), _p(v_cache), _p(out), _code(q), Sk, Hq, Hkv, k_cache.stride(0),
                                     v_cache.stride(0), hd, hd, hd, scale, _p(workspace), workspace.numel() * 4, _stream()),
          "v3d_attention_decode")
    return out


DEC_NONE, DEC_BIAS, DEC_RES, DEC_SWIGLU = range(4)


def linear_decode(x, w, out, norm_weight=None, eps=1e-6, bias=None, res=None, epilogue=DEC_NONE):
    """One activation row x [K] against w [N,K]; optional fused RMSNorm in front (see include/v3d.h)."""
    N, K = w.shape
    check(lib().v3d_linear_decode(_p(x), _p(norm_weight), eps, _p(w), w.stride(0), _p(bias), _p(res), _p(out), N, K,
                                  _code(x), epilogue, _stream()), "v3d_linear_decode")
    return out


def linear_decode_rows(x, w, out, norm_weight=None, eps=1e-6, bias=None, res=None, epilogue=DEC_NONE):
    """M activation rows x [M, K] against w [N, K] in ONE pass over the weights (scenes decoding together).  2 <= M <= 16
    (no fused norm, K % 128 == 0, N % 16 == 0): matrix-core form - a row's result depends neither on the other rows nor
    on M, and equals linear_decode(x[m], ...) up to the f32 summation order.  Other shapes: VALU form, M <= 4, rows
    bit-identical to linear_decode."""
    N, K = w.shape
    M = x.shape[0]
    check(lib().v3d_linear_decode_rows(_p(x), x.stride(0), M, _p(norm_weight), eps, _p(w), w.stride(0), _p(bias), _p(res),
                                       res.stride(0) if res is not None else 0, _p(out), out.stride(0), N, K, _code(x),
                                       epilogue, _stream()), "v3d_linear_decode_rows")
    return out


def linear_decode_rows_fuses_norm(M, N, K, epilogue=DEC_NONE):
    """True when linear_decode_rows(x [M, K], w [N, K], norm_weight=...) normalises the rows inside the matrix-core launch (bit-identical
    to rmsnorm() + the unfused call); otherwise normalise with rmsnorm() first (v3d_linear_decode_rows_fuses_norm)."""
    return bool(lib().v3d_linear_decode_rows_fuses_norm(int(M), int(N), int(K), int(epilogue)))


def linear_decode_fp8_rows(x, qw, sw, out, bias=None, res=None, epilogue=DEC_NONE):
    """x [M, K] 16-bit rows against e4m3 weights qw [N, K] (uint8) with row scales sw [N] (W8A16 decode, configs[3])."""
    N, K = qw.shape
    M = x.shape[0]
    check(lib().v3d_linear_decode_fp8_rows(_p(x), x.stride(0), M, _p(qw), qw.stride(0), _p(sw), _p(bias), _p(res),
                                           res.stride(0) if res is not None else 0, _p(out), out.stride(0), N, K, _code(x),
                                           epilogue, _stream()), "v3d_linear_decode_fp8_rows")
    return out


def rope_kv_append(qkv_row, n_q, n_kv, hd, table, pos, cache_row):
    check(lib().v3d_rope_kv_append(_p(qkv_row), n_q, n_kv, hd, _p(table.cos), _p(table.sin), table.n_pos, pos,
                                   _p(cache_row), _code(qkv_row), _stream()), "v3d_rope_kv_append")


def _host_ptrs(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _host_ints(values):
    return (ctypes.c_int * len(values))(*[int(v) for v in values])


def rope_kv_append_rows(qkv, n_q, n_kv, hd, table, positions, cache_rows):
    """qkv [M, n] rows of M scenes decoding together; positions / cache_rows: one per scene."""
    check(lib().v3d_rope_kv_append_rows(_p(qkv), qkv.stride(0), qkv.shape[0], n_q, n_kv, hd, _p(table.cos), _p(table.sin),
                                        table.n_pos, _host_ints(positions), _host_ptrs(cache_rows), _code(qkv), _stream()),
          "v3d_rope_kv_append_rows")


def attention_decode_rows(q, k_caches, v_caches, out, sk, n_heads, n_kv_heads, scale, workspace, prefix=0, prefix_kv=None):
    """q / out [M, Hq*128]; k_caches / v_caches: per-scene cache views [>=Sk, ...] sharing strides; sk: lengths.
    prefix > 0: the caches all start with the same `prefix` rows (questions about one scene): those keys are read from the
    first cache for every row - or from prefix_kv = (k, v) views of the scene's own cache with the same strides, when the questions'
    caches do not hold those rows at all - the rest from each row's own cache (bit-identical outputs, the prefix is streamed once)."""
    k0 = k_caches[0]
    if prefix > 0:
        if min(sk) < prefix:
            raise V3DError("attention_decode_rows: a scene is shorter than the shared prefix")
        kp, vp = prefix_kv if prefix_kv is not None else (k0, v_caches[0])
        if kp.stride(0) != k0.stride(0) or vp.stride(0) != v_caches[0].stride(0):
            raise V3DError("attention_decode_rows: the prefix cache must share the row stride of the scenes' caches")
        check(lib().v3d_attention_decode_rows_prefix(_p(q), q.stride(0), q.shape[0], _p(kp), _p(vp), int(prefix), _host_ptrs(k_caches),
                                                     _host_ptrs(v_caches), _host_ints(sk), _p(out), out.stride(0), _code(q), n_heads, n_kv_heads,
                                                     k0.stride(0), v_caches[0].stride(0), 128, 128, 128, float(scale), _p(workspace),
                                                     workspace.numel() * workspace.element_size(), _stream()), "v3d_attention_decode_rows_prefix")
        return out
    check(lib().v3d_attention_decode_rows(_p(q), q.stride(0), q.shape[0], _host_ptrs(k_caches), _host_ptrs(v_caches), _host_ints(sk),
                                          _p(out), out.stride(0), _code(q), n_heads, n_kv_heads, k0.stride(0), v_caches[0].stride(0),
                                          128, 128, 128, float(scale), _p(workspace), workspace.numel() * workspace.element_size(),
                                          _stream()), "v3d_attention_decode_rows")
    return out


def argmax_rows(x, out, workspace):
    """x [M, n] -> out int64 [M]; workspace >= M KiB."""
    check(lib().v3d_argmax_rows(_p(x), x.stride(0), x.shape[0], x.shape[1], _code(x), _p(out), _p(workspace), _stream()),
          "v3d_argmax_rows")
    return out


def eos_update(tokens, eos_ids, done, n_done):
    """tokens int64 [M] (device), eos_ids int64 [n] (device), done int32 [M], n_done int32 [1]: done |= tokens in eos_ids;
    n_done = done.sum() (v3d_eos_update)."""
    check(lib().v3d_eos_update(_p(tokens), tokens.numel(), _p(eos_ids), eos_ids.numel(), _p(done), _p(n_done), _stream()), "v3d_eos_update")
    return n_done


_argmax_ws = {}


def argmax(x, out, workspace=None):
    """out: int64 device tensor of one element; workspace: >= 1 KiB device scratch (one per stream in flight)."""
    if workspace is None:
        key = (x.device, torch.cuda.current_stream().cuda_stream)
        workspace = _argmax_ws.get(key)
        if workspace is None:
            workspace = _argmax_ws[key] = torch.empty(256, dtype=torch.float32, device=x.device)
    check(lib().v3d_argmax(_p(x), x.numel(), _code(x), _p(out), _p(workspace), _stream()), "v3d_argmax")
    return out


# ------------------------------------------------------------------------------ grounding


def object_patch_mask(coords, boxes, cell=14, thresh=None):
    """coords [F,S,S,3], boxes [n,6] (same dtype) -> uint8 [n, F, g, g]."""
    c, b = _dev(coords, "coords"), _dev(boxes, "boxes").to(coords.dtype)
    F_, S = c.shape[0], c.shape[1]
    g = (S - 6) // cell
    n = b.shape[0]
    thresh = int(cell * cell * 0.5) if thresh is None else thresh
    mask = torch.zeros((n, F_, g, g), dtype=torch.uint8, device=c.device)
    check(lib().v3d_object_patch_mask(_p(c), _code(c), F_, S, cell, _p(b), n, thresh, _p(mask), _stream()), "v3d_object_patch_mask")
    return mask


def masked_mean(feat, mask, add=None):
    """feat [T,C]; mask uint8 [n, T(flattened)] -> [n, C]."""
    f = _dev(feat, "feat")
    T_, C = f.shape
    n = mask.shape[0]
    out = torch.empty((n, C), dtype=f.dtype, device=f.device)
    check(lib().v3d_masked_mean(_p(f), _p(mask), n, T_, C, _p(add), _p(out), _code(f), _stream()), "v3d_masked_mean")
    return out


def ground_scores(obj, query):
    n, C = obj.shape
    out = torch.empty(n, dtype=obj.dtype, device=obj.device)
    check(lib().v3d_ground_scores(_p(obj), obj.stride(0), n, _p(query), C, _p(out), _code(obj), _stream()), "v3d_ground_scores")
    return out


def row_dots(x, q, bias=None, products_rounded=False):
    """out[i] = sum_c x[i, c] * q[c] (+ bias[0]) - the 'mlp' head's product-and-sum (products_rounded) or a Linear(C, 1) row."""
    n, C = x.shape
    if q.numel() != C or q.dtype != x.dtype or x.stride(1) != 1 or not q.is_contiguous():
        raise V3DError(f"row_dots: x {tuple(x.shape)} against q {tuple(q.shape)}")
    out = torch.empty(n, dtype=x.dtype, device=x.device)
    check(lib().v3d_row_dots(_p(x), x.stride(0), n, _p(q), C, _p(bias) if bias is not None else None, int(products_rounded), _p(out),
                             _code(x), _stream()), "v3d_row_dots")
    return out


def relu_mul_rows(x, row=None, relu=True):
    """In place: x[i, c] = relu?(x[i, c]) * (row[c] if row is given)."""
    if x.dim() != 2 or x.stride(1) != 1 or (row is not None and (row.numel() != x.shape[1] or row.dtype != x.dtype or not row.is_contiguous())):
        raise V3DError(f"relu_mul_rows: x {tuple(x.shape)}")
    check(lib().v3d_relu_mul_rows(_p(x), x.stride(0), x.shape[0], x.shape[1], _p(row) if row is not None else None, int(relu), _code(x),
                                  _stream()), "v3d_relu_mul_rows")
    return x


# ------------------------------------------------------------------------------ data movement


def copy_rows(src, dst, cols=None):
    rows = src.shape[0]
    cols = cols or src.shape[1]
    _rows_fit(dst, rows, cols, src.dtype, "copy_rows")
    check(lib().v3d_copy_rows(_p(src), src.stride(0), _p(dst), dst.stride(0), rows, cols, _code(src), _stream()),
          "v3d_copy_rows")
    return dst


def copy_rows_bcast(src, dst, cols=None):
    """src [rows, >=cols] -> dst [n_copies, >=rows, >=cols] (every copy gets the same rows)."""
    rows = src.shape[0]
    cols = cols or src.shape[1]
    if dst.dim() != 3 or dst.shape[1] < rows or dst.shape[2] < cols or dst.stride(2) != 1 or src.stride(1) != 1 or dst.dtype != src.dtype:
        raise V3DError(f"copy_rows_bcast: dst {tuple(dst.shape)} cannot hold copies of [{rows},{cols}]")
    check(lib().v3d_copy_rows_bcast(_p(src), src.stride(0), _p(dst), dst.stride(1), rows, cols, dst.shape[0], dst.stride(0), _code(src),
                                    _stream()), "v3d_copy_rows_bcast")
    return dst


def preprocess_rgb(frames, dtype=torch.float32, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5), rescale=1 / 255):
    """frames [F,H,W,3] uint8 (device) -> pixel_values [F,3,H,W] (SigLipImageProcessor's rescale + normalize + CHW)."""
    fr = _dev(frames, "frames")
    if fr.dtype != torch.uint8 or fr.dim() != 4 or fr.shape[-1] != 3:
        raise V3DError("preprocess_rgb wants [F,H,W,3] uint8 frames")
    fr = fr.contiguous()
    F_, H, W, _ = fr.shape
    out = torch.empty((F_, 3, H, W), dtype=dtype, device=fr.device)
    m = (ctypes.c_float * 3)(*mean)
    sd = (ctypes.c_float * 3)(*std)
    check(lib().v3d_preprocess_rgb_u8(_p(fr), F_, H, W, m, sd, float(rescale), _p(out), _DT[dtype], _stream()), "v3d_preprocess_rgb_u8")
    return out


# ---- Pillow's resampling tables (libImaging/Resample.c precompute_coeffs + normalize_coeffs_8bpc), evaluated in double exactly as
#      Pillow does: python floats ARE C doubles and every statement below is one of Resample.c's, in its order.  The integer
#      tables are all the device kernel needs (v3d_resize_bicubic_u8).
_PIL_PRECISION_BITS = 32 - 8 - 2


def _pil_bicubic(x):
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_resample_tables(in_size, out_size, support=2.0, filt=_pil_bicubic):
    """(bounds int32 [out,2], coeffs int32 [out,ksize], ksize) of one resize pass in_size -> out_size over the whole axis
    (box = (0, in_size)), BICUBIC by default (Image.resize's default filter for RGB)."""
    in0, in1 = 0.0, float(in_size)
    scale = filterscale = (in1 - in0) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    sup = support * filterscale
    ksize = int(math.ceil(sup)) * 2 + 1
    bounds = torch.zeros((out_size, 2), dtype=torch.int32)
    kk = torch.zeros((out_size, ksize), dtype=torch.int32)
    one = float(1 << _PIL_PRECISION_BITS)
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        ww = 0.0
        ss = 1.0 / filterscale
        xmin = int(center - sup + 0.5)            # (int) truncates toward zero, as the C cast does; the value is >= -sup + 0.5
        if xmin < 0:
            xmin = 0
        xmax = int(center + sup + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = []
        for x in range(xmax):
            w = filt((x + xmin - center + 0.5) * ss)
            k.append(w)
            ww += w
        for x in range(xmax):
            if ww != 0.0:
                k[x] /= ww
        bounds[xx, 0], bounds[xx, 1] = xmin, xmax
        for x in range(xmax):
            kk[xx, x] = int(-0.5 + k[x] * one) if k[x] < 0 else int(0.5 + k[x] * one)
    return bounds, kk, ksize


_RESIZE_TABLES = {}


def _resize_tables(in_size, out_size, device):
    key = (in_size, out_size, str(device))
    t = _RESIZE_TABLES.get(key)
    if t is None:
        b, k, ks = pil_resample_tables(in_size, out_size)
        t = _RESIZE_TABLES[key] = (b.to(device).contiguous(), k.to(device).contiguous(), ks)
        # the tables outlive this call and are read from OTHER streams later (two prefill streams): the one-off upload must have
        # landed before any of them can see the cached tensors
        if torch.device(device).type == "cuda":
            torch.cuda.current_stream(torch.device(device)).synchronize()
    return t


def resize_crop_rgb(frames, out_hw, crop=None, dtype=torch.uint8, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5), rescale=1 / 255):
    """PIL `Image.resize((OW, OH))` (default BICUBIC) + `.crop((left, top, left + cw, top + ch))` of RGB uint8 frames [F,H,W,3] on
    the device, bit for bit (video_utils.py:285-306).  crop = (top, left, ch, cw) inside the resized image (default: all of it).
    dtype uint8 -> [F,ch,cw,3] uint8; a float dtype -> SigLipImageProcessor's pixel_values [F,3,ch,cw] (rescale, normalise, CHW
    fused: v3d_preprocess_rgb_u8's arithmetic on the resized bytes)."""
    fr = _dev(frames, "frames")
    if fr.dtype != torch.uint8 or fr.dim() != 4 or fr.shape[-1] != 3:
        raise V3DError("resize_crop_rgb wants [F,H,W,3] uint8 frames")
    fr = fr.contiguous()
    F_, H, W, _ = fr.shape
    OH, OW = int(out_hw[0]), int(out_hw[1])
    top, left, ch, cw = crop if crop is not None else (0, 0, OH, OW)
    bh, kh, ksh = _resize_tables(W, OW, fr.device)
    bv, kv, ksv = _resize_tables(H, OH, fr.device)
    if dtype == torch.uint8:
        out = torch.empty((F_, ch, cw, 3), dtype=torch.uint8, device=fr.device)
        code, m, sd = 16, None, None
    else:
        out = torch.empty((F_, 3, ch, cw), dtype=dtype, device=fr.device)
        code, m, sd = _DT[dtype], (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    check(lib().v3d_resize_bicubic_u8(_p(fr), F_, H, W, OH, OW, _p(bh), _p(kh), ksh, _p(bv), _p(kv), ksv, top, left, ch, cw, m, sd,
                                      float(rescale), _p(out), code, _stream()), "v3d_resize_bicubic_u8")
    return out


def patchify(images, patch=14, kpad=640):
    B, C, S, _ = images.shape
    g = S // patch
    im = _dev(images, "images")
    out = torch.empty((B * g * g, kpad), dtype=im.dtype, device=im.device)
    check(lib().v3d_patchify(_p(im), _p(out), B, S, patch, kpad, _code(im), _stream()), "v3d_patchify")
    return out


# ------------------------------------------------------------------------------ training step (configs[4], first kernels)


def cross_entropy(logits, labels, ignore_index=-100):
    """Qwen2ForCausalLM's loss (modeling_qwen2.py:1195-1205): logits [S, vocab] (f32 / f16 / bf16), labels [S] int64 on the device;
    position t predicts label t + 1.  Returns (mean loss f32 scalar tensor, state for cross_entropy_grad)."""
    lg = _dev(logits, "logits")
    lb = _dev(labels, "labels").to(torch.int64).reshape(-1)
    S, V = lg.shape
    if lb.numel() != S or S < 2:
        raise V3DError("cross_entropy: labels must have one entry per position (>= 2 positions)")
    loss_rows = torch.empty(S - 1, dtype=torch.float32, device=lg.device)
    lse_rows = torch.empty(S - 1, dtype=torch.float32, device=lg.device)
    mean_count = torch.empty(2, dtype=torch.float32, device=lg.device)
    check(lib().v3d_cross_entropy(_p(lg), lg.stride(0), _code(lg), S, V, _p(lb), ignore_index, _p(loss_rows), _p(lse_rows), _p(mean_count),
                                  _stream()), "v3d_cross_entropy")
    return mean_count[0], (lg, lb, lse_rows, mean_count, ignore_index, loss_rows)


def cross_entropy_grad(state, upstream=1.0, dtype=None):
    """d loss / d logits for the state cross_entropy returned, [S, vocab] in `dtype` (default: the logits' dtype)."""
    lg, lb, lse_rows, mean_count, ignore_index, _ = state
    S, V = lg.shape
    dtype = dtype or lg.dtype
    out = torch.empty((S, V), dtype=dtype, device=lg.device)
    check(lib().v3d_cross_entropy_grad(_p(lg), lg.stride(0), _code(lg), S, V, _p(lb), ignore_index, _p(lse_rows), _p(mean_count),
                                       float(upstream), _p(out), out.stride(0), _DT[dtype], _stream()), "v3d_cross_entropy_grad")
    return out


def visual_tokens_grad(dout, V, side=27, n=14, newline=True):
    """Backward of visual_tokens(pool=True [, table] [, newline]) for dout [rows, C] (16-bit): returns (dfeat [V, side*side, C],
    dnewline f32 [C] or None)."""
    g = _dev(dout, "dout")
    rows, C = g.shape
    if rows != V * n * ((n + 1) if newline else n):
        raise V3DError(f"visual_tokens_grad: dout has {rows} rows, expected {V * n * ((n + 1) if newline else n)}")
    dfeat = torch.empty((V, side * side, C), dtype=g.dtype, device=g.device)
    dnl = torch.empty(C, dtype=torch.float32, device=g.device) if newline else None
    flags = VT_POOL | VT_PE | (VT_NEWLINE if newline else 0)
    check(lib().v3d_visual_tokens_grad(_p(g), g.stride(0), _p(dfeat), _p(dnl), _code(g), V, side, n, C, flags, _stream()), "v3d_visual_tokens_grad")
    return dfeat, dnl


# ------------------------------------------------------------------------------ backward of the dense blocks (configs[4])


def transpose(x, out_cols=None, out=None):
    """out[c, r] = x[r, c] for a 16-bit x [rows, cols]; out is [cols, out_cols] with columns >= rows zero (out_cols % 8 == 0)."""
    x = _dev(x, "x")
    rows, cols = x.shape
    out_cols = out_cols or (rows + 7) // 8 * 8
    if out is None:
        out = torch.empty((cols, out_cols), dtype=x.dtype, device=x.device)
    _rows_fit(out, cols, out_cols, x.dtype, "transpose")
    check(lib().v3d_transpose(_p(x), x.stride(0), rows, cols, _p(out), out.stride(0), out_cols, _code(x), _stream()), "v3d_transpose")
    return out


def _colsum_ws(rows, cols, device):
    return torch.empty(max(1, lib().v3d_colsum_workspace_bytes(rows, cols) // 4), dtype=torch.float32, device=device)


def colsum(x, dtype=None):
    """Sum over the rows of a 16-bit x [rows, cols] (f32 accumulation in a fixed order) -> [cols] in `dtype` (default x's)."""
    x = _dev(x, "x")
    rows, cols = x.shape
    out = torch.empty(cols, dtype=dtype or x.dtype, device=x.device)
    check(lib().v3d_colsum(_p(x), x.stride(0), rows, cols, _code(x), _p(_colsum_ws(rows, cols, x.device)), _p(out), _DT[out.dtype], _stream()),
          "v3d_colsum")
    return out


def rmsnorm_grad(x, weight, dy, eps=1e-6, add=None, dw_dtype=None):
    """Backward of rmsnorm: returns (dx [rows, cols] (+ add), dweight [cols])."""
    x, dy = _dev(x, "x"), _dev(dy, "dy")
    rows, cols = x.shape
    if tuple(dy.shape) != (rows, cols) or (add is not None and tuple(add.shape) != (rows, cols)):
        raise V3DError("rmsnorm_grad: dy / add must have x's shape")
    dx = torch.empty((rows, cols), dtype=x.dtype, device=x.device)
    dw = torch.empty(cols, dtype=dw_dtype or x.dtype, device=x.device)
    check(lib().v3d_rmsnorm_grad(_p(x), x.stride(0), _p(weight), _p(dy), dy.stride(0), _p(add), add.stride(0) if add is not None else 0,
                                 _p(dx), dx.stride(0), _p(_colsum_ws(rows, cols, x.device)), _p(dw), _DT[dw.dtype], rows, cols, eps,
                                 _code(x), _stream()), "v3d_rmsnorm_grad")
    return dx, dw


def swiglu(gu, out=None):
    """act_fn(gate) * up on planar rows gu = [gate | up] -> [rows, inter]."""
    gu = _dev(gu, "gu")
    rows, two = gu.shape
    inter = two // 2
    if out is None:
        out = torch.empty((rows, inter), dtype=gu.dtype, device=gu.device)
    check(lib().v3d_swiglu(_p(gu), gu.stride(0), _p(out), out.stride(0), rows, inter, _code(gu), _stream()), "v3d_swiglu")
    return out


def swiglu_grad(gu, dh):
    """Gradient of swiglu with respect to gu, planar [dgate | dup]."""
    gu, dh = _dev(gu, "gu"), _dev(dh, "dh")
    rows, two = gu.shape
    inter = two // 2
    if tuple(dh.shape) != (rows, inter):
        raise V3DError("swiglu_grad: dh must be [rows, inter]")
    dgu = torch.empty((rows, two), dtype=gu.dtype, device=gu.device)
    check(lib().v3d_swiglu_grad(_p(gu), gu.stride(0), _p(dh), dh.stride(0), _p(dgu), dgu.stride(0), rows, inter, _code(gu), _stream()),
          "v3d_swiglu_grad")
    return dgu


def causal_softmax_rows(s, n_keys, scale, offset=0, out=None):
    """p[i, j] = softmax over the keys j <= i + offset (j < n_keys) of s[i, j] * scale, zero in the other columns of s's width."""
    s = _dev(s, "s")
    rows, cols = s.shape
    if out is None:
        out = torch.empty((rows, cols), dtype=s.dtype, device=s.device)
    check(lib().v3d_causal_softmax_rows(_p(s), s.stride(0), _p(out), out.stride(0), rows, n_keys, cols, offset, float(scale), _code(s), _stream()),
          "v3d_causal_softmax_rows")
    return out


def softmax_grad_rows(p, dp, scale, out=None):
    """ds = p * (dp - rowsum(p * dp)) * scale."""
    p, dp = _dev(p, "p"), _dev(dp, "dp")
    rows, cols = p.shape
    if out is None:
        out = torch.empty((rows, cols), dtype=p.dtype, device=p.device)
    check(lib().v3d_softmax_grad_rows(_p(p), p.stride(0), _p(dp), dp.stride(0), _p(out), out.stride(0), rows, cols, float(scale), _code(p), _stream()),
          "v3d_softmax_grad_rows")
    return out


def attention_train(qkv, out, S, n_q, n_kv, scale, B=1, causal=True):
    """Attention (head dim 128) over B sequences of S rows whose q | k | v sit side by side in qkv [>= B S, (n_q + 2 n_kv) 128] (sequence b =
    rows [b S, (b + 1) S)); out [B S, n_q 128].  Returns the row log-sum-exp [B, n_q, S] f32 (scaled log2 units) for attention_backward."""
    qkv = _dev(qkv, "qkv")
    w = qkv.stride(0)
    lse = torch.empty((B, n_q, S), dtype=torch.float32, device=qkv.device)
    check(lib().v3d_attention_train(_p(qkv), _p(qkv[:, n_q * 128:]), _p(qkv[:, (n_q + n_kv) * 128:]), _p(out), _p(lse), _code(qkv), B, S, S,
                                    n_q, n_kv, w, w, w, out.stride(0), S * w, S * w, S * out.stride(0), 128, 128, 128, 1 if causal else 0, 0,
                                    float(scale), _stream()), "v3d_attention_train")
    return lse


def attention_backward(qkv, out, dout, lse, dqkv, S, n_q, n_kv, scale, B=1, causal=True):
    """Gradients of attention_train with respect to q | k | v, written side by side into dqkv [B S, (n_q + 2 n_kv) 128]."""
    qkv, dout = _dev(qkv, "qkv"), _dev(dout, "dout")
    w, dw, so, sd = qkv.stride(0), dqkv.stride(0), out.stride(0), dout.stride(0)
    nbytes = lib().v3d_attention_backward_workspace_bytes(B, S, n_q)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=qkv.device)
    check(lib().v3d_attention_backward(_p(qkv), _p(qkv[:, n_q * 128:]), _p(qkv[:, (n_q + n_kv) * 128:]), _p(out), _p(dout), _p(lse),
                                       _p(dqkv), _p(dqkv[:, n_q * 128:]), _p(dqkv[:, (n_q + n_kv) * 128:]), _code(qkv), B, S, n_q, n_kv,
                                       w, w, w, so, sd, dw, dw, dw, S * w, S * w, S * w, S * so, S * sd, S * dw, S * dw, S * dw,
                                       1 if causal else 0, float(scale), _p(ws), nbytes, _stream()), "v3d_attention_backward")
    return dqkv


def adamw_step(p32, m, v, grad, p16=None, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, step=1, grad_scale=1.0):
    """torch.optim.AdamW's update on flat f32 state (in place) from a gradient of any float dtype; p16: the 16-bit parameter copy."""
    n = p32.numel()
    if m.numel() != n or v.numel() != n or grad.numel() != n or (p16 is not None and p16.numel() != n):
        raise V3DError("adamw_step: p32 / m / v / grad / p16 must have the same number of elements")
    for t, name in ((p32, "p32"), (m, "m"), (v, "v")):
        if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
            raise V3DError(f"adamw_step: {name} must be a contiguous f32 tensor in HBM")
    g = _dev(grad, "grad")
    if not g.is_contiguous() or (p16 is not None and not p16.is_contiguous()):
        raise V3DError("adamw_step: grad / p16 must be contiguous")
    check(lib().v3d_adamw_step(_p(p32), _p(m), _p(v), _p(g), _DT[g.dtype], _p(p16), _DT[p16.dtype] if p16 is not None else 0, n, float(lr),
                               float(betas[0]), float(betas[1]), float(eps), float(weight_decay), int(step), float(grad_scale), _stream()),
          "v3d_adamw_step")


_sumsq_ws = {}


def sumsq(x, out=None, accumulate=False):
    """Sum of squares of a flat device tensor (f32 / f16 / bf16) in f32, deterministic (v3d_sumsq) -> out f32 [1] on the device
    (accumulate: added to it).  The global gradient norm of clip_grad_norm_ is sqrt of the sum over the gradient tensors."""
    t = _dev(x, "x")
    if not t.is_contiguous():
        raise V3DError("sumsq wants a contiguous tensor")
    if out is None:
        out = torch.empty(1, dtype=torch.float32, device=t.device)
        accumulate = False
    key = (t.device, torch.cuda.current_stream().cuda_stream)
    ws = _sumsq_ws.get(key)
    if ws is None:
        ws = _sumsq_ws[key] = torch.empty(lib().v3d_sumsq_workspace_bytes() // 4, dtype=torch.float32, device=t.device)
    if t.numel() == 0:
        if not accumulate:
            out.zero_()
        return out
    check(lib().v3d_sumsq(_p(t), t.numel(), _DT[t.dtype], _p(out), 1 if accumulate else 0, _p(ws), _stream()), "v3d_sumsq")
    return out


def embed_grad(dh, rows, ids, dE):
    """dE[ids[i]] = sum of dh[rows[j]] over the j with the same id (text rows of a sample); dE [vocab, H] pre-zeroed by the caller.
    ids outside [0, vocab) (e.g. IMAGE_TOKEN_INDEX of a raw prompt) and rows outside dh are skipped."""
    dh = _dev(dh, "dh")
    if rows.dtype != torch.int64 or ids.dtype != torch.int64 or rows.numel() != ids.numel() or not rows.is_cuda or not ids.is_cuda:
        raise V3DError("embed_grad: rows / ids must be device int64 tensors of one length")
    check(lib().v3d_embed_grad(_p(dh), dh.stride(0), dh.shape[0], _p(rows), _p(ids), rows.numel(), dh.shape[1], _p(dE), dE.stride(0), dE.shape[0],
                               _code(dh), _stream()), "v3d_embed_grad")
    return dE


def gelu(z, tanh_form=False):
    """tanh_form: False = erf GELU, True = tanh GELU, 2 = ReLU."""
    z = _dev(z, "z")
    out = torch.empty_like(z)
    check(lib().v3d_gelu(_p(z), z.stride(0), _p(out), out.stride(0), z.shape[0], z.shape[1], int(tanh_form), _code(z), _stream()), "v3d_gelu")
    return out


def gelu_grad(z, dy, tanh_form=False):
    z, dy = _dev(z, "z"), _dev(dy, "dy")
    if z.shape != dy.shape:
        raise V3DError("gelu_grad: dy must have z's shape")
    out = torch.empty_like(z)
    check(lib().v3d_gelu_grad(_p(z), z.stride(0), _p(dy), dy.stride(0), _p(out), out.stride(0), z.shape[0], z.shape[1], int(tanh_form),
                              _code(z), _stream()), "v3d_gelu_grad")
    return out


def layernorm_grad(x, weight, dy, eps=1e-6, add=None, dw_dtype=None):
    """Backward of layernorm: returns (dx [rows, cols] (+ add), dweight [cols], dbias [cols])."""
    x, dy = _dev(x, "x"), _dev(dy, "dy")
    rows, cols = x.shape
    if tuple(dy.shape) != (rows, cols) or (add is not None and tuple(add.shape) != (rows, cols)):
        raise V3DError("layernorm_grad: dy / add must have x's shape")
    dx = torch.empty((rows, cols), dtype=x.dtype, device=x.device)
    dw = torch.empty(cols, dtype=dw_dtype or x.dtype, device=x.device)
    db = torch.empty(cols, dtype=dw_dtype or x.dtype, device=x.device)
    ws = torch.empty(max(1, 2 * lib().v3d_colsum_workspace_bytes(rows, cols) // 4), dtype=torch.float32, device=x.device)
    check(lib().v3d_layernorm_grad(_p(x), x.stride(0), _p(weight), _p(dy), dy.stride(0), _p(add), add.stride(0) if add is not None else 0,
                                   _p(dx), dx.stride(0), _p(ws), _p(dw), _p(db), _DT[dw.dtype], rows, cols, eps, _code(x), _stream()),
          "v3d_layernorm_grad")
    return dx, dw, db


def axpy(y, x, alpha=1.0):
    """y += alpha * x in place (flat view of two contiguous 16-bit tensors of one shape)."""
    y, x = _dev(y, "y"), _dev(x, "x")
    if y.shape != x.shape or y.dtype != x.dtype or not y.is_contiguous() or not x.is_contiguous():
        raise V3DError("axpy: y and x must be contiguous tensors of one shape and dtype")
    check(lib().v3d_axpy(_p(y), _p(x), float(alpha), y.numel(), _code(y), _stream()), "v3d_axpy")
    return y


def ground_infonce(obj, query, positive, temperature):
    """The grounding loss and its gradients: obj [n, C] head outputs (zero-target row included), query [C], positive uint8 [n] on the device.
    Returns (loss f32 scalar tensor, scores f32 [n], dobj [n, C], dquery [C])."""
    obj, query = _dev(obj, "obj"), _dev(query, "query")
    n, C = obj.shape
    if positive.dtype != torch.uint8 or positive.numel() != n or not positive.is_cuda:
        raise V3DError("ground_infonce: positive must be a device uint8 tensor with one entry per row")
    loss = torch.empty(1, dtype=torch.float32, device=obj.device)
    scores = torch.empty(n, dtype=torch.float32, device=obj.device)
    dobj, dq = torch.empty_like(obj), torch.empty_like(query)
    check(lib().v3d_ground_infonce(_p(obj), obj.stride(0), n, _p(query), C, _p(positive), float(temperature), _p(loss), _p(scores), _p(dobj),
                                   dobj.stride(0), _p(dq), _code(obj), _stream()), "v3d_ground_infonce")
    return loss[0], scores, dobj, dq


def masked_mean_grad(mask, dobj, dfeat, accumulate=True):
    """dfeat [T, C] (+)= the gradient of masked_mean(feat, mask) for dobj [n, C]; mask uint8 [n, T]."""
    dobj = _dev(dobj, "dobj")
    n, C = dobj.shape
    T_ = dfeat.shape[0]
    if mask.dtype != torch.uint8 or mask.numel() != n * T_ or not dfeat.is_contiguous() or not dobj.is_contiguous():
        raise V3DError("masked_mean_grad: mask must be uint8 [n, T], dfeat [T, C] and dobj [n, C] contiguous")
    inv = torch.empty(n, dtype=torch.float32, device=dobj.device)
    check(lib().v3d_masked_mean_grad(_p(mask), n, T_, C, _p(dobj), _p(dfeat), 1 if accumulate else 0, _p(inv), _code(dobj), _stream()),
          "v3d_masked_mean_grad")
    return dfeat
