"""ctypes binding of libv3d_hip.so (include/v3d.h).  No fallback: if the library is missing the
import of any op fails loudly - the product path never computes on the CPU."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "libv3d_hip.so")

_lib = None
ABI_VERSION = 7      # include/v3d.h V3D_ABI_VERSION


class V3DError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise V3DError(
                f"{LIB_PATH} not found: build it with `make -C video-3d-llm_amd/csrc` "
                "(or __graft_entry__.build()).  There is no CPU fallback for this path.")
        _lib = ctypes.CDLL(LIB_PATH)
        _declare(_lib)
        if _lib.v3d_abi_version() != ABI_VERSION:
            raise V3DError(f"libv3d_hip.so has ABI version {_lib.v3d_abi_version()}, this binding needs {ABI_VERSION}: rebuild it")
    return _lib


c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_l = ctypes.c_int64
c_f = ctypes.c_float

# name -> (restype, argtypes); mirrors include/v3d.h one to one
SIGNATURES = {
    "v3d_abi_version": (c_i, []),
    "v3d_last_error": (ctypes.c_char_p, []),
    "v3d_unproject_f32": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "v3d_unproject_sampled_u16": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "v3d_unproject_resized_u16": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "v3d_clamp_xyz": (c_i, [c_p, c_l, c_p, c_p, c_i, c_p]),
    "v3d_unproject_bounds_workspace_bytes": (c_l, [c_i]),
    "v3d_unproject_bounds_u16": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_l, c_p]),
    "v3d_coord_pool_voxel": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_f, c_p, c_p, c_p, c_p]),
    "v3d_discrete_coords": (c_i, [c_p, c_i, c_l, c_p, c_p, c_f, c_p, c_p, c_p]),
    "v3d_sin3d_table_row_elems": (c_l, [c_i, c_i]),
    "v3d_sin3d_table_build": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_p]),
    "v3d_sin3d_pe": (c_i, [c_p, c_i, c_l, c_p, c_i, c_p, c_p]),
    "v3d_visual_tokens": (c_i, [c_p, c_p, c_p, c_i, c_p, c_p, c_l, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "v3d_embed_gather": (c_i, [c_p, c_l, c_i, c_p, c_l, c_p, c_l, c_i, c_p]),
    "v3d_gemm": (c_i, [c_p, c_l, c_p, c_l, c_p, c_p, c_l, c_i, c_p, c_l, c_i, c_i, c_i, c_i, c_i, c_p]),
    "v3d_quantize_fp8_rows": (c_i, [c_p, c_l, c_l, c_i, c_i, c_p, c_l, c_p, c_p]),
    "v3d_gemm_fp8": (c_i, [c_p, c_l, c_p, c_p, c_l, c_p, c_p, c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_i, c_i, c_p]),
    "v3d_rmsnorm": (c_i, [c_p, c_l, c_p, c_p, c_l, c_l, c_i, c_f, c_i, c_p]),
    "v3d_layernorm": (c_i, [c_p, c_l, c_p, c_p, c_p, c_l, c_l, c_i, c_f, c_i, c_p]),
    "v3d_rope_table_build": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_p]),
    "v3d_rope_apply": (c_i, [c_p, c_l, c_l, c_i, c_i, c_p, c_p, c_i, c_p, c_i, c_i, c_p]),
    "v3d_attention": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_l, c_l, c_l, c_l, c_l, c_l,
                            c_l, c_i, c_i, c_i, c_i, c_i, c_f, c_p]),
    "v3d_attention_shared_prefix": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_l, c_l, c_l, c_l, c_l, c_l, c_l,
                                          c_i, c_i, c_i, c_i, c_f, c_p]),
    "v3d_attention_decode_workspace_bytes": (c_l, [c_i, c_i]),
    "v3d_attention_decode": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_l, c_l, c_i, c_i, c_i, c_f, c_p, c_l, c_p]),
    "v3d_linear_decode": (c_i, [c_p, c_p, c_f, c_p, c_l, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "v3d_rmsnorm_quantize_fp8": (c_i, [c_p, c_l, c_p, c_f, c_l, c_i, c_i, c_p, c_l, c_p, c_p]),
    "v3d_linear_decode_fp8_rows": (c_i, [c_p, c_l, c_i, c_p, c_l, c_p, c_p, c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_i, c_p]),
    "v3d_linear_decode_rows_fuses_norm": (c_i, [c_i, c_i, c_i, c_i]),
    "v3d_linear_decode_rows": (c_i, [c_p, c_l, c_i, c_p, c_f, c_p, c_l, c_p, c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_i, c_p]),
    "v3d_rope_kv_append": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_p, c_i, c_p]),
    "v3d_preprocess_rgb_u8": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, ctypes.c_double, c_p, c_i, c_p]),
    "v3d_resize_bicubic_u8": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p,
                                    ctypes.c_double, c_p, c_i, c_p]),
    "v3d_argmax": (c_i, [c_p, c_i, c_i, c_p, c_p, c_p]),
    "v3d_argmax_rows": (c_i, [c_p, c_l, c_i, c_i, c_i, c_p, c_p, c_p]),
    "v3d_eos_update": (c_i, [c_p, c_i, c_p, c_i, c_p, c_p, c_p]),
    "v3d_rope_kv_append_rows": (c_i, [c_p, c_l, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_p, c_p, c_i, c_p]),
    "v3d_attention_decode_rows": (c_i, [c_p, c_l, c_i, c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_i, c_l, c_l, c_i, c_i, c_i, c_f,
                                        c_p, c_l, c_p]),
    "v3d_attention_decode_rows_prefix": (c_i, [c_p, c_l, c_i, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_i, c_l, c_l, c_i, c_i,
                                               c_i, c_f, c_p, c_l, c_p]),
    "v3d_object_patch_mask": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_p]),
    "v3d_masked_mean": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_i, c_p]),
    "v3d_ground_scores": (c_i, [c_p, c_l, c_i, c_p, c_i, c_p, c_i, c_p]),
    "v3d_row_dots": (c_i, [c_p, c_l, c_i, c_p, c_i, c_p, c_i, c_p, c_i, c_p]),
    "v3d_relu_mul_rows": (c_i, [c_p, c_l, c_i, c_i, c_p, c_i, c_i, c_p]),
    "v3d_rope_kv_store": (c_i, [c_p, c_l, c_l, c_i, c_i, c_i, c_p, c_p, c_i, c_p, c_i, c_p, c_l, c_p, c_l, c_i, c_p]),
    "v3d_add_row": (c_i, [c_p, c_l, c_p, c_i, c_i, c_p, c_i, c_p]),
    "v3d_copy_rows_bcast": (c_i, [c_p, c_l, c_p, c_l, c_l, c_i, c_i, c_l, c_i, c_p]),
    "v3d_copy_rows": (c_i, [c_p, c_l, c_p, c_l, c_l, c_i, c_i, c_p]),
    "v3d_patchify": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "v3d_cross_entropy": (c_i, [c_p, c_l, c_i, c_l, c_i, c_p, c_l, c_p, c_p, c_p, c_p]),
    "v3d_cross_entropy_grad": (c_i, [c_p, c_l, c_i, c_l, c_i, c_p, c_l, c_p, c_p, c_f, c_p, c_l, c_i, c_p]),
    "v3d_visual_tokens_grad": (c_i, [c_p, c_l, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "v3d_transpose": (c_i, [c_p, c_l, c_l, c_i, c_p, c_l, c_l, c_i, c_p]),
    "v3d_colsum_workspace_bytes": (c_l, [c_l, c_i]),
    "v3d_colsum": (c_i, [c_p, c_l, c_l, c_i, c_i, c_p, c_p, c_i, c_p]),
    "v3d_rmsnorm_grad": (c_i, [c_p, c_l, c_p, c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_p, c_i, c_l, c_i, c_f, c_i, c_p]),
    "v3d_swiglu": (c_i, [c_p, c_l, c_p, c_l, c_l, c_i, c_i, c_p]),
    "v3d_swiglu_grad": (c_i, [c_p, c_l, c_p, c_l, c_p, c_l, c_l, c_i, c_i, c_p]),
    "v3d_causal_softmax_rows": (c_i, [c_p, c_l, c_p, c_l, c_l, c_i, c_i, c_i, c_f, c_i, c_p]),
    "v3d_softmax_grad_rows": (c_i, [c_p, c_l, c_p, c_l, c_p, c_l, c_l, c_i, c_f, c_i, c_p]),
    "v3d_attention_train": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_l, c_l, c_l, c_l, c_l, c_l, c_l, c_i, c_i, c_i,
                                  c_i, c_i, c_f, c_p]),
    "v3d_attention_backward_workspace_bytes": (c_l, [c_i, c_i, c_i]),
    "v3d_attention_backward": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_l, c_l, c_l, c_l, c_l, c_l, c_l, c_l,
                                     c_l, c_l, c_l, c_l, c_l, c_l, c_l, c_l, c_i, c_f, c_p, c_l, c_p]),
    "v3d_adamw_step": (c_i, [c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_l, c_f, c_f, c_f, c_f, c_f, c_i, c_f, c_p]),
    "v3d_embed_grad": (c_i, [c_p, c_l, c_l, c_p, c_p, c_i, c_i, c_p, c_l, c_l, c_i, c_p]),
    "v3d_gelu": (c_i, [c_p, c_l, c_p, c_l, c_l, c_i, c_i, c_i, c_p]),
    "v3d_gelu_grad": (c_i, [c_p, c_l, c_p, c_l, c_p, c_l, c_l, c_i, c_i, c_i, c_p]),
    "v3d_layernorm_grad": (c_i, [c_p, c_l, c_p, c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_p, c_p, c_i, c_l, c_i, c_f, c_i, c_p]),
    "v3d_axpy": (c_i, [c_p, c_p, c_f, c_l, c_i, c_p]),
    "v3d_sumsq_workspace_bytes": (c_l, []),
    "v3d_sumsq": (c_i, [c_p, c_l, c_i, c_p, c_i, c_p, c_p]),
    "v3d_ground_infonce": (c_i, [c_p, c_l, c_i, c_p, c_i, c_p, c_f, c_p, c_p, c_p, c_l, c_p, c_i, c_p]),
    "v3d_masked_mean_grad": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_i, c_p, c_i, c_p]),
    "v3d_uniform_frame_indices_host": (c_i, [c_i, c_i, c_p]),
    "v3d_gemm_plan_host": (c_i, [c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p]),
    "v3d_voxel_keys_f32": (c_i, [c_p, c_l, c_f, c_p, c_p]),
    "v3d_greedy_cover_workspace_bytes": (c_l, [c_i, c_l]),
    "v3d_greedy_cover": (c_i, [c_p, c_i, c_l, c_p, c_l, c_i, c_p, c_p, c_p, c_p, c_p, c_l, c_p]),
    "v3d_greedy_cover_host": (c_i, [c_p, c_i, c_l, c_p, c_l, c_i, c_p, c_p, c_p, c_p]),
}


def _declare(l):
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(l, name)
        fn.restype = res
        fn.argtypes = args


def check(rc, what):
    if rc < 0:
        msg = lib().v3d_last_error().decode("utf-8", "replace")
        raise V3DError(f"{what} failed ({rc}): {msg}")
    return rc
