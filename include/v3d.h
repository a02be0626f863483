/*
 * v3d.h - C ABI of libv3d_hip.so: the MI355X (gfx950) hot path of Video-3D-LLM's
 * position-aware video -> LLM forward path.
 *
 * The reference (zd11024/Video-3D-LLM) is pure Python: it has no FFI for this path.  The
 * boundary is therefore the set of PyTorch calls its hot path makes; each entry point below
 * names the reference site (file:line under the reference checkout) whose arithmetic it
 * replaces.  The host-side mirror (video-3d-llm_amd/llava/...) binds these with ctypes; see
 * INTEGRATION.md for the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all entry points are
 *     asynchronous with respect to the host, allocate nothing, keep no pointer after return
 *     and are re-entrant - with one exception: v3d_gemm keeps one 64 MiB device workspace per stream for the
 *     split-K tail of its 256-wide kernel (accumulator images of cut tiles), allocated by the first launch that
 *     takes that path and never during stream capture (V3D_GEMM_STREAMK=0 switches the path, and the allocation, off);
 *   - return value: 0 = ok, negative = error (V3D_E_*); the message for the calling thread is
 *     returned by v3d_last_error();
 *   - dtype codes: V3D_F32 / V3D_F16 / V3D_BF16.  16-bit tensors are raw IEEE half / bfloat16.
 *   - tensors are dense row-major unless a row stride (in elements) is passed explicitly.
 */
#ifndef V3D_H_
#define V3D_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define V3D_ABI_VERSION 7   /* 2: *_rows decode entries, fp8 path, a7 / a3 device entries; 3: rope_kv_store, add_row, ...; 4: resize_bicubic, eos_update, embed_grad bounds; 6: sumsq, *_rows entries take 1..32 rows; 7: attention_shared_prefix */

enum { V3D_F32 = 0, V3D_F16 = 1, V3D_BF16 = 2 };
enum { V3D_U8_HWC = 16 };   /* v3d_resize_bicubic_u8 only: 8-bit interleaved output, no normalisation */

enum {
  V3D_OK = 0,
  V3D_E_INVALID = -1,   /* bad argument (null pointer, shape or alignment the kernel cannot take) */
  V3D_E_LAUNCH = -2,    /* HIP reported a launch/runtime error */
  V3D_E_UNSUPPORTED = -3
};

int v3d_abi_version(void);
const char* v3d_last_error(void);

/* ------------------------------------------------------------------ geometry (K1-K4) ---- */

/* K1  llava/video_utils.py:38-68 `unproject`.
 * depth_mm [V,H,W] f32 millimetres, intrinsics/poses [V,4,4] f32 -> world [V,H,W,3] f32. */
int v3d_unproject_f32(const float* depth_mm, const float* intrinsics, const float* poses,
                      float* world, int V, int H, int W, void* stream);

/* K1+K2  `unproject` evaluated only at the pixels that survive the nearest-neighbour resize
 * (cv2.resize INTER_NEAREST to (int(W*crop/H), crop)) and centre crop of
 * llava/video_utils.py:296-308.  depth [V,H,W] uint16 PNG values (the reference converts
 * u16 -> int32 -> f32, video_utils.py:217,230).  out [V,crop,crop,3] in out_dtype
 * (f32 = the tensor VideoProcessor returns; f16/bf16 = after the eval driver's
 * `.half()` / model-dtype cast, llava/eval/model_scanqa.py:163-165). */
int v3d_unproject_sampled_u16(const uint16_t* depth, const float* intrinsics, const float* poses,
                              void* out, int out_dtype, int V, int H, int W, int crop, void* stream);

/* VideoProcessor.preprocess, strategy "resize" (video_utils.py:293-296): the back-projection at the pixels cv2.resize(coords, (size, size),
 * INTER_NEAREST) keeps, both axes scaled independently, no crop: out [V,size,size,3].  (Index rule restated as for the centre-crop form:
 * parity unpinned, cv2 is absent.) */
int v3d_unproject_resized_u16(const uint16_t* depth, const float* intrinsics, const float* poses, void* out, int out_dtype, int V, int H,
                              int W, int size, void* stream);

/* calculate_world_coords(do_normalize=True) (video_utils.py:232-236; the "norm" frame-sampling strategies): xyz [n_points, 3] clamped
 * in place to [lo, hi] per axis (host float[3] each). */
int v3d_clamp_xyz(void* xyz, int64_t n_points, const float* lo_host, const float* hi_host, int dtype, void* stream);

/* llava/video_utils.py:268-273 `boundry`: [x_min, x_max, y_min, y_max, z_min, z_max] of `unproject` over ALL V*H*W pixels (taken
 * before the resize and crop), computed without materialising the full-resolution tensor: same per-pixel arithmetic as
 * v3d_unproject_f32 on the u16 depth, reduced on the fly.  bounds: 6 floats (device).  workspace: device scratch of at
 * least v3d_unproject_bounds_workspace_bytes(V) bytes. */
int64_t v3d_unproject_bounds_workspace_bytes(int V);
int v3d_unproject_bounds_u16(const uint16_t* depth, const float* intrinsics, const float* poses, int V, int H, int W,
                             float* bounds, void* workspace, int64_t workspace_bytes, void* stream);

/* K3+K4  llava/model/llava_arch.py:213-223 `average_coordinate_in_patch` and :259-272
 * `discrete_coords`.  coords [V,S,S,3] (dtype), the last S - n*patch rows/cols are dropped
 * (n = (S-6)/patch when S==384, patch==27 -> n = 14).  The 27x27 mean is accumulated in f32
 * in the reference's own (row, column) order so that voxel ids are bit-exact.
 * Outputs (any may be NULL): avg [V,n,n,3] dtype; vox [V,n,n,3] dtype (integer-valued floats,
 * what the reference returns); ids [V,n,n,3] int32.
 * min_xyz_host/max_xyz_host: 3 floats each on the HOST (config.min_xyz_range / max_xyz_range). */
int v3d_coord_pool_voxel(const void* coords, int dtype, int V, int S, int patch,
                         const float* min_xyz_host, const float* max_xyz_host, float voxel_size,
                         void* avg, void* vox, int32_t* ids, void* stream);

/* K4 alone on arbitrary points: xyz [N,3] dtype -> vox [N,3] dtype and/or ids [N,3] int32
 * (box centres / box_input path, llava_arch.py:416-420). */
int v3d_discrete_coords(const void* xyz, int dtype, int64_t N, const float* min_xyz_host,
                        const float* max_xyz_host, float voxel_size, void* vox, int32_t* ids,
                        void* stream);

/* ------------------------------------------------------------------ 3-D sinusoid (K5) --- */

/* Number of elements of one row of the shifted 16-byte-aligned PE table for `embedding_size`
 * channels of `dtype` (see v3d_sin3d_table_build). */
int64_t v3d_sin3d_table_row_elems(int embedding_size, int dtype);

/* llava/model/position_encoding.py:17-49 evaluated for every integer coordinate 0..n_ids-1.
 * dim_t [num_feats = embedding_size/3] f32 is supplied by the caller (the reference computes
 * `temperature ** (2*(i//2)/num_feats)` with torch's pow, :24-25; pow is not bit-reproducible
 * across libraries so it stays on the caller's side of the boundary).
 * table [3][n_ids][row_elems] dtype: axis a's copy is shifted by (a*num_feats) % (16/sizeof)
 * elements so that the fused kernel reads whole 16-byte vectors; padding is zero.
 * table_f32 (nullable) [n_ids][num_feats] f32: the unshifted f32 values. */
int v3d_sin3d_table_build(const float* dim_t, int embedding_size, int n_ids, int dtype,
                          void* table, float* table_f32, void* stream);

/* PositionEmbeddingSine3D.forward for arbitrary (also non-integer) coordinates:
 * xyz [N,3] dtype -> out [N,embedding_size] dtype; f32 math, output cast to dtype (:46-47). */
int v3d_sin3d_pe(const void* xyz, int dtype, int64_t N, const float* dim_t, int embedding_size,
                 void* out, void* stream);

/* ------------------------------------------------------------------ fusion (K5-K8) ------ */

/* The north-star kernel.  One pass over the projector output:
 *   K7  get_2dPool bilinear side x side -> n x n            llava_arch.py:191-210   (flag POOL)
 *   K5+K6  + PE(voxel ids) from the table                    llava_arch.py:506-517   (flag PE)
 *   K8  add_token_per_grid: `newline` after each row of n    llava_arch.py:307-328   (flag NEWLINE)
 * feat  [V, side*side, C] if POOL else [V, n*n, C]           (dtype)
 * ids   [V, n*n, 3] int32 voxel ids (PE only)
 * table from v3d_sin3d_table_build for (C, dtype), n_ids rows per axis (PE only)
 * newline [C] dtype (NEWLINE only)
 * out   rows of C elements, row stride out_stride elements: V*n*(n+1) rows if NEWLINE else V*n*n.
 * Rounding follows the reference: pooled value rounded to dtype, then the add rounded to dtype. */
enum { V3D_VT_POOL = 1, V3D_VT_PE = 2, V3D_VT_NEWLINE = 4 };
int v3d_visual_tokens(const void* feat, const int32_t* ids, const void* table, int n_ids,
                      const void* newline, void* out, int64_t out_stride, int dtype, int V, int side,
                      int n, int C, int flags, void* stream);

/* K9  embed_tokens gather for the text segments (llava_arch.py:693): out[i,:] = table[ids[i],:]. */
int v3d_embed_gather(const void* table, int64_t vocab, int C, const int64_t* ids, int64_t n,
                     void* out, int64_t out_stride, int dtype, void* stream);

/* ------------------------------------------------------------------ dense linears ------- */

/* K10-K12, K15, K17  every nn.Linear on the path (SigLIP q/k/v/out/fc1/fc2 and patch embedding,
 * siglip_encoder.py:156-174,197-265; mlp2x_gelu projector, multimodal_projector/builder.py:41-48;
 * Qwen2 q/k/v/o and gate/up/down, modeling_qwen2.py:177-189,237-240,322):
 *     out[M, N'] = epilogue( A[M,K] . W[N,K]^T ),   16-bit operands, f32 MFMA accumulation.
 * A row stride lda, W row stride ldw (elements).  N % 128 == 0 and K % 64 == 0 (pad weights with
 * zero rows / columns once at load); M is arbitrary (M <= 8 takes a weight-streaming path for decode).
 * Epilogues (rounding points as torch: linear output rounded to dtype, then each further op):
 *   NONE            out = acc
 *   BIAS            out = acc + bias[n]
 *   BIAS_GELU_ERF   out = gelu(acc + bias)            nn.GELU()            (projector)
 *   BIAS_GELU_TANH  out = gelu_tanh(acc + bias)       gelu_pytorch_tanh    (SigLIP fc1)
 *   BIAS_RES        out = res[m or m % res_mod, n] + (acc + bias)          (SigLIP out_proj/fc2, patch-embed + pos-emb)
 *   RES             out = res[m, n] + acc                                  (Qwen2 o_proj / down_proj)
 *   SWIGLU          out[m, j] = silu(gate_j) * up_j, N' = N/2; W rows are tile-interleaved: rows
 *                   [128t, 128t+64) = gate rows [64t, 64t+64), rows [128t+64, 128t+128) = the up rows.
 * Summation order: every kernel sums k in ascending order in one run per output - except tiles the split-K tail cuts
 * (M x N x K shapes whose 256 x 256 tiles leave the last round of the chip under half full, e.g. 6794 x 3584 x 18944),
 * which are the f32 sum of 2..4 runs: such an output may differ from the one-run result by one rounding of the
 * 16-bit value.  v3d_gemm_plan_host reports the plan; V3D_GEMM_STREAMK=0 keeps every output on one run. */
enum {
  V3D_EPI_NONE = 0, V3D_EPI_BIAS = 1, V3D_EPI_BIAS_GELU_ERF = 2, V3D_EPI_BIAS_GELU_TANH = 3,
  V3D_EPI_BIAS_RES = 4, V3D_EPI_RES = 5, V3D_EPI_SWIGLU = 6,
  V3D_EPI_BIAS_RELU = 7   /* out = relu(acc + bias)   grounding head, llava_qwen.py:93-104 */
};
int v3d_gemm(const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias, const void* res,
             int64_t ldr, int res_mod, void* out, int64_t ldo, int M, int N, int K, int dtype, int epilogue,
             void* stream);

/* FP8 (OCP e4m3) variant of the LLM linears - BASELINE configs[3]; not part of the reference (tolerance is
 * re-stated in tests/test_gpu_fp8.py).  y[m,n] = sa[m] * sw[n] * sum_k qa[m,k] qw[n,k] with per-row scales.
 * v3d_quantize_fp8_rows: x [rows, cols] (f16/bf16) -> q [rows, cols] e4m3 bytes (row stride ldq) + scale[rows] f32,
 *   scale = amax|row| / 448.  Used for activations per call and for weights once at load.
 * v3d_gemm_fp8: A [M,K] e4m3 (lda bytes), W [N,K] e4m3; N % 256 == 0, K % 128 == 0; epilogues NONE / BIAS / RES /
 *   SWIGLU (same codes and row interleave as v3d_gemm); output in out_dtype (f16 / bf16). */
int v3d_quantize_fp8_rows(const void* x, int64_t ldx, int64_t rows, int cols, int dtype, void* q, int64_t ldq,
                          float* scale, void* stream);
int v3d_gemm_fp8(const void* A, int64_t lda, const float* scale_a, const void* W, int64_t ldw, const float* scale_w,
                 const void* bias, const void* res, int64_t ldr, void* out, int64_t ldo, int M, int N, int K,
                 int out_dtype, int epilogue, void* stream);
/* v3d_rmsnorm fused with v3d_quantize_fp8_rows of its output (the 16-bit normalised row is formed and rounded exactly as
 * v3d_rmsnorm does, but never stored): q [rows, cols] e4m3 + scale[rows].  Bit-identical to the two calls in sequence
 * (rows > 4; v3d_rmsnorm's decode-order kernel for <= 4 rows sums in another order).  cols <= 4096. */
int v3d_rmsnorm_quantize_fp8(const void* x, int64_t ldx, const void* weight, float eps, int64_t rows, int cols, int dtype,
                             void* q, int64_t ldq, float* scale, void* stream);

/* ------------------------------------------------------------------ norms / rotary ------ */

/* K13  Qwen2RMSNorm, modeling_qwen2.py:85-90: out = weight * T(x * rsqrt(mean(x^2) + eps)), f32 inside. */
int v3d_rmsnorm(const void* x, int64_t ldx, const void* weight, void* out, int64_t ldo, int64_t rows,
                int cols, float eps, int dtype, void* stream);

/* nn.LayerNorm of the SigLIP encoder layers, siglip_encoder.py:272-274,292,300 (f32 statistics). */
int v3d_layernorm(const void* x, int64_t ldx, const void* weight, const void* bias, void* out,
                  int64_t ldo, int64_t rows, int cols, float eps, int dtype, void* stream);

/* K14  Qwen2RotaryEmbedding.forward, modeling_qwen2.py:106-129, for positions 0..n_pos-1:
 * cos/sin tables [n_pos, head_dim/2] in dtype.  inv_freq [head_dim/2] f32 comes from the caller
 * (the reference evaluates 1/(base**(arange/dim)) with torch on the host, :100). */
int v3d_rope_table_build(const float* inv_freq, int head_dim, int n_pos, int dtype, void* cos_table,
                         void* sin_table, void* stream);

/* apply_rotary_pos_emb, modeling_qwen2.py:141-173, in place on n_heads heads of head_dim laid out
 * contiguously from x in every token row (row stride ldx).  Token t uses positions[t] if positions
 * is non-NULL, else pos0 + t.  (The three M-RoPE position rows are identical on this path, :1003.) */
int v3d_rope_apply(void* x, int64_t ldx, int64_t tokens, int n_heads, int head_dim,
                   const void* cos_table, const void* sin_table, int n_pos, const int32_t* positions,
                   int pos0, int dtype, void* stream);

/* Prefill form of K14 + the KV-cache append in one pass (modeling_qwen2.py:141-173, then DynamicCache.update as called at
 * :283-285): rotary in place on the n_q query heads of every QKV row; the rotated key heads and the value heads of token t
 * go to cache + r*ldc, r = dst_rows[t] (device int64 array) or row0 + t, laid out [k heads | v heads].  The k/v columns of
 * the QKV buffer itself are left unrotated.  Token t uses positions[t] (device int32) or pos0 + t.  dst_rows lets several
 * questions of one scene, prefilled together, append to their own caches inside one allocation. */
int v3d_rope_kv_store(void* qkv, int64_t ldx, int64_t tokens, int n_q_heads, int n_kv_heads, int head_dim,
                      const void* cos_table, const void* sin_table, int n_pos, const int32_t* positions, int pos0,
                      void* cache, int64_t ldc, const int64_t* dst_rows, int64_t row0, int dtype, void* stream);

/* llava_arch.py:697-700: x[rows[i], 0:C] += add[0:C] (the box-centre PE on the <coord> token rows of inputs_embeds);
 * rows: device int64 [n_rows]; 16-bit dtypes; the sum is rounded once to dtype. */
int v3d_add_row(void* x, int64_t ldx, const int64_t* rows, int n_rows, int C, const void* add, int dtype, void* stream);

/* ------------------------------------------------------------------ attention ----------- */

/* K16 / K11  softmax(Q K^T * scale [+ causal mask]) V without materialising the scores.
 * Prefill: Qwen2 causal GQA (spec: eager Qwen2Attention, modeling_qwen2.py:289-311; runtime:
 * flash-attn-2 :567-574) with D = 128; SigLIP attention (siglip_encoder.py:213-239) non-causal with
 * D = 96 = the kernel's tile width for head dim 72: d_out = 72 is the number of valid dims - columns >= d_out of q are
 * treated as zero and only d_out columns of o are written, so heads may be packed at their true stride (hsq = hsk = 72:
 * the 24 extra columns a tile row covers then belong to the next head or to row padding; they must be readable).
 * Element (b, token s, head h, dim d) of q is at q + b*bsq + s*ldq + h*hsq + d (likewise k, v with
 * bsk/ldk|ldv/hsk and o with bso/ldo/hso); kv head = h / (Hq/Hkv).  With causal != 0 query i sits at
 * key position q_pos0 + i (q_pos0 = number of cached tokens).  Sq <= 8 with B == 1 takes the
 * cache-streaming decode path. */
int v3d_attention(const void* q, const void* k, const void* v, void* o, int dtype, int B, int Sq, int Sk,
                  int Hq, int Hkv, int D, int d_out, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                  int64_t bsq, int64_t bsk, int64_t bso, int hsq, int hsk, int hso, int causal, int q_pos0,
                  float scale, void* stream);

/* r04 (scene reuse): v3d_attention (causal, D = 128) for the question rows of B questions about ONE prefilled scene -
 * the reference recomputes the scene for every question (model_scanqa.py:130-185; the shared prompt: :46-60).  Key tiles below
 * shared_len (a multiple of 64, <= q_pos0) are read from the scene's cache k_shared / v_shared (ldk / ldv / hsk as k / v, no batch
 * stride); keys from shared_len on from k / v + b*bsk (each question's own cache, rows at their absolute positions: rows below
 * shared_len of those caches are never touched).  A workgroup's query slots hold the rows of all query heads of a kv head.  Every
 * output bit equals v3d_attention's on B full copies. */
int v3d_attention_shared_prefix(const void* q, const void* k, const void* v, const void* k_shared, const void* v_shared,
                                int shared_len, void* o, int dtype, int B, int Sq, int Sk, int Hq, int Hkv, int64_t ldq,
                                int64_t ldk, int64_t ldv, int64_t ldo, int64_t bsq, int64_t bsk, int64_t bso, int hsq, int hsk,
                                int hso, int q_pos0, float scale, void* stream);

/* Decode step of K16 (one new query row against the K/V cache, modeling_qwen2.py:282-311 with a
 * DynamicCache): split-KV so the cache is streamed once by the whole chip.  q [Hq*128] (head stride hsq),
 * caches [Sk, ...] with row strides ldk/ldv and kv-head stride hsk, o [Hq*128] (head stride hso).
 * workspace: device scratch of at least v3d_attention_decode_workspace_bytes(Hq, 1024/Hkv) bytes. */
int64_t v3d_attention_decode_workspace_bytes(int Hq, int max_splits);
int v3d_attention_decode(const void* q, const void* k_cache, const void* v_cache, void* o, int dtype, int Sk,
                         int Hq, int Hkv, int64_t ldk, int64_t ldv, int hsq, int hsk, int hso, float scale,
                         void* workspace, int64_t workspace_bytes, void* stream);
/* The same for M = 1..32 (r04; was 16) scenes decoding together in ONE launch pair (the single-scene kernels are latency-bound, not
 * bandwidth-bound): query rows q + m*q_stride, outputs o + m*o_stride (elements), per-scene cache pointers and lengths
 * in HOST arrays of M entries; workspace >= M * v3d_attention_decode_workspace_bytes(Hq, 1024/Hkv).  Scene m's output
 * is bit-identical to v3d_attention_decode on it alone, whatever the other scenes' lengths: every scene partitions its keys
 * by its own length (256 keys per split), the launch only adds empty splits for the shorter ones. */
int v3d_attention_decode_rows(const void* q, int64_t q_stride, int M, const void* const* k_caches,
                              const void* const* v_caches, const int* Sk, void* o, int64_t o_stride, int dtype, int Hq,
                              int Hkv, int64_t ldk, int64_t ldv, int hsq, int hsk, int hso, float scale, void* workspace,
                              int64_t workspace_bytes, void* stream);

/* Scene-level reuse: the M rows are questions about ONE scene whose caches all start with the same prefix_len rows
 * (the scene's [system | user | <image>] K/V, broadcast into each question's cache).  Keys < prefix_len are read from k_prefix /
 * v_prefix (same strides as the caches; any one of the copies) for every question, so the chip streams the prefix once per step
 * instead of M times; rows >= prefix_len come from the question's own cache.  Outputs are bit-identical to
 * v3d_attention_decode_rows on the same caches. */
int v3d_attention_decode_rows_prefix(const void* q, int64_t q_stride, int M, const void* k_prefix, const void* v_prefix,
                                     int prefix_len, const void* const* k_caches, const void* const* v_caches, const int* Sk,
                                     void* o, int64_t o_stride, int dtype, int Hq, int Hkv, int64_t ldk, int64_t ldv, int hsq,
                                     int hsk, int hso, float scale, void* workspace, int64_t workspace_bytes, void* stream);

/* Decode-step linear (one activation row; K15/K17/K18 at q_len == 1): y = epilogue(W . f(x)), W [N,K] row
 * stride ldw.  norm_weight != NULL fuses Qwen2RMSNorm (modeling_qwen2.py:85-90) in front: f(x) = w * T(x*rstd).
 * epilogue: 0 NONE, 1 BIAS (y += bias), 2 RES (y = res + T(y)), 3 SWIGLU (tile-interleaved gate|up rows as
 * v3d_gemm's SWIGLU; N' = N/2).  Weights are streamed once with non-temporal loads. */
enum { V3D_DEC_NONE = 0, V3D_DEC_BIAS = 1, V3D_DEC_RES = 2, V3D_DEC_SWIGLU = 3 };
int v3d_linear_decode(const void* x, const void* norm_weight, float eps, const void* W, int64_t ldw,
                      const void* bias, const void* res, void* out, int N, int K, int dtype, int epilogue,
                      void* stream);
/* The same linear for M activation rows at once (M scenes decoding together share one pass over the weights; the
 * step is HBM-bound on them).  x [M, K] row stride ldx, res [M, N] row stride ldr, out [M, N'] row stride ldo.
 * M = 1 is v3d_linear_decode.  M = 2..32 (r04; two 16-row blocks per weight fragment above 16; no fused norm, K % 128 == 0, N % 16 == 0): the weights feed the matrix cores
 * (v_mfma_f32_16x16x32, activation rows as the B operand), so the cost does not grow with M and a row's result does not
 * depend on the other rows or on M; against the one-row form it differs by the f32 summation order only.  Other shapes
 * (fused norm, odd sizes): VALU form, M = 1..4, each row bit-identical to v3d_linear_decode on it. */
int v3d_linear_decode_rows(const void* x, int64_t ldx, int M, const void* norm_weight, float eps, const void* W,
                           int64_t ldw, const void* bias, const void* res, int64_t ldr, void* out, int64_t ldo, int N,
                           int K, int dtype, int epilogue, void* stream);

/* r04: 1 when v3d_linear_decode_rows with a norm_weight and M rows takes the matrix-core form that normalises the rows itself (more than
 * four rows, K <= 4096, 9..32 K tiles of 128, no residual epilogue; the 1 / rms of a row is formed in v3d_rmsnorm's summation order for
 * that many rows, so the result equals v3d_rmsnorm followed by the unfused call bit for bit), else 0: such a call is then refused for
 * M > 4 and the caller normalises with v3d_rmsnorm first.  Host-only. */
int v3d_linear_decode_rows_fuses_norm(int M, int N, int K, int epilogue);

/* The decode linear over OCP e4m3 weights (BASELINE configs[3]; W8A16): W8 [N,K] bytes (row stride ldw) with one
 * f32 scale per output row as v3d_quantize_fp8_rows writes them; y[m,n] = scale_w[n] * sum_k W8[n,k] x[m,k], then
 * the epilogue of v3d_linear_decode.  K % 16 == 0.  M = 1: VALU form; M = 2..32 with K % 256 == 0, N % 16 == 0: matrix-core
 * form (weights widened to the activation type in registers; a row's result depends neither on the other rows nor on
 * M); other shapes: VALU form, M <= 4.  Not a reference code path (tolerance: tests/test_gpu_fp8.py). */
int v3d_linear_decode_fp8_rows(const void* x, int64_t ldx, int M, const void* W8, int64_t ldw, const float* scale_w,
                               const void* bias, const void* res, int64_t ldr, void* out, int64_t ldo, int N, int K,
                               int dtype, int epilogue, void* stream);

/* Decode-step K14 + cache append in one launch: rotary at position `pos` on the n_q + n_kv heads at the start
 * of the QKV row (in place), rotated k and v copied to cache_row = [k heads | v heads]. */
int v3d_rope_kv_append(void* qkv_row, int n_q_heads, int n_kv_heads, int head_dim, const void* cos_table,
                       const void* sin_table, int n_pos, int pos, void* cache_row, int dtype, void* stream);
/* M = 1..32 scenes in one launch: QKV rows qkv + m*qkv_stride, positions pos[m] and cache rows cache_rows[m] (HOST arrays). */
int v3d_rope_kv_append_rows(void* qkv, int64_t qkv_stride, int M, int n_q_heads, int n_kv_heads, int head_dim,
                            const void* cos_table, const void* sin_table, int n_pos, const int* pos,
                            void* const* cache_rows, int dtype, void* stream);

/* Greedy token choice: index of the maximum of x[0..n) (lowest index on ties, as torch.argmax).
 * workspace: 1 KiB of device scratch. */
int v3d_argmax(const void* x, int n, int dtype, int64_t* out_index, void* workspace, void* stream);
/* M rows x + m*ldx in one launch pair; out_index[M]; workspace >= M KiB. */
int v3d_argmax_rows(const void* x, int64_t ldx, int M, int n, int dtype, int64_t* out_index, void* workspace, void* stream);

/* The stop test of greedy decoding on the device (generate(..., eos_token_id), llava_qwen.py:208-236 -> HF greedy search):
 * done[m] |= tokens[m] in eos_ids[0:n_eos] (device int64 arrays; done: device int32 [M], kept across steps), *n_done = number
 * of rows that have finished.  The host polls n_done asynchronously instead of synchronising on every token. */
int v3d_eos_update(const int64_t* tokens, int M, const int64_t* eos_ids, int n_eos, int32_t* done, int32_t* n_done, void* stream);

/* ------------------------------------------------------------------ grounding (K19, K20) - */

/* K19  llava_arch.py:357-372, object_feature_type 'patch14': mask[o, f, py, px] = 1 iff at least `thresh`
 * (= int(14*14*0.5)) pixels of the cell x cell ViT patch lie inside box o = (centre xyz, size xyz);
 * 'patch27' (:367-371): cell = 27 (the pooled 14 x 14 token grid), thresh = int(27*27*0.25); cell * cell <= 768.
 * coords [F,S,S,3] dtype (first S-6 rows/cols used), boxes [n,6] dtype -> mask uint8 [n, F*g*g], g = (S-6)/cell. */
int v3d_object_patch_mask(const void* coords, int dtype, int F, int S, int cell, const void* boxes, int n_obj,
                          int thresh, uint8_t* mask, void* stream);

/* The 'mlp' and 'score' grounding heads (llava_qwen.py:59-86, 283-293; the shipped checkpoints use 'infonce', K20).
 * v3d_row_dots: out[i] = sum_c x[i,c] * q[c] (+ bias[0]); products_rounded != 0: every product is rounded to the dtype first
 * (`(ground_hidden * object_features).sum(dim=-1)`, :285), 0: a Linear(C, 1) row (:81-84).
 * v3d_relu_mul_rows: x[i,c] = relu?(x[i,c]) * (row ? row[c] : 1) in place (nn.ReLU; `obj_feat * query_feat`, :290). */
int v3d_row_dots(const void* x, int64_t ldx, int n_rows, const void* q, int C, const void* bias, int products_rounded,
                 void* out, int dtype, void* stream);
int v3d_relu_mul_rows(void* x, int64_t ldx, int n_rows, int cols, const void* row, int relu, int dtype, void* stream);

/* llava_arch.py:482-501: out[o] = mean of the rows t of feat [T,C] with mask[o,t] != 0 (zeros if none)
 * (+ add[o] if add != NULL: the box-centre PE). */
int v3d_masked_mean(const void* feat, const uint8_t* mask, int n_obj, int T, int C, const void* add, void* out,
                    int dtype, void* stream);

/* K20  predict_box 'infonce', llava_qwen.py:298-300: scores[i] = <normalize(obj[i]), normalize(query)>. */
int v3d_ground_scores(const void* obj, int64_t ldo, int n_rows, const void* query, int C, void* scores, int dtype,
                      void* stream);

/* ------------------------------------------------------------------ data movement ------- */

/* out[r, 0:cols] = in[r, 0:cols] for strided rows (KV-cache append). */
int v3d_copy_rows(const void* in, int64_t ldi, void* out, int64_t ldo, int64_t rows, int cols, int dtype,
                  void* stream);

/* The same rows to n_copies destinations: out[c*copy_stride + r*ldo + 0:cols] = in[r*ldi + 0:cols], c < n_copies (the cached
 * K/V prefix of a scene handed to the caches of the questions that are answered together; not a reference code path - the
 * reference recomputes the prefix per question, llava/eval/model_scanqa.py:130-185). */
int v3d_copy_rows_bcast(const void* in, int64_t ldi, void* out, int64_t ldo, int64_t rows, int cols, int n_copies,
                        int64_t copy_stride, int dtype, void* stream);

/* a7: SigLipImageProcessor.preprocess, siglip_encoder.py:47-67, for frames that already have the tower's size (the
 * 384 x 384 crops of VideoProcessor.preprocess, video_utils.py:292-308, for which its bicubic resize is the identity):
 * frames [F,H,W,3] u8 (device) -> out [F,3,H,W];  v = f32(f64(u8) * rescale);  out = T((v - mean[c]) / std[c]) in f32. */
int v3d_preprocess_rgb_u8(const uint8_t* frames, int F, int H, int W, const float* mean_host, const float* std_host,
                          double rescale, void* out, int dtype, void* stream);

/* a6 / a7, RGB half: `frame.resize((new_w, crop))` + centre crop of VideoProcessor.preprocess (video_utils.py:285-306), i.e.
 * Pillow's Image.resize for 8-bit RGB with its default BICUBIC filter, reproduced bit for bit (libImaging Resample.c: horizontal
 * pass, 8-bit intermediate, vertical pass; 22-bit fixed-point coefficients; clip8), optionally fused with v3d_preprocess_rgb_u8
 * (SigLipImageProcessor.preprocess, siglip_encoder.py:47-67).
 * frames [F,H,W,3] u8 (device).  The resized image would be OH x OW; only the window (crop_top, crop_left, crop_h, crop_w) of it is
 * computed.  bounds_* [n][2] int32 = (first source index, tap count) and coeffs_* [n][ksize_*] int32 per output column (h, n = OW)
 * / row (v, n = OH): DEVICE arrays holding what Pillow's precompute_coeffs + normalize_coeffs_8bpc produce for (W -> OW) and
 * (H -> OH) (v3d.ops.pil_resample_tables; the double arithmetic stays on the caller's side like dim_t / inv_freq do).
 * out_dtype V3D_U8_HWC: out [F,crop_h,crop_w,3] u8 (mean / std may be null);  V3D_F32 / F16 / BF16: out [F,3,crop_h,crop_w] =
 * T((f32(f64(u8) * rescale) - mean[c]) / std[c]).  At most 32 taps per pass (reductions up to ~7x). */
int v3d_resize_bicubic_u8(const uint8_t* frames, int F, int H, int W, int OH, int OW, const int32_t* bounds_h,
                          const int32_t* coeffs_h, int ksize_h, const int32_t* bounds_v, const int32_t* coeffs_v, int ksize_v,
                          int crop_top, int crop_left, int crop_h, int crop_w, const float* mean_host, const float* std_host,
                          double rescale, void* out, int out_dtype, void* stream);

/* K10 input: SigLipVisionEmbeddings' Conv2d(kernel = stride = patch), siglip_encoder.py:156-172, as a
 * GEMM: gathers images [B,3,S,S] into rows [B*(S/patch)^2, kpad], columns (c, ky, kx) zero padded. */
int v3d_patchify(const void* images, void* out, int B, int S, int patch, int kpad, int dtype, void* stream);

/* ------------------------------------------------------------------ training step (configs[4], first kernels) --- */

/* The loss of Qwen2ForCausalLM.forward, llava/model/language_model/qwen2/modeling_qwen2.py:1195-1205: logits [positions, vocab]
 * (dtype: f32 as the reference's `.float()`, or 16-bit; row stride ld), labels [positions] int64 (device); position t predicts
 * label t + 1 (the shift), rows whose label is ignore_index (-100) do not count.  Outputs (device, f32): loss_rows[positions-1]
 * (0 for ignored rows), lse_rows[positions-1] (log-sum-exp per row, reused by the gradient), mean_count[2] = {mean loss over the
 * valid rows (CrossEntropyLoss reduction 'mean'; NaN if none), number of valid rows}. */
int v3d_cross_entropy(const void* logits, int64_t ld, int dtype, int64_t positions, int vocab, const int64_t* labels,
                      int64_t ignore_index, float* loss_rows, float* lse_rows, float* mean_count, void* stream);
/* d loss / d logits: dlogits[t, :] = (softmax(logits[t, :]) - onehot(labels[t + 1])) * upstream / #valid, zero rows for ignored
 * labels and for the last position; written in grad_dtype (the logits' dtype or f32), row stride ldg. */
int v3d_cross_entropy_grad(const void* logits, int64_t ld, int dtype, int64_t positions, int vocab, const int64_t* labels,
                           int64_t ignore_index, const float* lse_rows, const float* mean_count, float upstream, void* dlogits,
                           int64_t ldg, int grad_dtype, void* stream);

/* Backward of v3d_visual_tokens (flags POOL [| PE] [| NEWLINE]) with respect to its parameter-free inputs: the PE add passes the
 * gradient through (llava_arch.py:515: coords are detached), get_2dPool's bilinear 27 -> 14 (llava_arch.py:191-210) scatters
 * it back onto the projector features with the forward's own tap weights (gather form, f32 accumulation in a fixed order), and
 * image_newline (llava_arch.py:307-328) collects the sum of its V * n rows.
 * dout rows as the forward wrote them (V*n*(n+1) with NEWLINE else V*n*n; row stride dout_stride), dfeat [V, side*side, C] dtype,
 * dnewline [C] f32 (NEWLINE only). */
int v3d_visual_tokens_grad(const void* dout, int64_t dout_stride, void* dfeat, float* dnewline, int dtype, int V, int side, int n,
                           int C, int flags, void* stream);

/* Backward of the decoder's dense blocks (r02).  With y = x . W^T (nn.Linear), dx = dy . W and dW = dy^T . x are v3d_gemm calls on
 * transposed operands (v3d/train.py: linear_backward), so the data movement the backward adds is this transpose:
 * out[c, r] = x[r, c] for x [rows, cols] (16-bit; cols % 8 == 0); columns [rows, out_cols) of out are zero-filled (out_cols % 8 == 0:
 * the k padding of the dW product, whose k runs over token rows). */
int v3d_transpose(const void* x, int64_t ldx, int64_t rows, int cols, void* out, int64_t ldo, int64_t out_cols, int dtype,
                  void* stream);
/* Column sums in f32 with a fixed summation order (32-row partials in `workspace`, then the partials in order): the bias gradient
 * of nn.Linear (sum over token rows of dy).  workspace: v3d_colsum_workspace_bytes(rows, cols) bytes; out [cols] in out_dtype. */
int64_t v3d_colsum_workspace_bytes(int64_t rows, int cols);
int v3d_colsum(const void* x, int64_t ldx, int64_t rows, int cols, int dtype, float* workspace, void* out, int out_dtype, void* stream);
/* Backward of v3d_rmsnorm (Qwen2RMSNorm, modeling_qwen2.py:76-90): with r = rsqrt(mean(x^2) + eps), n = T(x r), g = dy * weight:
 * dx = r (g - n mean(g n)), dweight = sum over rows of dy n (f32, fixed order; workspace as v3d_colsum's).  `add` (may be null):
 * a 16-bit gradient of the same shape added to dx - the residual branch that by-passes the norm (modeling_qwen2.py:771-789). */
int v3d_rmsnorm_grad(const void* x, int64_t ldx, const void* weight, const void* dy, int64_t ldy, const void* add, int64_t lda,
                     void* dx, int64_t ldd, float* workspace, void* dweight, int dw_dtype, int64_t rows, int cols, float eps, int dtype,
                     void* stream);
/* Qwen2MLP's act_fn(gate_proj(x)) * up_proj(x) (modeling_qwen2.py:177-189) on planar rows gu = [gate (inter) | up (inter)]
 * (the training forward keeps gate / up for the backward; the inference path forms the product in v3d_gemm's SWIGLU epilogue):
 * out = T(T(silu(g)) * u); the gradient dgu = [dh u silu'(g) | dh silu(g)]. */
int v3d_swiglu(const void* gu, int64_t ld, void* out, int64_t ldo, int64_t rows, int inter, int dtype, void* stream);
int v3d_swiglu_grad(const void* gu, int64_t ld, const void* dh, int64_t ldh, void* dgu, int64_t ldg, int64_t rows, int inter,
                    int dtype, void* stream);

/* Attention backward, first form: one head's probabilities are materialised (16-bit [queries, keys], the rounding points of the
 * reference's eager attention, modeling_qwen2.py:248-327) and the five products run on v3d_gemm (v3d/train.py: attention_backward);
 * these are the two row passes between them.  cols = the padded key count (% 8 == 0, <= 8192).
 * v3d_causal_softmax_rows: p[i, j] = softmax over the keys j <= i + offset, j < n_keys of T(s[i, j] * scale); 0 for the other j < cols.
 * v3d_softmax_grad_rows:   ds[i, j] = T(p (dp - sum_j p dp)) * scale (rounded to T). */
int v3d_causal_softmax_rows(const void* s, int64_t lds, void* p, int64_t ldp, int64_t rows, int n_keys, int cols, int offset,
                            float scale, int dtype, void* stream);
int v3d_softmax_grad_rows(const void* p, int64_t ldp, const void* dp, int64_t ldd, void* ds, int64_t lds, int64_t rows, int cols,
                          float scale, int dtype, void* stream);

/* Attention backward as tiled kernels (the [S, S] matrices never reach HBM): v3d_attention_train is v3d_attention at head dim 128
 * (outputs bit-identical) that also writes lse [B, Hq, Sq] f32, the row log-sum-exp in the kernel's scaled log2 units;
 * v3d_attention_backward (Sq = Sk = S, heads 128 columns apart inside a token row, B sequences with element strides bs*; causal =
 * the decoder, modeling_qwen2.py:248-482, non-causal = the SigLIP encoder with its 72-wide heads zero-padded to 128,
 * siglip_encoder.py:197-250) recomputes the probabilities from q, k and lse and writes dq [S, Hq 128], dk, dv [S, Hkv 128] per sequence
 * (the sum over a kv group's query heads is taken in f32, in a fixed order: no atomics).  q / k are what the forward multiplied (the
 * ROTATED projections in the decoder), o the forward's output, dout its gradient.  workspace:
 * v3d_attention_backward_workspace_bytes(B, S, Hq) bytes. */
int v3d_attention_train(const void* q, const void* k, const void* v, void* o, float* lse, int dtype, int B, int Sq, int Sk, int Hq,
                        int Hkv, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t bsq, int64_t bsk, int64_t bso, int hsq,
                        int hsk, int hso, int causal, int q_pos0, float scale, void* stream);
int64_t v3d_attention_backward_workspace_bytes(int B, int S, int Hq);
int v3d_attention_backward(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, void* dq,
                           void* dk, void* dv, int dtype, int B, int S, int Hq, int Hkv, int64_t ldq, int64_t ldk, int64_t ldv,
                           int64_t ldo, int64_t lddo, int64_t lddq, int64_t lddk, int64_t lddv, int64_t bsq, int64_t bsk, int64_t bsv,
                           int64_t bso, int64_t bsdo, int64_t bsdq, int64_t bsdk, int64_t bsdv, int causal, float scale,
                           void* workspace, int64_t workspace_bytes, void* stream);
/* One AdamW update of a flat parameter tensor (torch.optim.AdamW's single-tensor form = the reference's HF Trainer optimizer; ZeRO
 * runs the same update on its f32 master partition): p32 / m / v [n] f32 in place, grad [n] in grad_dtype (f32 / f16 / bf16) times
 * grad_scale, p16 (may be null) the 16-bit copy the next forward reads.  step counts from 1 (bias correction). */
int v3d_adamw_step(float* p32, float* m, float* v, const void* grad, int grad_dtype, void* p16, int p16_dtype, int64_t n, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);
/* Gradient of the embedding lookup for the text rows of a sample (llava_arch.py:650-700 embeds them with embed_tokens):
 * dE[ids[i], :] = sum over the j with ids[j] == ids[i], in order, of dh[rows[j], :] (f32 sums, one rounding); rows / ids: device
 * int64 [n].  Rows of dE that no id names are not touched (the caller zero-fills).
 * dh has n_rows rows, dE vocab rows: ids outside [0, vocab) (a raw prompt's IMAGE_TOKEN_INDEX = -200) and rows outside
 * [0, n_rows) are skipped - nothing is read or written for them. */
int v3d_embed_grad(const void* dh, int64_t ld, int64_t n_rows, const int64_t* rows, const int64_t* ids, int n, int H, void* dE, int64_t lde,
                   int64_t vocab, int dtype, void* stream);

/* GELU as a pass of its own for the training forward (which keeps the pre-activation z) and its gradient dz = dy * gelu'(z):
 * tanh_form 0 = nn.GELU() of the mm_projector (multimodal_projector/builder.py:41-48), 1 = gelu_pytorch_tanh of the SigLIP MLP
 * (siglip_encoder.py:253-262), 2 = nn.ReLU() of the grounding heads (llava_qwen.py:99-110; z may be the ReLU's output).  f32 inside,
 * one rounding. */
int v3d_gelu(const void* z, int64_t ldz, void* out, int64_t ldo, int64_t rows, int cols, int tanh_form, int dtype, void* stream);
int v3d_gelu_grad(const void* z, int64_t ldz, const void* dy, int64_t ldy, void* dz, int64_t ldo, int64_t rows, int cols, int tanh_form,
                  int dtype, void* stream);

/* Backward of v3d_layernorm (nn.LayerNorm of the SigLIP encoder layers, siglip_encoder.py:272-274,292,300): with xh = (x - mean) rstd
 * and g = dy * weight: dx = rstd (g - mean(g) - xh mean(g xh)) (+ add, the residual branch's gradient), dweight = sum_rows dy xh,
 * dbias = sum_rows dy (f32, fixed order).  workspace: 2 * v3d_colsum_workspace_bytes(rows, cols) bytes.  cols <= 3584. */
int v3d_layernorm_grad(const void* x, int64_t ldx, const void* weight, const void* dy, int64_t ldy, const void* add, int64_t lda, void* dx,
                       int64_t ldd, float* workspace, void* dweight, void* dbias, int dw_dtype, int64_t rows, int cols, float eps,
                       int dtype, void* stream);

/* y[i] = T(y[i] + alpha * x[i]) for flat 16-bit tensors: gradient accumulation over micro-batches (train_multi.sh:31-32, 60:
 * gradient_accumulation_steps = 2), in the arithmetic torch uses when it adds a new gradient to .grad (f32 add, one rounding). */
int v3d_axpy(void* y, const void* x, float alpha, int64_t n, int dtype, void* stream);

/* Sum of squares of a flat tensor x [n] (f32 / f16 / bf16; 16-byte aligned) in f32 -> out[0] (device; accumulate != 0: added to it),
 * deterministic (fixed partials folded in a fixed order).  The global gradient norm of gradient clipping - the HF Trainer's
 * max_grad_norm = 1.0 behind scripts/zero2.json:36 "gradient_clipping": "auto" and torch.nn.utils.clip_grad_norm_ - is the square root
 * of the sum of these over the gradient tensors.  workspace: v3d_sumsq_workspace_bytes() bytes of device scratch. */
int64_t v3d_sumsq_workspace_bytes(void);
int v3d_sumsq(const void* x, int64_t n, int dtype, float* out, int accumulate, float* workspace, void* stream);

/* The grounding loss (ScanRefer / Multi3DRefer samples of the joint training), predict_box 'infonce', llava_qwen.py:296-310: with
 * s_i = <normalize(obj[i]), normalize(query)> (v3d_ground_scores) over the n head outputs obj [n, C] (the zero-target row included),
 * loss = -log(sum_{positive i} e^{s_i / temperature} / sum_i e^{s_i / temperature}); positive: device uint8 [n].  Writes loss (f32
 * scalar), scores (f32 [n], may be null) and the gradients dobj [n, C], dquery [C].  n <= 1024. */
int v3d_ground_infonce(const void* obj, int64_t ldo, int n, const void* query, int C, const uint8_t* positive, float temperature,
                       float* loss, float* scores, void* dobj, int64_t ldd, void* dquery, int dtype, void* stream);
/* Backward of v3d_masked_mean (llava_arch.py:482-501): dfeat[t, :] = (accumulate ? dfeat[t, :] : 0) + sum over the objects o with
 * mask[o, t] != 0, in order, of dobj[o, :] / count[o]; dfeat [T, C] contiguous, inv_count: f32 [n_obj] scratch. */
int v3d_masked_mean_grad(const uint8_t* mask, int n_obj, int T, int C, const void* dobj, void* dfeat, int accumulate, float* inv_count,
                         int dtype, void* stream);

/* ------------------------------------------------------------------ host helpers -------- */

/* The launch plan v3d_gemm takes for an M x N x K product on a chip with `slots` compute units (pure host code, no device needed;
 * defaults of the V3D_GEMM_* switches): *kernel 0 = skinny (M <= 8), 1 = 128 x 128 tile, 2 = 256 x 256 ping-pong, 3 = 192 x 256
 * ping-pong; *tiles of that kernel; with the split-K tail (kernel 2 only) *dp = whole-tile rounds and *split = K-chunks per left-over
 * tile (2..4), else *dp = -1, *split = 1.  Invariants a test can hold it to: (tiles - dp * (slots & ~7)) * split <= slots & ~7 (every
 * chunk has a workgroup), K / 64 even and K / 128 >= 2 * split (a chunk is at least two 2-K-step granules). */
int v3d_gemm_plan_host(int M, int N, int K, int slots, int* kernel, int* tiles, int* dp, int* split);

/* a1  llava/video_utils.py:187  np.linspace(0, total-1, n, dtype=int).  out_host[n]. */
int v3d_uniform_frame_indices_host(int total_frames, int n, int32_t* out_host);

/* a3  scripts/3d/preprocessing/max_coverage_sampling.py:44-94 greedy max-coverage with the tie
 * rule "lowest frame position" (the reference draws random.choice, unseeded, :84).
 * keys_host [n_frames, pts_per_frame, 3] int32 voxel keys (round(xyz/voxel)), scene_host [m,3]
 * int32 scene voxel set.  Writes up to max_frames picks: sel_host (frame positions), gain_host
 * (voxel_nums), and the two totals.  Returns the number of picks or a negative error. */
int v3d_greedy_cover_host(const int32_t* keys_host, int n_frames, int64_t pts_per_frame,
                          const int32_t* scene_host, int64_t m, int max_frames, int32_t* sel_host,
                          int64_t* gain_host, int64_t* num_all_host, int64_t* num_sel_host);

/* ------------------------------------------------------------------ a3 on the device ---- */

/* scripts/3d/preprocessing/max_coverage_sampling.py:44-45: keys = round(xyz / voxel_size) as int32 (f32 division,
 * round half to even).  xyz / keys: n_values scalars (3 per point), device. */
int v3d_voxel_keys_f32(const float* xyz, int64_t n_values, float voxel_size, int32_t* keys, void* stream);

/* The greedy max-coverage loop of max_coverage_sampling.py:46-94 on the device, same contract and tie rule as
 * v3d_greedy_cover_host.  keys [n_frames, pts_per_frame, 3] and scene [m,3] int32 in HBM; outputs in HBM:
 * sel[max_frames] (frame positions), gain[max_frames] (voxel_nums), totals[2] = {num_all_voxels, num_select_voxels},
 * n_sel[1] = number of picks.  Stream-ordered, no host synchronisation.  Coordinates must lie in [-2^20, 2^20): a
 * scene key outside sets the int at workspace byte offset (workspace_bytes_needed - 256) to 1 (the wrapper raises);
 * frame keys outside can never match a scene voxel and are ignored. */
int64_t v3d_greedy_cover_workspace_bytes(int n_frames, int64_t m);
int v3d_greedy_cover(const int32_t* keys, int n_frames, int64_t pts_per_frame, const int32_t* scene, int64_t m,
                     int max_frames, int32_t* sel, int64_t* gain, int64_t* totals, int32_t* n_sel, void* workspace,
                     int64_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* V3D_H_ */
