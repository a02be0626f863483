// Object-proposal path of ScanRefer / Multi3DRefer (K19, K20): per-proposal ViT-patch masks from the world
// coordinates, masked mean of the projector rows, cosine scores of the infonce grounding head.
// All small and HBM/latency-bound; no MFMA.
#include "v3d_common.h"

namespace v3d {

// llava_arch.py:357-372 ('patch14'): cell (py,px) of frame f is selected for box o iff at least `thresh` of its
// cell x cell pixels satisfy lo <= xyz <= hi in all three axes (comparisons in the tensor dtype: the bounds
// lo = c - s/2, hi = c + s/2 are rounded to T like the reference's tensor arithmetic).
// ('patch27', :367-371: 27 x 27-pixel cells, the grid of the POOLED tokens, thresh = int(27*27*0.25).)
// One wave per cell; pixels over lanes (ITERS passes of 64: 4 for 196 pixels, 12 for 729), boxes looped, ballot + popcount.
template <typename T, int ITERS>
__global__ __launch_bounds__(256) void object_patch_mask_kernel(const T* __restrict__ coords, int F_, int S, int grid, int cell,
                                                                const T* __restrict__ boxes, int n_obj, int thresh,
                                                                uint8_t* __restrict__ mask) {
  const int lane = threadIdx.x & 63;
  const int cell_id = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int n_cells = F_ * grid * grid;
  if (cell_id >= n_cells) return;
  const int f = cell_id / (grid * grid), rem = cell_id - f * grid * grid, py = rem / grid, px = rem - py * grid;
  const int npix = cell * cell;
  float x[ITERS][3];
  bool valid[ITERS];
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int idx = it * 64 + lane;
    valid[it] = idx < npix;
    const int r = valid[it] ? idx / cell : 0, c = valid[it] ? idx - (idx / cell) * cell : 0;
    const T* p = coords + (((size_t)f * S + (py * cell + r)) * S + (px * cell + c)) * 3;
    x[it][0] = to_f32(p[0]); x[it][1] = to_f32(p[1]); x[it][2] = to_f32(p[2]);
  }
  for (int o = 0; o < n_obj; ++o) {
    float lo[3], hi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float c = to_f32(boxes[o * 6 + a]), h = round_to<T>(to_f32(boxes[o * 6 + 3 + a]) * 0.5f);
      lo[a] = round_to<T>(c - h);
      hi[a] = round_to<T>(c + h);
    }
    int cnt = 0;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const bool in = valid[it] && lo[0] <= x[it][0] && x[it][0] <= hi[0] && lo[1] <= x[it][1] && x[it][1] <= hi[1] &&
                      lo[2] <= x[it][2] && x[it][2] <= hi[2];
      cnt += __popcll(__ballot(in));
    }
    if (lane == 0) mask[(size_t)o * n_cells + cell_id] = cnt >= thresh ? 1 : 0;
  }
}

// llava_arch.py:482-501: mean over the selected rows of feat [T_, C] (f32 accumulation, one rounding), zeros if
// none, then `+ add[o]` (the box-centre PE) in the model dtype.  Block = (object, 2048-channel slab).
template <typename T>
__global__ __launch_bounds__(256) void masked_mean_kernel(const T* __restrict__ feat, const uint8_t* __restrict__ mask,
                                                          int T_, int C, const T* __restrict__ add, T* __restrict__ out) {
  const int o = blockIdx.x;
  const int k = blockIdx.y * 256 + threadIdx.x;      // 16-byte vector index within the row
  if (k * 8 >= C) return;
  const uint8_t* m = mask + (size_t)o * T_;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int cnt = 0;
  for (int t = 0; t < T_; ++t) {
    if (m[t]) {                                       // block-uniform
      const uint4 v = reinterpret_cast<const uint4*>(feat + (size_t)t * C)[k];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += vec_get<T>(v, j);
      ++cnt;
    }
  }
  float y[8];
  const float inv = cnt > 0 ? 1.0f / (float)cnt : 0.f;
  uint4 a4 = make_uint4(0, 0, 0, 0);
  if (add) a4 = reinterpret_cast<const uint4*>(add + (size_t)o * C)[k];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float mean = round_to<T>(cnt > 0 ? acc[j] / (float)cnt : 0.f);
    y[j] = add ? mean + vec_get<T>(a4, j) : mean;
  }
  (void)inv;
  reinterpret_cast<uint4*>(out + (size_t)o * C)[k] = vec_pack<T>(y);
}

// llava_qwen.py:298-300: scores[i] = <obj[i]/max(|obj[i]|,eps), q/max(|q|,eps)>, eps = 1e-12 (F.normalize).
// The reference normalises in the model dtype (each normalised vector rounded to T) then multiplies and sums.
template <typename T>
__global__ __launch_bounds__(256) void ground_scores_kernel(const T* __restrict__ obj, int64_t ldo, const T* __restrict__ q, int C,
                                                            T* __restrict__ scores) {
  __shared__ float red[3][4];
  const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T* row = obj + (size_t)i * ldo;
  float so = 0.f, sq = 0.f;
  for (int c = tid; c < C; c += 256) {
    const float a = to_f32(row[c]), b = to_f32(q[c]);
    so = fmaf(a, a, so); sq = fmaf(b, b, sq);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { so += __shfl_xor(so, off); sq += __shfl_xor(sq, off); }
  if (lane == 0) { red[0][wave] = so; red[1][wave] = sq; }
  __syncthreads();
  const float no = fmaxf(round_to<T>(sqrtf(red[0][0] + red[0][1] + red[0][2] + red[0][3])), 1e-12f);
  const float nq = fmaxf(round_to<T>(sqrtf(red[1][0] + red[1][1] + red[1][2] + red[1][3])), 1e-12f);
  float dot = 0.f;
  for (int c = tid; c < C; c += 256) dot += round_to<T>(round_to<T>(to_f32(row[c]) / no) * round_to<T>(to_f32(q[c]) / nq));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off);
  if (lane == 0) red[2][wave] = dot;
  __syncthreads();
  if (tid == 0) scores[i] = from_f32<T>(red[2][0] + red[2][1] + red[2][2] + red[2][3]);
}


// The 'mlp' and 'score' grounding heads (llava_qwen.py:59-86, 283-293) beside the shipped 'infonce' one - small row work:
//   row_dots:       out[i] = sum_c x[i, c] * q[c] (+ bias).  PRODUCTS_ROUNDED: `(ground_hidden * object_features).sum(-1)` of the 'mlp'
//                   head (an elementwise product in the tensor dtype, then a sum); otherwise a Linear(C, 1) (f32 products).
//   relu_mul_rows:  x[i, c] = relu?(x[i, c]) * (row ? row[c] : 1), in place (nn.ReLU; `obj_feat * query_feat`).
template <typename T, bool PRODUCTS_ROUNDED>
__global__ __launch_bounds__(256) void row_dots_kernel(const T* __restrict__ x, int64_t ldx, const T* __restrict__ q, int C,
                                                       const T* __restrict__ bias, T* __restrict__ out) {
  __shared__ float red[4];
  const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T* row = x + (size_t)i * ldx;
  float dot = 0.f;
  for (int c = tid; c < C; c += 256) {
    const float pr = to_f32(row[c]) * to_f32(q[c]);
    dot += PRODUCTS_ROUNDED ? round_to<T>(pr) : pr;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off);
  if (lane == 0) red[wave] = dot;
  __syncthreads();
  if (tid == 0) out[i] = from_f32<T>(red[0] + red[1] + red[2] + red[3] + (bias ? to_f32(bias[0]) : 0.f));
}

template <typename T>
__global__ __launch_bounds__(256) void relu_mul_rows_kernel(T* __restrict__ x, int64_t ldx, int cols, const T* __restrict__ row, int relu) {
  T* r = x + (size_t)blockIdx.x * ldx;
  for (int c = threadIdx.x; c < cols; c += 256) {
    float v = to_f32(r[c]);
    if (relu) v = fmaxf(v, 0.f);
    if (row) v = v * to_f32(row[c]);
    r[c] = from_f32<T>(v);
  }
}

}  // namespace v3d

using namespace v3d;

extern "C" int v3d_object_patch_mask(const void* coords, int dtype, int F_, int S, int cell, const void* boxes, int n_obj,
                                     int thresh, uint8_t* mask, void* stream) {
  V3D_REQUIRE(F_ > 0 && S > 0 && cell > 0 && cell * cell <= 768 && n_obj >= 0, "v3d_object_patch_mask: bad shape");
  if (n_obj == 0) return V3D_OK;                        // no proposals: nothing to write (boxes / mask may be empty, i.e. null)
  V3D_REQUIRE(coords && boxes && mask, "v3d_object_patch_mask: null pointer");
  const int grid = (S - 6) / cell;                       // [:378,:378] of 384 -> 27 cells of 14 (llava_arch.py:366)
  const int n_cells = F_ * grid * grid;
  if (cell * cell <= 256) {
    V3D_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((object_patch_mask_kernel<T, 4>), dim3((n_cells + 3) / 4), dim3(256), 0,
                                                 (hipStream_t)stream, (const T*)coords, F_, S, grid, cell, (const T*)boxes, n_obj,
                                                 thresh, mask));
  } else {
    V3D_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((object_patch_mask_kernel<T, 12>), dim3((n_cells + 3) / 4), dim3(256), 0,
                                                 (hipStream_t)stream, (const T*)coords, F_, S, grid, cell, (const T*)boxes, n_obj,
                                                 thresh, mask));
  }
  return check_launch("v3d_object_patch_mask");
}

extern "C" int v3d_masked_mean(const void* feat, const uint8_t* mask, int n_obj, int T_, int C, const void* add, void* out,
                               int dtype, void* stream) {
  V3D_REQUIRE(feat && mask && out && n_obj >= 0 && T_ > 0 && C > 0 && C % 8 == 0, "v3d_masked_mean: bad arguments");
  V3D_REQUIRE(dtype == V3D_F16 || dtype == V3D_BF16, "v3d_masked_mean: dtype must be f16 or bf16");
  V3D_REQUIRE(aligned16(feat) && aligned16(out) && (!add || aligned16(add)), "v3d_masked_mean: alignment");
  if (n_obj == 0) return V3D_OK;
  const dim3 grid(n_obj, (C / 8 + 255) / 256);
  if (dtype == V3D_BF16) hipLaunchKernelGGL(masked_mean_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)feat, mask, T_, C, (const bf16_t*)add, (bf16_t*)out);
  else hipLaunchKernelGGL(masked_mean_kernel<f16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const f16_t*)feat, mask, T_, C, (const f16_t*)add, (f16_t*)out);
  return check_launch("v3d_masked_mean");
}

extern "C" int v3d_ground_scores(const void* obj, int64_t ldo, int n_rows, const void* query, int C, void* scores, int dtype,
                                 void* stream) {
  V3D_REQUIRE(obj && query && scores && n_rows > 0 && C > 0 && ldo >= C, "v3d_ground_scores: bad arguments");
  if (dtype == V3D_BF16) hipLaunchKernelGGL(ground_scores_kernel<bf16_t>, dim3(n_rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)obj, ldo, (const bf16_t*)query, C, (bf16_t*)scores);
  else if (dtype == V3D_F16) hipLaunchKernelGGL(ground_scores_kernel<f16_t>, dim3(n_rows), dim3(256), 0, (hipStream_t)stream, (const f16_t*)obj, ldo, (const f16_t*)query, C, (f16_t*)scores);
  else { set_error("v3d_ground_scores: dtype must be f16 or bf16"); return V3D_E_INVALID; }
  return check_launch("v3d_ground_scores");
}

extern "C" int v3d_row_dots(const void* x, int64_t ldx, int n_rows, const void* q, int C, const void* bias, int products_rounded,
                            void* out, int dtype, void* stream) {
  V3D_REQUIRE(x && q && out && n_rows > 0 && C > 0 && ldx >= C, "v3d_row_dots: bad arguments");
  V3D_REQUIRE(dtype == V3D_F16 || dtype == V3D_BF16, "v3d_row_dots: dtype must be f16 or bf16");
#define V3D_RD(T, R) hipLaunchKernelGGL((row_dots_kernel<T, R>), dim3(n_rows), dim3(256), 0, (hipStream_t)stream, (const T*)x, ldx, (const T*)q, C, (const T*)bias, (T*)out)
  if (dtype == V3D_BF16) { if (products_rounded) V3D_RD(bf16_t, true); else V3D_RD(bf16_t, false); }
  else { if (products_rounded) V3D_RD(f16_t, true); else V3D_RD(f16_t, false); }
#undef V3D_RD
  return check_launch("v3d_row_dots");
}

extern "C" int v3d_relu_mul_rows(void* x, int64_t ldx, int n_rows, int cols, const void* row, int relu, int dtype, void* stream) {
  V3D_REQUIRE(x && n_rows > 0 && cols > 0 && ldx >= cols, "v3d_relu_mul_rows: bad arguments");
  V3D_REQUIRE(dtype == V3D_F16 || dtype == V3D_BF16, "v3d_relu_mul_rows: dtype must be f16 or bf16");
  if (dtype == V3D_BF16) hipLaunchKernelGGL(relu_mul_rows_kernel<bf16_t>, dim3(n_rows), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, ldx, cols, (const bf16_t*)row, relu);
  else hipLaunchKernelGGL(relu_mul_rows_kernel<f16_t>, dim3(n_rows), dim3(256), 0, (hipStream_t)stream, (f16_t*)x, ldx, cols, (const f16_t*)row, relu);
  return check_launch("v3d_relu_mul_rows");
}
