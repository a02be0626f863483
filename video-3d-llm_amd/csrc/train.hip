// First kernels of the training step (BASELINE configs[4]): the loss of Qwen2ForCausalLM.forward and the
// backward of the parameter-free fused visual-token kernel.  HBM-bound streaming; no MFMA.
//   v3d_cross_entropy         llava/model/language_model/qwen2/modeling_qwen2.py:1195-1205 (shifted CrossEntropyLoss, ignore_index)
//   v3d_cross_entropy_grad    its gradient with respect to the logits (softmax - onehot, scaled by 1 / #valid rows)
//   v3d_visual_tokens_grad    gradient of get_2dPool (bilinear 27 -> 14) + "+ PE" (no parameters: passes through) + image_newline
//                             insertion (llava_arch.py:191-210, 506-517, 307-328) with respect to the projector features and
//                             to image_newline
#include "v3d_common.h"

namespace v3d {

__device__ __forceinline__ float wave_max_f(float x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x = fmaxf(x, __shfl_xor(x, off));
  return x;
}
__device__ __forceinline__ float wave_sum_f(float x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
  return x;
}

template <typename T> __device__ __forceinline__ float ld1(const T* p) { return to_f32(*p); }

// One workgroup per (shifted) row r: label = labels[r + 1]; lse = log sum exp(logits[r, :]) in f32 (two passes: max, then
// sum of exp(x - max)), loss = lse - logits[r, label].  Rows whose label is ignore_index write loss 0 and lse anyway.
template <typename T>
__global__ __launch_bounds__(256) void cross_entropy_kernel(const T* __restrict__ logits, int64_t ld, int vocab,
                                                            const int64_t* __restrict__ labels, int64_t ignore_index,
                                                            float* __restrict__ loss_rows, float* __restrict__ lse_rows) {
  const int64_t r = blockIdx.x;
  const T* row = logits + r * ld;
  __shared__ float sm[4];
  float m = -INFINITY;
  for (int i = threadIdx.x; i < vocab; i += 256) m = fmaxf(m, ld1(row + i));
  m = wave_max_f(m);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
  __syncthreads();
  float s = 0.f;
  for (int i = threadIdx.x; i < vocab; i += 256) s += expf(ld1(row + i) - m);
  s = wave_sum_f(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float lse = m + logf((sm[0] + sm[1]) + (sm[2] + sm[3]));
    const int64_t lab = labels[r + 1];
    lse_rows[r] = lse;
    loss_rows[r] = (lab == ignore_index || lab < 0 || lab >= vocab) ? 0.f : lse - ld1(row + lab);
  }
}

// mean over the valid rows (CrossEntropyLoss reduction='mean'): one workgroup, fixed summation order (reproducible)
__global__ __launch_bounds__(256) void ce_mean_kernel(const float* __restrict__ loss_rows, const int64_t* __restrict__ labels, int64_t rows,
                                                      int64_t ignore_index, int vocab, float* __restrict__ out2) {
  __shared__ float ss[256];
  __shared__ int cc[256];
  float s = 0.f;
  int c = 0;
  for (int64_t r = threadIdx.x; r < rows; r += 256) {
    const int64_t lab = labels[r + 1];
    if (!(lab == ignore_index || lab < 0 || lab >= vocab)) { s += loss_rows[r]; ++c; }
  }
  ss[threadIdx.x] = s; cc[threadIdx.x] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f; int n = 0;
    for (int i = 0; i < 256; ++i) { t += ss[i]; n += cc[i]; }
    out2[0] = n > 0 ? t / (float)n : NAN;       // torch returns nan for an all-ignored batch
    out2[1] = (float)n;
  }
}

// dlogits[r, :] = (softmax(logits[r, :]) - onehot(label)) * scale for valid rows, 0 for ignored ones; the row that has no label
// (the last position: nothing to predict) is zero as well.  scale = upstream gradient / #valid rows.
template <typename T, typename TG>
__global__ __launch_bounds__(256) void cross_entropy_grad_kernel(const T* __restrict__ logits, int64_t ld, int vocab,
                                                                 const int64_t* __restrict__ labels, int64_t ignore_index, int64_t rows,
                                                                 const float* __restrict__ lse_rows, const float* __restrict__ mean_count,
                                                                 float upstream, TG* __restrict__ dlogits, int64_t ldg) {
  const int64_t r = blockIdx.x;
  TG* g = dlogits + r * ldg;
  const int64_t lab = r < rows ? labels[r + 1] : ignore_index;
  const bool valid = !(lab == ignore_index || lab < 0 || lab >= vocab);
  if (!valid) {
    for (int i = threadIdx.x; i < vocab; i += 256) g[i] = from_f32<TG>(0.f);
    return;
  }
  const float scale = upstream / mean_count[1];
  const float lse = lse_rows[r];
  const T* row = logits + r * ld;
  for (int i = threadIdx.x; i < vocab; i += 256) {
    const float p = expf(ld1(row + i) - lse);
    g[i] = from_f32<TG>((p - (i == lab ? 1.f : 0.f)) * scale);
  }
}

struct Taps2 { int i0, i1; float l0, l1; };
__device__ __forceinline__ Taps2 tap_of(int o, int n_in, int n_out) {       // the forward kernel's bilinear_tap
  const float scale = __fdiv_rn((float)n_in, (float)n_out);
  float src = fmaf(scale, (float)o + 0.5f, -0.5f);
  src = src < 0.0f ? 0.0f : src;
  Taps2 t;
  t.i0 = (int)src;
  t.i1 = t.i0 + (t.i0 < n_in - 1 ? 1 : 0);
  t.l1 = __fsub_rn(src, (float)t.i0);
  t.l0 = __fsub_rn(1.0f, t.l1);
  return t;
}

// Gather form of upsample_bilinear2d_backward: one workgroup per INPUT token (frame v, row y, column x) sums, in f32 and in a
// fixed order, weight_h * weight_w * dout[row of output (oh, ow)] over the <= 3 x 3 pooled outputs whose taps touch it.
template <typename T>
__global__ __launch_bounds__(256) void visual_tokens_grad_kernel(const T* __restrict__ dout, int64_t dstride, T* __restrict__ dfeat,
                                                                 int side, int n, int C, int cols) {
  const int tok = blockIdx.x;
  const int v = tok / (side * side);
  const int rem = tok - v * side * side;
  const int y = rem / side, x = rem - y * side;
  int oh[4], ow[4];
  float wh[4], ww[4];
  int nh = 0, nw = 0;
  for (int o = 0; o < n; ++o) {
    const Taps2 t = tap_of(o, side, n);
    float wy = 0.f, wx = 0.f;
    if (t.i0 == y) wy += t.l0;
    if (t.i1 == y) wy += t.l1;
    if (t.i0 == x) wx += t.l0;
    if (t.i1 == x) wx += t.l1;
    if ((t.i0 == y || t.i1 == y) && nh < 4) { oh[nh] = o; wh[nh++] = wy; }
    if ((t.i0 == x || t.i1 == x) && nw < 4) { ow[nw] = o; ww[nw++] = wx; }
  }
  constexpr int VEC = 16 / sizeof(T);
  uint4* dst = reinterpret_cast<uint4*>(dfeat + (size_t)tok * C);
  for (int k = threadIdx.x; k < C / VEC; k += 256) {
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    for (int a = 0; a < nh; ++a)
      for (int b = 0; b < nw; ++b) {
        const float w = wh[a] * ww[b];
        const uint4 g = reinterpret_cast<const uint4*>(dout + ((size_t)(v * n + oh[a]) * cols + ow[b]) * dstride)[k];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = fmaf(w, vec_get<T>(g, j), acc[j]);
      }
    dst[k] = vec_pack<T>(acc);
  }
}

// d image_newline[c] = sum over the V * n newline rows of dout[row, c] (f32, rows in order)
template <typename T>
__global__ __launch_bounds__(256) void newline_grad_kernel(const T* __restrict__ dout, int64_t dstride, float* __restrict__ dnl, int V,
                                                           int n, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int r = 0; r < V * n; ++r) s += to_f32(dout[((size_t)r * (n + 1) + n) * dstride + c]);
  dnl[c] = s;
}

}  // namespace v3d

using namespace v3d;

extern "C" int v3d_cross_entropy(const void* logits, int64_t ld, int dtype, int64_t positions, int vocab, const int64_t* labels,
                                 int64_t ignore_index, float* loss_rows, float* lse_rows, float* mean_count, void* stream) {
  V3D_REQUIRE(logits && labels && loss_rows && lse_rows && mean_count, "v3d_cross_entropy: null pointer");
  V3D_REQUIRE(positions >= 2 && vocab > 0 && ld >= vocab, "v3d_cross_entropy: needs >= 2 positions (position t predicts label t + 1)");
  const int64_t rows = positions - 1;
  V3D_REQUIRE(rows < (1ll << 31), "v3d_cross_entropy: too many rows");
  hipStream_t st = (hipStream_t)stream;
  V3D_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(cross_entropy_kernel<T>, dim3((unsigned)rows), dim3(256), 0, st, (const T*)logits, ld, vocab,
                                               labels, ignore_index, loss_rows, lse_rows));
  if (int e = check_launch("v3d_cross_entropy")) return e;
  hipLaunchKernelGGL(ce_mean_kernel, dim3(1), dim3(256), 0, st, (const float*)loss_rows, labels, rows, ignore_index, vocab, mean_count);
  return check_launch("v3d_cross_entropy");
}

extern "C" int v3d_cross_entropy_grad(const void* logits, int64_t ld, int dtype, int64_t positions, int vocab, const int64_t* labels,
                                      int64_t ignore_index, const float* lse_rows, const float* mean_count, float upstream,
                                      void* dlogits, int64_t ldg, int grad_dtype, void* stream) {
  V3D_REQUIRE(logits && labels && lse_rows && mean_count && dlogits, "v3d_cross_entropy_grad: null pointer");
  V3D_REQUIRE(positions >= 2 && vocab > 0 && ld >= vocab && ldg >= vocab, "v3d_cross_entropy_grad: bad shape");
  V3D_REQUIRE(grad_dtype == dtype || grad_dtype == V3D_F32, "v3d_cross_entropy_grad: the gradient is written in the logits' dtype or in f32");
  hipStream_t st = (hipStream_t)stream;
  const int64_t rows = positions - 1;
#define V3D_CEG(TG) hipLaunchKernelGGL((cross_entropy_grad_kernel<T, TG>), dim3((unsigned)positions), dim3(256), 0, st, (const T*)logits, ld, vocab, \
                                       labels, ignore_index, rows, lse_rows, mean_count, upstream, (TG*)dlogits, ldg)
  V3D_DISPATCH_DTYPE(dtype, { if (grad_dtype == V3D_F32) { V3D_CEG(float); } else { V3D_CEG(T); } });
#undef V3D_CEG
  return check_launch("v3d_cross_entropy_grad");
}

extern "C" int v3d_visual_tokens_grad(const void* dout, int64_t dout_stride, void* dfeat, float* dnewline, int dtype, int V, int side,
                                      int n, int C, int flags, void* stream) {
  V3D_REQUIRE(dout && dfeat, "v3d_visual_tokens_grad: null pointer");
  V3D_REQUIRE(dtype == V3D_F16 || dtype == V3D_BF16, "v3d_visual_tokens_grad: 16-bit dtypes (the training dtype is bf16)");
  V3D_REQUIRE((flags & V3D_VT_POOL) && V > 0 && side > 0 && n > 0 && n <= side && C > 0 && C % 8 == 0 && dout_stride % 8 == 0 && dout_stride >= C,
              "v3d_visual_tokens_grad: bad shape (POOL form only)");
  V3D_REQUIRE(aligned16(dout) && aligned16(dfeat), "v3d_visual_tokens_grad: pointers must be 16-byte aligned");
  V3D_REQUIRE(!(flags & V3D_VT_NEWLINE) || dnewline, "v3d_visual_tokens_grad: NEWLINE needs dnewline");
  const int cols = (flags & V3D_VT_NEWLINE) ? n + 1 : n;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == V3D_BF16) hipLaunchKernelGGL(visual_tokens_grad_kernel<bf16_t>, dim3(V * side * side), dim3(256), 0, st, (const bf16_t*)dout, dout_stride, (bf16_t*)dfeat, side, n, C, cols);
  else hipLaunchKernelGGL(visual_tokens_grad_kernel<f16_t>, dim3(V * side * side), dim3(256), 0, st, (const f16_t*)dout, dout_stride, (f16_t*)dfeat, side, n, C, cols);
  if (int e = check_launch("v3d_visual_tokens_grad")) return e;
  if (flags & V3D_VT_NEWLINE) {
    if (dtype == V3D_BF16) hipLaunchKernelGGL(newline_grad_kernel<bf16_t>, dim3((C + 255) / 256), dim3(256), 0, st, (const bf16_t*)dout, dout_stride, dnewline, V, n, C);
    else hipLaunchKernelGGL(newline_grad_kernel<f16_t>, dim3((C + 255) / 256), dim3(256), 0, st, (const f16_t*)dout, dout_stride, dnewline, V, n, C);
  }
  return check_launch("v3d_visual_tokens_grad");
}
