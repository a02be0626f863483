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


// ---------------------------------------------------------------------------------------------------------------------------
// Backward of the decoder's dense blocks (r02).  The matrix products of the backward pass run on v3d_gemm (out = A . W^T): with
// y = x . W^T,  dx = dy . (W^T)^T  and  dW = dy^T . (x^T)^T,  so the only new data movement is a 2-D transpose; everything else here
// is an HBM-bound row pass.
//   v3d_transpose       out[c, r] = x[r, c] (16-bit), columns [rows, out_cols) of out zero-filled (the k padding of the dW product)
//   v3d_colsum          bias gradients: column sums of dy in f32, fixed summation order (32-row partials, then the partials in order)
//   v3d_rmsnorm_grad    Qwen2RMSNorm (modeling_qwen2.py:76-90): dx (+ the residual branch's gradient) and dweight
//   v3d_swiglu / _grad  Qwen2MLP (modeling_qwen2.py:177-189): act_fn(gate) * up on a planar [gate | up] row, and its backward
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int TG_RPB = 32;         // rows per partial of the column reductions

template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ x, int64_t ldx, int64_t rows, int cols, T* __restrict__ out,
                                                        int64_t ldo, int64_t out_cols) {
  __shared__ uint16_t tile[64][64 + 2];
  const int64_t r0 = (int64_t)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + 256 * i, r = idx >> 3, ch = idx & 7;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r0 + r < rows && c0 + ch * 8 < cols) v = *reinterpret_cast<const uint4*>(x + (r0 + r) * ldx + c0 + ch * 8);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) { tile[r][ch * 8 + 2 * j] = (uint16_t)(w[j] & 0xffffu); tile[r][ch * 8 + 2 * j + 1] = (uint16_t)(w[j] >> 16); }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + 256 * i, c = idx >> 3, ch = idx & 7;          // output row c0 + c, output columns r0 + 8 ch ..
    if (c0 + c < cols && r0 + ch * 8 < out_cols) {
      uint32_t w[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) w[j] = (uint32_t)tile[ch * 8 + 2 * j][c] | ((uint32_t)tile[ch * 8 + 2 * j + 1][c] << 16);
      *reinterpret_cast<uint4*>(out + (int64_t)(c0 + c) * ldo + r0 + ch * 8) = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
}

// partial[b, c] = sum over rows [32 b, 32 b + 32) of x[r, c]   (f32, rows in order)
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, int64_t ldx, int64_t rows, int cols,
                                                             float* __restrict__ partial) {
  const int ch = blockIdx.x * 256 + threadIdx.x;
  if (ch * 8 >= cols) return;
  const int64_t rb = (int64_t)blockIdx.y * TG_RPB;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < TG_RPB && rb + i < rows; ++i) {
    const uint4 v = *reinterpret_cast<const uint4*>(x + (rb + i) * ldx + ch * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] += vec_get<T>(v, j);
  }
  float* o = partial + (int64_t)blockIdx.y * cols + ch * 8;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = s[j];
}

// out[c] = sum of the partials: thread (c, g) of a 16-column workgroup (64 groups) sums the partials b = g, g + 64, ... in order, the
// sixty-four group sums are added in order (a fixed tree: run-to-run identical).  16 columns per workgroup: cols / 16 workgroups
// (72 for the tower's 1152, 224 for 3584) instead of cols / 64 - the kernel is a latency chain per thread, so it wants the chip.
template <typename TO>
__global__ __launch_bounds__(1024) void colsum_final_kernel(const float* __restrict__ partial, int64_t n_part, int cols, TO* __restrict__ out) {
  __shared__ float red[64][16];
  const int cl = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float s = 0.f;
  if (c < cols)
    for (int64_t b = g; b < n_part; b += 64) s += partial[b * cols + c];
  red[g][cl] = s;
  __syncthreads();
  if (g == 0 && c < cols) {
    float t = red[0][cl];
#pragma unroll
    for (int i = 1; i < 64; ++i) t += red[i][cl];
    out[c] = from_f32<TO>(t);
  }
}

// y = w * T(x r), r = rsqrt(mean(x^2) + eps).  With n = x r and g = dy w:  dx = r (g - n mean(g n)) [+ add],  dw = sum_rows dy n.
// A workgroup takes 32 rows (a wave per row, eight rounds); the dw contributions of its rows are summed per lane in f32 (rows in
// order within a wave, the four waves through LDS in order) and leave as one partial row.
constexpr int TG_MAXV = 7;
// r04: MAXV = 16-byte vectors per lane and row the instantiation can hold (3: rows up to 1536 elements, 7: up to 3584).  The one-size kernel kept
// 7 x 8 f32 partial sums + 2 x 7 row vectors per lane whatever the width, ran at ONE wave per SIMD (256+ registers), and a wave walks its eight
// rows one after the other through three dependent reductions each: 178 us for 6794 x 3584 (0.8 TB/s).  Narrow rows now take the small form, and
// both are held to two (wide) / four (narrow) waves per SIMD; a row's arithmetic and the order of the partial sums are unchanged (bit-identical).
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void rmsnorm_grad_kernel(const T* __restrict__ x, int64_t ldx, const T* __restrict__ w, const T* __restrict__ dy,
                                                           int64_t ldy, const T* __restrict__ add, int64_t lda, T* __restrict__ dx, int64_t ldd,
                                                           float* __restrict__ partial, int64_t rows, int cols, float eps) {
  extern __shared__ float red[];              // [4][cols]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = cols / 8;
  float dwp[MAXV][8];
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) dwp[i][j] = 0.f;
#pragma unroll 1   // (unrolled, hipcc hoists every round's loads and spills: one row's registers at a time, the waves per SIMD cover the latency)
  for (int round = 0; round < TG_RPB / 4; ++round) {
    const int64_t row = (int64_t)blockIdx.x * TG_RPB + round * 4 + wave;
    if (row >= rows) break;
    uint4 xv[MAXV], gv[MAXV];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int k = i * 64 + lane;
      if (k < nv) {
        xv[i] = *reinterpret_cast<const uint4*>(x + row * ldx + k * 8);
        gv[i] = *reinterpret_cast<const uint4*>(dy + row * ldy + k * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float f = vec_get<T>(xv[i], j); ss = fmaf(f, f, ss); }
      }
    }
    ss = wave_sum_f(ss);
    const float r = 1.0f / sqrtf(ss / (float)cols + eps);
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int k = i * 64 + lane;
      if (k < nv) {
        const uint4 wv = *reinterpret_cast<const uint4*>(w + k * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float n = round_to<T>(vec_get<T>(xv[i], j) * r), d = vec_get<T>(gv[i], j);
          dot = fmaf(d * vec_get<T>(wv, j), n, dot);
          dwp[i][j] = fmaf(d, n, dwp[i][j]);
        }
      }
    }
    dot = wave_sum_f(dot) / (float)cols;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int k = i * 64 + lane;
      if (k < nv) {
        const uint4 wv = *reinterpret_cast<const uint4*>(w + k * 8);
        uint4 av = make_uint4(0, 0, 0, 0);
        if (add) av = *reinterpret_cast<const uint4*>(add + row * lda + k * 8);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float n = round_to<T>(vec_get<T>(xv[i], j) * r);
          const float g = vec_get<T>(gv[i], j) * vec_get<T>(wv, j);
          o[j] = r * (g - n * dot);
          if (add) o[j] = round_to<T>(o[j]) + vec_get<T>(av, j);        // the two branches' gradients are 16-bit tensors that torch adds
        }
        *reinterpret_cast<uint4*>(dx + row * ldd + k * 8) = vec_pack<T>(o);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int k = i * 64 + lane;
    if (k < nv)
#pragma unroll
      for (int j = 0; j < 8; ++j) red[wave * cols + k * 8 + j] = dwp[i][j];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < cols; c += 256)
    partial[(int64_t)blockIdx.x * cols + c] = ((red[c] + red[cols + c]) + red[2 * cols + c]) + red[3 * cols + c];
}

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }

// h = T(T(silu(g)) * u)  on rows [g (inter) | u (inter)]
template <typename T>
__global__ __launch_bounds__(256) void swiglu_kernel(const T* __restrict__ gu, int64_t ld, T* __restrict__ out, int64_t ldo, int64_t rows, int inter) {
  const int nv = inter / 8;
  const int64_t total = rows * nv;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t r = idx / nv;
    const int k = (int)(idx - r * nv);
    const uint4 g = *reinterpret_cast<const uint4*>(gu + r * ld + k * 8), u = *reinterpret_cast<const uint4*>(gu + r * ld + inter + k * 8);
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float x = vec_get<T>(g, j); o[j] = round_to<T>(x * sigmoid_f(x)) * vec_get<T>(u, j); }
    *reinterpret_cast<uint4*>(out + r * ldo + k * 8) = vec_pack<T>(o);
  }
}

// dg = dh u silu'(g),  du = dh silu(g);  silu'(g) = s (1 + g (1 - s)), s = sigmoid(g)
template <typename T>
__global__ __launch_bounds__(256) void swiglu_grad_kernel(const T* __restrict__ gu, int64_t ld, const T* __restrict__ dh, int64_t ldh,
                                                          T* __restrict__ dgu, int64_t ldg, int64_t rows, int inter) {
  const int nv = inter / 8;
  const int64_t total = rows * nv;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t r = idx / nv;
    const int k = (int)(idx - r * nv);
    const uint4 g = *reinterpret_cast<const uint4*>(gu + r * ld + k * 8), u = *reinterpret_cast<const uint4*>(gu + r * ld + inter + k * 8);
    const uint4 d = *reinterpret_cast<const uint4*>(dh + r * ldh + k * 8);
    float og[8], ou[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = vec_get<T>(g, j), s = sigmoid_f(x), dd = vec_get<T>(d, j);
      ou[j] = dd * round_to<T>(x * s);
      og[j] = round_to<T>(dd * vec_get<T>(u, j)) * (s * (1.0f + x * (1.0f - s)));
    }
    *reinterpret_cast<uint4*>(dgu + r * ldg + k * 8) = vec_pack<T>(og);
    *reinterpret_cast<uint4*>(dgu + r * ldg + inter + k * 8) = vec_pack<T>(ou);
  }
}


// Attention backward, first form (r02): the score matrix of one head is MATERIALISED ([queries, keys] 16-bit, the rounding points of the
// reference's eager attention, modeling_qwen2.py:248-327) so that all five products of the backward run on v3d_gemm; these two row
// passes are what is left.  One workgroup per query row, the row (<= 8192 keys) held in registers.
//   causal_softmax_rows   p[i, j] = softmax_j( T(s[i, j] * scale) ) over the keys j <= i + off, j < n_keys; 0 elsewhere in [0, cols)
//   softmax_grad_rows     ds[i, j] = T( T(p (dp - sum_j p dp)) * scale )
constexpr int SM_MAXV = 4;
template <typename T>
__global__ __launch_bounds__(256) void causal_softmax_rows_kernel(const T* __restrict__ s, int64_t lds, T* __restrict__ p, int64_t ldp, int n_keys,
                                                                  int cols, int off, float scale) {
  __shared__ float sm[4];
  const int64_t row = blockIdx.x;
  const int64_t lim = row + off + 1;
  const int valid = (int)(lim < n_keys ? lim : n_keys);         // keys [0, valid) are visible (>= 1 for off >= 0)
  const int nv = cols / 8, tid = threadIdx.x;
  float x[SM_MAXV][8];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < SM_MAXV; ++i) {
    const int k = i * 256 + tid;
    if (k < nv) {
      const uint4 v = *reinterpret_cast<const uint4*>(s + row * lds + k * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        x[i][j] = k * 8 + j < valid ? round_to<T>(vec_get<T>(v, j) * scale) : -INFINITY;
        m = fmaxf(m, x[i][j]);
      }
    }
  }
  m = wave_max_f(m);
  if ((tid & 63) == 0) sm[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
  __syncthreads();
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < SM_MAXV; ++i) {
    const int k = i * 256 + tid;
    if (k < nv)
#pragma unroll
      for (int j = 0; j < 8; ++j) { x[i][j] = m == -INFINITY ? 0.f : __expf(x[i][j] - m); sum += x[i][j]; }
  }
  sum = wave_sum_f(sum);
  if ((tid & 63) == 0) sm[tid >> 6] = sum;
  __syncthreads();
  sum = (sm[0] + sm[1]) + (sm[2] + sm[3]);
  const float inv = sum > 0.f ? 1.0f / sum : 0.f;
#pragma unroll
  for (int i = 0; i < SM_MAXV; ++i) {
    const int k = i * 256 + tid;
    if (k < nv) {
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = x[i][j] * inv;
      *reinterpret_cast<uint4*>(p + row * ldp + k * 8) = vec_pack<T>(o);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_grad_rows_kernel(const T* __restrict__ p, int64_t ldp, const T* __restrict__ dp, int64_t ldd,
                                                                T* __restrict__ ds, int64_t lds, int cols, float scale) {
  __shared__ float sm[4];
  const int64_t row = blockIdx.x;
  const int nv = cols / 8, tid = threadIdx.x;
  uint4 pv[SM_MAXV], dv[SM_MAXV];
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < SM_MAXV; ++i) {
    const int k = i * 256 + tid;
    if (k < nv) {
      pv[i] = *reinterpret_cast<const uint4*>(p + row * ldp + k * 8);
      dv[i] = *reinterpret_cast<const uint4*>(dp + row * ldd + k * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) dot = fmaf(vec_get<T>(pv[i], j), vec_get<T>(dv[i], j), dot);
    }
  }
  dot = wave_sum_f(dot);
  if ((tid & 63) == 0) sm[tid >> 6] = dot;
  __syncthreads();
  dot = (sm[0] + sm[1]) + (sm[2] + sm[3]);
#pragma unroll
  for (int i = 0; i < SM_MAXV; ++i) {
    const int k = i * 256 + tid;
    if (k < nv) {
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = round_to<T>(vec_get<T>(pv[i], j) * (vec_get<T>(dv[i], j) - dot)) * scale;
      *reinterpret_cast<uint4*>(ds + row * lds + k * 8) = vec_pack<T>(o);
    }
  }
}


// torch.optim.AdamW's single-tensor update (the optimizer of the reference's HF Trainer; DeepSpeed's FusedAdam in adam_w_mode computes the
// same thing on the f32 master partition of ZeRO): decoupled weight decay, bias-corrected moments, all in f32, and the 16-bit copy
// of the parameter the next forward reads.
template <typename TG>
__device__ __forceinline__ float adamw_one(float& p, float& mi, float& vi, float gr, float lr, float beta1, float beta2, float eps, float wd, float bc1,
                                           float bc2_sqrt) {
  // no fused-multiply-add contraction here: the scalar and the eight-wide kernel (and a partition of ZeRO's flat buffer against the
  // same elements inside a tensor) must round identically, whatever shape the surrounding loop gives the compiler
#pragma clang fp contract(off)
  p = p * (1.0f - lr * wd);
  mi = mi + (gr - mi) * (1.0f - beta1);                              // lerp, as torch: exp_avg.lerp_(grad, 1 - beta1)
  vi = vi * beta2 + (1.0f - beta2) * gr * gr;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  p = p - (lr / bc1) * (mi / denom);
  return p;
}

template <typename TG, typename TP>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p32, float* __restrict__ m, float* __restrict__ v, const TG* __restrict__ g,
                                                    TP* __restrict__ p16, int64_t n, float lr, float beta1, float beta2, float eps, float wd,
                                                    float bc1, float bc2_sqrt, float grad_scale) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gr = to_f32(g[i]) * grad_scale;
    float p = p32[i], mi = m[i], vi = v[i];
    adamw_one<TG>(p, mi, vi, gr, lr, beta1, beta2, eps, wd, bc1, bc2_sqrt);
    m[i] = mi; v[i] = vi; p32[i] = p;
    if (p16) p16[i] = from_f32<TP>(p);
  }
}

// r03: the same update, eight elements per thread through 16-byte loads and stores (28 bytes per parameter stream through HBM: the
// scalar form above reached 4.8 TB/s); 16-bit gradients and 16-bit parameter copies, n % 8 == 0, 16-byte aligned arrays.  Every
// element goes through adamw_one, so the results are the scalar kernel's bit for bit.
template <typename TG, typename TP>
__global__ __launch_bounds__(256) void adamw_vec8_kernel(float* __restrict__ p32, float* __restrict__ m, float* __restrict__ v, const TG* __restrict__ g,
                                                         TP* __restrict__ p16, int64_t n8, float lr, float beta1, float beta2, float eps, float wd,
                                                         float bc1, float bc2_sqrt, float grad_scale) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    float4 pa = reinterpret_cast<const float4*>(p32)[2 * i], pb = reinterpret_cast<const float4*>(p32)[2 * i + 1];
    float4 ma = reinterpret_cast<const float4*>(m)[2 * i], mb = reinterpret_cast<const float4*>(m)[2 * i + 1];
    float4 va = reinterpret_cast<const float4*>(v)[2 * i], vb = reinterpret_cast<const float4*>(v)[2 * i + 1];
    const uint4 gv = reinterpret_cast<const uint4*>(g)[i];
    float p[8] = {pa.x, pa.y, pa.z, pa.w, pb.x, pb.y, pb.z, pb.w};
    float mm[8] = {ma.x, ma.y, ma.z, ma.w, mb.x, mb.y, mb.z, mb.w};
    float vv[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) adamw_one<TG>(p[j], mm[j], vv[j], vec_get<TG>(gv, j) * grad_scale, lr, beta1, beta2, eps, wd, bc1, bc2_sqrt);
    reinterpret_cast<float4*>(p32)[2 * i] = make_float4(p[0], p[1], p[2], p[3]);
    reinterpret_cast<float4*>(p32)[2 * i + 1] = make_float4(p[4], p[5], p[6], p[7]);
    reinterpret_cast<float4*>(m)[2 * i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
    reinterpret_cast<float4*>(m)[2 * i + 1] = make_float4(mm[4], mm[5], mm[6], mm[7]);
    reinterpret_cast<float4*>(v)[2 * i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
    reinterpret_cast<float4*>(v)[2 * i + 1] = make_float4(vv[4], vv[5], vv[6], vv[7]);
    if (p16) reinterpret_cast<uint4*>(p16)[i] = vec_pack<TP>(p);
  }
}

// Gradient of the token-embedding lookup: dE[ids[i], :] = T(sum over the j with ids[j] == ids[i], in order, of dh[rows[j], :]); one
// workgroup per listed row, only the first occurrence of an id writes (n is the handful of text tokens of a sample).
template <typename T>
__global__ __launch_bounds__(256) void embed_grad_kernel(const T* __restrict__ dh, int64_t ld, const int64_t* __restrict__ rows, const int64_t* __restrict__ ids,
                                                         int n, int H, T* __restrict__ dE, int64_t lde, int64_t n_rows, int64_t vocab) {
  const int i = blockIdx.x;
  const int64_t id = ids[i];
  // ids outside [0, vocab) (IMAGE_TOKEN_INDEX = -200 of a raw prompt, ids >= the table's rows) and rows outside [0, n_rows) take no
  // part: no write, no read (these used to be unchecked global accesses)
  if (id < 0 || id >= vocab) return;
  for (int j = 0; j < i; ++j)
    if (ids[j] == id) return;                                        // an earlier workgroup owns this id (uniform branch)
  for (int c = threadIdx.x; c < H; c += 256) {
    float s = 0.f;
    for (int j = i; j < n; ++j) {
      const int64_t r = rows[j];
      if (ids[j] == id && r >= 0 && r < n_rows) s += to_f32(dh[r * ld + c]);
    }
    dE[id * lde + c] = from_f32<T>(s);
  }
}


// nn.GELU() of the mm_projector (erf form, multimodal_projector/builder.py:41-48) and the SigLIP MLP's gelu_pytorch_tanh
// (siglip_encoder.py:253-262) as passes of their own for the training forward (which keeps the pre-activation), and their gradients.
template <int TANH> __device__ __forceinline__ float gelu_f(float x) {
  if (TANH == 2) return fmaxf(x, 0.f);                  // kind 2: ReLU (the grounding heads, llava_qwen.py:99-110)
  if (TANH) { const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x); return 0.5f * x * (1.0f + tanhf(u)); }
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
template <int TANH> __device__ __forceinline__ float gelu_df(float x) {
  if (TANH == 2) return x > 0.f ? 1.0f : 0.f;           // (also right on the ReLU's OUTPUT: h > 0 iff z > 0)
  if (TANH) {
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x), t = tanhf(u);
    return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * x * x);
  }
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}
template <typename T, int TANH, bool GRAD>
__global__ __launch_bounds__(256) void gelu_kernel(const T* __restrict__ z, int64_t ldz, const T* __restrict__ dy, int64_t ldy, T* __restrict__ out,
                                                   int64_t ldo, int64_t rows, int cols) {
  const int nv = cols / 8;
  const int64_t total = rows * nv;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t r = idx / nv;
    const int k = (int)(idx - r * nv);
    const uint4 zv = *reinterpret_cast<const uint4*>(z + r * ldz + k * 8);
    uint4 dv = make_uint4(0, 0, 0, 0);
    if (GRAD) dv = *reinterpret_cast<const uint4*>(dy + r * ldy + k * 8);
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = vec_get<T>(zv, j);
      o[j] = GRAD ? vec_get<T>(dv, j) * gelu_df<TANH>(x) : gelu_f<TANH>(x);
    }
    *reinterpret_cast<uint4*>(out + r * ldo + k * 8) = vec_pack<T>(o);
  }
}


// nn.LayerNorm of the SigLIP encoder layers (siglip_encoder.py:272-274): y = xh w + b, xh = (x - mean) rstd.  With g = dy w:
// dx = rstd (g - mean(g) - xh mean(g xh)) [+ add],  dw = sum_rows dy xh,  db = sum_rows dy.   Same shape as rmsnorm_grad_kernel:
// 32 rows per workgroup, their dw / db contributions leave as one partial row each (partial_b follows partial_w's n_part rows).
constexpr int LN_MAXV = 7;       // cols <= 3584 (the grounding heads' LayerNorm is as wide as the LLM)
template <typename T, int MAXV>       // MAXV, occupancy: see rmsnorm_grad_kernel
__global__ __launch_bounds__(256) void layernorm_grad_kernel(const T* __restrict__ x, int64_t ldx, const T* __restrict__ w, const T* __restrict__ dy,
                                                             int64_t ldy, const T* __restrict__ add, int64_t lda, T* __restrict__ dx, int64_t ldd,
                                                             float* __restrict__ partial_w, float* __restrict__ partial_b, int64_t rows, int cols, float eps) {
  extern __shared__ float red[];              // [2][4][cols]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = cols / 8;
  float dwp[MAXV][8], dbp[MAXV][8];
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { dwp[i][j] = 0.f; dbp[i][j] = 0.f; }
#pragma unroll 1   // (unrolled, hipcc hoists every round's loads and spills: one row's registers at a time, the waves per SIMD cover the latency)
  for (int round = 0; round < TG_RPB / 4; ++round) {
    const int64_t row = (int64_t)blockIdx.x * TG_RPB + round * 4 + wave;
    if (row >= rows) break;
    uint4 xv[MAXV], gv[MAXV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int k = i * 64 + lane;
      if (k < nv) {
        xv[i] = *reinterpret_cast<const uint4*>(x + row * ldx + k * 8);
        gv[i] = *reinterpret_cast<const uint4*>(dy + row * ldy + k * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += vec_get<T>(xv[i], j);
      }
    }
    const float mean = wave_sum_f(sum) / (float)cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int k = i * 64 + lane;
      if (k < nv)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = vec_get<T>(xv[i], j) - mean; q = fmaf(d, d, q); }
    }
    const float rstd = 1.0f / sqrtf(wave_sum_f(q) / (float)cols + eps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int k = i * 64 + lane;
      if (k < nv) {
        const uint4 wv = *reinterpret_cast<const uint4*>(w + k * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xh = (vec_get<T>(xv[i], j) - mean) * rstd, d = vec_get<T>(gv[i], j), g = d * vec_get<T>(wv, j);
          sg += g;
          sgx = fmaf(g, xh, sgx);
          dwp[i][j] = fmaf(d, xh, dwp[i][j]);
          dbp[i][j] += d;
        }
      }
    }
    sg = wave_sum_f(sg) / (float)cols;
    sgx = wave_sum_f(sgx) / (float)cols;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int k = i * 64 + lane;
      if (k < nv) {
        const uint4 wv = *reinterpret_cast<const uint4*>(w + k * 8);
        uint4 av = make_uint4(0, 0, 0, 0);
        if (add) av = *reinterpret_cast<const uint4*>(add + row * lda + k * 8);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xh = (vec_get<T>(xv[i], j) - mean) * rstd, g = vec_get<T>(gv[i], j) * vec_get<T>(wv, j);
          o[j] = rstd * (g - sg - xh * sgx);
          if (add) o[j] = round_to<T>(o[j]) + vec_get<T>(av, j);
        }
        *reinterpret_cast<uint4*>(dx + row * ldd + k * 8) = vec_pack<T>(o);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int k = i * 64 + lane;
    if (k < nv)
#pragma unroll
      for (int j = 0; j < 8; ++j) { red[wave * cols + k * 8 + j] = dwp[i][j]; red[(4 + wave) * cols + k * 8 + j] = dbp[i][j]; }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < cols; c += 256) {
    partial_w[(int64_t)blockIdx.x * cols + c] = ((red[c] + red[cols + c]) + red[2 * cols + c]) + red[3 * cols + c];
    partial_b[(int64_t)blockIdx.x * cols + c] = ((red[4 * cols + c] + red[5 * cols + c]) + red[6 * cols + c]) + red[7 * cols + c];
  }
}


// y = T(y + alpha x): gradient accumulation over micro-batches (torch adds a new gradient to .grad in the parameter's dtype)
template <typename T>
__global__ __launch_bounds__(256) void axpy_kernel(T* __restrict__ y, const T* __restrict__ x, float alpha, int64_t n) {
  const int64_t nv = n / 8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
    const uint4 a = reinterpret_cast<const uint4*>(y)[i], b = reinterpret_cast<const uint4*>(x)[i];
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = vec_get<T>(a, j) + alpha * vec_get<T>(b, j);
    reinterpret_cast<uint4*>(y)[i] = vec_pack<T>(o);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    const int64_t i = nv * 8 + threadIdx.x;
    y[i] = from_f32<T>(to_f32(y[i]) + alpha * to_f32(x[i]));
  }
}


// Sum of squares of a flat tensor (f32 / f16 / bf16) in f32, deterministic: SUMSQ_BLOCKS fixed partials (a block walks a fixed strided
// set of 8-element vectors, lanes then waves reduced in a fixed order), then ONE block folds the partials in index order.  The global
// gradient norm of torch.nn.utils.clip_grad_norm_ (HF Trainer max_grad_norm = 1.0; scripts/zero2.json:36 "gradient_clipping": "auto")
// is sqrt of the sum of these over the gradient tensors.  `accumulate`: the result is added to *out (chaining over tensors in a fixed
// order) instead of overwriting it.
constexpr int SUMSQ_BLOCKS = 1024;
template <typename T>
__global__ __launch_bounds__(256) void sumsq_part_kernel(const T* __restrict__ x, int64_t n, float* __restrict__ part) {
  constexpr int VEC = sizeof(T) == 4 ? 4 : 8;
  const int64_t nv = n / VEC;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
    const uint4 a = reinterpret_cast<const uint4*>(x)[i];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { const float v = vec_get<T>(a, j); acc = fmaf(v, v, acc); }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n - nv * VEC)) { const float v = to_f32(x[nv * VEC + threadIdx.x]); acc = fmaf(v, v, acc); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  __shared__ float sw[4];
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}

__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ part, int n_part, float* __restrict__ out, int accumulate) {
  __shared__ float sm[256];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n_part; i += 256) acc += part[i];
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int s_ = 128; s_ > 0; s_ >>= 1) {
    if ((int)threadIdx.x < s_) sm[threadIdx.x] += sm[threadIdx.x + s_];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + sm[0];
}

// The grounding head's loss, predict_box 'infonce' (llava_qwen.py:296-310): s_i = <o_i / |o_i|, q / |q|>, logits e^{s_i / tau},
// loss = -log(sum over the positives / sum over all rows), and its gradient with respect to the head outputs o [n, C] and q [C]:
//   dloss / ds_i = (l_i / Z - [i positive] l_i / P) / tau,   do_i = ds_i (q^ - s_i o^_i) / |o_i|,   dq = sum_i ds_i (o^_i - s_i q^) / |q|.
// One workgroup; n <= 1024 rows.
template <typename T>
__global__ __launch_bounds__(256) void ground_infonce_kernel(const T* __restrict__ obj, int64_t ldo, int n, const T* __restrict__ query, int C,
                                                             const uint8_t* __restrict__ positive, float inv_tau, float* __restrict__ loss,
                                                             float* __restrict__ scores, T* __restrict__ dobj, int64_t ldd, T* __restrict__ dquery) {
  __shared__ float s_sc[1024], s_nrm[1024], s_ds[1024];
  __shared__ float s_q;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave == 0) {
    float qq = 0.f;
    for (int c = lane; c < C; c += 64) { const float v = to_f32(query[c]); qq = fmaf(v, v, qq); }
    qq = wave_sum_f(qq);
    if (lane == 0) s_q = fmaxf(sqrtf(qq), 1e-12f);               // F.normalize's eps
  }
  __syncthreads();
  const float qn = s_q;
  for (int i = wave; i < n; i += 4) {
    float oo = 0.f, oq = 0.f;
    for (int c = lane; c < C; c += 64) { const float o = to_f32(obj[i * ldo + c]); oo = fmaf(o, o, oo); oq = fmaf(o, to_f32(query[c]), oq); }
    oo = wave_sum_f(oo); oq = wave_sum_f(oq);
    if (lane == 0) { const float on = fmaxf(sqrtf(oo), 1e-12f); s_nrm[i] = on; s_sc[i] = oq / (on * qn); }
  }
  __syncthreads();
  if (tid == 0) {
    float m = -INFINITY;
    for (int i = 0; i < n; ++i) m = fmaxf(m, s_sc[i] * inv_tau);
    float Z = 0.f, P = 0.f;
    for (int i = 0; i < n; ++i) { const float l = __expf(s_sc[i] * inv_tau - m); Z += l; if (positive[i]) P += l; }
    *loss = -logf(P / Z);
    for (int i = 0; i < n; ++i) {
      const float l = __expf(s_sc[i] * inv_tau - m);
      s_ds[i] = (l / Z - (positive[i] ? l / P : 0.f)) * inv_tau;
      if (scores) scores[i] = s_sc[i];
    }
  }
  __syncthreads();
  for (int i = wave; i < n; i += 4) {
    const float ds = s_ds[i], sc = s_sc[i], on = s_nrm[i];
    for (int c = lane; c < C; c += 64) {
      const float oh = to_f32(obj[i * ldo + c]) / on, qh = to_f32(query[c]) / qn;
      dobj[i * ldd + c] = from_f32<T>(ds * (qh - sc * oh) / on);
    }
  }
  for (int c = tid; c < C; c += 256) {
    const float qh = to_f32(query[c]) / qn;
    float a = 0.f;
    for (int i = 0; i < n; ++i) a += s_ds[i] * (to_f32(obj[i * ldo + c]) / s_nrm[i] - s_sc[i] * qh);
    dquery[c] = from_f32<T>(a / qn);
  }
}

// inv_count[o] = 1 / #{t : mask[o, t] != 0} (0 if none): the mean's weights of v3d_masked_mean
__global__ __launch_bounds__(256) void mask_count_kernel(const uint8_t* __restrict__ mask, int T_, float* __restrict__ inv_count) {
  __shared__ int red[4];
  int cnt = 0;
  for (int t = threadIdx.x; t < T_; t += 256) cnt += mask[(int64_t)blockIdx.x * T_ + t] != 0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) { const int c = red[0] + red[1] + red[2] + red[3]; inv_count[blockIdx.x] = c > 0 ? 1.0f / (float)c : 0.f; }
}

// Backward of v3d_masked_mean: dfeat[t, :] (+)= sum over the objects o (in order) with mask[o, t] of dobj[o, :] / count[o]
template <typename T>
__global__ __launch_bounds__(256) void masked_mean_grad_kernel(const uint8_t* __restrict__ mask, const float* __restrict__ inv_count, const T* __restrict__ dobj,
                                                               int n, int T_, int C, T* __restrict__ dfeat, int accumulate) {
  const int nv = C / 8;
  const int64_t total = (int64_t)T_ * nv;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t t = idx / nv;
    const int k = (int)(idx - t * nv);
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bool any = false;
    for (int o = 0; o < n; ++o) {
      if (mask[(int64_t)o * T_ + t]) {
        any = true;
        const uint4 d = *reinterpret_cast<const uint4*>(dobj + (int64_t)o * C + k * 8);
        const float w = inv_count[o];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = fmaf(vec_get<T>(d, j), w, a[j]);
      }
    }
    uint4* dst = reinterpret_cast<uint4*>(dfeat + t * C + k * 8);
    if (accumulate) {
      if (!any) continue;
      const uint4 old = *dst;
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = round_to<T>(a[j]) + vec_get<T>(old, j);
    }
    *dst = vec_pack<T>(a);
  }
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

#define V3D_DISPATCH_HALF(dtype, ...)                                         \
  switch (dtype) {                                                            \
    case V3D_F16: { using T = v3d::f16_t; __VA_ARGS__; } break;               \
    case V3D_BF16: { using T = v3d::bf16_t; __VA_ARGS__; } break;             \
    default: v3d::set_error("16-bit dtypes only (got code %d)", (int)(dtype)); return V3D_E_INVALID; \
  }

extern "C" int v3d_transpose(const void* x, int64_t ldx, int64_t rows, int cols, void* out, int64_t ldo, int64_t out_cols, int dtype,
                             void* stream) {
  V3D_REQUIRE(x && out, "v3d_transpose: null pointer");
  V3D_REQUIRE(rows > 0 && cols > 0 && cols % 8 == 0 && ldx % 8 == 0 && ldx >= cols, "v3d_transpose: the source needs cols %% 8 == 0 and a row stride that is a multiple of 8");
  V3D_REQUIRE(out_cols >= rows && out_cols % 8 == 0 && ldo % 8 == 0 && ldo >= out_cols, "v3d_transpose: out_cols must cover the source rows and be a multiple of 8 (so must ldo)");
  V3D_REQUIRE(aligned16(x) && aligned16(out), "v3d_transpose: pointers must be 16-byte aligned");
  V3D_REQUIRE((out_cols + 63) / 64 < (1ll << 31) && (cols + 63) / 64 <= 65535, "v3d_transpose: too large");
  const dim3 grid((unsigned)((out_cols + 63) / 64), (unsigned)((cols + 63) / 64));
  V3D_DISPATCH_HALF(dtype, hipLaunchKernelGGL(transpose_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)x, ldx, rows, cols, (T*)out, ldo, out_cols));
  return check_launch("v3d_transpose");
}

extern "C" int64_t v3d_colsum_workspace_bytes(int64_t rows, int cols) {
  if (rows <= 0 || cols <= 0) return 0;
  return ((rows + TG_RPB - 1) / TG_RPB) * (int64_t)cols * 4;
}

static int colsum_final(const float* partial, int64_t n_part, int cols, void* out, int out_dtype, hipStream_t st, const char* what) {
  const dim3 grid((cols + 15) / 16);
  switch (out_dtype) {
    case V3D_F32: hipLaunchKernelGGL(colsum_final_kernel<float>, grid, dim3(1024), 0, st, partial, n_part, cols, (float*)out); break;
    case V3D_F16: hipLaunchKernelGGL(colsum_final_kernel<f16_t>, grid, dim3(1024), 0, st, partial, n_part, cols, (f16_t*)out); break;
    case V3D_BF16: hipLaunchKernelGGL(colsum_final_kernel<bf16_t>, grid, dim3(1024), 0, st, partial, n_part, cols, (bf16_t*)out); break;
    default: set_error("%s: unknown output dtype %d", what, out_dtype); return V3D_E_INVALID;
  }
  return check_launch(what);
}

extern "C" int v3d_colsum(const void* x, int64_t ldx, int64_t rows, int cols, int dtype, float* workspace, void* out, int out_dtype,
                          void* stream) {
  V3D_REQUIRE(x && workspace && out, "v3d_colsum: null pointer");
  V3D_REQUIRE(rows > 0 && cols > 0 && cols % 8 == 0 && ldx % 8 == 0 && ldx >= cols && aligned16(x), "v3d_colsum: cols and the row stride must be multiples of 8");
  const int64_t n_part = (rows + TG_RPB - 1) / TG_RPB;
  V3D_REQUIRE(n_part <= 65535, "v3d_colsum: too many rows");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((cols / 8 + 255) / 256, (unsigned)n_part);
  V3D_DISPATCH_HALF(dtype, hipLaunchKernelGGL(colsum_partial_kernel<T>, grid, dim3(256), 0, st, (const T*)x, ldx, rows, cols, workspace));
  if (int e = check_launch("v3d_colsum")) return e;
  return colsum_final(workspace, n_part, cols, out, out_dtype, st, "v3d_colsum");
}

extern "C" int v3d_rmsnorm_grad(const void* x, int64_t ldx, const void* weight, const void* dy, int64_t ldy, const void* add, int64_t lda,
                                void* dx, int64_t ldd, float* workspace, void* dweight, int dw_dtype, int64_t rows, int cols, float eps,
                                int dtype, void* stream) {
  V3D_REQUIRE(x && weight && dy && dx && workspace && dweight, "v3d_rmsnorm_grad: null pointer");
  V3D_REQUIRE(rows > 0 && cols > 0 && cols % 8 == 0 && cols <= TG_MAXV * 512, "v3d_rmsnorm_grad: cols=%d unsupported", cols);
  V3D_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && ldd % 8 == 0 && (!add || lda % 8 == 0) && aligned16(x) && aligned16(dy) && aligned16(dx) &&
              aligned16(weight) && aligned16(add), "v3d_rmsnorm_grad: alignment");
  const int64_t n_part = (rows + TG_RPB - 1) / TG_RPB;
  V3D_REQUIRE(n_part < (1ll << 31), "v3d_rmsnorm_grad: too many rows");
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)4 * cols * sizeof(float);
  V3D_DISPATCH_HALF(dtype, {
    if (cols <= 3 * 512) hipLaunchKernelGGL((rmsnorm_grad_kernel<T, 3>), dim3((unsigned)n_part), dim3(256), lds, st, (const T*)x, ldx, (const T*)weight,
                                            (const T*)dy, ldy, (const T*)add, lda, (T*)dx, ldd, workspace, rows, cols, eps);
    else hipLaunchKernelGGL((rmsnorm_grad_kernel<T, TG_MAXV>), dim3((unsigned)n_part), dim3(256), lds, st, (const T*)x, ldx, (const T*)weight,
                            (const T*)dy, ldy, (const T*)add, lda, (T*)dx, ldd, workspace, rows, cols, eps);
  });
  if (int e = check_launch("v3d_rmsnorm_grad")) return e;
  return colsum_final(workspace, n_part, cols, dweight, dw_dtype, st, "v3d_rmsnorm_grad");
}

extern "C" int v3d_swiglu(const void* gu, int64_t ld, void* out, int64_t ldo, int64_t rows, int inter, int dtype, void* stream) {
  V3D_REQUIRE(gu && out, "v3d_swiglu: null pointer");
  V3D_REQUIRE(rows > 0 && inter > 0 && inter % 8 == 0 && ld % 8 == 0 && ld >= 2 * (int64_t)inter && ldo % 8 == 0 && ldo >= inter && aligned16(gu) && aligned16(out),
              "v3d_swiglu: rows are [gate (inter) | up (inter)], inter %% 8 == 0");
  int64_t blocks = (rows * (inter / 8) + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  V3D_DISPATCH_HALF(dtype, hipLaunchKernelGGL(swiglu_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const T*)gu, ld, (T*)out, ldo, rows, inter));
  return check_launch("v3d_swiglu");
}

extern "C" int v3d_swiglu_grad(const void* gu, int64_t ld, const void* dh, int64_t ldh, void* dgu, int64_t ldg, int64_t rows, int inter,
                               int dtype, void* stream) {
  V3D_REQUIRE(gu && dh && dgu, "v3d_swiglu_grad: null pointer");
  V3D_REQUIRE(rows > 0 && inter > 0 && inter % 8 == 0 && ld % 8 == 0 && ld >= 2 * (int64_t)inter && ldg % 8 == 0 && ldg >= 2 * (int64_t)inter && ldh % 8 == 0 &&
              ldh >= inter && aligned16(gu) && aligned16(dh) && aligned16(dgu), "v3d_swiglu_grad: rows are [gate (inter) | up (inter)], inter %% 8 == 0");
  int64_t blocks = (rows * (inter / 8) + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  V3D_DISPATCH_HALF(dtype, hipLaunchKernelGGL(swiglu_grad_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const T*)gu, ld, (const T*)dh, ldh,
                                              (T*)dgu, ldg, rows, inter));
  return check_launch("v3d_swiglu_grad");
}

extern "C" int v3d_causal_softmax_rows(const void* s, int64_t lds, void* p, int64_t ldp, int64_t rows, int n_keys, int cols, int offset,
                                       float scale, int dtype, void* stream) {
  V3D_REQUIRE(s && p, "v3d_causal_softmax_rows: null pointer");
  V3D_REQUIRE(rows > 0 && rows < (1ll << 31) && n_keys > 0 && cols >= n_keys && cols % 8 == 0 && cols <= SM_MAXV * 2048 && offset >= 0,
              "v3d_causal_softmax_rows: cols %% 8 == 0, n_keys <= cols <= %d, offset >= 0", SM_MAXV * 2048);
  V3D_REQUIRE(lds % 8 == 0 && ldp % 8 == 0 && lds >= cols && ldp >= cols && aligned16(s) && aligned16(p), "v3d_causal_softmax_rows: alignment");
  V3D_DISPATCH_HALF(dtype, hipLaunchKernelGGL(causal_softmax_rows_kernel<T>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const T*)s, lds,
                                              (T*)p, ldp, n_keys, cols, offset, scale));
  return check_launch("v3d_causal_softmax_rows");
}

extern "C" int v3d_softmax_grad_rows(const void* p, int64_t ldp, const void* dp, int64_t ldd, void* ds, int64_t lds, int64_t rows, int cols,
                                     float scale, int dtype, void* stream) {
  V3D_REQUIRE(p && dp && ds, "v3d_softmax_grad_rows: null pointer");
  V3D_REQUIRE(rows > 0 && rows < (1ll << 31) && cols > 0 && cols % 8 == 0 && cols <= SM_MAXV * 2048, "v3d_softmax_grad_rows: cols %% 8 == 0, cols <= %d", SM_MAXV * 2048);
  V3D_REQUIRE(ldp % 8 == 0 && ldd % 8 == 0 && lds % 8 == 0 && ldp >= cols && ldd >= cols && lds >= cols && aligned16(p) && aligned16(dp) && aligned16(ds),
              "v3d_softmax_grad_rows: alignment");
  V3D_DISPATCH_HALF(dtype, hipLaunchKernelGGL(softmax_grad_rows_kernel<T>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const T*)p, ldp,
                                              (const T*)dp, ldd, (T*)ds, lds, cols, scale));
  return check_launch("v3d_softmax_grad_rows");
}

extern "C" int v3d_adamw_step(float* p32, float* m, float* v, const void* grad, int grad_dtype, void* p16, int p16_dtype, int64_t n, float lr,
                              float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream) {
  V3D_REQUIRE(p32 && m && v && grad, "v3d_adamw_step: null pointer");
  V3D_REQUIRE(n > 0 && step >= 1 && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f, "v3d_adamw_step: bad arguments");
  V3D_REQUIRE(!p16 || p16_dtype == V3D_F16 || p16_dtype == V3D_BF16, "v3d_adamw_step: the parameter copy is f16 or bf16");
  const float bc1 = 1.0f - powf(beta1, (float)step), bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
  int64_t blocks = (n + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipStream_t st = (hipStream_t)stream;
#define V3D_ADAMW(TG, TP) hipLaunchKernelGGL((adamw_kernel<TG, TP>), dim3((unsigned)blocks), dim3(256), 0, st, p32, m, v, (const TG*)grad, (TP*)p16, n, lr, \
                                             beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, grad_scale)
#define V3D_ADAMW_G(TG) { if (p16_dtype == V3D_F16 && p16) { V3D_ADAMW(TG, f16_t); } else { V3D_ADAMW(TG, bf16_t); } }
  if (grad_dtype != V3D_F32 && n % 8 == 0 && n >= 4096 && aligned16(p32) && aligned16(m) && aligned16(v) && aligned16(grad) && aligned16(p16)) {
    const int64_t n8 = n / 8;
    int64_t vb = (n8 + 255) / 256;
    if (vb > 256 * 16) vb = 256 * 16;
#define V3D_ADAMW8(TG, TP) hipLaunchKernelGGL((adamw_vec8_kernel<TG, TP>), dim3((unsigned)vb), dim3(256), 0, st, p32, m, v, (const TG*)grad, (TP*)p16, n8, \
                                              lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, grad_scale)
    if (grad_dtype == V3D_F16) { if (p16_dtype == V3D_F16 && p16) { V3D_ADAMW8(f16_t, f16_t); } else { V3D_ADAMW8(f16_t, bf16_t); } }
    else { if (p16_dtype == V3D_F16 && p16) { V3D_ADAMW8(bf16_t, f16_t); } else { V3D_ADAMW8(bf16_t, bf16_t); } }
#undef V3D_ADAMW8
    return check_launch("v3d_adamw_step");
  }
  switch (grad_dtype) {
    case V3D_F32: V3D_ADAMW_G(float) break;
    case V3D_F16: V3D_ADAMW_G(f16_t) break;
    case V3D_BF16: V3D_ADAMW_G(bf16_t) break;
    default: set_error("v3d_adamw_step: unknown gradient dtype %d", grad_dtype); return V3D_E_INVALID;
  }
#undef V3D_ADAMW_G
#undef V3D_ADAMW
  return check_launch("v3d_adamw_step");
}

extern "C" int v3d_embed_grad(const void* dh, int64_t ld, int64_t n_rows, const int64_t* rows, const int64_t* ids, int n, int H, void* dE, int64_t lde,
                              int64_t vocab, int dtype, void* stream) {
  V3D_REQUIRE(dh && rows && ids && dE, "v3d_embed_grad: null pointer");
  V3D_REQUIRE(n > 0 && n <= 65535 && H > 0 && ld >= H && lde >= H && n_rows > 0 && vocab > 0, "v3d_embed_grad: bad shape");
  V3D_DISPATCH_HALF(dtype, hipLaunchKernelGGL(embed_grad_kernel<T>, dim3(n), dim3(256), 0, (hipStream_t)stream, (const T*)dh, ld, rows, ids, n, H, (T*)dE, lde,
                                              n_rows, vocab));
  return check_launch("v3d_embed_grad");
}

static int gelu_launch(const void* z, int64_t ldz, const void* dy, int64_t ldy, void* out, int64_t ldo, int64_t rows, int cols, int tanh_form,
                       int dtype, void* stream, const char* what) {
  V3D_REQUIRE(z && out, "%s: null pointer", what);
  V3D_REQUIRE(rows > 0 && cols > 0 && cols % 8 == 0 && ldz % 8 == 0 && ldo % 8 == 0 && ldz >= cols && ldo >= cols && (!dy || (ldy % 8 == 0 && ldy >= cols)) &&
              aligned16(z) && aligned16(out) && aligned16(dy), "%s: cols and strides must be multiples of 8", what);
  int64_t blocks = (rows * (cols / 8) + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipStream_t st = (hipStream_t)stream;
#define V3D_GELU(TH, GR) hipLaunchKernelGGL((gelu_kernel<T, TH, GR>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)z, ldz, (const T*)dy, ldy, (T*)out, ldo, rows, cols)
  V3D_DISPATCH_HALF(dtype, {
    if (dy) { if (tanh_form == 2) { V3D_GELU(2, true); } else if (tanh_form) { V3D_GELU(1, true); } else { V3D_GELU(0, true); } }
    else { if (tanh_form == 2) { V3D_GELU(2, false); } else if (tanh_form) { V3D_GELU(1, false); } else { V3D_GELU(0, false); } }
  });
#undef V3D_GELU
  return check_launch(what);
}

extern "C" int v3d_gelu(const void* z, int64_t ldz, void* out, int64_t ldo, int64_t rows, int cols, int tanh_form, int dtype, void* stream) {
  return gelu_launch(z, ldz, nullptr, 0, out, ldo, rows, cols, tanh_form, dtype, stream, "v3d_gelu");
}

extern "C" int v3d_gelu_grad(const void* z, int64_t ldz, const void* dy, int64_t ldy, void* dz, int64_t ldo, int64_t rows, int cols, int tanh_form,
                             int dtype, void* stream) {
  V3D_REQUIRE(dy, "v3d_gelu_grad: null pointer");
  return gelu_launch(z, ldz, dy, ldy, dz, ldo, rows, cols, tanh_form, dtype, stream, "v3d_gelu_grad");
}

extern "C" int v3d_layernorm_grad(const void* x, int64_t ldx, const void* weight, const void* dy, int64_t ldy, const void* add, int64_t lda,
                                  void* dx, int64_t ldd, float* workspace, void* dweight, void* dbias, int dw_dtype, int64_t rows, int cols,
                                  float eps, int dtype, void* stream) {
  V3D_REQUIRE(x && weight && dy && dx && workspace && dweight && dbias, "v3d_layernorm_grad: null pointer");
  V3D_REQUIRE(rows > 0 && cols > 0 && cols % 8 == 0 && cols <= LN_MAXV * 512, "v3d_layernorm_grad: cols=%d unsupported", cols);
  V3D_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && ldd % 8 == 0 && (!add || lda % 8 == 0) && aligned16(x) && aligned16(dy) && aligned16(dx) &&
              aligned16(weight) && aligned16(add), "v3d_layernorm_grad: alignment");
  const int64_t n_part = (rows + TG_RPB - 1) / TG_RPB;
  V3D_REQUIRE(n_part < (1ll << 31), "v3d_layernorm_grad: too many rows");
  hipStream_t st = (hipStream_t)stream;
  float* pw = workspace;
  float* pb = workspace + n_part * cols;
  const size_t lds = (size_t)8 * cols * sizeof(float);
  V3D_DISPATCH_HALF(dtype, {
    static bool attr_done = false;
    if (!attr_done) {
      hipError_t e = hipFuncSetAttribute((const void*)layernorm_grad_kernel<T, LN_MAXV>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * LN_MAXV * 512 * 4);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)layernorm_grad_kernel<T, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 3 * 512 * 4);
      if (e != hipSuccess) { set_error("v3d_layernorm_grad: LDS attribute: %s", hipGetErrorString(e)); return V3D_E_LAUNCH; }
      attr_done = true;
    }
    if (cols <= 3 * 512) hipLaunchKernelGGL((layernorm_grad_kernel<T, 3>), dim3((unsigned)n_part), dim3(256), lds, st, (const T*)x, ldx, (const T*)weight,
                                            (const T*)dy, ldy, (const T*)add, lda, (T*)dx, ldd, pw, pb, rows, cols, eps);
    else hipLaunchKernelGGL((layernorm_grad_kernel<T, LN_MAXV>), dim3((unsigned)n_part), dim3(256), lds, st, (const T*)x, ldx, (const T*)weight,
                            (const T*)dy, ldy, (const T*)add, lda, (T*)dx, ldd, pw, pb, rows, cols, eps);
  });
  if (int e = check_launch("v3d_layernorm_grad")) return e;
  if (int e = colsum_final(pw, n_part, cols, dweight, dw_dtype, st, "v3d_layernorm_grad")) return e;
  return colsum_final(pb, n_part, cols, dbias, dw_dtype, st, "v3d_layernorm_grad");
}

extern "C" int v3d_axpy(void* y, const void* x, float alpha, int64_t n, int dtype, void* stream) {
  V3D_REQUIRE(y && x && n > 0 && aligned16(y) && aligned16(x), "v3d_axpy: bad arguments (16-byte aligned, n > 0)");
  int64_t blocks = (n / 8 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 256 * 16) blocks = 256 * 16;
  V3D_DISPATCH_HALF(dtype, hipLaunchKernelGGL(axpy_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (T*)y, (const T*)x, alpha, n));
  return check_launch("v3d_axpy");
}

extern "C" int64_t v3d_sumsq_workspace_bytes() { return (int64_t)SUMSQ_BLOCKS * (int64_t)sizeof(float); }

extern "C" int v3d_sumsq(const void* x, int64_t n, int dtype, float* out, int accumulate, float* workspace, void* stream) {
  V3D_REQUIRE(x && out && workspace && n > 0 && aligned16(x), "v3d_sumsq: bad arguments (16-byte aligned, n > 0)");
  hipStream_t st = (hipStream_t)stream;
  int64_t blocks = (n / 8 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > SUMSQ_BLOCKS) blocks = SUMSQ_BLOCKS;
  V3D_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(sumsq_part_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, st, (const T*)x, n, workspace));
  if (int e = check_launch("v3d_sumsq")) return e;
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, (int)blocks, out, accumulate);
  return check_launch("v3d_sumsq");
}

extern "C" int v3d_ground_infonce(const void* obj, int64_t ldo, int n, const void* query, int C, const uint8_t* positive, float temperature,
                                  float* loss, float* scores, void* dobj, int64_t ldd, void* dquery, int dtype, void* stream) {
  V3D_REQUIRE(obj && query && positive && loss && dobj && dquery, "v3d_ground_infonce: null pointer");
  V3D_REQUIRE(n >= 1 && n <= 1024 && C > 0 && ldo >= C && ldd >= C && temperature > 0.f, "v3d_ground_infonce: 1 to 1024 rows, temperature > 0");
  V3D_DISPATCH_HALF(dtype, hipLaunchKernelGGL(ground_infonce_kernel<T>, dim3(1), dim3(256), 0, (hipStream_t)stream, (const T*)obj, ldo, n, (const T*)query, C,
                                              positive, 1.0f / temperature, loss, scores, (T*)dobj, ldd, (T*)dquery));
  return check_launch("v3d_ground_infonce");
}

extern "C" int v3d_masked_mean_grad(const uint8_t* mask, int n_obj, int T_, int C, const void* dobj, void* dfeat, int accumulate, float* inv_count,
                                    int dtype, void* stream) {
  V3D_REQUIRE(mask && dobj && dfeat && inv_count, "v3d_masked_mean_grad: null pointer");
  V3D_REQUIRE(n_obj >= 1 && n_obj <= 65535 && T_ > 0 && C > 0 && C % 8 == 0 && aligned16(dobj) && aligned16(dfeat), "v3d_masked_mean_grad: bad shape");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(mask_count_kernel, dim3(n_obj), dim3(256), 0, st, mask, T_, inv_count);
  if (int e = check_launch("v3d_masked_mean_grad")) return e;
  int64_t blocks = ((int64_t)T_ * (C / 8) + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  V3D_DISPATCH_HALF(dtype, hipLaunchKernelGGL(masked_mean_grad_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, st, mask, (const float*)inv_count, (const T*)dobj,
                                              n_obj, T_, C, (T*)dfeat, accumulate));
  return check_launch("v3d_masked_mean_grad");
}
