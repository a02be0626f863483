// Row-wise normalisations and rotary embedding (K13, K14, LayerNorm of K11): HBM-bound,
// one wave per row, 16-byte loads, the row stays in registers between the reduction and the
// scaling pass (one read + one write of the tensor).
#include "v3d_common.h"

namespace v3d {

constexpr int NORM_MAXV = 8;   // vectors of 8 per lane -> rows up to 8*64*8 = 4096 elements

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
  return x;
}

// Qwen2RMSNorm (modeling_qwen2.py:85-90): f32 x * rsqrt(mean(x^2) + eps), cast to dtype, THEN * weight.
template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                      T* __restrict__ out, int64_t rows, int cols, int64_t ldx,
                                                      int64_t ldo, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nv = cols / 8;
  const uint4* xr = reinterpret_cast<const uint4*>(x + row * ldx);
  uint4 v[NORM_MAXV];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_MAXV; ++i) {
    const int k = i * 64 + lane;
    if (k < nv) {
      v[i] = xr[k];
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float f = vec_get<T>(v[i], j); ss = fmaf(f, f, ss); }
    }
  }
  ss = wave_sum(ss);
  const float r = 1.0f / sqrtf(ss / (float)cols + eps);
  uint4* orow = reinterpret_cast<uint4*>(out + row * ldo);
  const uint4* wr = reinterpret_cast<const uint4*>(w);
#pragma unroll
  for (int i = 0; i < NORM_MAXV; ++i) {
    const int k = i * 64 + lane;
    if (k < nv) {
      const uint4 wv = wr[k];
      float y[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) y[j] = vec_get<T>(wv, j) * round_to<T>(vec_get<T>(v[i], j) * r);
      orow[k] = vec_pack<T>(y);
    }
  }
}

// nn.LayerNorm (siglip_encoder.py:272,274): f32 statistics, (x-mean)*rstd*w + b, one rounding.
// One wave per R rows of at most MAXV * 512 elements.  R = 1, MAXV = 8 is the general form; SigLIP's 1152-wide rows are 2304 bytes,
// a third of what a Qwen2 row keeps in flight per wave, so the <3, 3> form loads three rows before it reduces any (the same
// per-row arithmetic in the same order: results are bit-identical) - the kernel is bound by bytes in flight, not by arithmetic.
template <typename T, int MAXV, int R>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                        const T* __restrict__ b, T* __restrict__ out, int64_t rows,
                                                        int cols, int64_t ldx, int64_t ldo, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  if (row0 >= rows) return;
  const int nv = cols / 8;
  uint4 v[R][MAXV];
  float s[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    s[r] = 0.f;
    const int64_t row = row0 + r < rows ? row0 + r : rows - 1;
    const uint4* xr = reinterpret_cast<const uint4*>(x + row * ldx);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int k = i * 64 + lane;
      if (k < nv) v[r][i] = xr[k];
    }
  }
  float mean[R], rstd[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int k = i * 64 + lane;
      if (k < nv) {
#pragma unroll
        for (int j = 0; j < 8; ++j) s[r] += vec_get<T>(v[r][i], j);
      }
    }
    mean[r] = wave_sum(s[r]) / (float)cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int k = i * 64 + lane;
      if (k < nv) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = vec_get<T>(v[r][i], j) - mean[r]; q = fmaf(d, d, q); }
      }
    }
    rstd[r] = 1.0f / sqrtf(wave_sum(q) / (float)cols + eps);
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int k = i * 64 + lane;
    if (k < nv) {
      const uint4 wv = reinterpret_cast<const uint4*>(w)[k];
      const uint4 bv = reinterpret_cast<const uint4*>(b)[k];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (row0 + r < rows) {
          float y[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) y[j] = (vec_get<T>(v[r][i], j) - mean[r]) * rstd[r] * vec_get<T>(wv, j) + vec_get<T>(bv, j);
          reinterpret_cast<uint4*>(out + (row0 + r) * ldo)[k] = vec_pack<T>(y);
        }
      }
    }
  }
}

// cos/sin table (modeling_qwen2.py:106-129): freqs = inv_freq[i] * pos in f32, cos/sin in f32, cast to the
// model dtype.  Trig evaluated in double and rounded once (= correctly rounded f32, then dtype).
template <typename T>
__global__ __launch_bounds__(256) void rope_table_kernel(const float* __restrict__ inv_freq, int half, int n_pos,
                                                         T* __restrict__ cos_t, T* __restrict__ sin_t) {
  const int64_t total = (int64_t)n_pos * half;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int pos = (int)(i / half), j = (int)(i - (int64_t)pos * half);
    const float f = inv_freq[j] * (float)pos;
    cos_t[i] = from_f32<T>((float)cos((double)f));
    sin_t[i] = from_f32<T>((float)sin((double)f));
  }
}

// apply_rotary_pos_emb (modeling_qwen2.py:141-173) with three identical position rows (the only case
// the eval path produces, :1003-1004), in place on `n_heads` heads of `hd` starting at x:
//   out = T(T(x*cos) + T(rotate_half(x)*sin)),  rotate_half(x)[i] = -x[i+hd/2] (i<hd/2) else x[i-hd/2]
// Thread = 8 consecutive i < hd/2 of one (token, head): the (i, i+hd/2) pairs, two 16-byte vectors.
template <typename T>
__global__ __launch_bounds__(256) void rope_kernel(T* __restrict__ x, int64_t ldx, int64_t tokens, int n_heads, int hd,
                                                   const T* __restrict__ cos_t, const T* __restrict__ sin_t,
                                                   const int32_t* __restrict__ positions, int pos0) {
  const int half = hd / 2, vper = half / 8;
  const int64_t total = tokens * n_heads * vper;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % vper);
    const int64_t th = i / vper;
    const int head = (int)(th % n_heads);
    const int64_t tok = th / n_heads;
    const int pos = positions ? positions[tok] : pos0 + (int)tok;
    T* base = x + tok * ldx + (int64_t)head * hd + c * 8;
    const uint4 lo = *reinterpret_cast<const uint4*>(base);
    const uint4 hi = *reinterpret_cast<const uint4*>(base + half);
    const uint4 cv = *reinterpret_cast<const uint4*>(cos_t + (int64_t)pos * half + c * 8);
    const uint4 sv = *reinterpret_cast<const uint4*>(sin_t + (int64_t)pos * half + c * 8);
    float ol[8], oh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float a = vec_get<T>(lo, j), b = vec_get<T>(hi, j);
      const float cs = vec_get<T>(cv, j), sn = vec_get<T>(sv, j);
      ol[j] = round_to<T>(a * cs) + round_to<T>(-b * sn);
      oh[j] = round_to<T>(b * cs) + round_to<T>(a * sn);
    }
    *reinterpret_cast<uint4*>(base) = vec_pack<T>(ol);
    *reinterpret_cast<uint4*>(base + half) = vec_pack<T>(oh);
  }
}

// Prefill K14 + KV-cache append in ONE pass over the QKV rows (replaces rope_kernel + copy_rows_kernel): rotary in place on
// the n_q query heads; the n_kv key heads are rotated and written straight to the cache row of their token together with the
// value heads (the QKV buffer's own k/v columns are left as the GEMM wrote them: attention reads K/V from the cache).
// Token t sits at position positions[t] (or pos0 + t) and its cache row is cache + dst_rows[t] * ldc (or (row0 + t) * ldc):
// the indexed form serves several questions of one scene prefilled together, each appending to its own cache
// (all caches live in one allocation, so a row index addresses any of them).
// Same arithmetic and rounding points as rope_kernel (modeling_qwen2.py:141-173).
template <typename T>
__global__ __launch_bounds__(256) void rope_kv_store_kernel(T* __restrict__ qkv, int64_t ldx, int64_t tokens, int n_q, int n_kv, int hd,
                                                            const T* __restrict__ cos_t, const T* __restrict__ sin_t,
                                                            const int32_t* __restrict__ positions, int pos0,
                                                            T* __restrict__ cache, int64_t ldc, const int64_t* __restrict__ dst_rows,
                                                            int64_t row0) {
  const int half = hd / 2, vper = half / 8;
  const int units = (n_q + 2 * n_kv) * vper;          // per token: (q | k) heads rotate a vector pair, v heads copy one
  const int64_t total = tokens * units;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t tok = i / units;
    const int u = (int)(i - tok * units);
    const int head = u / vper, c = u - head * vper;
    T* src = qkv + tok * ldx + (int64_t)head * hd + c * 8;
    const int64_t drow = dst_rows ? dst_rows[tok] : row0 + tok;
    const uint4 lo = *reinterpret_cast<const uint4*>(src);
    const uint4 hi = *reinterpret_cast<const uint4*>(src + half);
    if (head >= n_q + n_kv) {                          // value head: copy both halves
      T* d = cache + drow * ldc + (int64_t)(head - n_q) * hd + c * 8;
      *reinterpret_cast<uint4*>(d) = lo;
      *reinterpret_cast<uint4*>(d + half) = hi;
      continue;
    }
    const int pos = positions ? positions[tok] : pos0 + (int)tok;
    const uint4 cv = *reinterpret_cast<const uint4*>(cos_t + (int64_t)pos * half + c * 8);
    const uint4 sv = *reinterpret_cast<const uint4*>(sin_t + (int64_t)pos * half + c * 8);
    float ol[8], oh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float a = vec_get<T>(lo, j), b = vec_get<T>(hi, j);
      const float cs = vec_get<T>(cv, j), sn = vec_get<T>(sv, j);
      ol[j] = round_to<T>(a * cs) + round_to<T>(-b * sn);
      oh[j] = round_to<T>(b * cs) + round_to<T>(a * sn);
    }
    T* d = head < n_q ? src : cache + drow * ldc + (int64_t)(head - n_q) * hd + c * 8;
    *reinterpret_cast<uint4*>(d) = vec_pack<T>(ol);
    *reinterpret_cast<uint4*>(d + half) = vec_pack<T>(oh);
  }
}

// x[rows[i], :] = T(x[rows[i], :] + add[:])  (llava_arch.py:697-700: the box-centre PE added to the <coord> token rows)
template <typename T>
__global__ __launch_bounds__(256) void add_row_kernel(T* __restrict__ x, int64_t ldx, const int64_t* __restrict__ rows, int C,
                                                      const T* __restrict__ add) {
  T* r = x + rows[blockIdx.x] * ldx;
  for (int k = threadIdx.x; k < C / 8; k += blockDim.x) {
    const uint4 a = reinterpret_cast<const uint4*>(r)[k], b = reinterpret_cast<const uint4*>(add)[k];
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = vec_get<T>(a, j) + vec_get<T>(b, j);
    reinterpret_cast<uint4*>(r)[k] = vec_pack<T>(o);
  }
}

// strided row copy (KV-cache append, im2col-free patch gather, ...): out[r, :cols] = in[r, :cols]
template <typename T>
__global__ __launch_bounds__(256) void copy_rows_kernel(const T* __restrict__ in, int64_t ldi, T* __restrict__ out,
                                                        int64_t ldo, int64_t rows, int cols) {
  const int nv = cols / 8;
  const int64_t total = rows * nv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / nv;
    const int k = (int)(i - r * nv);
    reinterpret_cast<uint4*>(out + r * ldo)[k] = reinterpret_cast<const uint4*>(in + r * ldi)[k];
  }
}

// the same rows written to n_copies destinations copy_stride elements apart (the cached K/V prefix of a scene handed to every
// question context of a group): one read of the source per destination vector group, all copies in one launch
template <typename T>
__global__ __launch_bounds__(256) void copy_rows_bcast_kernel(const T* __restrict__ in, int64_t ldi, T* __restrict__ out, int64_t ldo,
                                                              int64_t rows, int cols, int n_copies, int64_t copy_stride) {
  const int nv = cols / 8;
  const int64_t total = rows * nv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / nv;
    const int k = (int)(i - r * nv);
    const uint4 v = reinterpret_cast<const uint4*>(in + r * ldi)[k];
    for (int c = 0; c < n_copies; ++c) reinterpret_cast<uint4*>(out + c * copy_stride + r * ldo)[k] = v;
  }
}

// SigLIP patch embedding input (siglip_encoder.py:156-172): Conv2d(k=s=14) == GEMM over the flattened
// (c, ky, kx) patch; this gathers pixels [B,3,S,S] into rows [B*g*g, kpad] (zero padded to kpad).
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const T* __restrict__ img, T* __restrict__ out, int B, int S,
                                                       int P, int g, int kpad) {
  const int kk = 3 * P * P;
  const int64_t total = (int64_t)B * g * g * kpad;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % kpad);
    const int64_t row = i / kpad;
    T val = from_f32<T>(0.f);
    if (k < kk) {
      const int c = k / (P * P), rem = k - c * P * P, ky = rem / P, kx = rem - ky * P;
      const int px = (int)(row % g), py = (int)((row / g) % g);
      const int64_t bb = row / ((int64_t)g * g);
      val = img[((bb * 3 + c) * S + (py * P + ky)) * S + px * P + kx];
    }
    out[i] = val;
  }
}

}  // namespace v3d

using namespace v3d;

#define V3D_DISPATCH_16(dtype, ...)                                           \
  switch (dtype) {                                                            \
    case V3D_F16: { using T = v3d::f16_t; __VA_ARGS__; } break;               \
    case V3D_BF16: { using T = v3d::bf16_t; __VA_ARGS__; } break;             \
    default: v3d::set_error("dtype must be f16 or bf16 (got %d)", (int)(dtype)); return V3D_E_INVALID; \
  }

// RMSNorm of a few rows in the SUMMATION ORDER of v3d_linear_decode's fused norm (thread t sums chunks t and
// t + 256, wave shuffle, four wave sums in order): scenes decoding together normalise once here and then use the
// unfused linear, and get the bits the fused single-scene path produces.  One workgroup per row, cols <= 4096.
template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_decode_kernel(const T* __restrict__ x, const T* __restrict__ w, T* __restrict__ out,
                                                             int cols, int64_t ldx, int64_t ldo, float eps) {
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kv = cols / 8;
  const uint4* xr = reinterpret_cast<const uint4*>(x + blockIdx.x * ldx);
  uint4 xn[2];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int k = tid + 256 * i;
    xn[i] = k < kv ? xr[k] : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float f = vec_get<T>(xn[i], j); ss = fmaf(f, f, ss); }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
  if (lane == 0) red[wave] = ss;
  __syncthreads();
  const float r = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)cols + eps);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int k = tid + 256 * i;
    if (k < kv) {
      const uint4 wv = reinterpret_cast<const uint4*>(w)[k];
      float y[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) y[j] = vec_get<T>(wv, j) * round_to<T>(vec_get<T>(xn[i], j) * r);
      reinterpret_cast<uint4*>(out + blockIdx.x * ldo)[k] = vec_pack<T>(y);
    }
  }
}

extern "C" int v3d_rmsnorm(const void* x, int64_t ldx, const void* weight, void* out, int64_t ldo, int64_t rows, int cols,
                           float eps, int dtype, void* stream) {
  V3D_REQUIRE(x && weight && out, "v3d_rmsnorm: null pointer");
  V3D_REQUIRE(rows >= 0 && cols > 0 && cols % 8 == 0 && cols <= NORM_MAXV * 512, "v3d_rmsnorm: cols=%d unsupported", cols);
  V3D_REQUIRE(ldx % 8 == 0 && ldo % 8 == 0 && aligned16(x) && aligned16(out) && aligned16(weight), "v3d_rmsnorm: alignment");
  if (rows == 0) return V3D_OK;
  if (rows <= 4 && cols <= 4096) {     // decode rows: the fused decode linear's summation order (see the kernel)
    V3D_DISPATCH_16(dtype, hipLaunchKernelGGL(rmsnorm_decode_kernel<T>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream,
                                              (const T*)x, (const T*)weight, (T*)out, cols, ldx, ldo, eps));
    return check_launch("v3d_rmsnorm");
  }
  V3D_DISPATCH_16(dtype, hipLaunchKernelGGL(rmsnorm_kernel<T>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                                            (hipStream_t)stream, (const T*)x, (const T*)weight, (T*)out, rows, cols, ldx, ldo, eps));
  return check_launch("v3d_rmsnorm");
}

extern "C" int v3d_layernorm(const void* x, int64_t ldx, const void* weight, const void* bias, void* out, int64_t ldo,
                             int64_t rows, int cols, float eps, int dtype, void* stream) {
  V3D_REQUIRE(x && weight && bias && out, "v3d_layernorm: null pointer");
  V3D_REQUIRE(rows >= 0 && cols > 0 && cols % 8 == 0 && cols <= NORM_MAXV * 512, "v3d_layernorm: cols=%d unsupported", cols);
  V3D_REQUIRE(ldx % 8 == 0 && ldo % 8 == 0 && aligned16(x) && aligned16(out) && aligned16(weight) && aligned16(bias), "v3d_layernorm: alignment");
  if (rows == 0) return V3D_OK;
  if (cols <= 3 * 512 && rows >= 4096) {       // short rows, many of them: three rows in flight per wave
    V3D_DISPATCH_16(dtype, hipLaunchKernelGGL((layernorm_kernel<T, 3, 3>), dim3((unsigned)((rows + 11) / 12)), dim3(256), 0,
                                              (hipStream_t)stream, (const T*)x, (const T*)weight, (const T*)bias, (T*)out, rows, cols, ldx, ldo, eps));
  } else {
    V3D_DISPATCH_16(dtype, hipLaunchKernelGGL((layernorm_kernel<T, NORM_MAXV, 1>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                                              (hipStream_t)stream, (const T*)x, (const T*)weight, (const T*)bias, (T*)out, rows, cols, ldx, ldo, eps));
  }
  return check_launch("v3d_layernorm");
}

extern "C" int v3d_rope_table_build(const float* inv_freq, int head_dim, int n_pos, int dtype, void* cos_table,
                                    void* sin_table, void* stream) {
  V3D_REQUIRE(inv_freq && cos_table && sin_table && head_dim > 0 && head_dim % 16 == 0 && n_pos > 0, "v3d_rope_table_build: bad arguments");
  int64_t blocks = ((int64_t)n_pos * (head_dim / 2) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  V3D_DISPATCH_16(dtype, hipLaunchKernelGGL(rope_table_kernel<T>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream,
                                            inv_freq, head_dim / 2, n_pos, (T*)cos_table, (T*)sin_table));
  return check_launch("v3d_rope_table_build");
}

extern "C" int v3d_rope_apply(void* x, int64_t ldx, int64_t tokens, int n_heads, int head_dim, const void* cos_table,
                              const void* sin_table, int n_pos, const int32_t* positions, int pos0, int dtype, void* stream) {
  V3D_REQUIRE(x && cos_table && sin_table, "v3d_rope_apply: null pointer");
  V3D_REQUIRE(tokens >= 0 && n_heads > 0 && head_dim % 16 == 0 && ldx % 8 == 0 && aligned16(x), "v3d_rope_apply: bad shape");
  V3D_REQUIRE(positions || (pos0 >= 0 && pos0 + tokens <= n_pos), "v3d_rope_apply: positions exceed the table (%d)", n_pos);
  if (tokens == 0) return V3D_OK;
  int64_t blocks = (tokens * n_heads * (head_dim / 16) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  V3D_DISPATCH_16(dtype, hipLaunchKernelGGL(rope_kernel<T>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, (T*)x, ldx,
                                            tokens, n_heads, head_dim, (const T*)cos_table, (const T*)sin_table, positions, pos0));
  return check_launch("v3d_rope_apply");
}

extern "C" int v3d_rope_kv_store(void* qkv, int64_t ldx, int64_t tokens, int n_q_heads, int n_kv_heads, int head_dim,
                                 const void* cos_table, const void* sin_table, int n_pos, const int32_t* positions, int pos0,
                                 void* cache, int64_t ldc, const int64_t* dst_rows, int64_t row0, int dtype, void* stream) {
  V3D_REQUIRE(qkv && cos_table && sin_table && cache, "v3d_rope_kv_store: null pointer");
  V3D_REQUIRE(tokens >= 0 && n_q_heads > 0 && n_kv_heads > 0 && head_dim % 16 == 0 && ldx % 8 == 0 && ldc % 8 == 0 && aligned16(qkv) &&
                  aligned16(cache), "v3d_rope_kv_store: bad shape");
  V3D_REQUIRE(ldx >= (int64_t)(n_q_heads + 2 * n_kv_heads) * head_dim && ldc >= (int64_t)2 * n_kv_heads * head_dim,
              "v3d_rope_kv_store: row strides shorter than the heads they hold");
  V3D_REQUIRE(positions || (pos0 >= 0 && pos0 + tokens <= n_pos), "v3d_rope_kv_store: positions exceed the table (%d)", n_pos);
  V3D_REQUIRE(dst_rows || row0 >= 0, "v3d_rope_kv_store: negative cache row");
  if (tokens == 0) return V3D_OK;
  int64_t blocks = (tokens * (n_q_heads + 2 * n_kv_heads) * (head_dim / 16) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  V3D_DISPATCH_16(dtype, hipLaunchKernelGGL(rope_kv_store_kernel<T>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, (T*)qkv, ldx,
                                            tokens, n_q_heads, n_kv_heads, head_dim, (const T*)cos_table, (const T*)sin_table, positions,
                                            pos0, (T*)cache, ldc, dst_rows, row0));
  return check_launch("v3d_rope_kv_store");
}

extern "C" int v3d_add_row(void* x, int64_t ldx, const int64_t* rows, int n_rows, int C, const void* add, int dtype, void* stream) {
  V3D_REQUIRE(x && rows && add && n_rows >= 0 && C > 0 && C % 8 == 0 && ldx % 8 == 0 && aligned16(x) && aligned16(add),
              "v3d_add_row: bad arguments");
  if (n_rows == 0) return V3D_OK;
  V3D_DISPATCH_16(dtype, hipLaunchKernelGGL(add_row_kernel<T>, dim3(n_rows), dim3(256), 0, (hipStream_t)stream, (T*)x, ldx, rows, C,
                                            (const T*)add));
  return check_launch("v3d_add_row");
}

extern "C" int v3d_copy_rows(const void* in, int64_t ldi, void* out, int64_t ldo, int64_t rows, int cols, int dtype, void* stream) {
  V3D_REQUIRE(in && out && rows >= 0 && cols > 0 && cols % 8 == 0 && ldi % 8 == 0 && ldo % 8 == 0 && aligned16(in) && aligned16(out),
              "v3d_copy_rows: bad arguments");
  if (rows == 0) return V3D_OK;
  int64_t blocks = (rows * (cols / 8) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  V3D_DISPATCH_16(dtype, hipLaunchKernelGGL(copy_rows_kernel<T>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream,
                                            (const T*)in, ldi, (T*)out, ldo, rows, cols));
  return check_launch("v3d_copy_rows");
}

extern "C" int v3d_copy_rows_bcast(const void* in, int64_t ldi, void* out, int64_t ldo, int64_t rows, int cols, int n_copies,
                                   int64_t copy_stride, int dtype, void* stream) {
  V3D_REQUIRE(in && out && rows >= 0 && cols > 0 && cols % 8 == 0 && ldi % 8 == 0 && ldo % 8 == 0 && copy_stride % 8 == 0 && n_copies >= 1 &&
                  aligned16(in) && aligned16(out), "v3d_copy_rows_bcast: bad arguments");
  if (rows == 0) return V3D_OK;
  int64_t blocks = (rows * (cols / 8) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  V3D_DISPATCH_16(dtype, hipLaunchKernelGGL(copy_rows_bcast_kernel<T>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream,
                                            (const T*)in, ldi, (T*)out, ldo, rows, cols, n_copies, copy_stride));
  return check_launch("v3d_copy_rows_bcast");
}

extern "C" int v3d_patchify(const void* images, void* out, int B, int S, int patch, int kpad, int dtype, void* stream) {
  V3D_REQUIRE(images && out && B > 0 && S >= patch && patch > 0 && kpad >= 3 * patch * patch, "v3d_patchify: bad arguments");  // padding="valid": trailing S % patch pixels are unused
  const int g = S / patch;
  int64_t blocks = ((int64_t)B * g * g * kpad + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  V3D_DISPATCH_16(dtype, hipLaunchKernelGGL(patchify_kernel<T>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream,
                                            (const T*)images, (T*)out, B, S, patch, g, kpad));
  return check_launch("v3d_patchify");
}
