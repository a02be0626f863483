// Decode-step kernels (one new token against resident weights and KV cache): everything here is
// HBM-bound weight / cache streaming, so the design goal is bytes in flight, not FLOPs.
//
//   v3d_linear_decode : y = epilogue( W . f(x) ), one activation row.  f = identity or Qwen2RMSNorm
//                       (fused: the row is normalised while it is staged into LDS, so the 57 per-token
//                       norm launches disappear).  One wave streams FOUR weight rows at a time with
//                       non-temporal 16-byte loads (read-once data, keep L2 for the activations);
//                       epilogues: bias / residual / SwiGLU over tile-interleaved gate|up rows.
//   v3d_rope_kv_append: rotary on the new q and k heads + copy of k,v into the cache row, one launch.
#include "v3d_common.h"

namespace v3d {

enum { DEC_EPI_NONE = 0, DEC_EPI_BIAS = 1, DEC_EPI_RES = 2, DEC_EPI_SWIGLU = 3 };

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

__device__ __forceinline__ uint4 ldg_nt(const uint4* p) {
  uint4 r;
  r.x = __builtin_nontemporal_load(&p->x);
  r.y = __builtin_nontemporal_load(&p->y);
  r.z = __builtin_nontemporal_load(&p->z);
  r.w = __builtin_nontemporal_load(&p->w);
  return r;
}

// Workgroup = 4 waves = FOUR weight rows; the K range is split over the 256 threads (chunk k = tid + 256 i), so
// every row is streamed by all four waves at once: N/4 workgroups (1152 for the QKV linear) keep the whole chip
// loading even for the small linears, and each thread has 8 independent 16-byte loads in flight.
// <= 64 VGPRs on purpose: one wave of this kernel then fits on every SIMD beside two resident 224-register GEMM
// waves, so the decode step of one scene streams weights while another scene's prefill owns the matrix cores.
// NORM (fused Qwen2RMSNorm, modeling_qwen2.py:85-90) needs K <= 4096: the thread keeps its 2 chunks of x in
// registers between the sum-of-squares pass and the multiply pass.
template <typename T, int EPI, bool NORM>
__global__ __launch_bounds__(256, 8) void linear_decode_kernel(const T* __restrict__ x, const T* __restrict__ norm_w, float eps,
                                                               const T* __restrict__ W, int64_t ldw, const T* __restrict__ bias,
                                                               const T* __restrict__ res, T* __restrict__ out, int N, int K) {
  __shared__ float red[4][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kv = K / 8;
  int rows[4];
  int o0;
  if (EPI == DEC_EPI_SWIGLU) {
    o0 = blockIdx.x * 2;                                   // two outputs = two (gate, up) row pairs
    rows[0] = (o0 >> 6) * 128 + (o0 & 63); rows[1] = rows[0] + 64;
    rows[2] = ((o0 + 1) >> 6) * 128 + ((o0 + 1) & 63); rows[3] = rows[2] + 64;
  } else {
    o0 = blockIdx.x * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) rows[r] = o0 + r;
  }
  const uint4* xr = reinterpret_cast<const uint4*>(x);
  uint4 xn[2];
  if (NORM) {                                              // kv <= 512: chunks tid and tid + 256
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
    if (lane == 0) red[0][wave] = ss;
    __syncthreads();
    const float r = 1.0f / sqrtf((red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (float)K + eps);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int k = tid + 256 * i;
      if (k < kv) {
        const uint4 w = reinterpret_cast<const uint4*>(norm_w)[k];
        float y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = vec_get<T>(w, j) * round_to<T>(vec_get<T>(xn[i], j) * r);
        xn[i] = vec_pack<T>(y);
      }
    }
  }
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  const uint4* w0 = reinterpret_cast<const uint4*>(W + (int64_t)rows[0] * ldw);
  const uint4* w1 = reinterpret_cast<const uint4*>(W + (int64_t)rows[1] * ldw);
  const uint4* w2 = reinterpret_cast<const uint4*>(W + (int64_t)rows[2] * ldw);
  const uint4* w3 = reinterpret_cast<const uint4*>(W + (int64_t)rows[3] * ldw);
  auto fma8 = [&](const uint4& xv, const uint4& a, const uint4& b, const uint4& c, const uint4& d) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xf = vec_get<T>(xv, j);
      s[0] = fmaf(vec_get<T>(a, j), xf, s[0]);
      s[1] = fmaf(vec_get<T>(b, j), xf, s[1]);
      s[2] = fmaf(vec_get<T>(c, j), xf, s[2]);
      s[3] = fmaf(vec_get<T>(d, j), xf, s[3]);
    }
  };
  if (NORM) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int k = tid + 256 * i;
      if (k < kv) fma8(xn[i], ldg_nt(w0 + k), ldg_nt(w1 + k), ldg_nt(w2 + k), ldg_nt(w3 + k));
    }
  } else {
#pragma unroll 2
    for (int k = tid; k < kv; k += 256) fma8(xr[k], ldg_nt(w0 + k), ldg_nt(w1 + k), ldg_nt(w2 + k), ldg_nt(w3 + k));
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s[r] += __shfl_xor(s[r], off);
  }
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) red[r][wave] = s[r];
  }
  __syncthreads();
  if (EPI == DEC_EPI_SWIGLU) {
    if (tid < 2) {
      const float g = round_to<T>(red[2 * tid][0] + red[2 * tid][1] + red[2 * tid][2] + red[2 * tid][3]);
      const float u = round_to<T>(red[2 * tid + 1][0] + red[2 * tid + 1][1] + red[2 * tid + 1][2] + red[2 * tid + 1][3]);
      out[o0 + tid] = from_f32<T>(round_to<T>(silu_f(g)) * u);
    }
  } else if (tid < 4) {
    const int n = o0 + tid;
    float v = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
    if (EPI == DEC_EPI_BIAS) v += to_f32(bias[n]);
    v = round_to<T>(v);                                   // the linear's own output rounding
    if (EPI == DEC_EPI_RES) v += to_f32(res[n]);
    out[n] = from_f32<T>(v);
  }
}

// rotary (apply_rotary_pos_emb, modeling_qwen2.py:141-173) on the new token's q and k heads, in place in the
// QKV row, and append of k (rotated) and v to cache row `pos`:  cache_row = [k heads | v heads].
template <typename T>
__global__ __launch_bounds__(256) void rope_kv_append_kernel(T* __restrict__ qkv, int n_q, int n_kv, int hd,
                                                             const T* __restrict__ cos_t, const T* __restrict__ sin_t,
                                                             int pos, T* __restrict__ cache_row) {
  const int half = hd / 2, vper = half / 8;
  const int n_rot = (n_q + n_kv) * vper;            // rotation work items (pairs of 16-byte vectors)
  const int n_v = n_kv * hd / 8;                    // v copy items
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_rot + n_v; i += gridDim.x * blockDim.x) {
    if (i < n_rot) {
      const int c = i % vper, head = i / vper;
      T* base = qkv + (int64_t)head * hd + c * 8;
      const uint4 lo = *reinterpret_cast<const uint4*>(base);
      const uint4 hi = *reinterpret_cast<const uint4*>(base + half);
      const uint4 cv = *reinterpret_cast<const uint4*>(cos_t + (int64_t)pos * half + c * 8);
      const uint4 sv = *reinterpret_cast<const uint4*>(sin_t + (int64_t)pos * half + c * 8);
      float ol[8], oh[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float a = vec_get<T>(lo, j), b = vec_get<T>(hi, j), cs = vec_get<T>(cv, j), sn = vec_get<T>(sv, j);
        ol[j] = round_to<T>(a * cs) + round_to<T>(-b * sn);
        oh[j] = round_to<T>(b * cs) + round_to<T>(a * sn);
      }
      const uint4 pl = vec_pack<T>(ol), ph = vec_pack<T>(oh);
      *reinterpret_cast<uint4*>(base) = pl;
      *reinterpret_cast<uint4*>(base + half) = ph;
      if (head >= n_q) {                            // k head -> cache
        T* dst = cache_row + (int64_t)(head - n_q) * hd + c * 8;
        *reinterpret_cast<uint4*>(dst) = pl;
        *reinterpret_cast<uint4*>(dst + half) = ph;
      }
    } else {
      const int k = i - n_rot;
      reinterpret_cast<uint4*>(cache_row + (int64_t)n_kv * hd)[k] =
          reinterpret_cast<const uint4*>(qkv + (int64_t)(n_q + n_kv) * hd)[k];
    }
  }
}

// argmax over the logits row (greedy decoding: generation_utils' argmax over logits[:, -1]).  Ties resolve to the
// lowest index, as torch.argmax does.  Two launches: per-block candidates (16-byte loads), then one small block.
__device__ __forceinline__ void amax_merge(float& bv, int& bi, float ov, int oi) {
  if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
}

template <typename T>
__global__ __launch_bounds__(256) void argmax_part_kernel(const T* __restrict__ x, int n, float* __restrict__ pv, int* __restrict__ pi) {
  __shared__ float sv[4];
  __shared__ int si[4];
  float best = -INFINITY;
  int idx = 0x7fffffff;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) amax_merge(best, idx, to_f32(x[i]), i);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) amax_merge(best, idx, __shfl_xor(best, off), __shfl_xor(idx, off));
  if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = idx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) amax_merge(best, idx, sv[w], si[w]);
    pv[blockIdx.x] = best;
    pi[blockIdx.x] = idx;
  }
}

__global__ __launch_bounds__(64) void argmax_final_kernel(const float* __restrict__ pv, const int* __restrict__ pi, int nb, int64_t* __restrict__ out) {
  float best = -INFINITY;
  int idx = 0x7fffffff;
  for (int i = threadIdx.x; i < nb; i += 64) amax_merge(best, idx, pv[i], pi[i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) amax_merge(best, idx, __shfl_xor(best, off), __shfl_xor(idx, off));
  if (threadIdx.x == 0) out[0] = idx;
}

}  // namespace v3d

using namespace v3d;

extern "C" int v3d_linear_decode(const void* x, const void* norm_weight, float eps, const void* W, int64_t ldw,
                                 const void* bias, const void* res, void* out, int N, int K, int dtype, int epilogue,
                                 void* stream) {
  V3D_REQUIRE(x && W && out, "v3d_linear_decode: null pointer");
  V3D_REQUIRE(dtype == V3D_F16 || dtype == V3D_BF16, "v3d_linear_decode: dtype must be f16 or bf16");
  V3D_REQUIRE(N > 0 && K > 0 && K % 8 == 0 && ldw % 8 == 0 && ldw >= K, "v3d_linear_decode: bad shape N=%d K=%d", N, K);
  V3D_REQUIRE(N % 4 == 0 && (epilogue != DEC_EPI_SWIGLU || N % 128 == 0), "v3d_linear_decode: N=%d not supported", N);
  V3D_REQUIRE(aligned16(x) && aligned16(W) && (!norm_weight || aligned16(norm_weight)), "v3d_linear_decode: alignment");
  V3D_REQUIRE(epilogue != DEC_EPI_BIAS || bias, "v3d_linear_decode: bias epilogue without bias");
  V3D_REQUIRE(epilogue != DEC_EPI_RES || res, "v3d_linear_decode: residual epilogue without residual");
  V3D_REQUIRE(!norm_weight || K / 8 <= 512, "v3d_linear_decode: fused RMSNorm needs K <= 4096 (got %d)", K);
  const int blocks = epilogue == DEC_EPI_SWIGLU ? N / 4 : N / 4;     // 4 weight rows per workgroup either way
  hipStream_t st = (hipStream_t)stream;
#define V3D_LD(TT, EE, NN)                                                                                                 \
  hipLaunchKernelGGL((linear_decode_kernel<TT, EE, NN>), dim3(blocks), dim3(256), 0, st, (const TT*)x, (const TT*)norm_weight, \
                     eps, (const TT*)W, ldw, (const TT*)bias, (const TT*)res, (TT*)out, N, K)
#define V3D_LD_N(TT, EE) { if (norm_weight) V3D_LD(TT, EE, true); else V3D_LD(TT, EE, false); }
#define V3D_LD_E(TT)                                                                                  \
  switch (epilogue) {                                                                                 \
    case DEC_EPI_NONE: V3D_LD_N(TT, DEC_EPI_NONE) break; case DEC_EPI_BIAS: V3D_LD_N(TT, DEC_EPI_BIAS) break; \
    case DEC_EPI_RES: V3D_LD_N(TT, DEC_EPI_RES) break; case DEC_EPI_SWIGLU: V3D_LD_N(TT, DEC_EPI_SWIGLU) break; \
    default: set_error("v3d_linear_decode: unknown epilogue %d", epilogue); return V3D_E_INVALID;     \
  }
  if (dtype == V3D_BF16) { V3D_LD_E(bf16_t) } else { V3D_LD_E(f16_t) }
#undef V3D_LD_E
#undef V3D_LD_N
#undef V3D_LD
  return check_launch("v3d_linear_decode");
}

extern "C" int v3d_rope_kv_append(void* qkv_row, int n_q_heads, int n_kv_heads, int head_dim, const void* cos_table,
                                  const void* sin_table, int n_pos, int pos, void* cache_row, int dtype, void* stream) {
  V3D_REQUIRE(qkv_row && cos_table && sin_table && cache_row, "v3d_rope_kv_append: null pointer");
  V3D_REQUIRE(head_dim % 16 == 0 && pos >= 0 && pos < n_pos, "v3d_rope_kv_append: pos %d outside the table (%d)", pos, n_pos);
  V3D_REQUIRE(aligned16(qkv_row) && aligned16(cache_row), "v3d_rope_kv_append: alignment");
  const int items = (n_q_heads + n_kv_heads) * (head_dim / 16) + n_kv_heads * head_dim / 8;
  const int blocks = (items + 255) / 256;
  if (dtype == V3D_BF16)
    hipLaunchKernelGGL(rope_kv_append_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (bf16_t*)qkv_row,
                       n_q_heads, n_kv_heads, head_dim, (const bf16_t*)cos_table, (const bf16_t*)sin_table, pos, (bf16_t*)cache_row);
  else if (dtype == V3D_F16)
    hipLaunchKernelGGL(rope_kv_append_kernel<f16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (f16_t*)qkv_row,
                       n_q_heads, n_kv_heads, head_dim, (const f16_t*)cos_table, (const f16_t*)sin_table, pos, (f16_t*)cache_row);
  else { set_error("v3d_rope_kv_append: dtype must be f16 or bf16"); return V3D_E_INVALID; }
  return check_launch("v3d_rope_kv_append");
}

extern "C" int v3d_argmax(const void* x, int n, int dtype, int64_t* out_index, void* workspace, void* stream) {
  V3D_REQUIRE(x && out_index && workspace && n > 0, "v3d_argmax: bad arguments");
  constexpr int NB = 128;                                    // workspace: NB floats + NB ints = 1 KiB
  float* pv = (float*)workspace;
  int* pi = (int*)(pv + NB);
  V3D_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(argmax_part_kernel<T>, dim3(NB), dim3(256), 0, (hipStream_t)stream, (const T*)x, n, pv, pi));
  if (int e = check_launch("v3d_argmax (partial)")) return e;
  hipLaunchKernelGGL(argmax_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, pv, pi, NB, out_index);
  return check_launch("v3d_argmax");
}
