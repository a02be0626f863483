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
#include <mutex>
#include <unordered_map>

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
// M activation rows (M scenes decoding together, 1 <= M <= 4) share ONE pass over the weights: the step is
// HBM-bound on the 15 GB of weights, so M scenes cost about what one does.  Each row's arithmetic is exactly the
// M = 1 arithmetic (same chunk order, same reductions), so batching scenes never changes a result bit.
// NORM (fused Qwen2RMSNorm, modeling_qwen2.py:85-90) needs K <= 4096: the thread keeps its 2 chunks of x in
// registers between the sum-of-squares pass and the multiply pass.
template <typename T, int EPI, bool NORM, int M>
__global__ __launch_bounds__(256, M == 1 ? 8 : 4) void linear_decode_kernel(
    const T* __restrict__ x, int64_t ldx, const T* __restrict__ norm_w, float eps, const T* __restrict__ W, int64_t ldw,
    const T* __restrict__ bias, const T* __restrict__ res, int64_t ldr, T* __restrict__ out, int64_t ldo, int N, int K) {
  __shared__ float red[M][4][4];
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
  uint4 xn[M][2];
  if (NORM) {                                              // kv <= 512: chunks tid and tid + 256
    float ss[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const uint4* xr = reinterpret_cast<const uint4*>(x + m * ldx);
      ss[m] = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int k = tid + 256 * i;
        xn[m][i] = k < kv ? xr[k] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float f = vec_get<T>(xn[m][i], j); ss[m] = fmaf(f, f, ss[m]); }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) ss[m] += __shfl_xor(ss[m], off);
      if (lane == 0) red[m][0][wave] = ss[m];
    }
    __syncthreads();
    float rstd[M];
#pragma unroll
    for (int m = 0; m < M; ++m)
      rstd[m] = 1.0f / sqrtf((red[m][0][0] + red[m][0][1] + red[m][0][2] + red[m][0][3]) / (float)K + eps);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int k = tid + 256 * i;
      if (k < kv) {
        const uint4 w = reinterpret_cast<const uint4*>(norm_w)[k];
#pragma unroll
        for (int m = 0; m < M; ++m) {
          float y[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) y[j] = vec_get<T>(w, j) * round_to<T>(vec_get<T>(xn[m][i], j) * rstd[m]);
          xn[m][i] = vec_pack<T>(y);
        }
      }
    }
  }
  float s[M][4];
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) s[m][r] = 0.f;
  const uint4* w0 = reinterpret_cast<const uint4*>(W + (int64_t)rows[0] * ldw);
  const uint4* w1 = reinterpret_cast<const uint4*>(W + (int64_t)rows[1] * ldw);
  const uint4* w2 = reinterpret_cast<const uint4*>(W + (int64_t)rows[2] * ldw);
  const uint4* w3 = reinterpret_cast<const uint4*>(W + (int64_t)rows[3] * ldw);
  auto fma8 = [&](int m, const uint4& xv, const uint4& a, const uint4& b, const uint4& c, const uint4& d) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xf = vec_get<T>(xv, j);
      s[m][0] = fmaf(vec_get<T>(a, j), xf, s[m][0]);
      s[m][1] = fmaf(vec_get<T>(b, j), xf, s[m][1]);
      s[m][2] = fmaf(vec_get<T>(c, j), xf, s[m][2]);
      s[m][3] = fmaf(vec_get<T>(d, j), xf, s[m][3]);
    }
  };
  if (NORM) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int k = tid + 256 * i;
      if (k < kv) {
        const uint4 a = ldg_nt(w0 + k), b = ldg_nt(w1 + k), c = ldg_nt(w2 + k), d = ldg_nt(w3 + k);
#pragma unroll
        for (int m = 0; m < M; ++m) fma8(m, xn[m][i], a, b, c, d);
      }
    }
  } else {
#pragma unroll 2
    for (int k = tid; k < kv; k += 256) {
      const uint4 a = ldg_nt(w0 + k), b = ldg_nt(w1 + k), c = ldg_nt(w2 + k), d = ldg_nt(w3 + k);
#pragma unroll
      for (int m = 0; m < M; ++m) fma8(m, reinterpret_cast<const uint4*>(x + m * ldx)[k], a, b, c, d);
    }
  }
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s[m][r] += __shfl_xor(s[m][r], off);
    }
  if (lane == 0) {
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[m][r][wave] = s[m][r];
  }
  __syncthreads();
  if (EPI == DEC_EPI_SWIGLU) {
    if (tid < 2 * M) {
      const int m = tid >> 1, t2 = tid & 1;
      const float g = round_to<T>(red[m][2 * t2][0] + red[m][2 * t2][1] + red[m][2 * t2][2] + red[m][2 * t2][3]);
      const float u = round_to<T>(red[m][2 * t2 + 1][0] + red[m][2 * t2 + 1][1] + red[m][2 * t2 + 1][2] + red[m][2 * t2 + 1][3]);
      out[m * ldo + o0 + t2] = from_f32<T>(round_to<T>(silu_f(g)) * u);
    }
  } else if (tid < 4 * M) {
    const int m = tid >> 2, r = tid & 3;
    const int n = o0 + r;
    float v = red[m][r][0] + red[m][r][1] + red[m][r][2] + red[m][r][3];
    if (EPI == DEC_EPI_BIAS) v += to_f32(bias[n]);
    v = round_to<T>(v);                                   // the linear's own output rounding
    if (EPI == DEC_EPI_RES) v += to_f32(res[m * ldr + n]);
    out[m * ldo + n] = from_f32<T>(v);
  }
}

// Matrix-core form of the decode linear for up to 16 activation rows (scenes decoding together): the weights are the
// MFMA A operand of v_mfma_f32_16x16x32 and the activation rows the B operand (x[m][k .. k+8) for column m < M, zero
// beyond), so 1..16 scenes cost the same VALU-free pass over the weights.  A workgroup owns 16 output rows (SWIGLU:
// 16 gate + 16 up rows sharing the B fragments) and its 8 waves split K in interleaved 128-element tiles.  A wave loads
// a tile with FULL-LINE coalescing (one load instruction = 4 rows x 256 contiguous bytes; the operand layout itself -
// lane (r, g) <- W[r][k0 + 8g .. +8) - would touch 16 rows x 64 B, i.e. half lines, and measured 3.5 TB/s), parks it in
// its private 4 KB LDS tile (272-byte row pitch: conflict-free ds_write_b128 / ds_read_b128) and reads the fragments
// back; two tiles of loads stay in flight in registers.  No workgroup barrier in the loop: the tile is wave-private and
// the LDS pipeline keeps a wave's writes and reads in order.  Partial 16 x 16 tiles meet in LDS at the end.  Column m of
// the result depends on row m of x only, so a scene's numbers do not depend on its group.
using dec_f32x4 = __attribute__((ext_vector_type(4))) float;
using dec_bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using dec_f16x8 = __attribute__((ext_vector_type(8))) _Float16;
__device__ __forceinline__ dec_f32x4 dec_mfma(bf16_t, const uint4& a, const uint4& b, dec_f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(dec_bf16x8, a), __builtin_bit_cast(dec_bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ dec_f32x4 dec_mfma(f16_t, const uint4& a, const uint4& b, dec_f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(dec_f16x8, a), __builtin_bit_cast(dec_f16x8, b), c, 0, 0, 0);
}

// MB (r04): 16-row blocks of activation rows - M <= 16 MB scenes / questions share the pass over the weights (a weight fragment feeds MB
// MFMAs; the accumulators of the blocks are independent, so a row's bits depend neither on M nor on MB).
// OG (r04): 16-row groups of OUTPUTS per workgroup.  Every workgroup reads all M activation rows (each wave its K slices), so the
// activation bytes pulled out of L2 per weight byte are M : 16 OG - at M = 32 and OG = 1 twice the weight stream itself (the gate/up
// linear then ran at 3.8 TB/s of weights).  More outputs per workgroup divide that; the k order of an output's sum does not change.
template <typename T, int EPI, int MB, int OG>
__global__ __launch_bounds__(512) void linear_decode_mfma_kernel(const T* __restrict__ x, int64_t ldx, int M, const T* __restrict__ W,
                                                                 int64_t ldw, const T* __restrict__ bias, const T* __restrict__ res,
                                                                 int64_t ldr, T* __restrict__ out, int64_t ldo, int N, int K) {
  constexpr int SW = EPI == DEC_EPI_SWIGLU ? 2 : 1;         // weight row groups per output group (gate + up)
  constexpr int RG = SW * OG;                               // 16-row weight groups per workgroup: [og][gate | up]
  constexpr int UN = RG * MB >= 8 ? 1 : 2;                  // 128-element K tiles in flight per wave (8 x 16 B of weights per lane and group)
  constexpr int PITCH = 272;                                // LDS row pitch in bytes (256 + 16)
  constexpr int TILE_B = 8 * RG * 16 * PITCH, PART_B = 8 * RG * MB * 256 * 4;
  __shared__ __attribute__((aligned(16))) char smem_[TILE_B > PART_B ? TILE_B : PART_B];
  auto tile = [&](int w, int q) -> char* { return smem_ + (size_t)(w * RG + q) * 16 * PITCH; };
  auto part = [&](int w, int q, int b_) -> float* { return reinterpret_cast<float*>(smem_) + (size_t)((w * RG + q) * MB + b_) * 256; };
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;                   // operand layout: row / scene r, k group g
  const int lr = lane >> 4, lc = lane & 15;                 // load layout: row 4j + lr of the tile, 16-byte chunk lc
  const int o0 = blockIdx.x * 16 * OG;                      // first output of this workgroup
  const T* wbase[RG];                                       // row lr of the group, chunk lc
#pragma unroll
  for (int og = 0; og < OG; ++og) {
    const int o = o0 + 16 * og;
    if (EPI == DEC_EPI_SWIGLU) {
      const int gate = (o >> 6) * 128 + (o & 63);           // tile-interleaved rows: 64 gate rows, then their 64 up rows
      wbase[SW * og] = W + (int64_t)(gate + lr) * ldw + 8 * lc;
      wbase[SW * og + SW - 1] = W + (int64_t)(gate + 64 + lr) * ldw + 8 * lc;
    } else {
      wbase[SW * og] = W + (int64_t)(o + lr) * ldw + 8 * lc;
    }
  }
  bool col_ok[MB];
  const T* xrow[MB];
#pragma unroll
  for (int b = 0; b < MB; ++b) {
    col_ok[b] = r + 16 * b < M;
    xrow[b] = x + (int64_t)(col_ok[b] ? r + 16 * b : 0) * ldx + 8 * g;
  }
  dec_f32x4 acc[RG][MB];
#pragma unroll
  for (int q = 0; q < RG; ++q)
#pragma unroll
    for (int b = 0; b < MB; ++b) acc[q][b] = dec_f32x4{0.f, 0.f, 0.f, 0.f};
  const int tiles = K / 128;
  // (the tile -> wave assignment does not depend on UN / MB / OG: tile s goes to wave s % 8 and a wave sums its tiles in ascending
  //  order, so an output's f32 summation order - and with it every bit - is the same in a group of 2 and in a group of 32)
  for (int s0 = wave; s0 < tiles; s0 += 8 * UN) {
    uint4 wreg[UN][RG][4], xb[UN][MB][4];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int s = s0 + 8 * u;
      const bool ok = s < tiles;
      const int k0 = 128 * (ok ? s : s0);                   // past the end: re-read a valid tile, its x is zeroed
#pragma unroll
      for (int q = 0; q < RG; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) wreg[u][q][j] = ldg_nt(reinterpret_cast<const uint4*>(wbase[q] + (int64_t)(4 * j) * ldw + k0));
#pragma unroll
      for (int b = 0; b < MB; ++b)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          xb[u][b][t] = make_uint4(0u, 0u, 0u, 0u);
          if (col_ok[b] && ok) xb[u][b][t] = *reinterpret_cast<const uint4*>(xrow[b] + k0 + 32 * t);     // lanes of absent scenes load nothing
        }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
#pragma unroll
      for (int q = 0; q < RG; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<uint4*>(tile(wave, q) + (4 * j + lr) * PITCH + lc * 16) = wreg[u][q][j];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < RG; ++q) {
          const uint4 a = *reinterpret_cast<const uint4*>(tile(wave, q) + r * PITCH + (4 * t + g) * 16);
#pragma unroll
          for (int b = 0; b < MB; ++b) acc[q][b] = dec_mfma(T{}, a, xb[u][b][t], acc[q][b]);
        }
    }
  }
  __syncthreads();                                          // the partial sums reuse the tiles' LDS
#pragma unroll
  for (int q = 0; q < RG; ++q)
#pragma unroll
    for (int b = 0; b < MB; ++b)
      *reinterpret_cast<float4*>(part(wave, q, b) + lane * 4) = make_float4(acc[q][b][0], acc[q][b][1], acc[q][b][2], acc[q][b][3]);
  __syncthreads();
  for (int e = tid; e < 256 * MB * OG; e += 512) {          // element (row 4 (l >> 4) + i, column l & 15) of block b, group og
    const int og = e / (256 * MB), e2 = e - og * 256 * MB;
    const int b = e2 >> 8, t8 = e2 & 255;
    const int l = t8 >> 2, i = t8 & 3;
    const int row = 4 * (l >> 4) + i, m = (l & 15) + 16 * b;
    float v[SW];
#pragma unroll
    for (int q = 0; q < SW; ++q) {
      v[q] = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) v[q] += part(w, SW * og + q, b)[t8];
    }
    if (m < M) {
      const int n = o0 + 16 * og + row;
      if (EPI == DEC_EPI_SWIGLU) {
        const float gt = round_to<T>(v[0]), up = round_to<T>(v[SW - 1]);
        out[m * ldo + n] = from_f32<T>(round_to<T>(silu_f(gt)) * up);
      } else {
        float y = v[0];
        if (EPI == DEC_EPI_BIAS) y += to_f32(bias[n]);
        y = round_to<T>(y);                                 // the linear's own output rounding
        if (EPI == DEC_EPI_RES) y += to_f32(res[m * ldr + n]);
        out[m * ldo + n] = from_f32<T>(y);
      }
    }
  }
}

// ---- r04 (second session): the matrix-core decode linear as a PIPELINED stream ----------------------------------------------------
// linear_decode_mfma_kernel above loads a wave's K tile, waits for it, parks it, multiplies, and only then asks for the next one: with
// one workgroup per CU (139 KB of parking tiles at 32 outputs) the CU's memory queue runs empty once per tile, and gate/up at 32 rows
// streams 4.4 TB/s where the one-row kernel reaches 6.5.  Two kernels replace it where they are faster (same arithmetic: slice w of K =
// the tiles s = w (mod 8) summed in ascending order, the eight slices added in order 0..7, so every output bit equals the kernel above):
//  * linear_decode_mfma2_kernel (K <= 4096): PERSISTENT workgroups walk 16-output groups; a wave's activation fragments - they depend on
//    its K slice only, not on the group - are loaded ONCE and stay in registers (no activation traffic in the loop at all: the r04 form
//    pulled M : 16 OG activation bytes per weight byte out of L2); its weight tiles run through two register buffers, tile j + 2 being
//    requested as soon as tile j has been parked in LDS, across group boundaries; one barrier per group (the partial sums meet in a
//    double-buffered LDS region); bias / residual values are requested at the start of a group so that the epilogue's wait is a counted
//    one and never drains the weight stream.
//  * linear_decode_mfma_stream_kernel (any K, here down_proj's 18944): one group per workgroup as before, weights AND activations of tile
//    j + 2 requested while tile j is multiplied.
// Buffer roles are static in both (two groups / two tiles per loop trip): hipcc's vmcnt bookkeeping merges the states of joining paths,
// and a run-time buffer choice made every wait a full drain.
// A tile's life: landed in registers -> parked in the wave's LDS tile (dec2_park) -> the registers are free, the tile after next is
// requested into them AT ONCE -> fragments read back and multiplied (dec2_mult).  __builtin_amdgcn_sched_barrier pins the requests there:
// left alone, hipcc's scheduler sank them to their first use, one tile later (load, vmcnt(0), ds_write, load, vmcnt(0), ...).
__device__ __forceinline__ void dec2_park(const uint4 (&wr)[4], char* tl, int lr, int lc) {
  constexpr int PITCH = 272;
#pragma unroll
  for (int j = 0; j < 4; ++j) *reinterpret_cast<uint4*>(tl + (4 * j + lr) * PITCH + lc * 16) = wr[j];
}
template <typename T, int MB>
__device__ __forceinline__ void dec2_mult(const char* tl, int r, int g, const uint4 (&xx)[MB][4], dec_f32x4 (&acc)[MB]) {
  constexpr int PITCH = 272;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const uint4 a = *reinterpret_cast<const uint4*>(tl + r * PITCH + (4 * t + g) * 16);
#pragma unroll
    for (int b = 0; b < MB; ++b) acc[b] = dec_mfma(T{}, a, xx[b][t], acc[b]);
  }
}

template <int EPI>
__device__ __forceinline__ int dec2_row0(int grp) {                      // first weight row of 16-output group grp (SWIGLU: its gate rows)
  const int o = 16 * grp;
  return EPI == DEC_EPI_SWIGLU ? (o >> 6) * 128 + (o & 63) : o;
}

// NW = K tiles of THIS wave (the workgroup's waves hold NT or NT - 1: two instantiations of the loop, chosen once per wave)
// NORM: x holds the rows BEFORE Qwen2RMSNorm; the workgroup forms every row's 1 / rms itself - in the summation order v3d_rmsnorm uses
// for more than four rows (rmsnorm_kernel's; the host sends up to four rows elsewhere), so the bits are those of the separate launch - and
// normalises its resident fragments in registers (weight * round(x * r), element by element as the kernels do).  One launch less per
// norm (a dependent launch costs >= 4.5 us here; the 32-row norm 7.3), for one more pass over the rows out of L2 per workgroup.
template <typename T, int EPI, int MB, int NW, bool SPLIT, bool NORM = false>
__device__ __forceinline__ void dec2_run(const T* __restrict__ x, int64_t ldx, int M, const T* __restrict__ W, int64_t ldw,
                                         const T* __restrict__ bias, const T* __restrict__ res, int64_t ldr, T* __restrict__ out,
                                         int64_t ldo, int groups, char* smem_, float* __restrict__ partial,
                                         const T* __restrict__ norm_w = nullptr, float eps = 0.f, int cols = 0, float* nrm_lds = nullptr) {
  constexpr int SW = EPI == DEC_EPI_SWIGLU ? 2 : 1;
  constexpr int PITCH = 272, TILE_B = 8 * SW * 16 * PITCH, PART_F = 8 * SW * MB * 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4, lr = lane >> 4, lc = lane & 15;
  auto tile = [&](int q) -> char* { return smem_ + (size_t)(wave * SW + q) * 16 * PITCH; };
  auto part = [&](int pb, int w, int q, int b_) -> float* {
    return reinterpret_cast<float*>(smem_ + TILE_B) + (size_t)pb * PART_F + (size_t)((w * SW + q) * MB + b_) * 256;
  };
  if (NORM) {   // the rows' 1 / rms first: wave w takes rows w, w + 8, w + 16, w + 24, all their loads in flight together (one round trip)
    float* rstd = nrm_lds;                                  // [32]
    const int nv = cols / 8;
    uint4 sv[4][8];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int row = wave + 8 * rr;
      const uint4* xr = reinterpret_cast<const uint4*>(x + (int64_t)(row < M ? row : 0) * ldx);
#pragma unroll
      for (int i = 0; i < 8; ++i) { const int k = i * 64 + lane; sv[rr][i] = xr[k < nv ? k : 0]; }
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {                        // rmsnorm_kernel's order: lane l sums chunks l, l + 64, ... , then the wave
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (i * 64 + lane < nv) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { const float f = vec_get<T>(sv[rr][i], j); ss = fmaf(f, f, ss); }
        }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
      if (lane == 0 && wave + 8 * rr < M) rstd[wave + 8 * rr] = 1.0f / sqrtf(ss / (float)cols + eps);
    }
  }
  // resident activation fragments of this wave's K slice
  uint4 xb[NW][MB][4];
#pragma unroll
  for (int u = 0; u < NW; ++u)
#pragma unroll
    for (int b = 0; b < MB; ++b) {
      // rows beyond M read row 0: column m of the product depends on row m alone, and the columns beyond M are never stored
      // (no branch around a load: see issue below)
      const T* xr = x + (int64_t)(r + 16 * b < M ? r + 16 * b : 0) * ldx + 8 * g + 128 * (wave + 8 * u);
#pragma unroll
      for (int t = 0; t < 4; ++t) xb[u][b][t] = *reinterpret_cast<const uint4*>(xr + 32 * t);
    }
  if (NORM) {
    float* rstd = nrm_lds;
    uint4 wv[NW][4];                                        // the norm weights of this wave's K slice: requested before the rendezvous
#pragma unroll
    for (int u = 0; u < NW; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t) wv[u][t] = *reinterpret_cast<const uint4*>(norm_w + 8 * g + 128 * (wave + 8 * u) + 32 * t);
    __syncthreads();
    float rr[MB];
#pragma unroll
    for (int b = 0; b < MB; ++b) rr[b] = rstd[r + 16 * b < M ? r + 16 * b : 0];
#pragma unroll
    for (int u = 0; u < NW; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int b = 0; b < MB; ++b) {
          float y[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) y[j] = vec_get<T>(wv[u][t], j) * round_to<T>(vec_get<T>(xb[u][b][t], j) * rr[b]);
          xb[u][b][t] = vec_pack<T>(y);
        }
  }
  const int my_groups = (groups - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const T* wlane = W + (int64_t)lr * ldw + 8 * lc + 128 * wave;
  int ig = 0, iu = 0;                                                    // issue cursor: (group of this workgroup, tile of this wave)
  // (unconditional: past the last group the cursor re-reads the last group's tiles into a buffer nobody multiplies - a branch around
  //  the loads would join a path WITH and a path WITHOUT them, and hipcc then waits for everything at the next use)
  // A buffer is ONE 16-row tile (4 x 16 B per lane): with gate and up rows (SW = 2) the stream alternates gate tile / up tile, so the two
  // buffers hold 32 registers, not 64 (beside 128 of resident activations at 32 rows the larger form spilled).
  int iq = 0;
  auto issue = [&](uint4 (&wr)[4]) {
    const int igc = ig < my_groups ? ig : my_groups - 1;
    const T* base = wlane + (int64_t)(dec2_row0<EPI>((int)blockIdx.x + igc * (int)gridDim.x) + 64 * iq) * ldw + 1024 * iu;
#pragma unroll
    for (int j = 0; j < 4; ++j) wr[j] = ldg_nt(reinterpret_cast<const uint4*>(base + (int64_t)(4 * j) * ldw));
    if (++iq == SW) { iq = 0; if (++iu == NW) { iu = 0; ++ig; } }
  };
  dec_f32x4 acc[SW][MB];
#pragma unroll
  for (int q = 0; q < SW; ++q)
#pragma unroll
    for (int b = 0; b < MB; ++b) acc[q][b] = dec_f32x4{0.f, 0.f, 0.f, 0.f};
  // epilogue element of this thread (MB = 1: threads 0..255 only): block b, row / column of the 16 x 16 tile
  const bool epi_on = tid < 256 * MB;
  const int eb = tid >> 8, t8 = tid & 255, el = t8 >> 2, ei = t8 & 3;
  const int erow = 4 * (el >> 4) + ei, em = (el & 15) + 16 * eb;
  auto group_end = [&](int pb, int gi, float side) {
#pragma unroll
    for (int q = 0; q < SW; ++q)
#pragma unroll
      for (int b = 0; b < MB; ++b) {
        *reinterpret_cast<float4*>(part(pb, wave, q, b) + lane * 4) = make_float4(acc[q][b][0], acc[q][b][1], acc[q][b][2], acc[q][b][3]);
        acc[q][b] = dec_f32x4{0.f, 0.f, 0.f, 0.f};
      }
    // LDS-only rendezvous: __syncthreads() is a workgroup-scope fence, i.e. s_waitcnt vmcnt(0) - it drained the weight stream every group
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (epi_on) {
      float v[SW];
#pragma unroll
      for (int q = 0; q < SW; ++q) {
        v[q] = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v[q] += part(pb, w, q, eb)[t8];
      }
      if (em < M) {
        const int n = 16 * ((int)blockIdx.x + gi * (int)gridDim.x) + erow;
        if (SPLIT) {                                        // this K chunk's f32 sum: decode_combine_kernel adds the chunks in order
          partial[(int64_t)em * (16 * groups) + n] = v[0];
        } else if (EPI == DEC_EPI_SWIGLU) {
          const float gt = round_to<T>(v[0]), up = round_to<T>(v[SW - 1]);
          out[em * ldo + n] = from_f32<T>(round_to<T>(silu_f(gt)) * up);
        } else {
          float y = v[0];
          if (EPI == DEC_EPI_BIAS) y += side;
          y = round_to<T>(y);                               // the linear's own output rounding
          if (EPI == DEC_EPI_RES) y += side;
          out[em * ldo + n] = from_f32<T>(y);
        }
      }
    }
  };
  auto side_load = [&](int gi) -> float {                  // requested a whole group ahead of its use; unconditional (rows beyond M read row 0)
    const int n = 16 * ((int)blockIdx.x + gi * (int)gridDim.x) + erow;
    if (EPI == DEC_EPI_BIAS && !SPLIT) return to_f32(bias[n]);
    if (EPI == DEC_EPI_RES && !SPLIT) return to_f32(res[(int64_t)(em < M ? em : 0) * ldr + n]);
    return 0.f;
  };
  uint4 wa[4], wb[4];
  issue(wa);
  issue(wb);
  auto one_group = [&](int h, int gi) {                     // h = parity of the group within its pair: buffer roles and LDS half are static
    const float side = side_load(gi);
#pragma unroll
    for (int u = 0; u < NW; ++u)
#pragma unroll
      for (int q = 0; q < SW; ++q) {
        if ((((h * NW + u) * SW + q) & 1) == 0) {
          dec2_park(wa, tile(q), lr, lc);
          __builtin_amdgcn_sched_barrier(0);
          issue(wa);
        } else {
          dec2_park(wb, tile(q), lr, lc);
          __builtin_amdgcn_sched_barrier(0);
          issue(wb);
        }
        __builtin_amdgcn_sched_barrier(0);
        dec2_mult<T, MB>(tile(q), r, g, xb[u], acc[q]);
        __builtin_amdgcn_sched_barrier(0);
      }
    group_end(h, gi, side);
  };
  int gi = 0;
  for (; gi + 1 < my_groups; gi += 2) {                     // whole pairs: no branch inside (see issue)
    one_group(0, gi);
    one_group(1, gi + 1);
  }
  if (gi < my_groups) one_group(0, gi);
}

template <typename T, int EPI, int MB, int NT, bool NORM = false>
__global__ __launch_bounds__(512) void linear_decode_mfma2_kernel(const T* __restrict__ x, int64_t ldx, int M, const T* __restrict__ W,
                                                                  int64_t ldw, const T* __restrict__ bias, const T* __restrict__ res,
                                                                  int64_t ldr, T* __restrict__ out, int64_t ldo, int K, int groups,
                                                                  const T* __restrict__ norm_w, float eps) {
  constexpr int SW = EPI == DEC_EPI_SWIGLU ? 2 : 1;
  __shared__ __attribute__((aligned(16))) char smem_[8 * SW * 16 * 272 + 2 * 8 * SW * MB * 256 * 4];
  __shared__ float nrm_[NORM ? 48 : 1];
  const int wave = threadIdx.x >> 6;
  const int nt_w = (K / 128 - wave + 7) / 8;               // NT or NT - 1 (host: 8 (NT - 1) < tiles <= 8 NT)
  if (nt_w == NT) dec2_run<T, EPI, MB, NT, false, NORM>(x, ldx, M, W, ldw, bias, res, ldr, out, ldo, groups, smem_, nullptr, norm_w, eps, K, nrm_);
  else dec2_run<T, EPI, MB, (NT > 1 ? NT - 1 : 1), false, NORM>(x, ldx, M, W, ldw, bias, res, ldr, out, ldo, groups, smem_, nullptr, norm_w, eps, K, nrm_);
}

// K > 4096 (down_proj, K = 18944): the activation fragments of a whole row of tiles do not fit the registers, and streamed per tile they
// are M : 16 activation bytes per weight byte out of L2 (the r04 kernel: 3.0 TB/s of weights at 32 rows).  So K is CUT into gridDim.y
// chunks of hi or hi - 1 tiles (the first n_hi chunks hold hi): workgroup (i, c) keeps chunk c's fragments resident and walks its groups
// exactly as above, but leaves the chunk's f32 sums in partial[c][m][n]; decode_combine_kernel adds the chunks in ascending order and
// applies the epilogue.  The cut depends on K, N and the chip only - never on M - so a row's bits still do not depend on its group.
template <typename T, int MB, int NT>
__global__ __launch_bounds__(512) void linear_decode_mfma2_split_kernel(const T* __restrict__ x, int64_t ldx, int M, const T* __restrict__ W,
                                                                        int64_t ldw, float* __restrict__ partial, int groups, int hi, int n_hi) {
  __shared__ __attribute__((aligned(16))) char smem_[8 * 16 * 272 + 2 * 8 * MB * 256 * 4];
  const int wave = threadIdx.x >> 6, c = blockIdx.y;
  const int t0 = c < n_hi ? c * hi : n_hi * hi + (c - n_hi) * (hi - 1), ct = c < n_hi ? hi : hi - 1;
  const int nt_w = (ct - wave + 7) / 8;                    // NT or NT - 1 (host: hi >= 9, NT = ceil(hi / 8))
  x += 128 * t0;
  W += 128 * t0;
  partial += (int64_t)c * (16 * MB) * (16 * groups);
  if (nt_w == NT) dec2_run<T, DEC_EPI_NONE, MB, NT, true>(x, ldx, M, W, ldw, nullptr, nullptr, 0, nullptr, 0, groups, smem_, partial);
  else dec2_run<T, DEC_EPI_NONE, MB, (NT > 1 ? NT - 1 : 1), true>(x, ldx, M, W, ldw, nullptr, nullptr, 0, nullptr, 0, groups, smem_, partial);
}

template <typename T, int EPI>
__global__ __launch_bounds__(256) void decode_combine_kernel(const float* __restrict__ partial, int chunks, int rows_pad, int M, int N,
                                                             const T* __restrict__ bias, const T* __restrict__ res, int64_t ldr,
                                                             T* __restrict__ out, int64_t ldo) {
  const int n = blockIdx.x * 256 + threadIdx.x, m = blockIdx.y;
  if (n >= N || m >= M) return;
  float y = 0.f;
  for (int c = 0; c < chunks; ++c) y += partial[((int64_t)c * rows_pad + m) * N + n];
  if (EPI == DEC_EPI_BIAS) y += to_f32(bias[n]);
  y = round_to<T>(y);                                       // the linear's own output rounding
  if (EPI == DEC_EPI_RES) y += to_f32(res[m * ldr + n]);
  out[m * ldo + n] = from_f32<T>(y);
}

// any K: one 16-output group per workgroup, weights and activations of tile j + 2 in flight while tile j is multiplied
template <typename T, int EPI, int MB>
__global__ __launch_bounds__(512) void linear_decode_mfma_stream_kernel(const T* __restrict__ x, int64_t ldx, int M, const T* __restrict__ W,
                                                                        int64_t ldw, const T* __restrict__ bias, const T* __restrict__ res,
                                                                        int64_t ldr, T* __restrict__ out, int64_t ldo, int K) {
  constexpr int SW = EPI == DEC_EPI_SWIGLU ? 2 : 1;
  constexpr int PITCH = 272, TILE_B = 8 * SW * 16 * PITCH, PART_B = 8 * SW * MB * 256 * 4;
  __shared__ __attribute__((aligned(16))) char smem_[TILE_B > PART_B ? TILE_B : PART_B];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4, lr = lane >> 4, lc = lane & 15;
  auto tile = [&](int q) -> char* { return smem_ + (size_t)(wave * SW + q) * 16 * PITCH; };
  auto part = [&](int w, int q, int b_) -> float* { return reinterpret_cast<float*>(smem_) + (size_t)((w * SW + q) * MB + b_) * 256; };
  const int tiles = K / 128;
  const int nt_w = tiles > wave ? (tiles - wave + 7) / 8 : 0;
  const T* wlane = W + (int64_t)(dec2_row0<EPI>((int)blockIdx.x) + lr) * ldw + 8 * lc + 128 * wave;
  const T* xlane[MB];
  bool col_ok[MB];
#pragma unroll
  for (int b = 0; b < MB; ++b) {
    col_ok[b] = r + 16 * b < M;
    xlane[b] = x + (int64_t)(col_ok[b] ? r + 16 * b : 0) * ldx + 8 * g + 128 * wave;
  }
  dec_f32x4 acc[SW][MB];
#pragma unroll
  for (int q = 0; q < SW; ++q)
#pragma unroll
    for (int b = 0; b < MB; ++b) acc[q][b] = dec_f32x4{0.f, 0.f, 0.f, 0.f};
  // unconditional (see linear_decode_mfma2_kernel): past the end, the last tile again
  auto issue_w = [&](uint4 (&wr)[SW][4], int u_) {
    const int u = u_ < nt_w ? u_ : nt_w - 1;
#pragma unroll
    for (int q = 0; q < SW; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) wr[q][j] = ldg_nt(reinterpret_cast<const uint4*>(wlane + (int64_t)(64 * q + 4 * j) * ldw + 1024 * u));
  };
  auto issue_x = [&](uint4 (&xr)[MB][4], int u_) {
    const int u = u_ < nt_w ? u_ : nt_w - 1;
#pragma unroll
    for (int b = 0; b < MB; ++b)
#pragma unroll
      for (int t = 0; t < 4; ++t) xr[b][t] = *reinterpret_cast<const uint4*>(xlane[b] + 1024 * u + 32 * t);   // absent rows: row 0, zeroed at use
  };
  auto step = [&](uint4 (&wr)[SW][4], uint4 (&xr)[MB][4], int u_next) {   // one tile: park, re-request the weights, multiply, re-request the rows
#pragma unroll
    for (int q = 0; q < SW; ++q) dec2_park(wr[q], tile(q), lr, lc);
    __builtin_amdgcn_sched_barrier(0);
    issue_w(wr, u_next);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int b = 0; b < MB; ++b)
      if (!col_ok[b]) {
#pragma unroll
        for (int t = 0; t < 4; ++t) xr[b][t] = make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
    for (int q = 0; q < SW; ++q) dec2_mult<T, MB>(tile(q), r, g, xr, acc[q]);
    __builtin_amdgcn_sched_barrier(0);
    issue_x(xr, u_next);
    __builtin_amdgcn_sched_barrier(0);
  };
  const bool epi_on = tid < 256 * MB;
  const int eb = tid >> 8, t8 = tid & 255, el = t8 >> 2, ei = t8 & 3;
  const int erow = 4 * (el >> 4) + ei, em = (el & 15) + 16 * eb;
  const int n = 16 * (int)blockIdx.x + erow;
  uint4 wa[SW][4], wb[SW][4], xa[MB][4], xb[MB][4];
  issue_w(wa, 0); issue_x(xa, 0);
  issue_w(wb, 1); issue_x(xb, 1);
  float side = 0.f;                                         // older than every later tile request: its wait is a counted one
  if (epi_on && em < M) {
    if (EPI == DEC_EPI_BIAS) side = to_f32(bias[n]);
    if (EPI == DEC_EPI_RES) side = to_f32(res[em * ldr + n]);
  }
  int u = 0;
  for (; u + 1 < nt_w; u += 2) {                            // whole pairs: no branch inside
    step(wa, xa, u + 2);
    step(wb, xb, u + 3);
  }
  if (u < nt_w) step(wa, xa, u);
  __syncthreads();                                          // the partial sums reuse the tiles' LDS
#pragma unroll
  for (int q = 0; q < SW; ++q)
#pragma unroll
    for (int b = 0; b < MB; ++b)
      *reinterpret_cast<float4*>(part(wave, q, b) + lane * 4) = make_float4(acc[q][b][0], acc[q][b][1], acc[q][b][2], acc[q][b][3]);
  __syncthreads();
  if (epi_on) {
    float v[SW];
#pragma unroll
    for (int q = 0; q < SW; ++q) {
      v[q] = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) v[q] += part(w, q, eb)[t8];
    }
    if (em < M) {
      if (EPI == DEC_EPI_SWIGLU) {
        const float gt = round_to<T>(v[0]), up = round_to<T>(v[SW - 1]);
        out[em * ldo + n] = from_f32<T>(round_to<T>(silu_f(gt)) * up);
      } else {
        float y = v[0];
        if (EPI == DEC_EPI_BIAS) y += side;
        y = round_to<T>(y);
        if (EPI == DEC_EPI_RES) y += side;
        out[em * ldo + n] = from_f32<T>(y);
      }
    }
  }
}

// The same weight-streaming linear over OCP e4m3 weights (BASELINE configs[3]): W8 [N, K] bytes with one f32 scale per
// output row (v3d_quantize_fp8_rows), activations stay 16-bit (W8A16): y[n] = sw[n] * sum_k q[n,k] x[k].  Half the
// bytes of the 16-bit kernel per step; a 16-byte chunk is 16 weights (two x vectors).  No fused norm (callers
// normalise with v3d_rmsnorm first).
template <typename T, int EPI, int M>
__global__ __launch_bounds__(256, 4) void linear_decode_fp8_kernel(
    const T* __restrict__ x, int64_t ldx, const uint8_t* __restrict__ W, int64_t ldw, const float* __restrict__ sw,
    const T* __restrict__ bias, const T* __restrict__ res, int64_t ldr, T* __restrict__ out, int64_t ldo, int N, int K) {
  __shared__ float red[M][4][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kv = K / 16;
  int rows[4];
  int o0;
  if (EPI == DEC_EPI_SWIGLU) {
    o0 = blockIdx.x * 2;
    rows[0] = (o0 >> 6) * 128 + (o0 & 63); rows[1] = rows[0] + 64;
    rows[2] = ((o0 + 1) >> 6) * 128 + ((o0 + 1) & 63); rows[3] = rows[2] + 64;
  } else {
    o0 = blockIdx.x * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) rows[r] = o0 + r;
  }
  float s[M][4];
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) s[m][r] = 0.f;
  const uint4* wr[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) wr[r] = reinterpret_cast<const uint4*>(W + (int64_t)rows[r] * ldw);
#pragma unroll 2
  for (int k = tid; k < kv; k += 256) {
    uint4 w4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) w4[r] = ldg_nt(wr[r] + k);
    float xf[M][16];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const uint4* xr = reinterpret_cast<const uint4*>(x + m * ldx) + 2 * k;
      const uint4 x0 = xr[0], x1 = xr[1];
#pragma unroll
      for (int j = 0; j < 8; ++j) { xf[m][j] = vec_get<T>(x0, j); xf[m][8 + j] = vec_get<T>(x1, j); }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t ww[4] = {w4[r].x, w4[r].y, w4[r].z, w4[r].w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const auto lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)ww[q], false);     // bytes 0,1
        const auto hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)ww[q], true);      // bytes 2,3
#pragma unroll
        for (int m = 0; m < M; ++m) {
          s[m][r] = fmaf(lo[0], xf[m][4 * q + 0], s[m][r]);
          s[m][r] = fmaf(lo[1], xf[m][4 * q + 1], s[m][r]);
          s[m][r] = fmaf(hi[0], xf[m][4 * q + 2], s[m][r]);
          s[m][r] = fmaf(hi[1], xf[m][4 * q + 3], s[m][r]);
        }
      }
    }
  }
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s[m][r] += __shfl_xor(s[m][r], off);
    }
  if (lane == 0) {
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[m][r][wave] = s[m][r];
  }
  __syncthreads();
  if (EPI == DEC_EPI_SWIGLU) {
    if (tid < 2 * M) {
      const int m = tid >> 1, t2 = tid & 1;
      const float g = round_to<T>((red[m][2 * t2][0] + red[m][2 * t2][1] + red[m][2 * t2][2] + red[m][2 * t2][3]) * sw[rows[2 * t2]]);
      const float u = round_to<T>((red[m][2 * t2 + 1][0] + red[m][2 * t2 + 1][1] + red[m][2 * t2 + 1][2] + red[m][2 * t2 + 1][3]) * sw[rows[2 * t2 + 1]]);
      out[m * ldo + o0 + t2] = from_f32<T>(round_to<T>(silu_f(g)) * u);
    }
  } else if (tid < 4 * M) {
    const int m = tid >> 2, r = tid & 3;
    const int n = o0 + r;
    float v = (red[m][r][0] + red[m][r][1] + red[m][r][2] + red[m][r][3]) * sw[n];
    if (EPI == DEC_EPI_BIAS) v += to_f32(bias[n]);
    v = round_to<T>(v);
    if (EPI == DEC_EPI_RES) v += to_f32(res[m * ldr + n]);
    out[m * ldo + n] = from_f32<T>(v);
  }
}

// rotary (apply_rotary_pos_emb, modeling_qwen2.py:141-173) on the new token's q and k heads, in place in the
// QKV row, and append of k (rotated) and v to cache row `pos`:  cache_row = [k heads | v heads].
constexpr int DEC_MAX_ROWS = 32;     // scenes / questions decoding together (r04: 32; the attention side: DEC_MAXROWS in attention.hip)
struct RopeRows {        // scenes decoding together (blockIdx.y = scene): own position and cache row
  int pos[DEC_MAX_ROWS];
  void* cache_row[DEC_MAX_ROWS];
  int64_t qkv_stride;
};

template <typename T>
__global__ __launch_bounds__(256) void rope_kv_append_kernel(T* __restrict__ qkv, int n_q, int n_kv, int hd,
                                                             const T* __restrict__ cos_t, const T* __restrict__ sin_t,
                                                             RopeRows rw) {
  const int pos = rw.pos[blockIdx.y];
  T* __restrict__ cache_row = (T*)rw.cache_row[blockIdx.y];
  qkv += blockIdx.y * rw.qkv_stride;
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
__global__ __launch_bounds__(256) void argmax_part_kernel(const T* __restrict__ x, int64_t ldx, int n, float* __restrict__ pv, int* __restrict__ pi) {
  __shared__ float sv[4];
  __shared__ int si[4];
  x += blockIdx.y * ldx;                     // blockIdx.y = row; its candidates live in workspace slice y (256 words)
  pv += blockIdx.y * 256;
  pi += blockIdx.y * 256;
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
  pv += blockIdx.x * 256;
  pi += blockIdx.x * 256;
  out += blockIdx.x;
  float best = -INFINITY;
  int idx = 0x7fffffff;
  for (int i = threadIdx.x; i < nb; i += 64) amax_merge(best, idx, pv[i], pi[i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) amax_merge(best, idx, __shfl_xor(best, off), __shfl_xor(idx, off));
  if (threadIdx.x == 0) out[0] = idx;
}

}  // namespace v3d

using namespace v3d;

// f32 partial sums of the K-split decode linear: one buffer per stream (launches on one stream are ordered; two streams never share it),
// grown when a larger product arrives (the old block is released once the stream has drained).
static float* decode_split_workspace(hipStream_t st, size_t bytes) {
  struct Block { void* p; size_t n; };
  static std::mutex mu;
  static std::unordered_map<hipStream_t, Block> blocks;
  std::lock_guard<std::mutex> lock(mu);
  Block& b = blocks[st];
  if (b.n < bytes) {
    if (b.p) { (void)hipStreamSynchronize(st); (void)hipFree(b.p); b.p = nullptr; b.n = 0; }
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    b.p = p; b.n = bytes;
  }
  return static_cast<float*>(b.p);
}

// Does a product with n_out outputs over K take the persistent matrix-core form (linear_decode_mfma2_kernel), and on how many workgroups?
// V3D_DEC_V2 (read per call: the tests switch it inside one process) = 0: never; 1 (default) / 2: where the shape allows (9..32 K tiles:
// 2..4 per wave, none without one); 3: the streaming form instead (tests, A/B).  The grid is balanced: ceil(groups / rounds) workgroups
// with rounds = ceil(groups / CUs), so every workgroup walks the same number of groups (288 groups: 144 x 2, not 224 x 1 + 32 x 2).
static bool dec_v2_plan(int n_out, int K, int* grid, int* v2_env_out, int* cus_out) {
  static int cus = 0;
  const char* v2e = getenv("V3D_DEC_V2");
  const int v2_env = v2e ? atoi(v2e) : 1;
  if (cus == 0) {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  if (v2_env_out) *v2_env_out = v2_env;
  if (cus_out) *cus_out = cus;
  const int tiles = K / 128, groups16 = n_out / 16;
  if (K % 128 != 0 || n_out % 16 != 0 || groups16 < 1) return false;
  const int rounds = (groups16 + cus - 1) / cus;
  if (grid) *grid = (groups16 + rounds - 1) / rounds;
  return v2_env >= 1 && v2_env <= 2 && tiles >= 9 && tiles <= 32;
}

static int linear_decode_rows(const void* x, int64_t ldx, int M, const void* norm_weight, float eps, const void* W, int64_t ldw,
                              const void* bias, const void* res, int64_t ldr, void* out, int64_t ldo, int N, int K, int dtype,
                              int epilogue, void* stream, const char* who) {
  V3D_REQUIRE(x && W && out, "%s: null pointer", who);
  V3D_REQUIRE(dtype == V3D_F16 || dtype == V3D_BF16, "%s: dtype must be f16 or bf16", who);
  V3D_REQUIRE(N > 0 && K > 0 && K % 8 == 0 && ldw % 8 == 0 && ldw >= K, "%s: bad shape N=%d K=%d", who, N, K);
  // one row: the VALU form (fastest for a single scene, fuses the norm); 2..16 rows: the matrix-core form, whose cost does
  // not grow with M and whose columns are independent - a scene's bits are the same in every group of two or more.
  // (Against the one-row form the values differ by the f32 summation order.)  Shapes the matrix-core form does not take
  // fall back to the VALU form with up to 4 rows.
  // r04: with norm_weight and MORE than four rows the persistent matrix-core form normalises its resident fragments itself (where that
  // form applies: v3d_linear_decode_rows_fuses_norm); up to four rows keep the VALU form, whose rows equal the one-row kernel's bit for bit.
  int v2_grid = 0, v2_env = 1, cus = 256;
  const bool mfma_shape = K % 128 == 0 && (epilogue == DEC_EPI_SWIGLU ? N % 128 == 0 : N % 16 == 0);
  const bool v2_ok = mfma_shape && dec_v2_plan(epilogue == DEC_EPI_SWIGLU ? N / 2 : N, K, &v2_grid, &v2_env, &cus);
  const bool fused_norm = norm_weight && M > 4 && v2_ok && epilogue != DEC_EPI_RES;
  const bool mfma_ok = mfma_shape && (!norm_weight || fused_norm);
  const bool mfma = mfma_ok && M >= 2;
  V3D_REQUIRE(M >= 1 && M <= (mfma_ok ? DEC_MAX_ROWS : 4), "%s: 1 to %d activation rows for this shape (got %d)", who, mfma_ok ? DEC_MAX_ROWS : 4, M);
  V3D_REQUIRE(N % 4 == 0 && (epilogue != DEC_EPI_SWIGLU || N % 128 == 0), "%s: N=%d not supported", who, N);
  V3D_REQUIRE(aligned16(x) && aligned16(W) && (!norm_weight || aligned16(norm_weight)), "%s: alignment", who);
  V3D_REQUIRE(M == 1 || (ldx % 8 == 0 && ldx >= K), "%s: activation row stride %lld", who, (long long)ldx);
  V3D_REQUIRE(epilogue != DEC_EPI_BIAS || bias, "%s: bias epilogue without bias", who);
  V3D_REQUIRE(epilogue != DEC_EPI_RES || res, "%s: residual epilogue without residual", who);
  V3D_REQUIRE(!norm_weight || K / 8 <= 512, "%s: fused RMSNorm needs K <= 4096 (got %d)", who, K);
  hipStream_t st = (hipStream_t)stream;
  if (mfma) {
    V3D_REQUIRE(aligned16(W) && (M == 1 || ldx % 8 == 0), "%s: alignment", who);
    const int n_out = epilogue == DEC_EPI_SWIGLU ? N / 2 : N;
    // outputs per workgroup: 16 (r03) or 32.  32 halves the activation bytes every workgroup pulls out of L2 per weight byte, but also the
    // number of workgroups that stream.  Measured (tools/time_decode_rows.py, M = 16 / 32 rows, us; profiles/r04_decode_rows.txt):
    //   LM head  152064 x 3584: 228 -> 188 / 330 -> 260      gate/up 37888 x 3584: 56 -> 60 / 70.5 -> 62      qkv 4608 x 3584: 14.0 -> 11.4 / 20.3 -> 15.4
    //   o_proj 3584 x 3584: 9.3 -> 11.2 / 11.7 -> 14.3      down 3584 x 18944: 32.6 -> 42.1 / 45.5 -> 57.1   (112 workgroups are too few)
    // hence: 32 outputs where at least 128 workgroups remain, except in the 256..1023-workgroup range at M <= 16.  V3D_DEC_OG=1 / 3: never / always.
    // r04, second session: the pipelined forms (linear_decode_mfma2_kernel / _stream_kernel above; bit-identical outputs).  V3D_DEC_V2 = 0: never;
    // 1 (default): where measured faster (tools/time_decode_rows.py, profiles/r04_decode_rows.txt); 2: wherever the shape allows (tests);
    // 3: the streaming form wherever the shape allows (tests, A/B).
    const int tiles = K / 128, groups16 = n_out / 16, nt = (tiles + 7) / 8;
    if (v2_ok) {
      const int grid = v2_grid;
      const bool nrm = fused_norm;
#define V3D_LD2_L(TT, EE, BB, NN, RR) hipLaunchKernelGGL((linear_decode_mfma2_kernel<TT, EE, BB, NN, RR>), dim3(grid), dim3(512), 0, st, (const TT*)x, ldx, M, \
                                                     (const TT*)W, ldw, (const TT*)bias, (const TT*)res, ldr, (TT*)out, ldo, K, groups16, (const TT*)norm_weight, eps)
#define V3D_LD2_K(TT, EE, BB, NN) do { if (nrm && EE != DEC_EPI_RES) V3D_LD2_L(TT, (EE == DEC_EPI_RES ? DEC_EPI_NONE : EE), BB, NN, true); else V3D_LD2_L(TT, EE, BB, NN, false); } while (0)
#define V3D_LD2_N(TT, EE, BB) { if (nt == 2) V3D_LD2_K(TT, EE, BB, 2); else if (nt == 3) V3D_LD2_K(TT, EE, BB, 3); else V3D_LD2_K(TT, EE, BB, 4); }
#define V3D_LD2_B(TT, EE) { if (M <= 16) V3D_LD2_N(TT, EE, 1) else V3D_LD2_N(TT, EE, 2) }
#define V3D_LD2_E(TT)                                                                                 \
  switch (epilogue) {                                                                                 \
    case DEC_EPI_NONE: V3D_LD2_B(TT, DEC_EPI_NONE); break; case DEC_EPI_BIAS: V3D_LD2_B(TT, DEC_EPI_BIAS); break; \
    case DEC_EPI_RES: V3D_LD2_B(TT, DEC_EPI_RES); break; case DEC_EPI_SWIGLU: V3D_LD2_B(TT, DEC_EPI_SWIGLU); break; \
    default: set_error("%s: unknown epilogue %d", who, epilogue); return V3D_E_INVALID;               \
  }
      if (dtype == V3D_BF16) { V3D_LD2_E(bf16_t) } else { V3D_LD2_E(f16_t) }
#undef V3D_LD2_E
#undef V3D_LD2_B
#undef V3D_LD2_N
#undef V3D_LD2_K
#undef V3D_LD2_L
      return check_launch(who);
    }
    if (v2_env >= 1 && v2_env <= 2 && tiles > 32 && epilogue != DEC_EPI_SWIGLU) {        // K cut into chunks whose activation fragments stay in registers
      int C = 0, hi = 0;
      double best = 0.0;
      for (int c = (tiles + 31) / 32; c <= 16; ++c) {                     // whole rounds of groups on as many CUs as possible
        const int h = (tiles + c - 1) / c;
        if (h > 32 || h < 9) continue;
        int G = cus / c;
        if (G < 1) break;
        if (G > groups16) G = groups16;
        const double eff = ((double)groups16 / G) / ((groups16 + G - 1) / G) * (double)(c * G) / cus;
        if (eff > best + 1e-9) { best = eff; C = c; hi = h; }
      }
      if (C > 0) {
        const int n_hi = tiles - C * (hi - 1), nt2 = (hi + 7) / 8, mb = M <= 16 ? 1 : 2;
        int G = cus / C;
        if (G > groups16) G = groups16;
        float* ws = decode_split_workspace(st, (size_t)C * 16 * mb * n_out * sizeof(float));
        if (!ws) { set_error("%s: no workspace for the K-split partial sums", who); return V3D_E_LAUNCH; }
#define V3D_LDK_K(TT, BB, NN) hipLaunchKernelGGL((linear_decode_mfma2_split_kernel<TT, BB, NN>), dim3(G, C), dim3(512), 0, st, (const TT*)x, ldx, M, \
                                                 (const TT*)W, ldw, ws, groups16, hi, n_hi)
#define V3D_LDK_N(TT, BB) { if (nt2 == 2) V3D_LDK_K(TT, BB, 2); else if (nt2 == 3) V3D_LDK_K(TT, BB, 3); else V3D_LDK_K(TT, BB, 4); }
#define V3D_LDK_C(TT, EE) hipLaunchKernelGGL((decode_combine_kernel<TT, EE>), dim3((n_out + 255) / 256, M), dim3(256), 0, st, ws, C, 16 * mb, M, n_out, \
                                             (const TT*)bias, (const TT*)res, ldr, (TT*)out, ldo)
#define V3D_LDK_T(TT) { if (mb == 1) V3D_LDK_N(TT, 1) else V3D_LDK_N(TT, 2)                                                         \
                        if (epilogue == DEC_EPI_BIAS) V3D_LDK_C(TT, DEC_EPI_BIAS); else if (epilogue == DEC_EPI_RES) V3D_LDK_C(TT, DEC_EPI_RES); \
                        else V3D_LDK_C(TT, DEC_EPI_NONE); }
        if (dtype == V3D_BF16) V3D_LDK_T(bf16_t) else V3D_LDK_T(f16_t)
#undef V3D_LDK_T
#undef V3D_LDK_C
#undef V3D_LDK_N
#undef V3D_LDK_K
        return check_launch(who);
      }
    }
    if (v2_env >= 2 && tiles >= 8) {     // the streaming form ties the r04 kernel on down_proj (47 vs 45.5 us at 32 rows): tests and A/B only
                                         // (every wave needs a tile: the unconditional requests clamp to its last one)
#define V3D_LDS_B(TT, EE) { if (M <= 16) hipLaunchKernelGGL((linear_decode_mfma_stream_kernel<TT, EE, 1>), dim3(groups16), dim3(512), 0, st, (const TT*)x, ldx, M, \
                                                     (const TT*)W, ldw, (const TT*)bias, (const TT*)res, ldr, (TT*)out, ldo, K);                            \
                            else hipLaunchKernelGGL((linear_decode_mfma_stream_kernel<TT, EE, 2>), dim3(groups16), dim3(512), 0, st, (const TT*)x, ldx, M,     \
                                                     (const TT*)W, ldw, (const TT*)bias, (const TT*)res, ldr, (TT*)out, ldo, K); }
#define V3D_LDS_E(TT)                                                                                 \
  switch (epilogue) {                                                                                 \
    case DEC_EPI_NONE: V3D_LDS_B(TT, DEC_EPI_NONE); break; case DEC_EPI_BIAS: V3D_LDS_B(TT, DEC_EPI_BIAS); break; \
    case DEC_EPI_RES: V3D_LDS_B(TT, DEC_EPI_RES); break; case DEC_EPI_SWIGLU: V3D_LDS_B(TT, DEC_EPI_SWIGLU); break; \
    default: set_error("%s: unknown epilogue %d", who, epilogue); return V3D_E_INVALID;               \
  }
      if (dtype == V3D_BF16) { V3D_LDS_E(bf16_t) } else { V3D_LDS_E(f16_t) }
#undef V3D_LDS_E
#undef V3D_LDS_B
      return check_launch(who);
    }
    static int og_env = -1;
    if (og_env < 0) { const char* e = getenv("V3D_DEC_OG"); og_env = e ? atoi(e) : 2; }
    const int wg32 = n_out / 32;
    const bool og_rule = wg32 >= 128 && !(wg32 >= 256 && wg32 < 1024 && M <= 16);
    const int og = (og_env >= 2 && n_out % 32 == 0 && (og_env >= 3 || og_rule)) ? 2 : 1;
    const int mblocks = n_out / (16 * og);
#define V3D_LDM_B(TT, EE, BB, GG) hipLaunchKernelGGL((linear_decode_mfma_kernel<TT, EE, BB, GG>), dim3(mblocks), dim3(512), 0, st, (const TT*)x, ldx, M, \
                                                     (const TT*)W, ldw, (const TT*)bias, (const TT*)res, ldr, (TT*)out, ldo, N, K)
#define V3D_LDM(TT, EE) { if (M <= 16) { if (og == 2) V3D_LDM_B(TT, EE, 1, 2); else V3D_LDM_B(TT, EE, 1, 1); }                       \
                          else { if (og == 2) V3D_LDM_B(TT, EE, 2, 2); else V3D_LDM_B(TT, EE, 2, 1); } }
#define V3D_LDM_E(TT)                                                                                 \
  switch (epilogue) {                                                                                 \
    case DEC_EPI_NONE: V3D_LDM(TT, DEC_EPI_NONE); break; case DEC_EPI_BIAS: V3D_LDM(TT, DEC_EPI_BIAS); break; \
    case DEC_EPI_RES: V3D_LDM(TT, DEC_EPI_RES); break; case DEC_EPI_SWIGLU: V3D_LDM(TT, DEC_EPI_SWIGLU); break; \
    default: set_error("%s: unknown epilogue %d", who, epilogue); return V3D_E_INVALID;               \
  }
    if (dtype == V3D_BF16) { V3D_LDM_E(bf16_t) } else { V3D_LDM_E(f16_t) }
#undef V3D_LDM_E
#undef V3D_LDM
#undef V3D_LDM_B
    return check_launch(who);
  }
  const int blocks = N / 4;                                   // 4 weight rows per workgroup (SWIGLU: 2 gate/up pairs)
#define V3D_LD(TT, EE, NN, MM)                                                                                                \
  hipLaunchKernelGGL((linear_decode_kernel<TT, EE, NN, MM>), dim3(blocks), dim3(256), 0, st, (const TT*)x, ldx, (const TT*)norm_weight, \
                     eps, (const TT*)W, ldw, (const TT*)bias, (const TT*)res, ldr, (TT*)out, ldo, N, K)
#define V3D_LD_M(TT, EE, NN)                                                                          \
  switch (M) { case 1: V3D_LD(TT, EE, NN, 1); break; case 2: V3D_LD(TT, EE, NN, 2); break;            \
               case 3: V3D_LD(TT, EE, NN, 3); break; default: V3D_LD(TT, EE, NN, 4); break; }
#define V3D_LD_N(TT, EE) { if (norm_weight) { V3D_LD_M(TT, EE, true) } else { V3D_LD_M(TT, EE, false) } }
#define V3D_LD_E(TT)                                                                                  \
  switch (epilogue) {                                                                                 \
    case DEC_EPI_NONE: V3D_LD_N(TT, DEC_EPI_NONE) break; case DEC_EPI_BIAS: V3D_LD_N(TT, DEC_EPI_BIAS) break; \
    case DEC_EPI_RES: V3D_LD_N(TT, DEC_EPI_RES) break; case DEC_EPI_SWIGLU: V3D_LD_N(TT, DEC_EPI_SWIGLU) break; \
    default: set_error("%s: unknown epilogue %d", who, epilogue); return V3D_E_INVALID;               \
  }
  if (dtype == V3D_BF16) { V3D_LD_E(bf16_t) } else { V3D_LD_E(f16_t) }
#undef V3D_LD_E
#undef V3D_LD_N
#undef V3D_LD_M
#undef V3D_LD
  return check_launch(who);
}

extern "C" int v3d_linear_decode(const void* x, const void* norm_weight, float eps, const void* W, int64_t ldw,
                                 const void* bias, const void* res, void* out, int N, int K, int dtype, int epilogue,
                                 void* stream) {
  return linear_decode_rows(x, 0, 1, norm_weight, eps, W, ldw, bias, res, 0, out, 0, N, K, dtype, epilogue, stream, "v3d_linear_decode");
}

extern "C" int v3d_linear_decode_rows_fuses_norm(int M, int N, int K, int epilogue) {
  const char* off = getenv("V3D_DEC_FUSE_NORM");           // "0": callers normalise first (A/B)
  if (off && atoi(off) == 0) return 0;
  const bool mfma_shape = K % 128 == 0 && (epilogue == DEC_EPI_SWIGLU ? N % 128 == 0 : N % 16 == 0);
  return (M > 4 && M <= DEC_MAX_ROWS && K <= 4096 && epilogue != DEC_EPI_RES && mfma_shape &&
          dec_v2_plan(epilogue == DEC_EPI_SWIGLU ? N / 2 : N, K, nullptr, nullptr, nullptr)) ? 1 : 0;
}

extern "C" int v3d_linear_decode_rows(const void* x, int64_t ldx, int M, const void* norm_weight, float eps, const void* W,
                                      int64_t ldw, const void* bias, const void* res, int64_t ldr, void* out, int64_t ldo, int N,
                                      int K, int dtype, int epilogue, void* stream) {
  return linear_decode_rows(x, ldx, M, norm_weight, eps, W, ldw, bias, res, ldr, out, ldo, N, K, dtype, epilogue, stream,
                            "v3d_linear_decode_rows");
}

// Matrix-core form of the e4m3 decode linear (W8A16, 2..16 scenes): as linear_decode_mfma_kernel, but a tile is 16 rows x
// 256 BYTES = 256 weights, loaded with full-line coalescing into the wave-private LDS tile; the fragment of a k step is 8
// bytes per lane, widened to the activation type by v_cvt_scalef32_pk_{bf16,f16}_fp8 (exact: e4m3 has 3 mantissa bits)
// and fed to v_mfma_f32_16x16x32 against the 16-bit activation rows.  The per-row weight scale is applied in the epilogue.
__device__ __forceinline__ uint4 dec_widen_fp8(bf16_t, uint2 w8) {
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  const b2 p0 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)w8.x, 1.0f, false), p1 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)w8.x, 1.0f, true);
  const b2 p2 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)w8.y, 1.0f, false), p3 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)w8.y, 1.0f, true);
  return make_uint4(__builtin_bit_cast(uint32_t, p0), __builtin_bit_cast(uint32_t, p1), __builtin_bit_cast(uint32_t, p2), __builtin_bit_cast(uint32_t, p3));
}
__device__ __forceinline__ uint4 dec_widen_fp8(f16_t, uint2 w8) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const h2 p0 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((int)w8.x, 1.0f, false), p1 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((int)w8.x, 1.0f, true);
  const h2 p2 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((int)w8.y, 1.0f, false), p3 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((int)w8.y, 1.0f, true);
  return make_uint4(__builtin_bit_cast(uint32_t, p0), __builtin_bit_cast(uint32_t, p1), __builtin_bit_cast(uint32_t, p2), __builtin_bit_cast(uint32_t, p3));
}

template <typename T, int EPI, int MB>
__global__ __launch_bounds__(512) void linear_decode_fp8_mfma_kernel(const T* __restrict__ x, int64_t ldx, int M, const uint8_t* __restrict__ W,
                                                                     int64_t ldw, const float* __restrict__ sw, const T* __restrict__ bias,
                                                                     const T* __restrict__ res, int64_t ldr, T* __restrict__ out, int64_t ldo,
                                                                     int N, int K) {
  constexpr int RG = EPI == DEC_EPI_SWIGLU ? 2 : 1;
  constexpr int PITCH = 272;
  __shared__ __attribute__((aligned(16))) char tile[8][RG][16 * PITCH];
  __shared__ float part[8][RG][MB][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int lr = lane >> 4, lc = lane & 15;
  const int o0 = blockIdx.x * 16;
  int row0[RG];
  if (EPI == DEC_EPI_SWIGLU) { row0[0] = (o0 >> 6) * 128 + (o0 & 63); row0[RG - 1] = row0[0] + 64; }
  else row0[0] = o0;
  const uint8_t* wbase[RG];
#pragma unroll
  for (int q = 0; q < RG; ++q) wbase[q] = W + (int64_t)(row0[q] + lr) * ldw + 16 * lc;
  bool col_ok[MB];
  const T* xrow[MB];
#pragma unroll
  for (int b = 0; b < MB; ++b) {
    col_ok[b] = r + 16 * b < M;
    xrow[b] = x + (int64_t)(col_ok[b] ? r + 16 * b : 0) * ldx + 8 * g;
  }
  dec_f32x4 acc[RG][MB];
#pragma unroll
  for (int q = 0; q < RG; ++q)
#pragma unroll
    for (int b = 0; b < MB; ++b) acc[q][b] = dec_f32x4{0.f, 0.f, 0.f, 0.f};
  const int tiles = K / 256;
  for (int s = wave; s < tiles; s += 8) {
    const int k0 = 256 * s;
    uint4 wreg[RG][4], xb[MB][8];
#pragma unroll
    for (int q = 0; q < RG; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) wreg[q][j] = ldg_nt(reinterpret_cast<const uint4*>(wbase[q] + (int64_t)(4 * j) * ldw + k0));
#pragma unroll
    for (int b = 0; b < MB; ++b)
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        xb[b][t] = make_uint4(0u, 0u, 0u, 0u);
        if (col_ok[b]) xb[b][t] = *reinterpret_cast<const uint4*>(xrow[b] + k0 + 32 * t);
      }
#pragma unroll
    for (int q = 0; q < RG; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<uint4*>(&tile[wave][q][(4 * j + lr) * PITCH + lc * 16]) = wreg[q][j];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int q = 0; q < RG; ++q) {
        const uint2 w8 = *reinterpret_cast<const uint2*>(&tile[wave][q][r * PITCH + (4 * t + g) * 8]);
        const uint4 a = dec_widen_fp8(T{}, w8);
#pragma unroll
        for (int b = 0; b < MB; ++b) acc[q][b] = dec_mfma(T{}, a, xb[b][t], acc[q][b]);
      }
  }
#pragma unroll
  for (int q = 0; q < RG; ++q)
#pragma unroll
    for (int b = 0; b < MB; ++b)
      *reinterpret_cast<float4*>(&part[wave][q][b][lane * 4]) = make_float4(acc[q][b][0], acc[q][b][1], acc[q][b][2], acc[q][b][3]);
  __syncthreads();
  for (int e = tid; e < 256 * MB; e += 512) {
    const int b = e >> 8, t8 = e & 255;
    const int l = t8 >> 2, i = t8 & 3;
    const int row = 4 * (l >> 4) + i, m = (l & 15) + 16 * b;
    float v[RG];
#pragma unroll
    for (int q = 0; q < RG; ++q) {
      v[q] = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) v[q] += part[w][q][b][t8];
      v[q] *= sw[row0[q] + row];
    }
    if (m < M) {
      const int n = o0 + row;
      if (EPI == DEC_EPI_SWIGLU) {
        const float gt = round_to<T>(v[0]), up = round_to<T>(v[RG - 1]);
        out[m * ldo + n] = from_f32<T>(round_to<T>(silu_f(gt)) * up);
      } else {
        float y = v[0];
        if (EPI == DEC_EPI_BIAS) y += to_f32(bias[n]);
        y = round_to<T>(y);
        if (EPI == DEC_EPI_RES) y += to_f32(res[m * ldr + n]);
        out[m * ldo + n] = from_f32<T>(y);
      }
    }
  }
}

extern "C" int v3d_linear_decode_fp8_rows(const void* x, int64_t ldx, int M, const void* W8, int64_t ldw, const float* scale_w,
                                          const void* bias, const void* res, int64_t ldr, void* out, int64_t ldo, int N, int K,
                                          int dtype, int epilogue, void* stream) {
  const char* who = "v3d_linear_decode_fp8_rows";
  V3D_REQUIRE(x && W8 && scale_w && out, "%s: null pointer", who);
  V3D_REQUIRE(dtype == V3D_F16 || dtype == V3D_BF16, "%s: dtype must be f16 or bf16", who);
  V3D_REQUIRE(N > 0 && K > 0 && K % 16 == 0 && ldw % 16 == 0 && ldw >= K, "%s: bad shape N=%d K=%d", who, N, K);
  // one row: VALU form; 2..16 rows: matrix-core form when the shape allows (a row's bits then depend neither on the other
  // rows nor on M); other shapes: VALU form with up to 4 rows
  const bool mfma_ok = K % 256 == 0 && (epilogue == DEC_EPI_SWIGLU ? N % 128 == 0 : N % 16 == 0);
  const bool mfma = mfma_ok && M >= 2;
  V3D_REQUIRE(M >= 1 && M <= (mfma_ok ? DEC_MAX_ROWS : 4), "%s: 1 to %d activation rows for this shape (got %d)", who, mfma_ok ? DEC_MAX_ROWS : 4, M);
  V3D_REQUIRE(N % 4 == 0 && (epilogue != DEC_EPI_SWIGLU || N % 128 == 0), "%s: N=%d not supported", who, N);
  V3D_REQUIRE(aligned16(x) && aligned16(W8) && (M == 1 || (ldx % 8 == 0 && ldx >= K)), "%s: alignment", who);
  V3D_REQUIRE(epilogue != DEC_EPI_BIAS || bias, "%s: bias epilogue without bias", who);
  V3D_REQUIRE(epilogue != DEC_EPI_RES || res, "%s: residual epilogue without residual", who);
  hipStream_t st = (hipStream_t)stream;
  if (mfma) {
    const int mblocks = (epilogue == DEC_EPI_SWIGLU ? N / 2 : N) / 16;
#define V3D_LD8M_B(TT, EE, BB) hipLaunchKernelGGL((linear_decode_fp8_mfma_kernel<TT, EE, BB>), dim3(mblocks), dim3(512), 0, st, (const TT*)x, ldx, M, \
                                                  (const uint8_t*)W8, ldw, scale_w, (const TT*)bias, (const TT*)res, ldr, (TT*)out, ldo, N, K)
#define V3D_LD8M(TT, EE) { if (M <= 16) V3D_LD8M_B(TT, EE, 1); else V3D_LD8M_B(TT, EE, 2); }
#define V3D_LD8M_E(TT)                                                                                \
  switch (epilogue) {                                                                                 \
    case DEC_EPI_NONE: V3D_LD8M(TT, DEC_EPI_NONE); break; case DEC_EPI_BIAS: V3D_LD8M(TT, DEC_EPI_BIAS); break; \
    case DEC_EPI_RES: V3D_LD8M(TT, DEC_EPI_RES); break; case DEC_EPI_SWIGLU: V3D_LD8M(TT, DEC_EPI_SWIGLU); break; \
    default: set_error("%s: unknown epilogue %d", who, epilogue); return V3D_E_INVALID;               \
  }
    if (dtype == V3D_BF16) { V3D_LD8M_E(bf16_t) } else { V3D_LD8M_E(f16_t) }
#undef V3D_LD8M_E
#undef V3D_LD8M
#undef V3D_LD8M_B
    return check_launch(who);
  }
  const int blocks = N / 4;
#define V3D_LD8(TT, EE, MM)                                                                                                   \
  hipLaunchKernelGGL((linear_decode_fp8_kernel<TT, EE, MM>), dim3(blocks), dim3(256), 0, st, (const TT*)x, ldx, (const uint8_t*)W8, ldw, \
                     scale_w, (const TT*)bias, (const TT*)res, ldr, (TT*)out, ldo, N, K)
#define V3D_LD8_M(TT, EE)                                                                             \
  switch (M) { case 1: V3D_LD8(TT, EE, 1); break; case 2: V3D_LD8(TT, EE, 2); break;                  \
               case 3: V3D_LD8(TT, EE, 3); break; default: V3D_LD8(TT, EE, 4); break; }
#define V3D_LD8_E(TT)                                                                                 \
  switch (epilogue) {                                                                                 \
    case DEC_EPI_NONE: V3D_LD8_M(TT, DEC_EPI_NONE) break; case DEC_EPI_BIAS: V3D_LD8_M(TT, DEC_EPI_BIAS) break; \
    case DEC_EPI_RES: V3D_LD8_M(TT, DEC_EPI_RES) break; case DEC_EPI_SWIGLU: V3D_LD8_M(TT, DEC_EPI_SWIGLU) break; \
    default: set_error("%s: unknown epilogue %d", who, epilogue); return V3D_E_INVALID;               \
  }
  if (dtype == V3D_BF16) { V3D_LD8_E(bf16_t) } else { V3D_LD8_E(f16_t) }
#undef V3D_LD8_E
#undef V3D_LD8_M
#undef V3D_LD8
  return check_launch(who);
}

static int rope_kv_append_rows(void* qkv, int64_t qkv_stride, int M, int n_q_heads, int n_kv_heads, int head_dim, const void* cos_table,
                               const void* sin_table, int n_pos, const int* pos, void* const* cache_rows, int dtype, void* stream,
                               const char* who) {
  V3D_REQUIRE(qkv && cos_table && sin_table && pos && cache_rows, "%s: null pointer", who);
  V3D_REQUIRE(M >= 1 && M <= DEC_MAX_ROWS && head_dim % 16 == 0 && aligned16(qkv) && qkv_stride % 8 == 0, "%s: bad arguments", who);
  RopeRows rw{};
  for (int m = 0; m < M; ++m) {
    V3D_REQUIRE(pos[m] >= 0 && pos[m] < n_pos, "%s: pos %d outside the table (%d)", who, pos[m], n_pos);
    V3D_REQUIRE(cache_rows[m] && aligned16(cache_rows[m]), "%s: cache row %d", who, m);
    rw.pos[m] = pos[m]; rw.cache_row[m] = cache_rows[m];
  }
  rw.qkv_stride = qkv_stride;
  const int items = (n_q_heads + n_kv_heads) * (head_dim / 16) + n_kv_heads * head_dim / 8;
  const dim3 grid((items + 255) / 256, M);
  if (dtype == V3D_BF16)
    hipLaunchKernelGGL(rope_kv_append_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (bf16_t*)qkv,
                       n_q_heads, n_kv_heads, head_dim, (const bf16_t*)cos_table, (const bf16_t*)sin_table, rw);
  else if (dtype == V3D_F16)
    hipLaunchKernelGGL(rope_kv_append_kernel<f16_t>, grid, dim3(256), 0, (hipStream_t)stream, (f16_t*)qkv,
                       n_q_heads, n_kv_heads, head_dim, (const f16_t*)cos_table, (const f16_t*)sin_table, rw);
  else { set_error("%s: dtype must be f16 or bf16", who); return V3D_E_INVALID; }
  return check_launch(who);
}

extern "C" int v3d_rope_kv_append(void* qkv_row, int n_q_heads, int n_kv_heads, int head_dim, const void* cos_table,
                                  const void* sin_table, int n_pos, int pos, void* cache_row, int dtype, void* stream) {
  return rope_kv_append_rows(qkv_row, 0, 1, n_q_heads, n_kv_heads, head_dim, cos_table, sin_table, n_pos, &pos, &cache_row, dtype,
                             stream, "v3d_rope_kv_append");
}

extern "C" int v3d_rope_kv_append_rows(void* qkv, int64_t qkv_stride, int M, int n_q_heads, int n_kv_heads, int head_dim,
                                       const void* cos_table, const void* sin_table, int n_pos, const int* pos,
                                       void* const* cache_rows, int dtype, void* stream) {
  return rope_kv_append_rows(qkv, qkv_stride, M, n_q_heads, n_kv_heads, head_dim, cos_table, sin_table, n_pos, pos, cache_rows,
                             dtype, stream, "v3d_rope_kv_append_rows");
}

static int argmax_rows(const void* x, int64_t ldx, int M, int n, int dtype, int64_t* out_index, void* workspace, void* stream, const char* who) {
  V3D_REQUIRE(x && out_index && workspace && n > 0 && M >= 1 && M <= 64, "%s: bad arguments", who);
  constexpr int NB = 128;                                    // workspace per row: NB floats + NB ints = 1 KiB
  float* pv = (float*)workspace;
  int* pi = (int*)(pv + NB);
  V3D_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(argmax_part_kernel<T>, dim3(NB, M), dim3(256), 0, (hipStream_t)stream, (const T*)x, ldx, n, pv, pi));
  if (int e = check_launch(who)) return e;
  hipLaunchKernelGGL(argmax_final_kernel, dim3(M), dim3(64), 0, (hipStream_t)stream, pv, pi, NB, out_index);
  return check_launch(who);
}

extern "C" int v3d_argmax(const void* x, int n, int dtype, int64_t* out_index, void* workspace, void* stream) {
  return argmax_rows(x, 0, 1, n, dtype, out_index, workspace, stream, "v3d_argmax");
}

extern "C" int v3d_argmax_rows(const void* x, int64_t ldx, int M, int n, int dtype, int64_t* out_index, void* workspace, void* stream) {
  return argmax_rows(x, ldx, M, n, dtype, out_index, workspace, stream, "v3d_argmax_rows");
}

// ------------------------------------------------------------------------------------------
// Greedy decoding's stop test on the device (r03): generate() ends a sequence at its first EOS id (llava_qwen.py:208-236 -> HF
// greedy search with eos_token_id).  done[m] |= tokens[m] in eos_ids; *n_done = number of finished rows.  The host reads n_done
// through a pinned copy a step or two later instead of synchronising on every token.
__global__ __launch_bounds__(64) void eos_update_kernel(const int64_t* __restrict__ tokens, int M, const int64_t* __restrict__ eos, int n_eos,
                                                        int32_t* __restrict__ done, int32_t* __restrict__ n_done) {
  int cnt = 0;
  for (int m = threadIdx.x; m < M; m += 64) {
    int d = done[m];
    const int64_t t = tokens[m];
    for (int e = 0; e < n_eos; ++e) d |= (t == eos[e]) ? 1 : 0;
    done[m] = d;
    cnt += d ? 1 : 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
  if (threadIdx.x == 0) *n_done = cnt;
}

extern "C" int v3d_eos_update(const int64_t* tokens, int M, const int64_t* eos_ids, int n_eos, int32_t* done, int32_t* n_done, void* stream) {
  V3D_REQUIRE(tokens && done && n_done && (eos_ids || n_eos == 0), "v3d_eos_update: null pointer");
  V3D_REQUIRE(M > 0 && n_eos >= 0, "v3d_eos_update: bad shape");
  hipLaunchKernelGGL(eos_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, tokens, M, eos_ids, n_eos, done, n_done);
  return check_launch("v3d_eos_update");
}
