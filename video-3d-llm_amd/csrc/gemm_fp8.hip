// FP8 (OCP e4m3) path for the LLM linears - BASELINE configs[3] ("fp8 MFMA LLM GEMMs, tolerance re-stated").
// The reference has no fp8 code: this is the bf16 linear  y = x W^T  evaluated as
//     y[m,n] = sa[m] * sw[n] * sum_k qa[m,k] * qw[n,k]          qa = e4m3(x / sa), sa[m] = amax_k |x[m,k]| / 448
// with per-row (token) activation scales and per-output-channel weight scales, f32 accumulation on
// v_mfma_scale_f32_16x16x128_f8f6f4 (block scales pinned to 2^0): twice the MFMA rate of bf16 and half the
// operand bytes through LDS.  Lane layout of that instruction was measured (tools/probes/mfma_scale_f8_layout.hip):
// lane l holds row l&15, k-bytes [32*(l>>4), +32) of A and of B; C/D as every 16x16 MFMA.
//
// Kernel structure = gemm256x256_kernel (gemm.hip): 256(192) x 256 tile, K-step 128 (a 128-byte LDS row, i.e. the
// same LDS image, DMA staging and XOR swizzle as the bf16 kernel), 8 waves of 128 x 64, four phases of 8 MFMAs
// with the A fragments of phase p+1 in flight under phase p, B fragments double-buffered across K-steps.
#include "v3d_common.h"

namespace v3d {

using v8i = __attribute__((ext_vector_type(8))) int;
using v4i_ = __attribute__((ext_vector_type(4))) int;
using f32x4_ = __attribute__((ext_vector_type(4))) float;

constexpr int F8_BK = 128;                               // k elements (= bytes) per K-step
constexpr int F8_TILE = 256 * F8_BK;                     // 32 KiB per operand tile
constexpr int F8_STAGE = 2 * F8_TILE;
constexpr int F8_LDS = 2 * F8_STAGE;                     // 128 KiB
constexpr int F8_BN = 256;

enum { F8_EPI_NONE = 0, F8_EPI_BIAS = 1, F8_EPI_RES = 5, F8_EPI_SWIGLU = 6 };

struct Fp8GemmArgs {
  const uint8_t* A; const uint8_t* W; const float* sa; const float* sw;
  const void* bias; const void* res; void* out;
  int M, N, K;
  int64_t lda, ldw, ldr, ldo;
  int tiles_m, tiles_n;
};

__device__ __forceinline__ float silu8(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }

__device__ __forceinline__ void glds16f(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ void tile_of_block8(int bid, int nblocks, int tiles_m, int tiles_n, int& tm, int& tn) {
  const int q = nblocks >> 3, r = nblocks & 7, xcd = bid & 7;
  const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * tiles_n;
  const int g = L / per_group, in_g = L - g * per_group;
  const int first_m = g * GROUP_M;
  const int gsz = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
  tm = first_m + in_g % gsz;
  tn = in_g / gsz;
}

template <typename T, int EPI, int MT>
__global__ __launch_bounds__(512, 2) void gemm_fp8_kernel(Fp8GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int MG = MT / 4;              // m-tiles per phase (4 phases); only MT = 8 (BM = 256) is instantiated
  constexpr int WROWS = MT * 16;
  constexpr int BM = 2 * WROWS;
  constexpr int APW = BM / 8 / 8;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  int tm, tn;
  tile_of_block8(blockIdx.x, gridDim.x, p.tiles_m, p.tiles_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * F8_BN;

  unsigned a_off[APW], w_off[4];
#pragma unroll
  for (int i = 0; i < APW; ++i) {
    const int row = wave * (APW * 8) + i * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    int gm = m0 + row;
    gm = gm < p.M ? gm : p.M - 1;
    a_off[i] = (unsigned)(gm * (int)p.lda + chunk * 16);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 32 + i * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    // SwiGLU (tile columns = two groups of [gate64 | up64]): wave column wn = row >> 6 gets the gate columns and the MATCHING up
    // columns of one 32-wide output block, so silu(gate) * up is formed in registers (as gemm256pp_kernel)
    const int wcol = EPI == F8_EPI_SWIGLU ? 128 * (row >> 7) + ((row & 32) ? 64 : 0) + 32 * ((row >> 6) & 1) + (row & 31) : row;
    w_off[i] = (unsigned)((n0 + wcol) * (int)p.ldw + chunk * 16);
  }
  auto stage = [&](int buf, int kt) {
    char* ba = smem + buf * F8_STAGE + (wave * APW * 8) * F8_BK;
    char* bw = smem + buf * F8_STAGE + F8_TILE + (wave * 32) * F8_BK;
    const char* Ak = (const char*)p.A + (size_t)kt * F8_BK;
    const char* Wk = (const char*)p.W + (size_t)kt * F8_BK;
#pragma unroll
    for (int i = 0; i < APW; ++i) glds16f(Ak + a_off[i], ba + i * 8 * F8_BK);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16f(Wk + w_off[i], bw + i * 8 * F8_BK);
  };

  // fragment = 32 bytes = logical chunks 2g, 2g+1 of the lane's row (g = lane>>4), XOR-swizzled like the bf16 kernel
  const int sw = (lane >> 1) & 7, g2 = 2 * (lane >> 4);
  const int frow = (lane & 15) * F8_BK;
  const unsigned fo_lo = frow + (((g2) ^ sw) << 4), fo_hi = frow + (((g2 + 1) ^ sw) << 4);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned offA = lds0 + (wm * WROWS) * F8_BK, offW = lds0 + F8_TILE + (wn * 64) * F8_BK;

  f32x4_ acc[4][MT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4_{0.f, 0.f, 0.f, 0.f};

  // register sets: B fragments of the K-step (4 n-tiles), A ring of two groups of MG m-tiles; a fragment = lo|hi v4i
  // (two ds_read_b128; the 8-register MFMA operand is assembled by the compiler).  One barrier per K-step:
  //   top: read B + A(0,1), issue the DMA of tile t+1 into the other buffer, then four phases of 8 MFMAs with
  //   the A fragments of phase p+1 in flight under phase p; the last phase also retires the DMA (vmcnt(0)).
  v4i_ A0[MG][2], A1[MG][2], B0[4][2];
#define F8_DSR(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:" #imm : "=v"(dst) : "v"(addr))
#define F8_RDT(f, lo, hi, imm) { F8_DSR(f[0], lo, imm); F8_DSR(f[1], hi, imm); }
#define F8_RD_A(F, lo, hi, tile0)                                                                                 \
  {                                                                                                               \
    if constexpr ((tile0) == 0) { F8_RDT(F[0], lo, hi, 0); F8_RDT(F[1], lo, hi, 2048); }                          \
    else if constexpr ((tile0) == 2) { F8_RDT(F[0], lo, hi, 4096); F8_RDT(F[1], lo, hi, 6144); }                  \
    else if constexpr ((tile0) == 4) { F8_RDT(F[0], lo, hi, 8192); F8_RDT(F[1], lo, hi, 10240); }                 \
    else { F8_RDT(F[0], lo, hi, 12288); F8_RDT(F[1], lo, hi, 14336); }                                            \
  }
#define F8_RD_B(F, lo, hi) { F8_RDT(F[0], lo, hi, 0); F8_RDT(F[1], lo, hi, 2048); F8_RDT(F[2], lo, hi, 4096); F8_RDT(F[3], lo, hi, 6144); }
#define F8_WA(cnt, F) asm volatile("s_waitcnt " cnt : "+v"(F[0][0]), "+v"(F[0][1]), "+v"(F[1][0]), "+v"(F[1][1]) : : "memory")
#define F8_WAB(cnt, F, G)                                                                                          \
  asm volatile("s_waitcnt " cnt : "+v"(F[0][0]), "+v"(F[0][1]), "+v"(F[1][0]), "+v"(F[1][1]), "+v"(G[0][0]), "+v"(G[0][1]), \
               "+v"(G[1][0]), "+v"(G[1][1]), "+v"(G[2][0]), "+v"(G[2][1]), "+v"(G[3][0]), "+v"(G[3][1]) : : "memory")
#define F8_FRAG(f) (v8i{f[0][0], f[0][1], f[0][2], f[0][3], f[1][0], f[1][1], f[1][2], f[1][3]})
#define F8_MMA(FA, FB, tile0)                                                                                       \
  {                                                                                                                 \
    __builtin_amdgcn_s_setprio(1);                                                                                  \
    _Pragma("unroll") for (int i = 0; i < MG; ++i)                                                                  \
    _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                                                \
        acc[ni][(tile0) + i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(F8_FRAG(FB[ni]), F8_FRAG(FA[i]),    \
                                                                                  acc[ni][(tile0) + i], 0, 0, 0, 127, 0, 127); \
    __builtin_amdgcn_s_setprio(0);                                                                                  \
  }

  const int nt = p.K / F8_BK;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    const unsigned alo = offA + cur * F8_STAGE + fo_lo, ahi = offA + cur * F8_STAGE + fo_hi;
    const unsigned wlo = offW + cur * F8_STAGE + fo_lo, whi = offW + cur * F8_STAGE + fo_hi;
    F8_RD_B(B0, wlo, whi);
    F8_RD_A(A0, alo, ahi, 0);
    if (t + 1 < nt) stage(cur ^ 1, t + 1);          // buffer cur^1 was released by the barrier that ended step t-1
    F8_RD_A(A1, alo, ahi, 2);
    F8_WAB("lgkmcnt(4)", A0, B0);
    F8_MMA(A0, B0, 0);
    F8_RD_A(A0, alo, ahi, 4);
    F8_WA("lgkmcnt(4)", A1);
    F8_MMA(A1, B0, 2);
    F8_RD_A(A1, alo, ahi, 6);
    F8_WA("lgkmcnt(4)", A0);
    F8_MMA(A0, B0, 4);
    F8_WA("vmcnt(0) lgkmcnt(0)", A1);               // last fragments + this wave's share of tile t+1
    F8_MMA(A1, B0, 6);
    __builtin_amdgcn_s_barrier();                    // tile t fully read by everyone, tile t+1 visible
  }
  __syncthreads();

  // epilogue: dequantise (sa[m] * sw[n]), then as gemm256pp_kernel: each wave turns its 128 x 64 part of the C tile around in a
  // private 8 KiB of LDS, 32 rows at a time - no workgroup barrier, whole row segments out; SwiGLU formed in registers first
  const T* bias = (const T*)p.bias;
  T* out = (T*)p.out;
  {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int l15 = ln & 15, l4 = ln >> 4;
    char* const reg = smem + wave * 8192;
    if constexpr (EPI == F8_EPI_SWIGLU) {
      constexpr int CP = 80;
      const int rrow = ln >> 2, rch = ln & 3;
      const int gcol = n0 + 128 * (wn >> 1) + 32 * (wn & 1) + 4 * l4;       // + 16 n2 (+ 64 for up): W rows behind acc[n2] / acc[2 + n2]
      float swg[2][4], swu[2][4];
#pragma unroll
      for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
        for (int r = 0; r < 4; ++r) { swg[n2][r] = p.sw[gcol + 16 * n2 + r]; swu[n2][r] = p.sw[gcol + 64 + 16 * n2 + r]; }
      T* const obase = out + tn * 128 + (wn >> 1) * 64 + (wn & 1) * 32 + rch * 8;
#pragma unroll
      for (int q = 0; q < MT / 2; ++q) {
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2) {
          const int mi = 2 * q + m2;
          int gmc = m0 + wm * WROWS + 16 * mi + l15;
          gmc = gmc < p.M ? gmc : p.M - 1;
          const float sa = p.sa[gmc];
#pragma unroll
          for (int n2 = 0; n2 < 2; ++n2) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
              v[r] = round_to<T>(silu8(round_to<T>(acc[n2][mi][r] * (sa * swg[n2][r])))) * round_to<T>(acc[2 + n2][mi][r] * (sa * swu[n2][r]));
            uint2 pk;
            pk.x = pack2<T>(v[0], v[1]); pk.y = pack2<T>(v[2], v[3]);
            *reinterpret_cast<uint2*>(reg + (m2 * 16 + l15) * CP + (16 * n2 + 4 * l4) * 2) = pk;
          }
        }
        uint4 cq[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) cq[j] = *reinterpret_cast<const uint4*>(reg + (rrow + 16 * j) * CP + rch * 16);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int gm = m0 + wm * WROWS + 32 * q + rrow + 16 * j;
          if (gm < p.M) *reinterpret_cast<uint4*>(obase + (int64_t)gm * p.ldo) = cq[j];
        }
      }
    } else {
      constexpr int CP = 144;
      const int rrow = ln >> 3, rch = ln & 7;
      const int colr = n0 + wn * 64 + rch * 8;
      float swv[4][4], bv[4][4];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = n0 + wn * 64 + 16 * ni + 4 * l4 + r;
          swv[ni][r] = p.sw[c];
          bv[ni][r] = bias != nullptr ? to_f32(bias[c]) : 0.f;
        }
#pragma unroll
      for (int q = 0; q < MT / 2; ++q) {
        uint4 rr[4];
        if constexpr (EPI == F8_EPI_RES) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            int gc = m0 + wm * WROWS + 32 * q + rrow + 8 * j;
            gc = gc < p.M ? gc : p.M - 1;
            rr[j] = *reinterpret_cast<const uint4*>((const T*)p.res + (int64_t)gc * p.ldr + colr);
          }
        }
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2) {
          const int mi = 2 * q + m2;
          int gmc = m0 + wm * WROWS + 16 * mi + l15;
          gmc = gmc < p.M ? gmc : p.M - 1;
          const float sa = p.sa[gmc];
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            uint2 pk;
            pk.x = pack2<T>(acc[ni][mi][0] * (sa * swv[ni][0]) + bv[ni][0], acc[ni][mi][1] * (sa * swv[ni][1]) + bv[ni][1]);
            pk.y = pack2<T>(acc[ni][mi][2] * (sa * swv[ni][2]) + bv[ni][2], acc[ni][mi][3] * (sa * swv[ni][3]) + bv[ni][3]);
            *reinterpret_cast<uint2*>(reg + (m2 * 16 + l15) * CP + (16 * ni + 4 * l4) * 2) = pk;
          }
        }
        uint4 cq[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) cq[j] = *reinterpret_cast<const uint4*>(reg + (rrow + 8 * j) * CP + rch * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          uint4 c = cq[j];
          const int gm = m0 + wm * WROWS + 32 * q + rrow + 8 * j;
          if constexpr (EPI == F8_EPI_RES) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = vec_get<T>(c, e) + vec_get<T>(rr[j], e);
            c = vec_pack<T>(v);
          }
          if (gm < p.M) *reinterpret_cast<uint4*>(out + (int64_t)gm * p.ldo + colr) = c;
        }
      }
    }
  }
#undef F8_DSR
#undef F8_RDT
#undef F8_RD_A
#undef F8_RD_B
#undef F8_WA
#undef F8_WAB
#undef F8_FRAG
#undef F8_MMA
}

// Row-wise e4m3 quantisation: scale[r] = amax|x[r,:]| / 448 (1 if the row is zero), q = sat_e4m3(x / scale).
// One wave per row, two passes over the row (second pass from L2).
template <typename T>
__global__ __launch_bounds__(256) void quantize_rows_fp8_kernel(const T* __restrict__ x, int64_t ldx, int64_t rows, int cols,
                                                                uint8_t* __restrict__ q, int64_t ldq, float* __restrict__ scale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const uint4* xr = reinterpret_cast<const uint4*>(x + row * ldx);
  const int nv = cols / 8;
  float amax = 0.f;
  for (int k = lane; k < nv; k += 64) {
    const uint4 v = xr[k];
#pragma unroll
    for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(vec_get<T>(v, j)));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
  const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
  const float inv = 1.0f / sc;
  if (lane == 0) scale[row] = sc;
  uint2* qr = reinterpret_cast<uint2*>(q + row * ldq);
  for (int k = lane; k < nv; k += 64) {
    const uint4 v = xr[k];
    int w0 = 0, w1 = 0;
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(vec_get<T>(v, 0) * inv, vec_get<T>(v, 1) * inv, w0, false);
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(vec_get<T>(v, 2) * inv, vec_get<T>(v, 3) * inv, w0, true);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(vec_get<T>(v, 4) * inv, vec_get<T>(v, 5) * inv, w1, false);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(vec_get<T>(v, 6) * inv, vec_get<T>(v, 7) * inv, w1, true);
    qr[k] = make_uint2((uint32_t)w0, (uint32_t)w1);
  }
}

// Long rows (4096 < cols <= 32768, the SwiGLU output): one WORKGROUP per row with the row held in registers between the
// amax reduction and the conversion - one read of the row instead of two.  Same arithmetic as the wave-per-row kernel
// (amax is order-independent), so the results are bit-identical.
template <typename T>
__global__ __launch_bounds__(256) void quantize_row_block_fp8_kernel(const T* __restrict__ x, int64_t ldx, int cols, uint8_t* __restrict__ q,
                                                                     int64_t ldq, float* __restrict__ scale) {
  constexpr int MAXV = 16;                          // 16-byte vectors per thread: 256 x 16 x 8 = 32768 columns
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const int64_t row = blockIdx.x;
  const uint4* xr = reinterpret_cast<const uint4*>(x + row * ldx);
  const int nv = cols / 8;
  uint4 v[MAXV];
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int k = i * 256 + tid;
    if (k < nv) {
      v[i] = xr[k];
#pragma unroll
      for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(vec_get<T>(v[i], j)));
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
  if ((tid & 63) == 0) red[tid >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
  const float inv = 1.0f / sc;
  if (tid == 0) scale[row] = sc;
  uint2* qr = reinterpret_cast<uint2*>(q + row * ldq);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int k = i * 256 + tid;
    if (k < nv) {
      int w0 = 0, w1 = 0;
      w0 = __builtin_amdgcn_cvt_pk_fp8_f32(vec_get<T>(v[i], 0) * inv, vec_get<T>(v[i], 1) * inv, w0, false);
      w0 = __builtin_amdgcn_cvt_pk_fp8_f32(vec_get<T>(v[i], 2) * inv, vec_get<T>(v[i], 3) * inv, w0, true);
      w1 = __builtin_amdgcn_cvt_pk_fp8_f32(vec_get<T>(v[i], 4) * inv, vec_get<T>(v[i], 5) * inv, w1, false);
      w1 = __builtin_amdgcn_cvt_pk_fp8_f32(vec_get<T>(v[i], 6) * inv, vec_get<T>(v[i], 7) * inv, w1, true);
      qr[k] = make_uint2((uint32_t)w0, (uint32_t)w1);
    }
  }
}

// Qwen2RMSNorm (modeling_qwen2.py:85-90) fused with the row-wise e4m3 quantisation of its output: the normalised row
// y = w * T(x * rstd) is formed exactly as v3d_rmsnorm forms it (rounded to T), its amax gives the scale, and only the
// e4m3 image + scale leave the chip (one read of x, half a write) - the 16-bit y is never stored.  One wave per row,
// cols <= 4096.
template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_quantize_fp8_kernel(const T* __restrict__ x, int64_t ldx, const T* __restrict__ w, float eps,
                                                                   int64_t rows, int cols, uint8_t* __restrict__ q, int64_t ldq,
                                                                   float* __restrict__ scale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nv = cols / 8;
  const uint4* xr = reinterpret_cast<const uint4*>(x + row * ldx);
  uint4 v[8];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = i * 64 + lane;
    if (k < nv) {
      v[i] = xr[k];
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float f = vec_get<T>(v[i], j); ss = fmaf(f, f, ss); }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
  const float r = 1.0f / sqrtf(ss / (float)cols + eps);
  float y[8][8];
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = i * 64 + lane;
    if (k < nv) {
      const uint4 wv = reinterpret_cast<const uint4*>(w)[k];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        y[i][j] = round_to<T>(vec_get<T>(wv, j) * round_to<T>(vec_get<T>(v[i], j) * r));
        amax = fmaxf(amax, fabsf(y[i][j]));
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
  const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
  const float inv = 1.0f / sc;
  if (lane == 0) scale[row] = sc;
  uint2* qr = reinterpret_cast<uint2*>(q + row * ldq);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = i * 64 + lane;
    if (k < nv) {
      int w0 = 0, w1 = 0;
      w0 = __builtin_amdgcn_cvt_pk_fp8_f32(y[i][0] * inv, y[i][1] * inv, w0, false);
      w0 = __builtin_amdgcn_cvt_pk_fp8_f32(y[i][2] * inv, y[i][3] * inv, w0, true);
      w1 = __builtin_amdgcn_cvt_pk_fp8_f32(y[i][4] * inv, y[i][5] * inv, w1, false);
      w1 = __builtin_amdgcn_cvt_pk_fp8_f32(y[i][6] * inv, y[i][7] * inv, w1, true);
      qr[k] = make_uint2((uint32_t)w0, (uint32_t)w1);
    }
  }
}

template <typename T, int MT>
static int launch_fp8(Fp8GemmArgs p, int epi, hipStream_t st) {
  constexpr int BM = MT * 32;
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = p.N / F8_BN;
#define F8_CASE(E)                                                                                        \
  case E: {                                                                                               \
    auto k = gemm_fp8_kernel<T, E, MT>;                                                                   \
    static bool attr_done = false;                                                                        \
    if (!attr_done) {                                                                                     \
      hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, F8_LDS); \
      if (e != hipSuccess) { set_error("v3d_gemm_fp8: LDS attribute: %s", hipGetErrorString(e)); return V3D_E_LAUNCH; } \
      attr_done = true;                                                                                   \
    }                                                                                                     \
    hipLaunchKernelGGL(k, dim3(p.tiles_m * p.tiles_n), dim3(512), F8_LDS, st, p);                         \
  } break;
  switch (epi) {
    F8_CASE(F8_EPI_NONE)
    F8_CASE(F8_EPI_BIAS)
    F8_CASE(F8_EPI_RES)
    F8_CASE(F8_EPI_SWIGLU)
    default: set_error("v3d_gemm_fp8: epilogue %d unsupported (NONE, BIAS, RES, SWIGLU)", epi); return V3D_E_INVALID;
  }
#undef F8_CASE
  return check_launch("v3d_gemm_fp8");
}

}  // namespace v3d

using namespace v3d;

extern "C" int v3d_quantize_fp8_rows(const void* x, int64_t ldx, int64_t rows, int cols, int dtype, void* q, int64_t ldq,
                                     float* scale, void* stream) {
  V3D_REQUIRE(x && q && scale, "v3d_quantize_fp8_rows: null pointer");
  V3D_REQUIRE(rows >= 0 && cols > 0 && cols % 8 == 0 && ldx % 8 == 0 && ldq % 8 == 0 && ldq >= cols, "v3d_quantize_fp8_rows: bad shape");
  V3D_REQUIRE(aligned16(x) && (reinterpret_cast<uintptr_t>(q) & 7) == 0, "v3d_quantize_fp8_rows: alignment");
  if (rows == 0) return V3D_OK;
  if (cols > 4096 && cols <= 32768 && rows < (1ll << 31)) {       // long rows: one workgroup per row, single read
    if (dtype == V3D_BF16) hipLaunchKernelGGL(quantize_row_block_fp8_kernel<bf16_t>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, cols, (uint8_t*)q, ldq, scale);
    else if (dtype == V3D_F16) hipLaunchKernelGGL(quantize_row_block_fp8_kernel<f16_t>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const f16_t*)x, ldx, cols, (uint8_t*)q, ldq, scale);
    else { set_error("v3d_quantize_fp8_rows: dtype must be f16 or bf16"); return V3D_E_INVALID; }
    return check_launch("v3d_quantize_fp8_rows");
  }
  const unsigned blocks = (unsigned)((rows + 3) / 4);
  if (dtype == V3D_BF16) hipLaunchKernelGGL(quantize_rows_fp8_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, rows, cols, (uint8_t*)q, ldq, scale);
  else if (dtype == V3D_F16) hipLaunchKernelGGL(quantize_rows_fp8_kernel<f16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16_t*)x, ldx, rows, cols, (uint8_t*)q, ldq, scale);
  else { set_error("v3d_quantize_fp8_rows: dtype must be f16 or bf16"); return V3D_E_INVALID; }
  return check_launch("v3d_quantize_fp8_rows");
}

extern "C" int v3d_gemm_fp8(const void* A, int64_t lda, const float* scale_a, const void* W, int64_t ldw, const float* scale_w,
                            const void* bias, const void* res, int64_t ldr, void* out, int64_t ldo, int M, int N, int K,
                            int out_dtype, int epilogue, void* stream) {
  V3D_REQUIRE(A && W && scale_a && scale_w && out, "v3d_gemm_fp8: null pointer");
  V3D_REQUIRE(out_dtype == V3D_F16 || out_dtype == V3D_BF16, "v3d_gemm_fp8: output dtype must be f16 or bf16");
  V3D_REQUIRE(M > 0 && N > 0 && K > 0 && N % F8_BN == 0 && K % F8_BK == 0, "v3d_gemm_fp8: N=%d must be a multiple of 256 and K=%d of 128", N, K);
  V3D_REQUIRE(lda >= K && ldw >= K && lda % 16 == 0 && ldw % 16 == 0 && ldo % 8 == 0, "v3d_gemm_fp8: leading dimensions");
  V3D_REQUIRE(aligned16(A) && aligned16(W) && aligned16(out), "v3d_gemm_fp8: pointers must be 16-byte aligned");
  V3D_REQUIRE((int64_t)M * lda < (1ll << 31) && (int64_t)N * ldw < (1ll << 31), "v3d_gemm_fp8: operand larger than 2 GiB");
  V3D_REQUIRE(epilogue != F8_EPI_BIAS || bias, "v3d_gemm_fp8: bias epilogue without bias");
  V3D_REQUIRE(epilogue != F8_EPI_RES || (res && aligned16(res) && ldr % 8 == 0), "v3d_gemm_fp8: residual epilogue without residual");
  Fp8GemmArgs p;
  p.A = (const uint8_t*)A; p.W = (const uint8_t*)W; p.sa = scale_a; p.sw = scale_w;
  p.bias = epilogue == F8_EPI_BIAS ? bias : nullptr; p.res = epilogue == F8_EPI_RES ? res : nullptr; p.out = out;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldr = ldr; p.ldo = ldo;
  hipStream_t st = (hipStream_t)stream;
  return out_dtype == V3D_BF16 ? launch_fp8<bf16_t, 8>(p, epilogue, st) : launch_fp8<f16_t, 8>(p, epilogue, st);
}

extern "C" int v3d_rmsnorm_quantize_fp8(const void* x, int64_t ldx, const void* weight, float eps, int64_t rows, int cols, int dtype,
                                        void* q, int64_t ldq, float* scale, void* stream) {
  V3D_REQUIRE(x && weight && q && scale, "v3d_rmsnorm_quantize_fp8: null pointer");
  V3D_REQUIRE(rows >= 0 && cols > 0 && cols % 8 == 0 && cols <= 4096, "v3d_rmsnorm_quantize_fp8: cols=%d unsupported", cols);
  V3D_REQUIRE(ldx % 8 == 0 && ldq % 8 == 0 && aligned16(x) && aligned16(weight) && (reinterpret_cast<uintptr_t>(q) & 7) == 0,
              "v3d_rmsnorm_quantize_fp8: alignment");
  if (rows == 0) return V3D_OK;
  const unsigned blocks = (unsigned)((rows + 3) / 4);
  if (dtype == V3D_BF16) hipLaunchKernelGGL(rmsnorm_quantize_fp8_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, (const bf16_t*)weight, eps, rows, cols, (uint8_t*)q, ldq, scale);
  else if (dtype == V3D_F16) hipLaunchKernelGGL(rmsnorm_quantize_fp8_kernel<f16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16_t*)x, ldx, (const f16_t*)weight, eps, rows, cols, (uint8_t*)q, ldq, scale);
  else { set_error("v3d_rmsnorm_quantize_fp8: dtype must be f16 or bf16"); return V3D_E_INVALID; }
  return check_launch("v3d_rmsnorm_quantize_fp8");
}
