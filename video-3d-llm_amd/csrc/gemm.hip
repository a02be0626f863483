// Dense 16-bit GEMM on MFMA for the ViT / projector / LLM linears (K10-K12, K15, K17):
//     out[M, N'] = epilogue( A[M,K] . W[N,K]^T )        A, W: K-contiguous (nn.Linear layout)
//
// Two tile shapes, picked per problem by a small cost model (launch_gemm):
//   gemm_kernel         128 x 128 x 64, 4 waves (2 x 2) of 64 x 64, two workgroups per CU - fine-grained, for
//                       shapes whose 256 x 256 tile count quantises badly against the 256 CUs;
//   gemm256x256_kernel  256 x 256 x 64, 8 waves (2 x 4) of 128 x 64, one workgroup per CU - twice the operand
//                       reuse per LDS byte, ~1.15-1.35 PF on the path's large shapes.
// Both operand tiles are staged HBM -> LDS with
// global_load_lds_dwordx4 (LDS-DMA, no VGPR round trip) into a 2-deep ring; the LDS image is
// XOR-swizzled through the per-lane SOURCE address (the DMA's destination is lane-linear) so that
// the ds_read_b128 fragment reads are bank-conflict free.  The accumulators are produced
// TRANSPOSED (mfma(W-frag, A-frag)): a lane then owns 4 consecutive output columns of one row,
// i.e. one 8-byte LDS write, and the tile leaves through LDS as whole 256-byte row segments.
//
// Rounding points follow the reference (torch): the linear output (acc + bias, f32) is rounded to
// the 16-bit dtype once; activation / gate product / residual add each take 16-bit inputs,
// compute in f32 and round again.
#include <stdlib.h>

#include "v3d_common.h"
#include <mutex>
#include <unordered_map>

namespace v3d {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <typename T> struct Mfma16;
template <> struct Mfma16<bf16_t> {
  using frag = bf16x8;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mfma16<f16_t> {
  using frag = f16x8;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

template <int N> struct IntC { static constexpr int value = N; };

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_A_BYTES = BM * BK * 2;            // 16 KiB
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;      // 32 KiB
constexpr int GEMM_LDS_BYTES = 2 * STAGE_BYTES;      // 64 KiB -> 2 workgroups per CU
constexpr int C_ROW_BYTES = BN * 2 + 16;             // padded C-tile row in LDS

struct GemmArgs {
  const void* A; const void* W; const void* bias; const void* res; void* out;
  int M, N, K;
  int64_t lda, ldw, ldr, ldo;
  int res_mod;        // >0: residual row = m % res_mod (ViT position embedding)
  int tiles_m, tiles_n;
  int dma_late;       // 256-wide kernel: restage after (1) or before (0) the last phase's MFMAs of a K-step
  int skew;           // ping-pong kernel: largest start delay of a workgroup, in units of 1024 shader cycles (0 = none)
  // ping-pong kernel, split-K tail (sk_dp >= 0): per XCD sk_dp rounds of whole tiles, then each remaining tile is cut into
  // sk_split K-chunks run by neighbouring workgroups of the XCD AT THE SAME TIME (all chunk-c workgroups walk the same k range in
  // lock step, so operand panels are still shared in L2 - a stream-K walk with staggered k offsets ran at half speed: HBM-bound);
  // the chunk-0 workgroup adds the others' accumulator images and runs the epilogue
  int tile_base, tile_count;   // ping-pong kernel: the launch covers the logical tiles [tile_base, tile_base + tile_count) (count 0 = all)
  int sk_dp;          // -1: off
  int sk_split;       // 2 .. 4
  float* sk_ws;       // one accumulator image (32 x 512 x 16 bytes) per workgroup
  unsigned* sk_flags; // one word per workgroup: == sk_epoch once its image is written
  unsigned sk_epoch;
};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_tanh(float x) {
  // 0.5 x (1 + tanh(u)) = x * sigmoid(2u) = x / (1 + exp(-2u)),  u = sqrt(2/pi) (x + 0.044715 x^3):
  // 5 VALU + v_exp_f32 + v_rcp_f32 (each ~1 ulp; the result is rounded to 16 bits) instead of the branchy libm tanhf,
  // which cost the SigLIP fc1 GEMM a third of its time in the epilogue.
  const float k = -2.0f * 0.7978845608028654f * 1.4426950408889634f;        // -2 sqrt(2/pi) log2(e)
  const float z = x * fmaf(k * 0.044715f, x * x, k);
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z));
}
// x * sigmoid(x) with v_exp_f32 + v_rcp_f32 (the IEEE division sequence costs ~10 VALU per element of the SwiGLU epilogue)
__device__ __forceinline__ float silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// XCD-aware logical block id (blocks b and b+8 share an XCD's L2), then grouped row-major tiles:
// the ~64 workgroups resident on one XCD cover an 8 x 8 patch of tiles and share operand panels.
__device__ __forceinline__ void tile_of_logical(int L, int tiles_m, int tiles_n, int& tm, int& tn);
__device__ __forceinline__ int xcd_first_logical(int xcd, int nblocks) {     // first logical tile of an XCD's share, shares: q + (xcd < r)
  const int q = nblocks >> 3, r = nblocks & 7;
  return xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
}
__device__ __forceinline__ void tile_of_block(int bid, int nblocks, int tiles_m, int tiles_n, int& tm, int& tn) {
  tile_of_logical(xcd_first_logical(bid & 7, nblocks) + (bid >> 3), tiles_m, tiles_n, tm, tn);
}
__device__ __forceinline__ void tile_of_logical(int L, int tiles_m, int tiles_n, int& tm, int& tn) {
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * tiles_n;
  const int g = L / per_group, in_g = L - g * per_group;
  const int first_m = g * GROUP_M;
  const int gsz = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
  tm = first_m + in_g % gsz;
  tn = in_g / gsz;
}

enum { EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_GELU_ERF = 2, EPI_BIAS_GELU_TANH = 3, EPI_BIAS_RES = 4, EPI_RES = 5, EPI_SWIGLU = 6, EPI_BIAS_RELU = 7 };
__host__ __device__ constexpr bool epi_has_bias(int e) { return (e >= EPI_BIAS && e <= EPI_BIAS_RES) || e == EPI_BIAS_RELU; }

template <typename T, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using M16 = Mfma16<T>;
  using frag = typename M16::frag;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int tm, tn;
  tile_of_block(blockIdx.x, gridDim.x, p.tiles_m, p.tiles_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- staging: wave w moves rows [32w, 32w+32) of both tiles, 8 rows (1 KiB) per DMA ----
  const uint16_t* a_src[4];
  const uint16_t* w_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 32 + i * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);          // logical 16-B chunk that lands in slot lane&7
    int gm = m0 + row;
    gm = gm < p.M ? gm : p.M - 1;                             // M tail: re-read the last row, never stored
    a_src[i] = (const uint16_t*)p.A + (int64_t)gm * p.lda + chunk * 8;
    w_src[i] = (const uint16_t*)p.W + (int64_t)(n0 + row) * p.ldw + chunk * 8;
  }
  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * STAGE_BYTES + (wave * 32) * (BK * 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(a_src[i] + kt * BK, base + i * 8 * (BK * 2));
      glds16(w_src[i] + kt * BK, base + TILE_A_BYTES + i * 8 * (BK * 2));
    }
  };

  // ---- fragment read offsets (bytes inside one operand tile) ----
  // lane reads row (16*t + lane&15), logical chunk 4*ks + lane>>4; ((row>>1)&7) == ((lane>>1)&7)
  const int sw = (lane >> 1) & 7;
  const int frow = (lane & 15) * (BK * 2);
  const int fo0 = frow + (((0 + (lane >> 4)) ^ sw) << 4);
  const int fo1 = frow + (((4 + (lane >> 4)) ^ sw) << 4);

  f32x4 acc[4][4];   // [ni][mi], transposed product: D[row = n][col = m]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Software pipeline over the two 32-deep k-halves of each tile: the fragment reads of one half are
  // in flight under the 16 MFMAs of the other (two register sets), ONE barrier per K-step:
  //   read half1(t) -> fb | lgkmcnt(8): fa landed | MFMA half0 (fa) | lgkmcnt(0) vmcnt(0) barrier
  //   (tile t+1 visible, tile t fully read) | stage tile t+2 over tile t | read half0(t+1) -> fa | MFMA half1 (fb)
  // hipcc's waitcnt pass only emits lgkmcnt(0) for this loop-carried pattern, so the fragment reads are
  // inline asm (not tracked) and the counted waits are ours; every wait names the registers it
  // retires as "+v" operands so no consumer can be scheduled above it (cdna guide 5.7, form ii).
  const int nt = p.K / BK;
  using v4i = __attribute__((ext_vector_type(4))) int;
  v4i fa[8], fb[8];      // [0..3] = A-tile (m) fragments, [4..7] = W-tile (n) fragments
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned offA = lds0 + (wm * 64) * (BK * 2), offW = lds0 + TILE_A_BYTES + (wn * 64) * (BK * 2);
#define V3D_DSR(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:" #imm : "=v"(dst) : "v"(addr))
#define V3D_READ_HALF(f, buf, fo)                                         \
  {                                                                       \
    const unsigned aA = offA + (buf) * STAGE_BYTES + (fo), aW = offW + (buf) * STAGE_BYTES + (fo); \
    V3D_DSR(f[0], aA, 0); V3D_DSR(f[1], aA, 2048); V3D_DSR(f[4], aW, 0); V3D_DSR(f[5], aW, 2048);  \
    V3D_DSR(f[2], aA, 4096); V3D_DSR(f[3], aA, 6144); V3D_DSR(f[6], aW, 4096); V3D_DSR(f[7], aW, 6144); \
  }
#define V3D_WAIT(cnt, f)                                                                             \
  asm volatile("s_waitcnt " cnt : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), \
               "+v"(f[6]), "+v"(f[7]) : : "memory")
  auto mma = [&](const v4i* f) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
        acc[ni][mi] = M16::run(__builtin_bit_cast(frag, f[4 + ni]), __builtin_bit_cast(frag, f[mi]), acc[ni][mi]);
    __builtin_amdgcn_s_setprio(0);
  };
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (nt > 1) stage(1, 1);
  V3D_READ_HALF(fa, 0, fo0);
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    V3D_READ_HALF(fb, cur, fo1);
    V3D_WAIT("lgkmcnt(8)", fa);                 // the 8 older reads (fa) have landed, fb still in flight
    mma(fa);
    V3D_WAIT("vmcnt(0) lgkmcnt(0)", fb);        // fb landed (tile t fully read by this wave), tile t+1 landed
    __builtin_amdgcn_s_barrier();
    if (t + 2 < nt) stage(cur, t + 2);
    V3D_READ_HALF(fa, cur ^ 1, fo0);            // last step: stale bytes, never used
    mma(fb);
  }
  V3D_WAIT("lgkmcnt(0)", fa);
#undef V3D_DSR
#undef V3D_READ_HALF
#undef V3D_WAIT
  __syncthreads();   // every wave is done with the ring before the C tile overwrites it

  // ---- epilogue: accumulators -> LDS C tile (16-bit, bias added in f32 first) -> coalesced rows ----
  const T* bias = (const T*)p.bias;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int nl = wn * 64 + ni * 16 + 4 * (lane >> 4);
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (epi_has_bias(EPI)) {
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[r] = to_f32(bias[n0 + nl + r]);
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int ml = wm * 64 + mi * 16 + (lane & 15);
      uint2 pk;
      pk.x = pack2<T>(acc[ni][mi][0] + bv[0], acc[ni][mi][1] + bv[1]);
      pk.y = pack2<T>(acc[ni][mi][2] + bv[2], acc[ni][mi][3] + bv[3]);
      *reinterpret_cast<uint2*>(smem + ml * C_ROW_BYTES + nl * 2) = pk;
    }
  }
  __syncthreads();

  T* out = (T*)p.out;
  if (EPI == EPI_SWIGLU) {
    // tile columns [0,64) = gate_j, [64,128) = up_j of the same 64 j's; output width N/2
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (tid >> 3) + 32 * i, ch = tid & 7;
      const int gm = m0 + row;
      const uint4 g = *reinterpret_cast<const uint4*>(smem + row * C_ROW_BYTES + ch * 16);
      const uint4 u = *reinterpret_cast<const uint4*>(smem + row * C_ROW_BYTES + 128 + ch * 16);
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = round_to<T>(silu(vec_get<T>(g, j))) * vec_get<T>(u, j);
      if (gm < p.M) *reinterpret_cast<uint4*>(out + (int64_t)gm * p.ldo + tn * 64 + ch * 8) = vec_pack<T>(v);
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = (tid >> 4) + 16 * i, ch = tid & 15;
    const int gm = m0 + row;
    if (gm >= p.M) continue;
    uint4 c = *reinterpret_cast<const uint4*>(smem + row * C_ROW_BYTES + ch * 16);
    if (EPI == EPI_BIAS_GELU_ERF || EPI == EPI_BIAS_GELU_TANH || EPI == EPI_BIAS_RES || EPI == EPI_RES || EPI == EPI_BIAS_RELU) {
      float v[8];
      uint4 rr = make_uint4(0, 0, 0, 0);
      if (EPI == EPI_BIAS_RES || EPI == EPI_RES) {
        const int64_t rm = p.res_mod > 0 ? (gm % p.res_mod) : gm;
        rr = *reinterpret_cast<const uint4*>((const T*)p.res + rm * p.ldr + n0 + ch * 8);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float x = vec_get<T>(c, j);
        v[j] = EPI == EPI_BIAS_GELU_ERF ? gelu_erf(x) : EPI == EPI_BIAS_GELU_TANH ? gelu_tanh(x) : EPI == EPI_BIAS_RELU ? fmaxf(x, 0.f) : x + vec_get<T>(rr, j);
      }
      c = vec_pack<T>(v);
    }
    *reinterpret_cast<uint4*>(out + (int64_t)gm * p.ldo + n0 + ch * 8) = c;
  }
}


// ------------------------------------------------------------------------------------------
// v3: 256 x 256 x 64 tile, 512 threads = 8 waves (2 x 4), each wave 128 x 64 (8 x 4 accumulators).
// Twice the operand reuse of the 64 x 64 wave tile: 24 ds_read_b128 feed 64 MFMAs per K-step, which
// takes the LDS pipe (fragment reads + DMA writes) off the critical path.  Four phases of 16 MFMAs per
// K-step; the fragment reads of phase p+1 are in flight under the MFMAs of phase p (two A and two B
// register sets, counted lgkmcnt), one barrier and one 8-DMA stage per K-step on a 2-deep ring.
// ------------------------------------------------------------------------------------------
constexpr int B3N = 256;
constexpr int T3_BYTES = 256 * BK * 2;                  // 32 KiB per operand tile (A tile uses BM rows of it)
constexpr int S3_BYTES = 2 * T3_BYTES;                  // 64 KiB per stage
constexpr int GEMM3_LDS_BYTES = 2 * S3_BYTES;           // 128 KiB

// MT = 16-row accumulator tiles per wave along M: 8 -> BM = 256, 6 -> BM = 192 (picked so that the tile count
// fills whole rounds of the 256 CUs, e.g. M = 6794, N = 3584: 27 x 14 = 378 tiles (74 %) vs 36 x 14 = 504 (98 %)).
// Register budget: <= 224 per lane (2 waves/SIMD = 448 of 512) leaves one 64-register slot per SIMD, which is
// what a linear_decode wave of ANOTHER scene needs to stream weights beside this kernel (bench.py run_pipelined).
#ifdef V3D_GEMM_PROF   // tools/probes/gemm_prof.hip only: per-wave shader-clock split of the K loop (never in the product build)
__device__ unsigned long long g_gemm_prof[8 * 64];
#define V3D_GSTAMP(v) unsigned long long v; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory")
#else
#define V3D_GSTAMP(v)
#endif

template <typename T, int EPI, int MT>
__global__ __launch_bounds__(512, 2) void gemm256x256_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using M16 = Mfma16<T>;
  using frag = typename M16::frag;
  using v4i = __attribute__((ext_vector_type(4))) int;
  constexpr int MG = MT / 2;              // m-tiles per phase
  constexpr int WROWS = MT * 16;          // rows per wave
  constexpr int BM = 2 * WROWS;
  constexpr int APW = BM / 8 / 8;         // A DMA pieces (8 rows) per wave: 4 or 3

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  int tm, tn;
  tile_of_block(blockIdx.x, gridDim.x, p.tiles_m, p.tiles_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * B3N;

  // per-lane source offsets in elements (32 bit: M*lda and N*ldw stay below 2^31 on this path; checked by the host)
  // Staging alternates between the two waves that share a SIMD (w and w ^ 4): in K-step t the waves with wm == (t & 1)
  // issue the DMAs of BOTH (16 instructions), the others none.  When all eight waves restage right after the barrier the
  // vector-memory path backs up and a wave stuck on DMA issue cannot issue the MFMAs behind it - with both waves of a SIMD
  // stuck at once the matrix pipe idles (tools/probes/gemm_prof.hip); now one of them always runs MFMAs.
  unsigned a_off[2][APW], w_off[2][4];   // [own / partner] BYTE offsets, unsigned: SGPR-base + 32-bit VGPR-offset DMA form
#pragma unroll
  for (int o = 0; o < 2; ++o) {
    const int wv = o == 0 ? wave : (wave ^ 4);
#pragma unroll
    for (int i = 0; i < APW; ++i) {
      const int row = wv * (APW * 8) + i * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      int gm = m0 + row;
      gm = gm < p.M ? gm : p.M - 1;
      a_off[o][i] = (unsigned)(gm * (int)p.lda + chunk * 8) * 2u;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wv * 32 + i * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      w_off[o][i] = (unsigned)((n0 + row) * (int)p.ldw + chunk * 8) * 2u;
    }
  }
  auto stage_one = [&](int buf, int kt, int o) {
    const int wv = o == 0 ? wave : (wave ^ 4);
    char* ba = smem + buf * S3_BYTES + (wv * APW * 8) * (BK * 2);
    char* bw = smem + buf * S3_BYTES + T3_BYTES + (wv * 32) * (BK * 2);
    const char* Ak = (const char*)p.A + (size_t)kt * (BK * 2);
    const char* Wk = (const char*)p.W + (size_t)kt * (BK * 2);
#pragma unroll
    for (int i = 0; i < APW; ++i) glds16(Ak + a_off[o][i], ba + i * 8 * (BK * 2));
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(Wk + w_off[o][i], bw + i * 8 * (BK * 2));
  };
  auto stage = [&](int buf, int kt) {          // every wave its own share (prologue)
    stage_one(buf, kt, 0);
  };
  auto stage_alt = [&](int buf, int kt, int turn) {      // steady state: the waves whose turn it is stage for the pair
    if (wm == (turn & 1)) { stage_one(buf, kt, 0); stage_one(buf, kt, 1); }
  };

  // all wave row bases are multiples of 16, so ((row>>1)&7) == ((lane>>1)&7)
  const int sw = (lane >> 1) & 7;
  const int frow = (lane & 15) * (BK * 2);
  const unsigned fo0 = frow + (((0 + (lane >> 4)) ^ sw) << 4);
  const unsigned fo1 = frow + (((4 + (lane >> 4)) ^ sw) << 4);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned offA = lds0 + (wm * WROWS) * (BK * 2), offW = lds0 + T3_BYTES + (wn * 64) * (BK * 2);
  const unsigned offA_hi = offA + MG * 2048;             // second group of m-tiles

  f32x4 acc[4][MT];   // [nt][mt]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  v4i A0[4], A1[4], B0[4], B1[4];
#define V3D_DSR(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:" #imm : "=v"(dst) : "v"(addr))
#define V3D_RD4(f, base) { V3D_DSR(f[0], base, 0); V3D_DSR(f[1], base, 2048); V3D_DSR(f[2], base, 4096); V3D_DSR(f[3], base, 6144); }
#define V3D_RD3(f, base) { V3D_DSR(f[0], base, 0); V3D_DSR(f[1], base, 2048); V3D_DSR(f[2], base, 4096); }
#define V3D_RDA(f, base) { if constexpr (MG == 4) V3D_RD4(f, base) else V3D_RD3(f, base) }
#ifdef V3D_GEMM_NOWAIT   // probe only (WRONG RESULTS): fragment waits removed, to see how much of a K-step is LDS latency
#define V3D_W4(cnt, f) asm volatile("s_nop 0" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : : "memory")
#define V3D_W8(cnt, f, g) asm volatile("s_nop 0" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]) : : "memory")
#else
#define V3D_W4(cnt, f) asm volatile("s_waitcnt " cnt : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : : "memory")
#define V3D_W8(cnt, f, g) asm volatile("s_waitcnt " cnt : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]) : : "memory")
#endif
#define V3D_MMA(FA, FB, G)                                                                             \
  {                                                                                                    \
    __builtin_amdgcn_s_setprio(1);                                                                     \
    _Pragma("unroll") for (int i = 0; i < MG; ++i)                                                     \
    _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                                   \
        acc[ni][(G) * MG + i] = M16::run(__builtin_bit_cast(frag, FB[ni]), __builtin_bit_cast(frag, FA[i]), acc[ni][(G) * MG + i]); \
    __builtin_amdgcn_s_setprio(0);                                                                     \
  }

  if constexpr (MG == 3) { A0[3] = v4i{0, 0, 0, 0}; A1[3] = v4i{0, 0, 0, 0}; }
  const int nt = p.K / BK;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (nt > 1) stage(1, 1);
  {
    const unsigned a = offA + fo0, w = offW + fo0;
    V3D_RDA(A0, a);
    V3D_RD4(B0, w);
  }
#ifdef V3D_GEMM_PROF
  unsigned long long gp_mma = 0, gp_wait = 0, gp_bar = 0, gp_tail = 0;
  V3D_GSTAMP(gp_t0);
  unsigned long long gp_prev = gp_t0;
#endif
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    const unsigned ah0 = offA_hi + cur * S3_BYTES + fo0, al1 = offA + cur * S3_BYTES + fo1, ah1 = offA_hi + cur * S3_BYTES + fo1;
    const unsigned w1 = offW + cur * S3_BYTES + fo1;
    // phase 0: k-half 0, first group of m-tiles
    V3D_RDA(A1, ah0);
    if constexpr (MG == 4) V3D_W8("lgkmcnt(4)", A0, B0); else V3D_W8("lgkmcnt(3)", A0, B0);
    V3D_MMA(A0, B0, 0);
    // phase 1: k-half 0, second group
    V3D_RDA(A0, al1);
    V3D_RD4(B1, w1);
    if constexpr (MG == 4) V3D_W4("lgkmcnt(8)", A1); else V3D_W4("lgkmcnt(7)", A1);
    V3D_MMA(A1, B0, 1);
    // phase 2: k-half 1, first group
    V3D_RDA(A1, ah1);
    if constexpr (MG == 4) V3D_W8("lgkmcnt(4)", A0, B1); else V3D_W8("lgkmcnt(3)", A0, B1);
    V3D_MMA(A0, B1, 0);
    // phase 3: k-half 1, second group; tile t fully read, tile t+1 landed -> rendezvous, restage
    V3D_GSTAMP(gp_a);
    V3D_W4("vmcnt(0) lgkmcnt(0)", A1);
    V3D_GSTAMP(gp_b);
    __builtin_amdgcn_s_barrier();
    V3D_GSTAMP(gp_c);
    if (!p.dma_late && t + 2 < nt) stage_alt(cur, t + 2, t);
    {
      const unsigned a = offA + (cur ^ 1) * S3_BYTES + fo0, w = offW + (cur ^ 1) * S3_BYTES + fo0;
      V3D_RDA(A0, a);      // next tile's phase-0 fragments (last step: stale, unused)
      V3D_RD4(B0, w);
    }
    V3D_MMA(A1, B1, 1);
    // dma_late (A/B option): restage after the phase's MFMAs instead of before them
    if (p.dma_late && t + 2 < nt) stage_alt(cur, t + 2, t);
#ifdef V3D_GEMM_PROF
    { V3D_GSTAMP(gp_d);
      gp_mma += gp_a - gp_prev; gp_wait += gp_b - gp_a; gp_bar += gp_c - gp_b; gp_tail += gp_d - gp_c; gp_prev = gp_d; }
#endif
  }
#ifdef V3D_GEMM_PROF
  if (lane == 0 && blockIdx.x < 8) {
    unsigned long long* pr = g_gemm_prof + (blockIdx.x * 8 + wave) * 8;
    pr[0] = gp_mma; pr[1] = gp_wait; pr[2] = gp_bar; pr[3] = gp_tail; pr[4] = gp_prev - gp_t0; pr[5] = (unsigned long long)nt;
  }
#endif
  V3D_W8("lgkmcnt(0)", A0, B0);
#undef V3D_DSR
#undef V3D_RD4
#undef V3D_RD3
#undef V3D_RDA
#undef V3D_W4
#undef V3D_W8
#undef V3D_MMA
  __syncthreads();

  // epilogue in two halves of WROWS rows (the whole C tile does not fit the ring with padding)
  constexpr int C3_ROW = B3N * 2 + 16;
  const T* bias = (const T*)p.bias;
  T* out = (T*)p.out;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (wm == half) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int nl = wn * 64 + ni * 16 + 4 * (lane >> 4);
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (epi_has_bias(EPI) || bias != nullptr) {   // (bias is NULL for the bias-free epilogues; keeping the
#pragma unroll                                          //  load site in every variant keeps hipcc at <= 224 VGPRs)
          for (int r = 0; r < 4; ++r) bv[r] = to_f32(bias[n0 + nl + r]);
        }
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
          const int ml = mi * 16 + (lane & 15);
          uint2 pk;
          pk.x = pack2<T>(acc[ni][mi][0] + bv[0], acc[ni][mi][1] + bv[1]);
          pk.y = pack2<T>(acc[ni][mi][2] + bv[2], acc[ni][mi][3] + bv[3]);
          *reinterpret_cast<uint2*>(smem + ml * C3_ROW + nl * 2) = pk;
          __builtin_amdgcn_sched_barrier(0);      // keep the packs in order: bounds the live temporaries (<= 224 VGPRs)
        }
      }
    }
    __syncthreads();
    if (EPI == EPI_SWIGLU) {
      // tile columns: per 128-column group [gate64 | up64]; two groups per tile -> 128 output columns
      for (int row = tid >> 4; row < WROWS; row += 32) {
        const int c16 = tid & 15, grp = c16 >> 3, ch = c16 & 7;
        const int gm = m0 + half * WROWS + row;
        const uint4 g = *reinterpret_cast<const uint4*>(smem + row * C3_ROW + grp * 256 + ch * 16);
        const uint4 u = *reinterpret_cast<const uint4*>(smem + row * C3_ROW + grp * 256 + 128 + ch * 16);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = round_to<T>(silu(vec_get<T>(g, j))) * vec_get<T>(u, j);
        if (gm < p.M) *reinterpret_cast<uint4*>(out + (int64_t)gm * p.ldo + tn * 128 + grp * 64 + ch * 8) = vec_pack<T>(v);
      }
    } else {
      for (int row = tid >> 5; row < WROWS; row += 16) {
        const int ch = tid & 31;
        const int gm = m0 + half * WROWS + row;
        if (gm < p.M) {
          uint4 c = *reinterpret_cast<const uint4*>(smem + row * C3_ROW + ch * 16);
          if (EPI == EPI_BIAS_GELU_ERF || EPI == EPI_BIAS_GELU_TANH || EPI == EPI_BIAS_RES || EPI == EPI_RES || EPI == EPI_BIAS_RELU) {
            float v[8];
            uint4 rr = make_uint4(0, 0, 0, 0);
            if (EPI == EPI_BIAS_RES || EPI == EPI_RES) {
              const int64_t rm = p.res_mod > 0 ? (gm % p.res_mod) : gm;
              rr = *reinterpret_cast<const uint4*>((const T*)p.res + rm * p.ldr + n0 + ch * 8);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float x = vec_get<T>(c, j);
              v[j] = EPI == EPI_BIAS_GELU_ERF ? gelu_erf(x) : EPI == EPI_BIAS_GELU_TANH ? gelu_tanh(x) : EPI == EPI_BIAS_RELU ? fmaxf(x, 0.f) : x + vec_get<T>(rr, j);
            }
            c = vec_pack<T>(v);
          }
          *reinterpret_cast<uint4*>(out + (int64_t)gm * p.ldo + n0 + ch * 8) = c;
        }
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// v4: the same 256 x 256 x 64 macro tile and 128 x 64 wave tile, scheduled as a PING-PONG between the two waves that
// share a SIMD (waves w and w + 4, i.e. the wave rows wm = 0 / 1): a K-step is four phases of 16 MFMAs (one 64 x 32
// quadrant of the wave tile x both k-halves); every phase is { load segment: the quadrant's new fragments by ds_read_b128 +
// two LDS-DMA pieces of a later half-tile | barrier | lgkmcnt(0) | 16 MFMAs | barrier }, and the wm = 1 waves run ONE barrier
// behind the wm = 0 waves, so that on every SIMD one wave is inside its MFMA cluster while its partner issues its loads:
// fragment-read latency and DMA issue cost (60-185 cycles per piece) sit in the shadow of the partner's 256 MFMA cycles
// instead of in front of the wave's own MFMAs (v3 keeps ~30 % of its wave cycles at the K-step rendezvous).
//
// Operand tiles are staged in HALF-tiles of 128 rows (16 KiB = 2 pieces per wave), ordered as they are consumed:
//   H(4t+0) = A rows {128 wm + r}, r < 64   (quadrants with hm = 0)      H(4t+1) = W rows {64 wn + r}, r < 32  (hn = 0)
//   H(4t+2) = W rows {64 wn + 32 + r}                      (hn = 1)      H(4t+3) = A rows {128 wm + 64 + r}    (hm = 1)
// phases of K-step t: (hm, hn) = (0,0) (0,1) (1,1) (1,0) - each needs at most one new half of each operand; W(hn = 0) stays in
// registers for the fourth.  Ring: 8 half-tile slots (2 K-steps, 128 KiB); phase P issues H(P + 5) and waits vmcnt(6), i.e.
// until everything up to H(P + 2) - what phase P + 1 reads - has landed (its own pieces; the barriers publish the others').
// Hazards, with B(k) the k-th workgroup barrier, phase P of the wm = 0 waves between B(2P) and B(2P+2), of the wm = 1
// waves between B(2P+1) and B(2P+3):
//   RAW  H(j) is read in phase c(j) >= j - 1 >= (its wait's phase) + 1, i.e. after a barrier every wave passed after its wait;
//   WAR  H(P + 5) overwrites the slot of H(P - 3), last read in phase <= P - 3: those reads were retired (lgkmcnt(0)) before
//        B(2P - 3) by both wave rows, and the earliest overwrite is issued after B(2P).
// ------------------------------------------------------------------------------------------
constexpr int PP_HALF = 128 * BK * 2;                   // 16 KiB per half-tile
constexpr int PP_BUF = 4 * PP_HALF;                     // one K-step: A0 | W0 | W1 | A1
constexpr int PP_LDS_BYTES = 256 * (B3N * 2 + 16);      // 132 KiB: the C tile of the epilogue (the 128 KiB ring fits inside)

#ifdef V3D_PP_TIMELINE   // tools/probes/gemm_pp_timeline.hip only: shader-clock stamps around a tile of the ping-pong kernel
__device__ unsigned long long g_pp_tl[256 * 8 * 8];     // [workgroup][wave][segment sums: 0 zero+first wait, 1 K loop, 2 ring-free barrier, 3 next-tile issue, 4 epilogue, 5 store drain, 6 tiles]
#define V3D_TL(v) unsigned long long v; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory")
#define V3D_TL_ACC(i, a, b) tl_acc[i] += (b) - (a)
#else
#define V3D_TL(v)
#define V3D_TL_ACC(i, a, b)
#endif

// SKT: the split-K tail's exchange code is compiled in.  (With an early `break` / `continue` out of the image dump that code made
// hipcc spill 40-90 accumulator registers in EVERY tile, cut or not - which is why it lives in an instantiation of its own; written
// as straight-line flags neither instantiation spills.)
template <typename T, int EPI, int MT, bool SKT>
__global__ __launch_bounds__(512, 2) void gemm256pp_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using M16 = Mfma16<T>;
  using frag = typename M16::frag;
  using v4i = __attribute__((ext_vector_type(4))) int;
  // MT = 16-row accumulator tiles per wave along M: 8 -> BM = 256; 6 -> BM = 192 (A half-tiles of 96 rows = 12 pieces: the
  // wm = 0 waves move two of them, the wm = 1 waves one, so the counted waits differ between the wave rows)
  constexpr int MH = MT / 2;                // m-tiles per quadrant
  constexpr int HR = MH * 16;               // A rows per wave row and half
  constexpr int WROWS = MT * 16, BM = 2 * WROWS;
  constexpr int NPA = 2 * HR / 8;           // pieces of an A half-tile: 16 or 12

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int n_tiles = p.tile_count > 0 ? p.tile_count : p.tiles_m * p.tiles_n;      // tiles of THIS launch

  // PERSISTENT: the grid is one workgroup per CU (or per tile when there are fewer); workgroup b walks the logical tiles
  // b, b + grid, b + 2 grid, ... (the XCD-aware grouped order of tile_of_block: the stride is a multiple of 8, so a workgroup
  // stays in its XCD's column of the order).  After a tile's K loop the K-step-0 half-tiles of the NEXT tile are issued
  // before the epilogue, so their HBM latency (and the launch / dispatch gap between rounds) hides under the C-tile stores.
  int bid = blockIdx.x;
  // Start skew: with one tile per CU per round and equal tile times every CU reaches its epilogue at the same moment, and the
  // chip alternates between "all CUs on the matrix cores, HBM nearly idle" and "all CUs storing C tiles, matrix cores idle"
  // (the burst is HBM-bound: 256 x 128 KiB at once).  Delaying workgroup b by ((5 b) mod 16) / 16 of `skew` spreads the
  // epilogues of later rounds over the K loops of the other CUs; equal tile times keep the offsets for the whole launch.
  if (p.skew > 0) {
    const int units = (((int)blockIdx.x * 5) & 15) * p.skew / 16;
    for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(16);
  }
  int tm, tn, m0, n0;
  // ---- the work of this workgroup: a list of SEGMENTS (tile, K-step range).  Plain persistent walk: whole tiles b, b + grid, ...
  // Split-K tail (p.sk_dp >= 0): with g = xcd wx + idx the XCD-major id of workgroup (xcd = b & 7, idx = b >> 3; wx = grid / 8), the
  // first sk_dp grid tiles of the logical order are whole tiles - XCD x owns the block [x dp wx, (x + 1) dp wx), round r tile
  // r wx + idx of it - and the T = n_tiles - dp grid tiles left go out as items (chunk c, tile j) = (g / T, g mod T), g < split T:
  // an XCD's workgroups take neighbouring tiles of (mostly) ONE chunk, so they walk the same k range in lock step and share operand
  // panels in L2 exactly as in a whole-tile round.  Chunk c covers the granules (2 K-steps) [c gpt / split, (c + 1) gpt / split).
  // A chunk c > 0 ends by writing its accumulators to the workspace and raising its flag - it never waits - and chunk 0 adds the
  // images of the workgroups that ran the tile's other chunks and runs the epilogue.
  const int nt = p.K / BK;       // >= 2 (the launcher sends shorter K to the v3 kernel)
  const bool sk = SKT && p.sk_dp >= 0;
  const int xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3, wx = (int)gridDim.x >> 3;
  const int gpt = nt >> 1;                                                   // granules per tile
  const int sk_g = xcd * wx + idx, sk_T = n_tiles - p.sk_dp * (int)gridDim.x;
  int dp_round = 0;
  bool sk_item = sk && sk_g < p.sk_split * sk_T;                             // this workgroup has a chunk of the tail
  int k_begin = 0, k_end = nt;
  // next segment -> (L, k range); false when the list is exhausted
  auto next_segment = [&](int& L) -> bool {
    if (!sk) { if (bid >= n_tiles) return false; L = p.tile_base + xcd_first_logical(bid & 7, n_tiles) + (bid >> 3); bid += (int)gridDim.x; k_begin = 0; k_end = nt; return true; }
    if (dp_round < p.sk_dp) { L = p.tile_base + (xcd * p.sk_dp + dp_round) * wx + idx; ++dp_round; k_begin = 0; k_end = nt; return true; }
    if (!sk_item) return false;
    sk_item = false;
    const int c = sk_g / sk_T;
    L = p.tile_base + p.sk_dp * (int)gridDim.x + (sk_g - c * sk_T);
    k_begin = 2 * (c * gpt / p.sk_split); k_end = 2 * ((c + 1) * gpt / p.sk_split);
    return true;
  };
  // ---- staging: piece q = wave + 8 i (8 rows, 1 KiB) of every half-tile; W halves have 16 pieces, A halves NPA ----
  unsigned a_off[2][2], w_off[2][2];        // [half][piece] byte offsets of this lane's 16-byte source chunk
  auto setup = [&](int tm_, int tn_) {         // per-lane source offsets of tile (tm_, tn_)
    tm = tm_; tn = tn_;
#if defined(V3D_PP_PROBE) && defined(V3D_PP_SAMETILE)   // probe: every workgroup reads tile (0, 0)'s operands - the K loop with everything L2-resident
    tm = 0; tn = 0;
#endif
    m0 = tm * BM; n0 = tn * B3N;
    int ln = lane;
    asm volatile("" : "+v"(ln));               // recompute the lane constants here: hoisted, they live across the K loop and spill
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int lr = (wave + 8 * i) * 8 + (ln >> 3);
      const int chunk = (ln & 7) ^ ((lr >> 1) & 7);
      const int awm = lr / HR, ar = lr - awm * HR;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int gm = m0 + WROWS * awm + HR * h + ar;
        gm = gm < p.M ? gm : p.M - 1;                            // M tail: re-read the last row, never stored
        a_off[h][i] = (unsigned)(gm * (int)p.lda + chunk * 8) * 2u;
        // W half hn holds, for every wave column wn = lr >> 5, the 32 tile columns the wave multiplies in the (., hn) phases.
        // SwiGLU (tile columns = two groups of [gate64 | up64]): wave wn gets gate columns (hn 0) and the MATCHING up columns
        // (hn 1) of one 32-wide output block, so silu(gate) * up is formed in registers, without an exchange between waves.
        const int gn = EPI == EPI_SWIGLU ? n0 + 128 * (lr >> 6) + 64 * h + 32 * ((lr >> 5) & 1) + (lr & 31)
                                         : n0 + 64 * (lr >> 5) + 32 * h + (lr & 31);
        w_off[h][i] = (unsigned)(gn * (int)p.ldw + chunk * 8) * 2u;
      }
    }
  };
  const bool a_two = wave + 8 < NPA;        // this wave moves a second A piece (always for MT = 8; the wm = 0 waves for MT = 6)
  // half-tile kinds in stream order: 0 = A(hm 0), 1 = W(hn 0), 2 = W(hn 1), 3 = A(hm 1)
  auto stage = [&](auto kind_c, int kt) {
    constexpr int KIND = decltype(kind_c)::value;
    constexpr bool IS_A = KIND == 0 || KIND == 3;
    char* dst = smem + (kt & 1) * PP_BUF + KIND * PP_HALF + (wave * 8) * (BK * 2);
    const char* src = IS_A ? (const char*)p.A : (const char*)p.W;
    src += (size_t)kt * (BK * 2);
    const unsigned* off = KIND == 0 ? a_off[0] : KIND == 3 ? a_off[1] : KIND == 1 ? w_off[0] : w_off[1];
    glds16(src + off[0], dst);
    if (!IS_A || NPA == 16 || a_two) glds16(src + off[1], dst + 64 * (BK * 2));
  };

  const int sw = (lane >> 1) & 7;
  const unsigned frow = (lane & 15) * (BK * 2);
  const unsigned fo0 = frow + (((0 + (lane >> 4)) ^ sw) << 4);
  const unsigned fo1 = frow + (((4 + (lane >> 4)) ^ sw) << 4);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned aA = lds0 + (wm * HR) * (BK * 2);        // + buf * PP_BUF + {0 | 3 * PP_HALF} + fo{0,1}; m-tile: + 2048 each
  const unsigned aW = lds0 + PP_HALF + (wn * 32) * (BK * 2);   // + buf * PP_BUF + {0 | PP_HALF}

  f32x4 acc[4][MT];   // [n-tile][m-tile] of the transposed product
  v4i FA[2 * MH], W0[4], W1[4];     // FA[2 * mt + kh], W[2 * nt2 + kh]
#define V3D_DSR(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(imm))
#define V3D_RDA(b0, b1) { V3D_DSR(FA[0], b0, 0); V3D_DSR(FA[1], b1, 0); V3D_DSR(FA[2], b0, 2048); V3D_DSR(FA[3], b1, 2048); \
                          V3D_DSR(FA[4], b0, 4096); V3D_DSR(FA[5], b1, 4096);                                                 \
                          if constexpr (MH == 4) { V3D_DSR(FA[6], b0, 6144); V3D_DSR(FA[7], b1, 6144); } }
#define V3D_RDW4(f, b0, b1) { V3D_DSR(f[0], b0, 0); V3D_DSR(f[1], b1, 0); V3D_DSR(f[2], b0, 2048); V3D_DSR(f[3], b1, 2048); }
#define V3D_LG0_A() { if constexpr (MH == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(FA[0]), "+v"(FA[1]), "+v"(FA[2]), "+v"(FA[3]), "+v"(FA[4]), "+v"(FA[5]), "+v"(FA[6]), "+v"(FA[7]) : : "memory"); \
                      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(FA[0]), "+v"(FA[1]), "+v"(FA[2]), "+v"(FA[3]), "+v"(FA[4]), "+v"(FA[5]) : : "memory"); }
#define V3D_LG0_W(f) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : : "memory")
#define V3D_PPMMA(HM, HN, WF)                                                                          \
  {                                                                                                    \
    __builtin_amdgcn_s_setprio(1);                                                                     \
    _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                                   \
    _Pragma("unroll") for (int n2 = 0; n2 < 2; ++n2)                                                   \
    _Pragma("unroll") for (int mt = 0; mt < MH; ++mt)                                                  \
        acc[2 * (HN) + n2][MH * (HM) + mt] = M16::run(__builtin_bit_cast(frag, WF[2 * n2 + kh]), __builtin_bit_cast(frag, FA[2 * mt + kh]), \
                                                      acc[2 * (HN) + n2][MH * (HM) + mt]);             \
    __builtin_amdgcn_s_setprio(0);                                                                     \
  }
  // counted wait: n0 for the waves that move two pieces of every half-tile, n1 for those with a single A piece (MT = 6, wm = 1)
#define V3D_VM(n0, n1) { if (NPA == 16 || a_two) asm volatile("s_waitcnt vmcnt(" #n0 ")" ::: "memory"); else asm volatile("s_waitcnt vmcnt(" #n1 ")" ::: "memory"); }
#define V3D_BAR() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }

  // MODE 0: steady state; 1: second-last K-step (its fourth phase has nothing left to issue); 2: last K-step.
  // After phase P's issue everything up to H(P + 2) must have landed: H(P+3..P+5) may stay in flight - kinds (A A W), (A W W),
  // (W W A), (W A A) for P mod 4 = 0..3, i.e. 6 pieces, or 4 / 5 / 5 / 4 for a wave with single A pieces.
  auto kstep = [&](auto mode_c, int t) {
    constexpr int MODE = decltype(mode_c)::value;
    const unsigned bo = (unsigned)(t & 1) * PP_BUF;
    // phase 0: quadrant (hm 0, hn 0)
    { const unsigned w0 = aW + bo + fo0, w1 = aW + bo + fo1, a0 = aA + bo + fo0, a1 = aA + bo + fo1;
      V3D_RDW4(W0, w0, w1); V3D_RDA(a0, a1); }
    if constexpr (MODE < 2) { stage(IntC<1>{}, t + 1); V3D_VM(6, 4); } else { V3D_VM(2, 1); }
    V3D_BAR(); V3D_LG0_W(W0); V3D_LG0_A();
    V3D_PPMMA(0, 0, W0);
    V3D_BAR();
    // phase 1: quadrant (hm 0, hn 1)
    { const unsigned w0 = aW + bo + PP_HALF + fo0, w1 = aW + bo + PP_HALF + fo1;
      V3D_RDW4(W1, w0, w1); }
    if constexpr (MODE < 2) { stage(IntC<2>{}, t + 1); V3D_VM(6, 5); } else { V3D_VM(0, 0); }
    V3D_BAR(); V3D_LG0_W(W1);
    V3D_PPMMA(0, 1, W1);
    V3D_BAR();
    // phase 2: quadrant (hm 1, hn 1)
    { const unsigned a0 = aA + bo + 3 * PP_HALF + fo0, a1 = aA + bo + 3 * PP_HALF + fo1;
      V3D_RDA(a0, a1); }
    if constexpr (MODE < 2) { stage(IntC<3>{}, t + 1); V3D_VM(6, 5); }
    V3D_BAR(); V3D_LG0_A();
    V3D_PPMMA(1, 1, W1);
    V3D_BAR();
    // phase 3: quadrant (hm 1, hn 0): no new fragments
    if constexpr (MODE == 0) { stage(IntC<0>{}, t + 2); V3D_VM(6, 4); } else if constexpr (MODE == 1) { V3D_VM(4, 3); }
    V3D_BAR();
    V3D_PPMMA(1, 0, W0);
    V3D_BAR();
  };

  const T* bias = (const T*)p.bias;
  T* out = (T*)p.out;

  {
    int L, t0, t1;
    if (!next_segment(L)) return;               // (split tail with fewer items than workgroups: nothing to do)
    tile_of_logical(L, p.tiles_m, p.tiles_n, t0, t1);
    setup(t0, t1);
  }
  stage(IntC<0>{}, k_begin); stage(IntC<1>{}, k_begin); stage(IntC<2>{}, k_begin); stage(IntC<3>{}, k_begin);
#ifdef V3D_PP_TIMELINE
  unsigned long long tl_acc[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
  while (true) {
    V3D_TL(tl0);
    // the NEXT segment is located here, before the accumulators are live (its integer divisions want registers)
    const int cur_kb = k_begin, cur_ke = k_end;
    int n_tm = 0, n_tn = 0;
    bool has_next;
    {
      int L;
      has_next = next_segment(L);                 // workgroup-uniform; k_begin / k_end now describe the NEXT segment (k_begin even)
      if (has_next) tile_of_logical(L, p.tiles_m, p.tiles_n, n_tm, n_tn);
    }
    const int n_kb = k_begin, n_ke = k_end;
    k_begin = cur_kb; k_end = cur_ke;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    stage(IntC<0>{}, k_begin + 1);                // H4 (slot 4: free - the previous tile's epilogue has left it; k_begin is even)
    V3D_VM(6, 4);                                 // H0, H1 landed; (W A A) in flight (later tiles: everything but H4 landed long ago)
    V3D_BAR();
    if (wm == 1) V3D_BAR();                      // the wm = 1 waves run one barrier behind
    V3D_TL(tl1);
    {
      int t = k_begin;
      for (; t < k_end - 2; ++t) kstep(IntC<0>{}, t);
      kstep(IntC<1>{}, t);
      kstep(IntC<2>{}, t + 1);
    }
    V3D_TL(tl2);
    // Epilogue operands - bias, and the residual rows of the first two 32-row quarters in the epilogue's read-back layout - are
    // requested here and COMPLETED right after the ring-free barrier, before the next tile's DMAs are issued: the compiler waits for a
    // load with a vmcnt that counts only the memory operations it knows, and a load issued after those DMAs would wait for their
    // HBM latency too (16 k cycles per tile for a bias epilogue, 35-40 k for a residual one: tools/probes/gemm_pp_timeline.hip).
    // (Asm loads with hand-counted waits were tried: hipcc spills or copies their destination registers while the data is in flight.)
    constexpr bool HAS_RES = EPI == EPI_BIAS_RES || EPI == EPI_RES;
    typedef int v2i_t __attribute__((ext_vector_type(2)));
    v2i_t e_bias[4];
    v4i e_res[HAS_RES ? 2 : 1][4];
    auto res_load = [&](int q, v4i (&dst)[4], int ln) {
      const int colr = n0 + wn * 64 + (ln & 7) * 8;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int gm = m0 + wm * WROWS + 32 * q + (ln >> 3) + 8 * j;
        const int gc = gm < p.M ? gm : p.M - 1;
        const int64_t rm = p.res_mod > 0 ? (gc % p.res_mod) : gc;
        dst[j] = *reinterpret_cast<const v4i*>((const T*)p.res + rm * p.ldr + colr);
      }
    };
    if constexpr (EPI != EPI_SWIGLU) {
      int ln = lane;
      asm volatile("" : "+v"(ln));
      if constexpr (epi_has_bias(EPI)) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) e_bias[ni] = *reinterpret_cast<const v2i_t*>(bias + n0 + wn * 64 + 16 * ni + 4 * (ln >> 4));
      }
      if constexpr (HAS_RES) { res_load(0, e_res[0], ln); res_load(1, e_res[1], ln); }
    }
    if (wm == 0) V3D_BAR();
    __syncthreads();                              // every fragment read of this tile is done: the whole ring is free
    if constexpr (EPI != EPI_SWIGLU) {
      if constexpr (epi_has_bias(EPI)) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) asm volatile("" : "+v"(e_bias[ni]));
      }
      if constexpr (HAS_RES) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { asm volatile("" : "+v"(e_res[0][j])); asm volatile("" : "+v"(e_res[1][j])); }
      }
    }
    V3D_TL(tl3);
#if defined(V3D_PP_PROBE) && V3D_PP_PROBE == 3          // probe: every tile stores to tile (0, 0)'s output (C traffic stays in L2)
    const int cm0 = 0, cn0 = 0, ctn = 0;
#else
    const int cm0 = m0, cn0 = n0, ctn = tn;
#endif
    const bool seg_tail = k_begin > 0, seg_head = k_end < nt;        // split tail: this segment lacks the tile's first / last K-steps
    k_begin = n_kb; k_end = n_ke;
    if (has_next) {                               // next segment's first K-step -> buffer 0, in flight under the epilogue
      setup(n_tm, n_tn);
      stage(IntC<0>{}, k_begin); stage(IntC<1>{}, k_begin); stage(IntC<2>{}, k_begin); stage(IntC<3>{}, k_begin);
    }
    // The exchange uses no fences (an agent-scope release / acquire is a whole-L2 write-back / invalidate per wave on this chip and
    // cost more than the idle round it removes): the image and the flag travel with device-scope instructions instead - stores
    // written through (sc0 sc1), s_waitcnt vmcnt(0) = acknowledged, barrier, flag; the reader polls the flag with a device-scope
    // load and reads the image with loads that bypass the caches a stale copy could sit in.
    V3D_TL(tl4);
    if constexpr (SKT) {
    if (seg_tail) {
      // a chunk c > 0: hand the accumulators to the tile's chunk-0 workgroup (f32, lane-major: the reader has the same layout)
      const float* img = p.sk_ws + (size_t)blockIdx.x * (32 * 512 * 4);
      unsigned off = (unsigned)tid * 16u;
      asm volatile("" : "+v"(off));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) {
          // (s_nop: a VALU write to the data registers of a > 64-bit store needs 2 wait states after it; hipcc pads that hazard for
          //  its own stores but cannot see into an asm statement - without it the next address add corrupted the image)
          asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1\n\ts_nop 1" : : "v"(off), "v"(acc[i][j]), "s"(img) : "memory");
          off += 512u * 16u;
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(p.sk_flags + blockIdx.x, p.sk_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }                                              // (no early exit: a chunk c > 0 skips the epilogue below by the same flag -
    if (seg_head && !seg_tail) {                   //  with break / continue here hipcc spilled accumulators in every tile;
                                                   //  a MIDDLE chunk has both flags set and is a tail: it must never wait)
      for (int c = 1; c < p.sk_split; ++c) {
        const int fg = c * sk_T + sk_g;                                  // chunk c of this tile (this workgroup is chunk 0: sk_g = j)
        const unsigned from = (unsigned)((fg % wx) * 8 + fg / wx);
        if (tid == 0)
          while (__hip_atomic_load(p.sk_flags + from, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.sk_epoch) __builtin_amdgcn_s_sleep(8);
        __syncthreads();
        // The image comes in through LDS (the ring is free: this is the workgroup's last segment): LDS-DMA needs no registers, and
        // the adds then take one 16-byte LDS read at a time - with the image loaded into registers (even four vectors at a time)
        // hipcc spilled 40-90 accumulator registers in every tile of this instantiation.  Two halves of sixteen vectors per lane
        // (16 KiB per wave and half: the whole 128 KiB ring), device-scope loads (sc0 sc1: bypass what a stale copy could sit in).
        const char* img = (const char*)(p.sk_ws + (size_t)from * (32 * 512 * 4)) + (size_t)tid * 16;
        char* const lw = smem + wave * 16384;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            const int c = hf * 16 + u;
            if (c < 4 * MT)
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(img + (size_t)c * (512 * 16)),
                                               (__attribute__((address_space(3))) void*)(lw + u * 1024), 16, 0, 17);
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            const int c = hf * 16 + u;
            if (c < 4 * MT) {
              const f32x4 part = *reinterpret_cast<const f32x4*>(lw + u * 1024 + lane * 16);
              f32x4& a = acc[c / MT][c % MT];
              a[0] += part[0]; a[1] += part[1]; a[2] += part[2]; a[3] += part[3];
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the reads are done before the next half's DMA lands on them
        }
      }
      __syncthreads();       // the epilogue's wave-private regions lie inside other waves' bounce regions: everyone is done with those
    }
    }   // SKT

    // Epilogue (r02b): each wave turns its own 128|96 x 64 part of the C tile around in a PRIVATE 6 KiB of LDS, 32 rows at a time -
    // no workgroup barrier, and whole 128-byte row segments (16 bytes per lane, 8 rows per instruction) go out.  The two earlier forms
    // both cost 5-12 us per tile (tools/probes/gemm_pp_timeline.hip): the workgroup-wide LDS staging in barriers with the matrix cores
    // idle, 8-byte stores straight from the accumulators (a lane owns 4 columns) in 4096 32-byte write requests per tile.  The private
    // regions sit in K-step buffer 1 above its first half-tile slot: buffer 0 receives the next tile's first K-step meanwhile, and the
    // first DMA of the next K loop (H4 -> that first slot) may start while slower waves are still here.  Same arithmetic and rounding
    // points as the other forms (product + bias rounded to 16 bits, activation / residual on the rounded value, rounded again).
#ifdef V3D_PP_PROBE   // tools/probes/gemm_pp_probe.hip only (WRONG RESULTS): 1 = K loop alone, the accumulators folded into one store
    if (V3D_PP_PROBE == 1) {
      float sacc = 0.f;
      for (int i = 0; i < 4; ++i) for (int j = 0; j < MT; ++j) sacc += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
      if (sacc == 12345.678f) out[tid] = from_f32<T>(sacc);
    } else
#endif
    if (!(SKT && seg_tail)) {
      int ln = lane;
      asm volatile("" : "+v"(ln));             // (as in setup: keep the epilogue's lane constants out of the K loop's registers)
      const int l15 = ln & 15, l4 = ln >> 4;
      char* const reg = smem + PP_BUF + PP_HALF + wave * 6144;
      if constexpr (EPI == EPI_SWIGLU) {
        // silu(gate) * up in registers (the wave holds both, see setup), 32 output columns per wave: rows of 64 bytes, pitch 80
        constexpr int CP = 80;
        const int rrow = ln >> 2, rch = ln & 3;                 // read-back: 16 rows x 4 chunks per instruction
        T* const obase = out + ctn * 128 + (wn >> 1) * 64 + (wn & 1) * 32 + rch * 8;
#pragma unroll
        for (int q = 0; q < MT / 2; ++q) {
#pragma unroll
          for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2) {
              const int mi = 2 * q + m2;
              float v[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = round_to<T>(silu(round_to<T>(acc[n2][mi][r]))) * round_to<T>(acc[2 + n2][mi][r]);
              uint2 pk;
              pk.x = pack2<T>(v[0], v[1]); pk.y = pack2<T>(v[2], v[3]);
              *reinterpret_cast<uint2*>(reg + (m2 * 16 + l15) * CP + (16 * n2 + 4 * l4) * 2) = pk;
            }
          uint4 cq[2];                                          // all read-backs first (one LDS round trip), then the predicated stores
#pragma unroll
          for (int j = 0; j < 2; ++j) cq[j] = *reinterpret_cast<const uint4*>(reg + (rrow + 16 * j) * CP + rch * 16);
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int gm = cm0 + wm * WROWS + 32 * q + rrow + 16 * j;
            if (gm < p.M) *reinterpret_cast<uint4*>(obase + (int64_t)gm * p.ldo) = cq[j];
          }
        }
      } else {
        constexpr int CP = 144;                                 // 64 columns = 128 bytes per row + 16
        const int rrow = ln >> 3, rch = ln & 7;                 // read-back: 8 rows x 8 chunks (whole 128-byte lines) per instruction
        const int colr = cn0 + wn * 64 + rch * 8;
        auto res_load_c = [&](int q, v4i (&dst)[4]) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int gm = cm0 + wm * WROWS + 32 * q + rrow + 8 * j;
            const int gc = gm < p.M ? gm : p.M - 1;
            const int64_t rm = p.res_mod > 0 ? (gc % p.res_mod) : gc;
            dst[j] = *reinterpret_cast<const v4i*>((const T*)p.res + rm * p.ldr + colr);
          }
        };
#pragma unroll
        for (int q = 0; q < MT / 2; ++q) {
          uint4 rr[4];
          if constexpr (HAS_RES) {
            // quarters 0, 1 were requested after the K loop; 2 (, 3) are requested when 1 has been consumed (the next tile's DMAs,
            // issued before these loads, have landed or nearly so by then)
            if (q == 2) { res_load_c(2, e_res[0]); if (MT / 2 > 3) res_load_c(3, e_res[1]); }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              rr[j] = make_uint4((unsigned)e_res[q & 1][j][0], (unsigned)e_res[q & 1][j][1], (unsigned)e_res[q & 1][j][2], (unsigned)e_res[q & 1][j][3]);
            }
          }
#pragma unroll
          for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
              const int mi = 2 * q + m2;
              float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;      // (widened where used: sixteen live floats push accumulators to scratch)
              if (epi_has_bias(EPI)) {
                const unsigned wlo = (unsigned)e_bias[ni][0], whi = (unsigned)e_bias[ni][1];
                b0 = pair_lo<T>(wlo); b1 = pair_hi<T>(wlo); b2 = pair_lo<T>(whi); b3 = pair_hi<T>(whi);
              }
              uint2 pk;
              pk.x = pack2<T>(acc[ni][mi][0] + b0, acc[ni][mi][1] + b1);
              pk.y = pack2<T>(acc[ni][mi][2] + b2, acc[ni][mi][3] + b3);
              *reinterpret_cast<uint2*>(reg + (m2 * 16 + l15) * CP + (16 * ni + 4 * l4) * 2) = pk;
            }
          uint4 cq[4];                                          // all read-backs first (one LDS round trip), then the predicated stores
#pragma unroll
          for (int j = 0; j < 4; ++j) cq[j] = *reinterpret_cast<const uint4*>(reg + (rrow + 8 * j) * CP + rch * 16);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int row = rrow + 8 * j;
            uint4 c = cq[j];
            const int gm = cm0 + wm * WROWS + 32 * q + row;
            if constexpr (EPI != EPI_NONE && EPI != EPI_BIAS) {
              float v[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float x = vec_get<T>(c, e);
                v[e] = EPI == EPI_BIAS_GELU_ERF ? gelu_erf(x) : EPI == EPI_BIAS_GELU_TANH ? gelu_tanh(x) : EPI == EPI_BIAS_RELU ? fmaxf(x, 0.f)
                                                                                                                                : x + (HAS_RES ? vec_get<T>(rr[j], e) : 0.f);
              }
              c = vec_pack<T>(v);
            }
#if defined(V3D_PP_TIMELINE) && defined(V3D_PP_NOSTORE)    // probe: the epilogue without its global stores
            if (gm < p.M && c.x == 0x12345678u) *reinterpret_cast<uint4*>(out + (int64_t)gm * p.ldo + colr) = c;
#else
            if (gm < p.M) *reinterpret_cast<uint4*>(out + (int64_t)gm * p.ldo + colr) = c;
#endif
          }
        }
      }
    }
#ifdef V3D_PP_PROBE
    if (V3D_PP_PROBE == 2) { if (!has_next) break; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); continue; }
#endif
    V3D_TL(tl5);
#ifdef V3D_PP_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    { V3D_TL(tl6);
      V3D_TL_ACC(0, tl0, tl1); V3D_TL_ACC(1, tl1, tl2); V3D_TL_ACC(2, tl2, tl3); V3D_TL_ACC(3, tl3, tl4); V3D_TL_ACC(4, tl4, tl5); V3D_TL_ACC(5, tl5, tl6);
      tl_acc[6] += 1; }
    if (!has_next) {
      if (lane == 0 && blockIdx.x < 256)
        for (int i = 0; i < 7; ++i) g_pp_tl[((int)blockIdx.x * 8 + wave) * 8 + i] = tl_acc[i];
    }
#endif
    if (!has_next) break;
    // the counted waits of the K loop assume that only this wave's staging DMAs are outstanding, in issue order: retire the
    // epilogue's stores / residual loads (and with them the prefetch, long since landed) before the next tile starts counting
#if !(defined(V3D_PP_PROBE) && V3D_PP_PROBE == 4)        // probe 4 (timing only, races): do not wait for the epilogue's stores
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  }
#undef V3D_DSR
#undef V3D_RDA
#undef V3D_RDW4
#undef V3D_LG0_A
#undef V3D_LG0_W
#undef V3D_PPMMA
#undef V3D_VM
#undef V3D_BAR
}

// ------------------------------------------------------------------------------------------
// Skinny GEMM for decode (M <= 8): out[m, n] = A[m,:] . W[n,:] (+bias)(+res).  Pure weight
// streaming: one wave per output column group, 16-byte loads, f32 accumulate, shuffle reduce.
// ------------------------------------------------------------------------------------------
template <typename T, int MAXM>
__global__ __launch_bounds__(256) void gemv_kernel(const T* __restrict__ A, const T* __restrict__ W,
                                                   const T* __restrict__ bias, const T* __restrict__ res,
                                                   T* __restrict__ out, int M, int N, int K, int64_t lda, int64_t ldw,
                                                   int64_t ldr, int64_t ldo, int epi, int res_mod) {
  const int lane = threadIdx.x & 63;
  const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int n_waves = (gridDim.x * blockDim.x) >> 6;
  const int kv = K / 8;
  for (int n = wave_global; n < N; n += n_waves) {
    float s[MAXM];
#pragma unroll
    for (int m = 0; m < MAXM; ++m) s[m] = 0.f;
    const uint4* wrow = reinterpret_cast<const uint4*>(W + (int64_t)n * ldw);
    for (int k = lane; k < kv; k += 64) {
      const uint4 w = wrow[k];
#pragma unroll
      for (int m = 0; m < MAXM; ++m) {
        if (m < M) {
          const uint4 a = reinterpret_cast<const uint4*>(A + (int64_t)m * lda)[k];
#pragma unroll
          for (int j = 0; j < 8; ++j) s[m] = fmaf(vec_get<T>(w, j), vec_get<T>(a, j), s[m]);
        }
      }
    }
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s[m] += __shfl_xor(s[m], off);
    }
    if (lane == 0) {
      for (int m = 0; m < M; ++m) {
        float x = s[m];
        if (bias) x += to_f32(bias[n]);
        x = round_to<T>(x);
        if (epi == EPI_BIAS_GELU_ERF) x = gelu_erf(x);
        if (epi == EPI_BIAS_GELU_TANH) x = gelu_tanh(x);
        if (epi == EPI_BIAS_RELU) x = fmaxf(x, 0.f);
        if (epi == EPI_BIAS_GELU_ERF || epi == EPI_BIAS_GELU_TANH) x = round_to<T>(x);
        if (res) x = x + to_f32(res[(int64_t)(res_mod > 0 ? m % res_mod : m) * ldr + n]);
        out[(int64_t)m * ldo + n] = from_f32<T>(x);
      }
    }
  }
}

// decode SwiGLU: out[m, j] = silu(A.Wg_j) * (A.Wu_j) with the tile-interleaved [gate64 | up64] row order
template <typename T, int MAXM>
__global__ __launch_bounds__(256) void gemv_swiglu_kernel(const T* __restrict__ A, const T* __restrict__ W,
                                                          T* __restrict__ out, int M, int N, int K, int64_t lda,
                                                          int64_t ldw, int64_t ldo) {
  const int lane = threadIdx.x & 63;
  const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int n_waves = (gridDim.x * blockDim.x) >> 6;
  const int kv = K / 8;
  const int half = N / 2;
  for (int j = wave_global; j < half; j += n_waves) {
    const int ng = (j >> 6) * 128 + (j & 63), nu = ng + 64;
    float sg[MAXM], su[MAXM];
#pragma unroll
    for (int m = 0; m < MAXM; ++m) sg[m] = su[m] = 0.f;
    const uint4* wg = reinterpret_cast<const uint4*>(W + (int64_t)ng * ldw);
    const uint4* wu = reinterpret_cast<const uint4*>(W + (int64_t)nu * ldw);
    for (int k = lane; k < kv; k += 64) {
      const uint4 g = wg[k], u = wu[k];
#pragma unroll
      for (int m = 0; m < MAXM; ++m) {
        if (m < M) {
          const uint4 a = reinterpret_cast<const uint4*>(A + (int64_t)m * lda)[k];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float av = vec_get<T>(a, e);
            sg[m] = fmaf(vec_get<T>(g, e), av, sg[m]);
            su[m] = fmaf(vec_get<T>(u, e), av, su[m]);
          }
        }
      }
    }
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) { sg[m] += __shfl_xor(sg[m], off); su[m] += __shfl_xor(su[m], off); }
    }
    if (lane == 0)
      for (int m = 0; m < M; ++m)
        out[(int64_t)m * ldo + j] = from_f32<T>(round_to<T>(silu(round_to<T>(sg[m]))) * round_to<T>(su[m]));
  }
}

static int gemm_variant() {   // 0 = auto, 1 = force 128x128, 3 = force 256x256 (where N allows), 4 = force 192x256; read per call (A/B runs)
  const char* e = getenv("V3D_GEMM_VARIANT");
  return e ? atoi(e) : 0;
}

static int gemm_dma_mode() {   // -1 = auto, 0 / 1 = force (A/B runs)
  static int v = -2;
  if (v == -2) { const char* e = getenv("V3D_GEMM_DMA"); v = e ? atoi(e) : -1; }
  return v;
}

template <typename T, int MT>
static int launch_gemm256x256(GemmArgs p, int epi, hipStream_t st) {
  constexpr int BM = MT * 32;
  // (V3D_GEMM_DMA=1 moves the restage behind the last phase's MFMAs; with the alternating staging it no longer pays)
  p.dma_late = gemm_dma_mode() > 0 ? 1 : 0;
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = p.N / B3N;
#define V3D_GEMM3_CASE(E)                                                                                 \
  case E: {                                                                                               \
    auto k = gemm256x256_kernel<T, E, MT>;                                                                \
    static bool attr_done = false;                                                                        \
    if (!attr_done) {                                                                                     \
      hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM3_LDS_BYTES); \
      if (e != hipSuccess) { set_error("v3d_gemm: LDS attribute: %s", hipGetErrorString(e)); return V3D_E_LAUNCH; } \
      attr_done = true;                                                                                   \
    }                                                                                                     \
    hipLaunchKernelGGL(k, dim3(p.tiles_m * p.tiles_n), dim3(512), GEMM3_LDS_BYTES, st, p);                \
  } break;
  switch (epi) {
    V3D_GEMM3_CASE(EPI_NONE)
    V3D_GEMM3_CASE(EPI_BIAS)
    V3D_GEMM3_CASE(EPI_BIAS_GELU_ERF)
    V3D_GEMM3_CASE(EPI_BIAS_GELU_TANH)
    V3D_GEMM3_CASE(EPI_BIAS_RES)
    V3D_GEMM3_CASE(EPI_RES)
    V3D_GEMM3_CASE(EPI_SWIGLU)
    V3D_GEMM3_CASE(EPI_BIAS_RELU)
    default: set_error("v3d_gemm: unknown epilogue %d", epi); return V3D_E_INVALID;
  }
#undef V3D_GEMM3_CASE
  return check_launch("v3d_gemm (256-wide)");
}

static int gemm_pp_mode() {   // 1 (default) = ping-pong v4 for the 256 x 256 tile, 0 = the v3 kernel; read per call so that a
  const char* e = getenv("V3D_GEMM_PP");     // test can A/B the two in one process
  return e ? atoi(e) : 1;
}

static int pp_slots() {       // workgroups the chip holds at once (one per CU: 132 KiB of LDS each); V3D_GEMM_PP_GRID overrides (tests)
  const char* e = getenv("V3D_GEMM_PP_GRID");
  if (e && atoi(e) > 0) return atoi(e);
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

// ---- split-K tail of the ping-pong kernel: host side ----
// V3D_GEMM_STREAMK: 1 (default) = where the time model says it pays; 0 = never (every output then sums k in one run, whatever M); 2 = wherever legal.
static int gemm_sk_mode() {
  const char* e = getenv("V3D_GEMM_STREAMK");
  return e ? atoi(e) : 1;
}
// Legal: grid a multiple of 8 (the XCD-major order), an even number of K-steps (granule = 2) with at least two granules per chunk,
// and the tiles left after the whole-tile rounds fit the grid at least twice.  Returns the whole-tile rounds (or -1) and the split.
static int sk_plan(int n_tiles, int grid, int nt, int* split) {
  *split = 1;
  if (grid < 8 || (grid & 7) || (nt & 1)) return -1;
  const int dp = n_tiles / grid, rest = n_tiles - dp * grid;      // (dp = 0: fewer tiles than workgroups - the whole launch is the tail)
  if (rest == 0) return -1;                  // whole rounds only
  int sp = grid / rest;
  sp = sp > 4 ? 4 : sp;
  if (sp < 2 || (nt >> 1) < 2 * sp) return -1;
  *split = sp;
  return dp;
}
static bool sk_pays(int n_tiles, int grid, int nt) {           // the time model of launch_gemm: whole rounds vs whole rounds + split tail
  int sp = 1;
  const int dp = sk_plan(n_tiles, grid, nt, &sp);
  if (dp < 0) return false;
  const double ks = (double)nt, round = 8.0 + 1.263 * ks;
  return dp * round + (dp > 0 ? 5.0 : 0.0) + 22.0 + (dp > 0 ? 1.7 : 1.3) * ks / sp + 40.0 < (double)((n_tiles + grid - 1) / grid) * round;
}
struct SkWorkspace { float* ws = nullptr; unsigned* flags = nullptr; unsigned epoch = 0; int slots = 0; };
// One workspace per stream (launches on one stream are ordered; two streams must not share accumulator images).  Allocated on
// first use, never during stream capture (the caller then gets the plain walk).
static SkWorkspace* sk_workspace(hipStream_t st, int grid, unsigned* epoch) {      // *epoch: this launch's flag value (taken under the lock)
  static std::mutex mu;
  static std::unordered_map<hipStream_t, SkWorkspace> map;
  std::lock_guard<std::mutex> lock(mu);
  SkWorkspace& w = map[st];
  if (w.slots >= grid) { *epoch = ++w.epoch; return &w; }
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return nullptr; }
  if (w.ws) { (void)hipStreamSynchronize(st); (void)hipFree(w.ws); w = SkWorkspace{}; }
  const size_t img = (size_t)32 * 512 * 16, bytes = (size_t)grid * img + (size_t)grid * sizeof(unsigned);
  void* ptr = nullptr;
  if (hipMalloc(&ptr, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  w.ws = (float*)ptr;
  w.flags = (unsigned*)((char*)ptr + (size_t)grid * img);
  if (hipMemsetAsync(w.flags, 0, (size_t)grid * sizeof(unsigned), st) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(ptr); w = SkWorkspace{}; return nullptr; }
  w.slots = grid; w.epoch = 0;
  *epoch = ++w.epoch;
  return &w;
}

template <typename T, int MT>
static int launch_gemm256pp(GemmArgs p, int epi, hipStream_t st, int sk_allow = -1) {   // sk_allow: -1 decide here (forced tile), 0 / 1 = the caller's model decided
  p.tiles_m = (p.M + MT * 32 - 1) / (MT * 32);
  p.tiles_n = p.N / B3N;
  const bool part = sk_allow == -2;                                 // a launch over the tile range the caller set (no split of its own)
  if (!part) { p.tile_base = 0; p.tile_count = 0; }
  const int n_tiles = part ? p.tile_count : p.tiles_m * p.tiles_n;
  int grid = n_tiles < pp_slots() ? n_tiles : pp_slots();           // persistent: one workgroup per CU walks its tiles
  { const char* e = getenv("V3D_GEMM_SKEW"); p.skew = e ? atoi(e) : 0; }
  if (n_tiles <= grid) p.skew = 0;                                  // a single round: nothing to spread
  p.sk_dp = -1; p.sk_split = 1; p.sk_ws = nullptr; p.sk_flags = nullptr; p.sk_epoch = 0;
  {
    int split = 1;
    const int slots = pp_slots() & ~7;
    const int mode = gemm_sk_mode(), dp = sk_plan(n_tiles, slots, p.K / BK, &split);
    if (!part && MT == 8 && mode != 0 && dp >= 0 && (mode == 2 || (sk_allow < 0 ? sk_pays(n_tiles, slots, p.K / BK) : sk_allow != 0))) {
      unsigned epoch = 0;
      if (SkWorkspace* w = sk_workspace(st, slots, &epoch)) {
        // The whole-tile rounds go out as a launch of their own with the instantiation that has no exchange code (no spills), the
        // tail as a second launch: its chunks then all start together and stay in lock step (tail K-steps 1.3 us instead of 1.7).
        if (dp > 0) {
          GemmArgs pw = p;
          pw.tile_base = 0; pw.tile_count = dp * slots;
          const int rc = launch_gemm256pp<T, MT>(pw, epi, st, -2);
          if (rc != V3D_OK) return rc;
          p.tile_base = dp * slots; p.tile_count = n_tiles - dp * slots;
        }
        p.sk_dp = 0; p.sk_split = split; p.sk_ws = w->ws; p.sk_flags = w->flags; p.sk_epoch = epoch;
        p.skew = 0;
        grid = slots;                                               // (also when n_tiles < slots: the chunks fill the chip)
      }
    }
  }
#define V3D_GEMM4_LAUNCH(E, SKT)                                                                          \
  {                                                                                                       \
    auto k = gemm256pp_kernel<T, E, MT, SKT>;                                                             \
    static bool attr_done = false;                                                                        \
    if (!attr_done) {                                                                                     \
      hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES); \
      if (e != hipSuccess) { set_error("v3d_gemm: LDS attribute: %s", hipGetErrorString(e)); return V3D_E_LAUNCH; } \
      attr_done = true;                                                                                   \
    }                                                                                                     \
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), PP_LDS_BYTES, st, p);                                    \
  }
  // in the tail launch a chunk-0 workgroup waits (unbounded) for flags its partner chunks raise.  That is deadlock-free
  // only while every workgroup of the launch becomes resident - true for ONE tail launch on the chip (grid <= one workgroup per CU),
  // not for two that interleave their dispatch (two streams): waiting chunk-0 workgroups of both could fill an XCD whose
  // writers then never start.  So tail launches of this process are CHAINED: each waits for the previous one's completion event,
  // whatever stream it was on.  (Two PROCESSES sharing one card - the gloo rehearsals - run with V3D_GEMM_STREAMK=0.)
  struct TailChain { std::mutex mu; hipEvent_t done = nullptr; hipStream_t last = nullptr; bool recorded = false; };
  static TailChain chain;
  const bool is_tail = p.sk_dp >= 0;
  std::unique_lock<std::mutex> tail_lock(chain.mu, std::defer_lock);
  if (is_tail) {
    tail_lock.lock();
    if (!chain.done && hipEventCreateWithFlags(&chain.done, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); chain.done = nullptr; }
    if (chain.done && chain.recorded && chain.last != st) (void)hipStreamWaitEvent(st, chain.done, 0);
  }
#define V3D_GEMM4_CASE(E)                                                                                 \
  case E: {                                                                                               \
    if constexpr (MT == 8) { if (p.sk_dp >= 0) V3D_GEMM4_LAUNCH(E, true) else V3D_GEMM4_LAUNCH(E, false) } \
    else V3D_GEMM4_LAUNCH(E, false)                                                                        \
  } break;
  switch (epi) {
    V3D_GEMM4_CASE(EPI_NONE)
    V3D_GEMM4_CASE(EPI_BIAS)
    V3D_GEMM4_CASE(EPI_BIAS_GELU_ERF)
    V3D_GEMM4_CASE(EPI_BIAS_GELU_TANH)
    V3D_GEMM4_CASE(EPI_BIAS_RES)
    V3D_GEMM4_CASE(EPI_RES)
    V3D_GEMM4_CASE(EPI_SWIGLU)
    V3D_GEMM4_CASE(EPI_BIAS_RELU)
    default: set_error("v3d_gemm: unknown epilogue %d", epi); return V3D_E_INVALID;
  }
#undef V3D_GEMM4_CASE
#undef V3D_GEMM4_LAUNCH
  if (is_tail && chain.done) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone && hipEventRecord(chain.done, st) == hipSuccess) {
      chain.recorded = true; chain.last = st;
    } else {
      (void)hipGetLastError();
    }
  }
  return check_launch("v3d_gemm (256-wide, ping-pong)");
}

// Tile choice (speed only), from a time model fitted on MI355X to the path's shapes (tools/time_gemm_ab.py; microseconds):
//   a kernel takes  rounds x (fixed + K-steps x per-step),  rounds = ceil(tiles / workgroup slots of the chip)
//   256 x 256 ping-pong (256 slots): fixed 8 (epilogue, rendezvous, first DMA latency; 13 with SwiGLU), 1.26 per K-step
//   192 x 256 ping-pong (256 slots): fixed 4.5, 1.22 per K-step   (3/4 of the tile at the same step time: it only wins where
//                                    256-row tiles quantise badly against the 256 CUs, e.g. M = 6794, N = 3584, K = 3584)
//   128 x 128           (512 slots): fixed 1, 0.98 per K-step     (many small tiles: small M, or N % 256 != 0)
//   256 x 256 with the split-K tail: whole rounds as above (a launch of their own) + 5 for the second launch + one round of
//                                    (22 + step x K-steps / split + 40 for the exchange),
//                                    step 1.3 when the launch is only the tail, 1.7 behind whole rounds (measured: 6794 x 3584 x 18944
//                                    780 -> 685 us, 960 x 3584 x 18944 330 -> 152 us; no gain at K = 3584)
struct GemmPlan { int kernel; int sk; int dp; int split; int tiles; };    // kernel: 1 = 128 x 128, 2 = 256 x 256, 3 = 192 x 256 (2, 3: ping-pong or v3)
static GemmPlan gemm_plan(int M, int N, int K, int slots, int var, bool pp_on, int sk_mode) {
  GemmPlan g = {1, 0, -1, 1, ((M + BM - 1) / BM) * (N / BN)};
  if (N % B3N != 0 || var == 1) return g;
  const double ks = (double)(K / BK);
  const int tiles256 = ((M + 255) / 256) * (N / B3N), tiles192 = ((M + 191) / 192) * (N / B3N);
  const bool pp = pp_on && K >= 2 * BK;
  double t256 = (double)((tiles256 + 255) / 256) * (pp ? 8.0 + 1.263 * ks : 9.0 + 1.36 * ks);
  const double t192 = (double)((tiles192 + 255) / 256) * (pp ? 4.5 + 1.2225 * ks : 2.0 + 1.32 * ks);
  const int r1 = (g.tiles + 511 - 25) / 512;                     // (a last round under 5 % full is not felt)
  const double t1 = (double)(r1 > 0 ? r1 : 1) * (1.0 + 0.98 * ks);
  int sk_use = 0, sp = 1, dp = -1;
  if (pp && sk_mode != 0) {
    dp = sk_plan(tiles256, slots & ~7, K / BK, &sp);
    if (dp >= 0) {
      const double tsk = dp * (8.0 + 1.263 * ks) + (dp > 0 ? 5.0 : 0.0) + 22.0 + (dp > 0 ? 1.7 : 1.3) * ks / sp + 40.0;
      const double best = t256 < t192 ? (t256 < t1 ? t256 : t1) : (t192 < t1 ? t192 : t1);
      if (tsk < 0.98 * best || sk_mode == 2) { t256 = tsk < t256 ? tsk : t256; sk_use = 1; }    // (a 2 % margin: the model is coarse)
    }
  }
  if (var == 4) { g.kernel = 3; g.tiles = tiles192; g.sk = -1; return g; }       // forced tiles decide the split themselves (sk = -1)
  if (var == 3) { g.kernel = 2; g.tiles = tiles256; g.sk = -1; return g; }
  if (t192 < t256 && t192 < t1) { g.kernel = 3; g.tiles = tiles192; return g; }
  if (t256 < t1) { g.kernel = 2; g.tiles = tiles256; g.sk = sk_use; if (sk_use) { g.dp = dp; g.split = sp; } return g; }
  return g;
}

template <typename T>
static int launch_gemm(const GemmArgs& p, int epi, hipStream_t st) {
  const bool pp_on = gemm_pp_mode() != 0;
  const GemmPlan g = gemm_plan(p.M, p.N, p.K, pp_slots(), gemm_variant(), pp_on, gemm_sk_mode());
  const bool pp = pp_on && p.K >= 2 * BK;
  if (g.kernel == 3) return pp ? launch_gemm256pp<T, 6>(p, epi, st, g.sk < 0 ? -1 : 0) : launch_gemm256x256<T, 6>(p, epi, st);
  if (g.kernel == 2) return pp ? launch_gemm256pp<T, 8>(p, epi, st, g.sk) : launch_gemm256x256<T, 8>(p, epi, st);
#define V3D_GEMM_CASE(E)                                                                                  \
  case E: {                                                                                               \
    auto k = gemm_kernel<T, E>;                                                                           \
    static bool attr_done = false;                                                                        \
    if (!attr_done) {                                                                                     \
      hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES); \
      if (e != hipSuccess) { set_error("v3d_gemm: LDS attribute: %s", hipGetErrorString(e)); return V3D_E_LAUNCH; } \
      attr_done = true;                                                                                   \
    }                                                                                                     \
    hipLaunchKernelGGL(k, dim3(p.tiles_m * p.tiles_n), dim3(256), GEMM_LDS_BYTES, st, p);                 \
  } break;
  switch (epi) {
    V3D_GEMM_CASE(EPI_NONE)
    V3D_GEMM_CASE(EPI_BIAS)
    V3D_GEMM_CASE(EPI_BIAS_GELU_ERF)
    V3D_GEMM_CASE(EPI_BIAS_GELU_TANH)
    V3D_GEMM_CASE(EPI_BIAS_RES)
    V3D_GEMM_CASE(EPI_RES)
    V3D_GEMM_CASE(EPI_SWIGLU)
    V3D_GEMM_CASE(EPI_BIAS_RELU)
    default: set_error("v3d_gemm: unknown epilogue %d", epi); return V3D_E_INVALID;
  }
#undef V3D_GEMM_CASE
  return check_launch("v3d_gemm");
}

template <typename T>
static int launch_gemv(const GemmArgs& p, int epi, hipStream_t st) {
  const int waves_needed = epi == EPI_SWIGLU ? p.N / 2 : p.N;
  int blocks = (waves_needed + 3) / 4;
  if (blocks > 256 * 8) blocks = 256 * 8;
  if (epi == EPI_SWIGLU) {
    hipLaunchKernelGGL((gemv_swiglu_kernel<T, 8>), dim3(blocks), dim3(256), 0, st, (const T*)p.A, (const T*)p.W,
                       (T*)p.out, p.M, p.N, p.K, p.lda, p.ldw, p.ldo);
  } else {
    const bool has_bias = epi_has_bias(epi);
    const bool has_res = epi == EPI_BIAS_RES || epi == EPI_RES;
    hipLaunchKernelGGL((gemv_kernel<T, 8>), dim3(blocks), dim3(256), 0, st, (const T*)p.A, (const T*)p.W,
                       has_bias ? (const T*)p.bias : nullptr, has_res ? (const T*)p.res : nullptr, (T*)p.out, p.M, p.N,
                       p.K, p.lda, p.ldw, p.ldr, p.ldo, epi, p.res_mod);
  }
  return check_launch("v3d_gemm (skinny)");
}

}  // namespace v3d

using namespace v3d;

extern "C" int v3d_gemm_plan_host(int M, int N, int K, int slots, int* kernel, int* tiles, int* dp, int* split) {
  V3D_REQUIRE(M > 0 && N > 0 && K > 0 && N % BN == 0 && K % BK == 0 && slots > 0, "v3d_gemm_plan_host: bad shape");
  V3D_REQUIRE(kernel && tiles && dp && split, "v3d_gemm_plan_host: null pointer");
  if (M <= 8) { *kernel = 0; *tiles = 0; *dp = -1; *split = 1; return V3D_OK; }
  const GemmPlan g = gemm_plan(M, N, K, slots, 0, true, 1);
  *kernel = g.kernel; *tiles = g.tiles; *dp = g.sk > 0 ? g.dp : -1; *split = g.sk > 0 ? g.split : 1;
  return V3D_OK;
}

extern "C" int v3d_gemm(const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias, const void* res,
                        int64_t ldr, int res_mod, void* out, int64_t ldo, int M, int N, int K, int dtype, int epilogue,
                        void* stream) {
  V3D_REQUIRE(A && W && out, "v3d_gemm: null pointer");
  V3D_REQUIRE(dtype == V3D_F16 || dtype == V3D_BF16, "v3d_gemm: dtype must be f16 or bf16");
  V3D_REQUIRE(M > 0 && N > 0 && K > 0, "v3d_gemm: bad shape M=%d N=%d K=%d", M, N, K);
  V3D_REQUIRE(N % BN == 0 && K % BK == 0, "v3d_gemm: N=%d must be a multiple of %d and K=%d of %d (pad the weights)", N, BN, K, BK);
  V3D_REQUIRE(lda >= K && ldw >= K && lda % 8 == 0 && ldw % 8 == 0 && ldo % 8 == 0, "v3d_gemm: leading dimensions must be >= K and multiples of 8");
  V3D_REQUIRE(aligned16(A) && aligned16(W) && aligned16(out), "v3d_gemm: pointers must be 16-byte aligned");
  const bool need_bias = epi_has_bias(epilogue);
  const bool need_res = epilogue == EPI_BIAS_RES || epilogue == EPI_RES;
  V3D_REQUIRE(!need_bias || bias, "v3d_gemm: epilogue %d needs a bias", epilogue);
  V3D_REQUIRE(!need_res || (res && aligned16(res) && ldr % 8 == 0), "v3d_gemm: epilogue %d needs an aligned residual", epilogue);
  V3D_REQUIRE(ldo >= (epilogue == EPI_SWIGLU ? N / 2 : N), "v3d_gemm: ldo too small");
  V3D_REQUIRE((int64_t)M * lda < (1ll << 31) && (int64_t)N * ldw < (1ll << 31), "v3d_gemm: operand larger than 2^31 elements (4 GiB)");
  GemmArgs p;
  p.A = A; p.W = W; p.bias = need_bias ? bias : nullptr; p.res = need_res ? res : nullptr; p.out = out;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldr = ldr; p.ldo = ldo; p.res_mod = res_mod;
  p.tiles_m = (M + BM - 1) / BM; p.tiles_n = N / BN; p.skew = 0; p.dma_late = 0; p.tile_base = 0; p.tile_count = 0;
  hipStream_t st = (hipStream_t)stream;
  if (M <= 8) return dtype == V3D_BF16 ? launch_gemv<bf16_t>(p, epilogue, st) : launch_gemv<f16_t>(p, epilogue, st);
  return dtype == V3D_BF16 ? launch_gemm<bf16_t>(p, epilogue, st) : launch_gemm<f16_t>(p, epilogue, st);
}
