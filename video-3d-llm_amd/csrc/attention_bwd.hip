// Backward of the causal GQA flash attention (training step, BASELINE configs[4]; reference: Qwen2Attention /
// Qwen2FlashAttention2, llava/model/language_model/qwen2/modeling_qwen2.py:248-482, differentiated).  Head dim 128 (narrower heads zero-padded), Sq = Sk = S,
// causal (the decoder) or not (the SigLIP encoder, siglip_encoder.py:197-250), a batch of sequences.  The S x S matrices never leave the chip: the probabilities are RECOMPUTED from q, k and the forward's row
// log-sum-exp (v3d_attention_train writes it), tile by tile, with the forward kernel's own primitives - 64-row LDS images staged by
// LDS-DMA under one XOR swizzle, row fragments by ds_read_b128, transposed fragments by ds_read_b64_tr_b16, 32x32x16 MFMAs whose
// accumulator registers are re-used as the next product's B operand (both operands in the same permuted k order).
//
//   delta[q]   = sum_d dO[q][d] O[q][d]                                                     attn_delta_kernel
//   dQ kernel  (workgroup = 128 queries of one head, lane = query; loop over key tiles up to the diagonal)
//       S'^T = K (cQ)^T - L      P^T = exp2(S'^T)      dP^T = V dO^T      dS^T = P^T (dP^T - delta)      dQ^T += K^T dS^T
//   dK/dV kernel (workgroup = 128 keys of one QUERY head, lane = key; loop over query tiles from the diagonal on)
//       S = Q K^T      P = exp2(c S - L)      dP = dO V^T      dS = P (dP - delta)      dV^T += dO^T P      dK^T += Q^T dS
//       -> f32 partials per query head; attn_bwd_reduce_kernel sums the heads of a kv group in a fixed order (no atomics).
// Seven products instead of the minimal five (S and dP are formed in both kernels): that buys a dQ without atomics and a
// deterministic result.  The images are double-buffered (the next tile's LDS-DMA flies under the current tile's MFMAs) and the
// fragments are read through two register rings by asm statements with counted waits (hipcc puts a vmcnt(0) in front of LDS reads it
// can see, which would drain the prefetch): eight reads fly under the eight MFMAs of the previous ring.  One workgroup per CU (the accumulators of both gradients live in AGPRs); no finer software pipeline yet.  The softmax is taken a quarter
// (16 rows of the tile) at a time, each quarter's MFMAs issued before the next quarter's exponentials (measured: no faster than all
// four quarters first - with one wave per SIMD the kernel is bound by exposed LDS / issue latency, 5.5-6 k cycles per tile against
// 2 k of MFMA; two waves per SIMD need the accumulators of one gradient only, i.e. a third recomputation of S).
#include "v3d_common.h"

namespace v3d {
namespace bwd {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using v4i = __attribute__((ext_vector_type(4))) int;
using v2i = __attribute__((ext_vector_type(2))) int;

template <typename T> struct Mfma32;
template <> struct Mfma32<bf16_t> {
  static __device__ __forceinline__ f32x16 run(v4i a, v4i b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mfma32<f16_t> {
  static __device__ __forceinline__ f32x16 run(v4i a, v4i b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};

constexpr int BW_ROW = 256;                 // LDS row bytes (128 x 16 bit)
constexpr int BW_TILE = 64 * BW_ROW;        // one 64-row image
constexpr int BW_BUF = 2 * BW_TILE;         // one stage: two images
constexpr int BW_STAT = 2 * BW_BUF;         // 2 x (64 L + 64 delta) floats: row statistics of the dK/dV kernel's query tiles
constexpr int BW_LDS = BW_STAT + 1024;      // 65 KiB (the dQ kernel's output transpose, 34 KiB, re-uses the stages)

struct BwdArgs {
  const void* q; const void* k; const void* v; const void* o; const void* dout;
  const float* lse; float* delta;            // [Hq, S]
  void* dq; float* dk_part; float* dv_part;  // dq [S, .] 16 bit; partials [Hq, S, 128] f32
  int64_t ldq, ldk, ldv, ldo, lddo, lddq;    // token strides (elements); heads are 128 apart
  int64_t bsq, bsk, bsv, bso, bsdo, bsdq;    // batch strides (elements); lse / delta / the partials are [B, Hq, S(, 128)]
  int S, Hq, group, B, causal;
  float scale, scale_log2;
};

using f32x4_ = __attribute__((ext_vector_type(4))) float;
// Row statistics of one softmax quarter (accumulator registers 8 (I & 1) .. + 8 of row block I >> 1): x[g][j] = stat[32 (I >> 1) + 16 (I & 1) +
// 8 g + 4 h + j], for the L and the delta array (256 bytes apart), one asm statement with its wait.
template <int I> __device__ __forceinline__ void stat_quarter(unsigned stat_lds, int lane, f32x4_ (&L)[2], f32x4_ (&D)[2]) {
  const unsigned b = stat_lds + 16u * (unsigned)(lane >> 5);
  constexpr int o = (32 * (I >> 1) + 16 * (I & 1)) * 4;
  asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&v"(L[0]), "=&v"(L[1]), "=&v"(D[0]), "=&v"(D[1])
               : "v"(b), "n"(o), "n"(o + 32), "n"(o + 256), "n"(o + 256 + 32)
               : "memory");
}

template <int N> struct IntC { static constexpr int value = N; };

__device__ __forceinline__ int swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Stage rows [row_first, row_first + 64) of a [S, ld] 16-bit matrix (head columns at src) into an image: wave w moves rows
// [16w, 16w + 16) in four DMA pieces of 4 rows x 256 B; row r lands with its 16-byte chunks XORed by swz(r) (rows past the end
// re-read the last row: the callers mask them).
__device__ __forceinline__ void stage_image(const uint16_t* src, unsigned ld_bytes, char* image, int row_first, int S, int wave, int lane) {
  const int srow = lane >> 4;
  const int chunk0 = (lane & 15) ^ (srow << 2);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int row = row_first + wave * 16 + 4 * i + srow;
    row = row < S ? row : S - 1;
    glds16((const char*)src + ((unsigned)row * ld_bytes + (unsigned)((chunk0 ^ i) * 16)), image + (wave * 16 + 4 * i) * BW_ROW);
  }
}

// A-operand fragments f[ks] of image row (32 half + (lane & 31)), d = 16 ks + 8 (lane >> 5) .. + 8, ks = 0..7: the k-step toggles address
// bits 5..7 (chunk (2 ks + h) ^ swz = 2 ks ^ (h ^ swz)).  ISSUE only: the data is complete after frags_wait<N>(f) with N = the LDS
// reads issued after these eight (LDS returns in order).  The reads are asm statements because hipcc puts a vmcnt(0) in front of LDS
// reads it can see (which would drain the prefetch) and serialises read -> wait -> MFMA; the wait ties the eight registers ("+v") so
// that nothing reads them before it (the pattern of attention.hip's K / V fragment rings).
__device__ __forceinline__ void row_frags_issue(unsigned image_lds, int lane, int half, v4i (&f)[8]) {
  const int r = lane & 31, h = lane >> 5;
  const unsigned b = image_lds + (32 * half + r) * BW_ROW + ((h ^ swz(r)) << 4);
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) asm volatile("ds_read_b128 %0, %1" : "=v"(f[ks]) : "v"(b ^ ((unsigned)ks << 5)));
}
template <int N> __device__ __forceinline__ void frags_wait(v4i (&f)[8]) {
  asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void frags_wait(v2i (&f)[8]) {
  asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "n"(N) : "memory");
}

// Row statistics of the accumulator registers' rows: x[4 qh + g][j] = stat[32 qh + 8 g + 4 h + j] (register r = 4 g + j of row block qh)
using f32x4 = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ void stat_frags(unsigned stat_lds, int lane, f32x4 (&x)[8]) {
  const unsigned b = stat_lds + 16u * (unsigned)(lane >> 5);
  asm volatile(
      "ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:32\n\tds_read_b128 %2, %8 offset:64\n\tds_read_b128 %3, %8 offset:96\n\t"
      "ds_read_b128 %4, %8 offset:128\n\tds_read_b128 %5, %8 offset:160\n\tds_read_b128 %6, %8 offset:192\n\tds_read_b128 %7, %8 offset:224\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7])
      : "v"(b)
      : "memory");
}

// A-operand fragments of the image's TRANSPOSE for the 32 columns d = 32 dt ..: f[2 s4 + {0,1}] cover the 16 rows of block s4 in the
// k order the accumulator registers have ((j & 3) + 8 (j >> 2) + 4 h), so that {f[2 s4], f[2 s4 + 1]} multiplies a B operand packed
// straight from accumulator registers 8 (s4 & 1) .. + 8 of row block s4 >> 1.   (address form of attention.hip's V^T reads; ISSUE only)
__device__ __forceinline__ void tr_frags_issue(unsigned image_lds, int lane, int dt, v2i (&f)[8]) {
  const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3, h = lane >> 5;
  const int chunk_lo = 2 * (g & 1) + (pp >> 1), byte = 8 * (pp & 1);
  const int row0 = 4 * h + qq;
  const unsigned a0 = (image_lds + row0 * BW_ROW + byte + ((chunk_lo ^ swz(row0)) << 4)) ^ (dt << 6);
  const unsigned a1 = (image_lds + (row0 + 8) * BW_ROW + byte + ((chunk_lo ^ swz(row0 + 8)) << 4)) ^ (dt << 6);
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f[0]) : "v"(a0));
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f[1]) : "v"(a1));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(f[2]) : "v"(a0));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(f[3]) : "v"(a1));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8192" : "=v"(f[4]) : "v"(a0));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8192" : "=v"(f[5]) : "v"(a1));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:12288" : "=v"(f[6]) : "v"(a0));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:12288" : "=v"(f[7]) : "v"(a1));
}

// The same fragments by ROW BLOCK: f[2 dt + {0,1}] = the 16 rows of block S4 for the four d tiles dt - what the four products of one
// softmax quarter need (the quarter's MFMAs then run while the next quarter's exponentials are computed).
template <int S4> __device__ __forceinline__ void tr_block_issue(unsigned image_lds, int lane, v2i (&f)[8]) {
  const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3, h = lane >> 5;
  const int chunk_lo = 2 * (g & 1) + (pp >> 1), byte = 8 * (pp & 1);
  const int row0 = 4 * h + qq;
  const unsigned a0 = image_lds + row0 * BW_ROW + byte + ((chunk_lo ^ swz(row0)) << 4);
  const unsigned a1 = image_lds + (row0 + 8) * BW_ROW + byte + ((chunk_lo ^ swz(row0 + 8)) << 4);
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[2 * dt]) : "v"(a0 ^ ((unsigned)dt << 6)), "n"(S4 * 4096));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[2 * dt + 1]) : "v"(a1 ^ ((unsigned)dt << 6)), "n"(S4 * 4096));
  }
}

template <typename T> __device__ __forceinline__ v4i pack8(const float* e) {
  v4i r;
  r[0] = (int)pack2<T>(e[0], e[1]); r[1] = (int)pack2<T>(e[2], e[3]); r[2] = (int)pack2<T>(e[4], e[5]); r[3] = (int)pack2<T>(e[6], e[7]);
  return r;
}

// delta[head, t] = sum_d dO[t][head][d] * O[t][head][d]; a wave per (token, head)
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_kernel(BwdArgs p) {
  const int lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= (int64_t)p.B * p.S * p.Hq) return;
  const int head = (int)(item % p.Hq);
  const int64_t bt = item / p.Hq;
  const int t = (int)(bt % p.S), bi = (int)(bt / p.S);
  const uint32_t a = *reinterpret_cast<const uint32_t*>((const uint16_t*)p.o + bi * p.bso + (int64_t)t * p.ldo + head * 128 + 2 * lane);
  const uint32_t b = *reinterpret_cast<const uint32_t*>((const uint16_t*)p.dout + bi * p.bsdo + (int64_t)t * p.lddo + head * 128 + 2 * lane);
  float s = pair_lo<T>(a) * pair_lo<T>(b) + pair_hi<T>(a) * pair_hi<T>(b);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) p.delta[((int64_t)bi * p.Hq + head) * p.S + t] = s;
}

template <typename T>
__global__ __launch_bounds__(256, 1) void attn_bwd_dq_kernel(BwdArgs p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  using M = Mfma32<T>;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ql = lane & 31, h = lane >> 5;
  const int qt = (int)gridDim.y - 1 - (int)blockIdx.y;            // heaviest query tiles first
  const int head = blockIdx.x, hk = head / p.group, bi = blockIdx.z;
  const int q0 = qt * 128;
  const uint16_t* Q = (const uint16_t*)p.q + bi * p.bsq + (int64_t)head * 128;
  const uint16_t* K = (const uint16_t*)p.k + bi * p.bsk + (int64_t)hk * 128;
  const uint16_t* V = (const uint16_t*)p.v + bi * p.bsv + (int64_t)hk * 128;
  const uint16_t* DO = (const uint16_t*)p.dout + bi * p.bsdo + (int64_t)head * 128;
  const int qi = q0 + wave * 32 + ql;
  const int qi_ld = qi < p.S ? qi : p.S - 1;
  v4i qf[8], dof[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    const uint4 raw = *reinterpret_cast<const uint4*>(Q + (int64_t)qi_ld * p.ldq + ks * 16 + h * 8);
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = vec_get<T>(raw, j) * p.scale_log2;          // as the forward kernel: c Q rounded to 16 bit
    qf[ks] = pack8<T>(f);
    dof[ks] = *reinterpret_cast<const v4i*>(DO + (int64_t)qi_ld * p.lddo + ks * 16 + h * 8);
  }
  const int64_t stat0 = ((int64_t)bi * p.Hq + head) * p.S;
  const float L = p.lse[stat0 + qi_ld], dl = p.delta[stat0 + qi_ld];
  const int last_q = q0 + 127 < p.S ? q0 + 127 : p.S - 1;
  const int n_tiles = p.causal ? last_q / 64 + 1 : (p.S + 63) / 64;
  const int n_wave = p.causal ? min((q0 + wave * 32 + 31) / 64 + 1, n_tiles) : n_tiles;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  stage_image(K, (unsigned)p.ldk * 2u, smem, 0, p.S, wave, lane);
  stage_image(V, (unsigned)p.ldv * 2u, smem + BW_TILE, 0, p.S, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < n_tiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < n_tiles) {            // the next tile flies under this tile's MFMAs (its stage was last read two barriers ago)
      stage_image(K, (unsigned)p.ldk * 2u, smem + (buf ^ 1) * BW_BUF, (t + 1) * 64, p.S, wave, lane);
      stage_image(V, (unsigned)p.ldv * 2u, smem + (buf ^ 1) * BW_BUF + BW_TILE, (t + 1) * 64, p.S, wave, lane);
    }
    const unsigned lds_k = lds0 + buf * BW_BUF, lds_v = lds_k + BW_TILE;
    {   // (no branch on t < n_wave: tiles past a wave's diagonal are masked to zeros; see the dK/dV kernel)
      f32x16 s[2], dp[2];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[kt][r] = -L; dp[kt][r] = 0.f; }
      v4i fa[8];                                          // one fragment ring (see the dK/dV kernel)
      row_frags_issue(lds_k, lane, 0, fa);
      frags_wait<0>(fa);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) s[0] = M::run(fa[ks], qf[ks], s[0]);
      row_frags_issue(lds_v, lane, 0, fa);
      frags_wait<0>(fa);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) dp[0] = M::run(fa[ks], dof[ks], dp[0]);
      row_frags_issue(lds_k, lane, 1, fa);
      frags_wait<0>(fa);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) s[1] = M::run(fa[ks], qf[ks], s[1]);
      row_frags_issue(lds_v, lane, 1, fa);
      v2i f0[8], f1[8];
      tr_block_issue<0>(lds_k, lane, f0);                 // K^T fragments of the first 16 keys, all four d tiles: in flight under dp
      frags_wait<8>(fa);                                  // (lgkmcnt counts to 15: never more than two rings in flight)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) dp[1] = M::run(fa[ks], dof[ks], dp[1]);
      const int limit = ((p.causal && qi < p.S - 1) ? qi : p.S - 1) - t * 64 - 4 * h;       // visible iff tile-local key offset <= limit
      // a quarter (16 keys) at a time: its exponentials, then its four MFMAs - which run while the next quarter's exponentials are computed
      auto quarter = [&](auto i_c, const v2i (&f)[8]) {
        constexpr int i = decltype(i_c)::value;
        float e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int r = 8 * (i & 1) + j;
          const bool vis = ((i >> 1) * 32 + (r & 3) + 8 * (r >> 2)) <= limit;
          const float pr = vis ? __builtin_amdgcn_exp2f(s[i >> 1][r]) : 0.f;
          e[j] = pr * (dp[i >> 1][r] - dl);
        }
        return pack8<T>(e);
      };
      auto mma_block = [&](const v2i (&f)[8], const v4i& b) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const v4i a = {f[2 * dt][0], f[2 * dt][1], f[2 * dt + 1][0], f[2 * dt + 1][1]};
          acc[dt] = M::run(a, b, acc[dt]);
        }
      };
      v4i pf = quarter(IntC<0>{}, f0);
      tr_block_issue<1>(lds_k, lane, f1);
      frags_wait<8>(f0); mma_block(f0, pf);
      pf = quarter(IntC<1>{}, f1);
      tr_block_issue<2>(lds_k, lane, f0);
      frags_wait<8>(f1); mma_block(f1, pf);
      pf = quarter(IntC<2>{}, f0);
      tr_block_issue<3>(lds_k, lane, f1);
      frags_wait<8>(f0); mma_block(f0, pf);
      pf = quarter(IntC<3>{}, f1);
      frags_wait<0>(f1); mma_block(f1, pf);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                  // the next tile has landed and is visible; everyone is done with this one
  }
  // dQ = scale * acc, transposed through LDS, whole rows out
  constexpr int OROW = 128 * 2 + 16;
  static_assert(4 * 32 * OROW <= BW_LDS, "output transpose must fit the allocation");
  char* so = smem + wave * 32 * OROW;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int d = 32 * dt + 8 * r4 + 4 * h;
      uint2 pk;
      pk.x = pack2<T>(acc[dt][4 * r4 + 0] * p.scale, acc[dt][4 * r4 + 1] * p.scale);
      pk.y = pack2<T>(acc[dt][4 * r4 + 2] * p.scale, acc[dt][4 * r4 + 3] * p.scale);
      *reinterpret_cast<uint2*>(so + ql * OROW + d * 2) = pk;
    }
  __syncthreads();
  uint16_t* DQ = (uint16_t*)p.dq + bi * p.bsdq + (int64_t)head * 128;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = i * 64 + lane, row = idx >> 4, ch = idx & 15;
    const int q = q0 + wave * 32 + row;
    if (q < p.S) *reinterpret_cast<uint4*>(DQ + (int64_t)q * p.lddq + ch * 8) = *reinterpret_cast<const uint4*>(so + row * OROW + ch * 16);
  }
}

template <typename T>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv_kernel(BwdArgs p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  using M = Mfma32<T>;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kl = lane & 31, h = lane >> 5;
  const int head = blockIdx.x, hk = head / p.group, bi = blockIdx.z;
  const int k0 = (int)blockIdx.y * 128;
  const uint16_t* Q = (const uint16_t*)p.q + bi * p.bsq + (int64_t)head * 128;
  const uint16_t* K = (const uint16_t*)p.k + bi * p.bsk + (int64_t)hk * 128;
  const uint16_t* V = (const uint16_t*)p.v + bi * p.bsv + (int64_t)hk * 128;
  const uint16_t* DO = (const uint16_t*)p.dout + bi * p.bsdo + (int64_t)head * 128;
  const int64_t stat0 = ((int64_t)bi * p.Hq + head) * p.S;
  const int key = k0 + wave * 32 + kl;
  const int key_ld = key < p.S ? key : p.S - 1;
  v4i kf[8], vf[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    kf[ks] = *reinterpret_cast<const v4i*>(K + (int64_t)key_ld * p.ldk + ks * 16 + h * 8);
    vf[ks] = *reinterpret_cast<const v4i*>(V + (int64_t)key_ld * p.ldv + ks * 16 + h * 8);
  }
  float* stat = reinterpret_cast<float*>(smem + BW_STAT);           // per stage: [0, 64) L, [64, 128) delta of the tile's queries
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  f32x16 dk[4], dv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[i][r] = 0.f; dv[i][r] = 0.f; }
  const int n_qt = (p.S + 63) / 64;
  const int wave_first_key = k0 + wave * 32;
  const int t0 = p.causal ? k0 / 64 : 0;
  auto stat_load = [&](int t) -> float {                            // threads 0..127: L or delta of one query of tile t
    const int qrow = t * 64 + (tid & 63);
    const float* src = tid < 64 ? p.lse : p.delta;
    return (tid < 128 && qrow < p.S) ? src[stat0 + qrow] : 0.f;
  };
  {
    const float sv = stat_load(t0);
    stage_image(Q, (unsigned)p.ldq * 2u, smem + (t0 & 1) * BW_BUF, t0 * 64, p.S, wave, lane);
    stage_image(DO, (unsigned)p.lddo * 2u, smem + (t0 & 1) * BW_BUF + BW_TILE, t0 * 64, p.S, wave, lane);
    if (tid < 128) stat[(t0 & 1) * 128 + tid] = sv;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  for (int t = t0; t < n_qt; ++t) {
    const int buf = t & 1;
    float sv = 0.f;
    const bool more = t + 1 < n_qt;
    if (more) {
      sv = stat_load(t + 1);                                        // (older than the DMAs below: its wait does not cover them)
      stage_image(Q, (unsigned)p.ldq * 2u, smem + (buf ^ 1) * BW_BUF, (t + 1) * 64, p.S, wave, lane);
      stage_image(DO, (unsigned)p.lddo * 2u, smem + (buf ^ 1) * BW_BUF + BW_TILE, (t + 1) * 64, p.S, wave, lane);
    }
    const unsigned lds_q = lds0 + buf * BW_BUF, lds_do = lds_q + BW_TILE;
    {   // every wave computes every tile: a tile none of whose queries sees this wave's keys (the first one, for the upper half of the
        // workgroup) is masked to zeros - a branch around the MFMAs makes hipcc copy all 128 accumulator registers AGPR <-> VGPR per tile
      f32x16 s[2], dp[2];
#pragma unroll
      for (int qh = 0; qh < 2; ++qh)
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[qh][r] = 0.f; dp[qh][r] = 0.f; }
      v4i fa[8];                                          // ONE fragment ring: the next eight reads are issued right behind the eight MFMAs
      row_frags_issue(lds_q, lane, 0, fa);               // that consume the ring (they return while the matrix pipe is still busy)
      frags_wait<0>(fa);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) s[0] = M::run(fa[ks], kf[ks], s[0]);
      row_frags_issue(lds_do, lane, 0, fa);
      frags_wait<0>(fa);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) dp[0] = M::run(fa[ks], vf[ks], dp[0]);
      row_frags_issue(lds_q, lane, 1, fa);
      frags_wait<0>(fa);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) s[1] = M::run(fa[ks], kf[ks], s[1]);
      row_frags_issue(lds_do, lane, 1, fa);
      frags_wait<0>(fa);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) dp[1] = M::run(fa[ks], vf[ks], dp[1]);
      const unsigned stat_lds = lds0 + BW_STAT + buf * 512;
      v2i f0[8], f1[8];
      tr_block_issue<0>(lds_do, lane, f0);               // dO^T and Q^T fragments of the first 16 queries, all four d tiles
      tr_block_issue<0>(lds_q, lane, f1);
      // a quarter (16 queries) at a time: its probabilities, then its eight MFMAs - which run while the next quarter is computed
      auto quarter = [&](auto i_c, v4i& bp, v4i& bs) {
        constexpr int i = decltype(i_c)::value;
        f32x4_ Lq[2], Dq[2];
        stat_quarter<i>(stat_lds, lane, Lq, Dq);
        float ep[8], es[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int r = 8 * (i & 1) + j;
          const int row = (i >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;        // query within the tile
          const int qidx = t * 64 + row;
          const bool vis = (!p.causal || key <= qidx) && qidx < p.S;
          const float pr = vis ? __builtin_amdgcn_exp2f(fmaf(s[i >> 1][r], p.scale_log2, -Lq[j >> 2][j & 3])) : 0.f;
          ep[j] = pr;
          es[j] = pr * (dp[i >> 1][r] - Dq[j >> 2][j & 3]);
        }
        bp = pack8<T>(ep);
        bs = pack8<T>(es);
      };
      auto mma_block = [&](const v2i (&f)[8], const v4i& b, f32x16 (&acc)[4]) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const v4i a = {f[2 * dt][0], f[2 * dt][1], f[2 * dt + 1][0], f[2 * dt + 1][1]};
          acc[dt] = M::run(a, b, acc[dt]);
        }
      };
      v4i bp, bs;
      quarter(IntC<0>{}, bp, bs);
      frags_wait<8>(f0); mma_block(f0, bp, dv);           // dV^T += dO^T P
      tr_block_issue<1>(lds_do, lane, f0);
      frags_wait<8>(f1); mma_block(f1, bs, dk);           // dK^T += Q^T dS
      tr_block_issue<1>(lds_q, lane, f1);
      quarter(IntC<1>{}, bp, bs);
      frags_wait<8>(f0); mma_block(f0, bp, dv);
      tr_block_issue<2>(lds_do, lane, f0);
      frags_wait<8>(f1); mma_block(f1, bs, dk);
      tr_block_issue<2>(lds_q, lane, f1);
      quarter(IntC<2>{}, bp, bs);
      frags_wait<8>(f0); mma_block(f0, bp, dv);
      tr_block_issue<3>(lds_do, lane, f0);
      frags_wait<8>(f1); mma_block(f1, bs, dk);
      tr_block_issue<3>(lds_q, lane, f1);
      quarter(IntC<3>{}, bp, bs);
      frags_wait<8>(f0); mma_block(f0, bp, dv);
      frags_wait<0>(f1); mma_block(f1, bs, dk);
    }
    if (more && tid < 128) stat[(buf ^ 1) * 128 + tid] = sv;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (key < p.S) {
    float* ok = p.dk_part + (stat0 + key) * 128;
    float* ov = p.dv_part + (stat0 + key) * 128;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int d = 32 * dt + 8 * r4 + 4 * h;
        *reinterpret_cast<float4*>(ok + d) = make_float4(dk[dt][4 * r4] * p.scale, dk[dt][4 * r4 + 1] * p.scale, dk[dt][4 * r4 + 2] * p.scale, dk[dt][4 * r4 + 3] * p.scale);
        *reinterpret_cast<float4*>(ov + d) = make_float4(dv[dt][4 * r4], dv[dt][4 * r4 + 1], dv[dt][4 * r4 + 2], dv[dt][4 * r4 + 3]);
      }
  }
}

// dk[b][t][g, d] = T(sum over the query heads of kv group g, in order, of dk_part[b][head][t][d]); likewise dv
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_reduce_kernel(const float* __restrict__ dk_part, const float* __restrict__ dv_part, T* __restrict__ dk,
                                                              int64_t lddk, int64_t bsdk, T* __restrict__ dv, int64_t lddv, int64_t bsdv, int B, int S,
                                                              int Hkv, int group) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;       // (b, t, g, d / 8)
  const int64_t total = (int64_t)B * S * Hkv * 16;
  if (idx >= total) return;
  const int ch = (int)(idx & 15);
  const int g = (int)((idx >> 4) % Hkv);
  const int64_t bt = (idx >> 4) / Hkv;
  const int t = (int)(bt % S), bi = (int)(bt / S);
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, b[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < group; ++j) {
    const int64_t off = ((((int64_t)bi * Hkv + g) * group + j) * S + t) * 128 + ch * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] += dk_part[off + i]; b[i] += dv_part[off + i]; }
  }
  *reinterpret_cast<uint4*>(dk + bi * bsdk + (int64_t)t * lddk + g * 128 + ch * 8) = vec_pack<T>(a);
  *reinterpret_cast<uint4*>(dv + bi * bsdv + (int64_t)t * lddv + g * 128 + ch * 8) = vec_pack<T>(b);
}

}  // namespace bwd
}  // namespace v3d

using namespace v3d;
using namespace v3d::bwd;

extern "C" int64_t v3d_attention_backward_workspace_bytes(int B, int S, int Hq) {
  if (B <= 0 || S <= 0 || Hq <= 0) return 0;
  return ((int64_t)B * Hq * S + 2 * (int64_t)B * Hq * S * 128) * (int64_t)sizeof(float);
}

extern "C" int v3d_attention_backward(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                                      void* dq, void* dk, void* dv, int dtype, int B, int S, int Hq, int Hkv, int64_t ldq, int64_t ldk,
                                      int64_t ldv, int64_t ldo, int64_t lddo, int64_t lddq, int64_t lddk, int64_t lddv, int64_t bsq,
                                      int64_t bsk, int64_t bsv, int64_t bso, int64_t bsdo, int64_t bsdq, int64_t bsdk, int64_t bsdv,
                                      int causal, float scale, void* workspace, int64_t workspace_bytes, void* stream) {
  V3D_REQUIRE(q && k && v && o && dout && lse && dq && dk && dv && workspace, "v3d_attention_backward: null pointer");
  V3D_REQUIRE(dtype == V3D_F16 || dtype == V3D_BF16, "v3d_attention_backward: dtype must be f16 or bf16");
  V3D_REQUIRE(B > 0 && B <= 65535 && S > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0, "v3d_attention_backward: bad shape");
  V3D_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0 && lddo % 8 == 0 && lddq % 8 == 0 && lddk % 8 == 0 && lddv % 8 == 0 &&
                  bsq % 8 == 0 && bsk % 8 == 0 && bsv % 8 == 0 && bso % 8 == 0 && bsdo % 8 == 0 && bsdq % 8 == 0 && bsdk % 8 == 0 && bsdv % 8 == 0,
              "v3d_attention_backward: strides must be multiples of 8 elements");
  V3D_REQUIRE(ldq >= (int64_t)Hq * 128 && ldo >= (int64_t)Hq * 128 && lddo >= (int64_t)Hq * 128 && lddq >= (int64_t)Hq * 128 &&
                  ldk >= (int64_t)Hkv * 128 && ldv >= (int64_t)Hkv * 128 && lddk >= (int64_t)Hkv * 128 && lddv >= (int64_t)Hkv * 128,
              "v3d_attention_backward: heads are 128 columns apart inside a token row");
  V3D_REQUIRE(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(o) && aligned16(dout) && aligned16(dq) && aligned16(dk) && aligned16(dv) &&
                  aligned16(workspace), "v3d_attention_backward: pointers must be 16-byte aligned");
  V3D_REQUIRE(workspace_bytes >= v3d_attention_backward_workspace_bytes(B, S, Hq), "v3d_attention_backward: workspace too small");
  V3D_REQUIRE((int64_t)S * ldq < (1ll << 31) && (int64_t)S * ldk < (1ll << 31) && (int64_t)S * ldv < (1ll << 31) && (int64_t)S * lddo < (1ll << 31),
              "v3d_attention_backward: row offsets inside one sequence must fit 32 bits");
  BwdArgs p;
  p.q = q; p.k = k; p.v = v; p.o = o; p.dout = dout; p.lse = lse;
  p.delta = (float*)workspace;
  p.dk_part = p.delta + (int64_t)B * Hq * S;
  p.dv_part = p.dk_part + (int64_t)B * Hq * S * 128;
  p.dq = dq;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.lddo = lddo; p.lddq = lddq;
  p.bsq = bsq; p.bsk = bsk; p.bsv = bsv; p.bso = bso; p.bsdo = bsdo; p.bsdq = bsdq;
  p.S = S; p.Hq = Hq; p.group = Hq / Hkv; p.B = B; p.causal = causal ? 1 : 0;
  p.scale = scale; p.scale_log2 = scale * 1.44269504088896340736f;
  hipStream_t st = (hipStream_t)stream;
  const unsigned n128 = (unsigned)((S + 127) / 128);
#define V3D_BWD(TT)                                                                                                              \
  {                                                                                                                              \
    static bool attr_done = false;                                                                                               \
    if (!attr_done) {                                                                                                            \
      hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<TT>, hipFuncAttributeMaxDynamicSharedMemorySize, BW_LDS); \
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<TT>, hipFuncAttributeMaxDynamicSharedMemorySize, BW_LDS); \
      if (e != hipSuccess) { set_error("v3d_attention_backward: LDS attribute: %s", hipGetErrorString(e)); return V3D_E_LAUNCH; } \
      attr_done = true;                                                                                                          \
    }                                                                                                                            \
    hipLaunchKernelGGL(attn_delta_kernel<TT>, dim3((unsigned)(((int64_t)B * S * Hq + 3) / 4)), dim3(256), 0, st, p);              \
    hipLaunchKernelGGL(attn_bwd_dq_kernel<TT>, dim3(Hq, n128, B), dim3(256), BW_LDS, st, p);                                      \
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<TT>, dim3(Hq, n128, B), dim3(256), BW_LDS, st, p);                                     \
    hipLaunchKernelGGL(attn_bwd_reduce_kernel<TT>, dim3((unsigned)(((int64_t)B * S * Hkv * 16 + 255) / 256)), dim3(256), 0, st,   \
                       (const float*)p.dk_part, (const float*)p.dv_part, (TT*)dk, lddk, bsdk, (TT*)dv, lddv, bsdv, B, S, Hkv, p.group); \
  }
  if (dtype == V3D_BF16) V3D_BWD(bf16_t) else V3D_BWD(f16_t)
#undef V3D_BWD
  return check_launch("v3d_attention_backward");
}
