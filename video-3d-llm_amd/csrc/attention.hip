// Flash attention for prefill (K16 causal GQA, hd 128) and for the ViT (K11, non-causal,
// hd 72 zero-padded to 96 by the QKV weight layout).  S x S scores never leave the chip.
//
// Workgroup = 4 waves = 128 queries of one (batch, head); each wave owns 32 queries.
// KV tile = 64 keys, staged HBM -> LDS by LDS-DMA into a 2-deep ring (K and V both row-major,
// 256-byte rows, one XOR swizzle that serves K's row reads and V's transposed reads).
//
//   S^T[key][q]  = K . Q^T      v_mfma_f32_32x32x16: A = K rows (ds_read_b128), B = Q (registers)
//                               -> a lane owns ONE query column and 16 keys per 32-key tile:
//                               softmax is in-lane plus one exchange with lane^32.
//   O^T[d][q]   += V^T . P^T    A = V^T via ds_read_b64_tr_b16 (hardware transpose of the row-major
//                               V image), B = P^T straight from the S^T accumulator registers
//                               (converted to 16 bit; k order of both operands permuted alike)
//                               -> O^T keeps the query on the lane: the online-softmax rescale
//                               is a per-lane multiply.
// The O tile is transposed once through LDS on the way out.
#include <stdlib.h>

#include "v3d_common.h"

namespace v3d {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using i16x4 = __attribute__((ext_vector_type(4))) short;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using v4i = __attribute__((ext_vector_type(4))) int;
using v2i = __attribute__((ext_vector_type(2))) int;

template <typename T> struct Mfma32;
template <> struct Mfma32<bf16_t> {
  using frag = bf16x8;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mfma32<f16_t> {
  using frag = f16x8;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

using f32x4 = __attribute__((ext_vector_type(4))) float;
template <typename T> struct Mfma16;
template <> struct Mfma16<bf16_t> {
  using frag = bf16x8;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Mfma16<f16_t> {
  using frag = f16x8;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

union Frag16 {   // 16 bytes viewed as MFMA fragment / raw words
  uint4 u;
  bf16x8 b;
  f16x8 h;
  i16x4 s[2];
  v4i i4;
};
template <typename T> __device__ __forceinline__ typename Mfma32<T>::frag as_frag(const Frag16& f);
template <> __device__ __forceinline__ bf16x8 as_frag<bf16_t>(const Frag16& f) { return f.b; }
template <> __device__ __forceinline__ f16x8 as_frag<f16_t>(const Frag16& f) { return f.h; }
template <typename T> __device__ __forceinline__ typename Mfma16<T>::frag as_frag16(const Frag16& f) { return as_frag<T>(f); }

constexpr int AT_BQ = 128, AT_BKV = 64, AT_ROW = 256;           // LDS row bytes (D padded to 128)
constexpr int AT_TILE = AT_BKV * AT_ROW;                        // 16 KiB per K or V tile
constexpr int AT_LDS = 4 * AT_TILE;                             // K,V x 2 stages = 64 KiB

struct AttnArgs {
  const void* q; const void* k; const void* v; void* o;
  int64_t ldq, ldk, ldv, ldo;        // token strides (elements)
  int64_t bsq, bsk, bso;             // batch strides (elements) for q / k,v / o
  int hsq, hsk, hso;                 // head strides (elements)
  int Sq, Sk, Hq, group;             // group = Hq / Hkv
  int d_out;                         // valid head dim written to o (<= D)
  int q_pos0;                        // causal: query i sits at key position q_pos0 + i
  float scale_log2;                  // softmax scale * log2(e)
  float* lse;                        // training only (v3d_attention_train): [B, Hq, Sq] row log-sum-exp in scaled log2 units, or null
  int xcd_p;                         // > 0: 1-D grid, XCD-aware (head, query tile) mapping with xcd_p XCDs per kv head (see attn_prefill_kernel)
  int n_qt;                          // query tiles per head (the mapped form needs it; the 3-D grid reads gridDim.y)
  // attn_prefill16_kernel<PART> only (r04, shared-prefix decode attention): the "queries" of kernel head hk are the q_rpg query heads of
  // kv head hk of each of the M decode rows (query i = row i / q_rpg, head hk q_rpg + i % q_rpg: address q + (i / q_rpg) ldq + hsq hk +
  // 128 (i % q_rpg)); blockIdx.z = key chunk c: keys [c Sk, min((c + 1) Sk, sk_total)); the result is the decode kernels' f32 partial
  // (o[128], m, l) in split slot part_split0 + c of row i / q_rpg's workspace slice (part_ws_stride floats apart), head hk q_rpg + i % q_rpg
  int q_rpg, sk_total, part_split0, part_hq;
  int64_t part_ws_stride;
  // attn_prefill_kernel<QPK> only (r04, the question rows of an answer batch over ONE copy of the scene's prefix): key tiles below
  // sh_tiles are read from k_sh / v_sh (no batch stride: the scene's cache), the rest from k / v (each question's own cache)
  const void* k_sh; const void* v_sh;
  int sh_tiles;
};

#ifndef V3D_ATTN_VPF
#define V3D_ATTN_VPF 1   // r04: first V^T fragment reads of a steady-state step issued inside the score phase (0: behind it, the r03 order; bit-identical)
#endif
#ifdef V3D_ATTN_PROF   // tools/probes/attn_prof.hip only: per-wave cycle split of the tile loop (never in the product build)
__device__ unsigned long long g_attn_prof[4 * 4096];
__device__ unsigned long long g_attn_blocks[4 * 4096];   // per workgroup: realtime start, end, shader-clock delta, HW_ID
// per-phase shader-clock stamps of the pipeline step (one asm statement each, so the wait stays with the stamp)
#define V3D_STAMP(v) unsigned long long v; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory")
#define V3D_ACC(i, a, b) prof_acc[i] += (b) - (a)
#else
#define V3D_STAMP(v)
#define V3D_ACC(i, a, b)
#endif

__device__ __forceinline__ int kv_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

__device__ __forceinline__ void glds16a(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int N> struct IntC { static constexpr int value = N; };

// Softmax bookkeeping is in the log2 domain on PRE-SCALED scores: Q is multiplied by scale*log2(e) once when it is
// loaded, and the S^T accumulators START at -m_run (the lane's running maximum), so a score leaves the MFMA chain as
// s' = q.k*c - m_run and p = exp2(s') needs no subtract.  The maximum is only RAISED when some s' exceeds
// AT_RAISE (p <= 2^AT_RAISE stays far inside f32 / 16-bit range) - after the first tiles that is rare, so the common
// tile costs one max3 chain, 32 v_exp, 32 adds and 16 packs per lane and no cross-lane traffic at all.
constexpr float AT_RAISE = 8.0f;
// r03: the raise is DETECTED from the tile's row-sum share instead of a max3 chain over the scores: every p is positive, so a lane
// whose 32 probabilities sum to <= AT_LS_LIMIT holds no p > AT_LS_LIMIT, i.e. no score above log2(AT_LS_LIMIT) < AT_RAISE (255
// rather than 256 leaves v_exp_f32's last-place error on the safe side).  Only when some lane's share exceeds the limit does the
// wave compute the maxima and take the decision exactly as before (per lane: raise iff the query's maximum > AT_RAISE) - so the
// outputs are bit-identical to the max-first form, the common tile loses 18 dependent VALU operations and, above all, the
// exponentials no longer wait for the end of the score MFMA chain plus a reduction.
constexpr float AT_LS_LIMIT = 255.0f;

// KSV: k-steps of QK^T that can be non-zero (head dims >= d_out are zeroed in Q): SigLIP's 72-wide heads on the 96-wide tile
// need 5 of the 6 (the sixth multiplies zeros)
// QPK (r04, v3d_attention_shared_prefix: the question rows of an answer batch): the workgroup's 128 query slots are the rows of ALL the
// query heads of one kv head (slot i = head i / Sq of the group, row i % Sq: 7 x 60 rows fill four tiles where one head per workgroup
// filled 60 of 128 slots seven times), and key tiles below p.sh_tiles come from the scene's one cache instead of the batch entry's copy.
// A row's arithmetic is untouched - same tiles, same order, the raise decided per lane - so its bits equal the unpacked kernel's.
template <typename T, int D, bool CAUSAL, int KSV = D / 16, bool LSE = false, bool QPK = false>
__global__ __launch_bounds__(256, 2) void attn_prefill_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];   // 1 KiB: fragment addresses are formed by XOR
  using M = Mfma32<T>;
  constexpr int KS = D / 16;       // k-steps of QK^T
  static_assert(KSV == KS || (KS == 6 && KSV == 5), "only the 96-wide tile has a short form");
  constexpr int DT = D / 32;       // 32-wide d tiles of O^T
  constexpr int CH = D / 8;        // 16-byte chunks per K/V row actually present

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ql = lane & 31, h = lane >> 5;
  // Grid = (head, query tile, batch): workgroups are dispatched x-fastest, so ALL heads' heaviest (last, causal)
  // query tiles start first and the lightest ones fill the tail (longest-processing-time order; with the query
  // tile fastest the last heads' 107-tile workgroups started late and left 40 % of the launch half empty).
  // r03, xcd_p > 0 (causal GQA prefill with 8 % Hkv == 0): 1-D grid.  Workgroups go to the XCDs round-robin by their linear id, and
  // each XCD has its own L2, so id -> (head, query tile) is chosen such that an XCD only ever works on ONE kv head (xcd_p XCDs
  // per kv head): the 7 query heads x all query tiles that share a K/V stream (3.5 MB at S = 6.8k) meet in one 4 MB L2 instead of
  // all four kv heads' K/V passing through every L2.  Within an XCD the heaviest query tiles still come first.
  int qt, head, b;
  if (p.xcd_p > 0) {
    const int n_kv = p.Hq / p.group, per_b = ((p.Hq * p.n_qt + 7) >> 3) << 3;
    b = (int)blockIdx.x / per_b;
    const int l = (int)blockIdx.x - b * per_b;
    const int xcd = l & 7, kvh = xcd % n_kv, slot = xcd / n_kv;
    const int jj = (l >> 3) * p.xcd_p + slot;
    if (jj >= p.group * p.n_qt) return;          // padding of the grid to a multiple of 8 (whole workgroup)
    head = kvh * p.group + jj % p.group;
    qt = p.n_qt - 1 - jj / p.group;
  } else {
    qt = (int)gridDim.y - 1 - (int)blockIdx.y;
    head = blockIdx.x; b = blockIdx.z;            // QPK: blockIdx.x is the kv head
  }
  const int hk = QPK ? head : head / p.group;
  const int q0 = qt * AT_BQ;

  const uint16_t* K = (const uint16_t*)p.k + b * p.bsk + (int64_t)hk * p.hsk;
  const uint16_t* V = (const uint16_t*)p.v + b * p.bsk + (int64_t)hk * p.hsk;
  const uint16_t* Ksh = QPK ? (const uint16_t*)p.k_sh + (int64_t)hk * p.hsk : K;
  const uint16_t* Vsh = QPK ? (const uint16_t*)p.v_sh + (int64_t)hk * p.hsk : V;

  // ---- Q fragments: B operand, lane (q, h) holds c * Q[q][16ks + 8h .. +8) ----
  int qi = q0 + wave * 32 + ql;
  int qi_ld = qi < p.Sq ? qi : p.Sq - 1;
  if (QPK) {                                     // slot -> (head of the group, row); slots past the last row repeat it and store nothing
    const int slots = p.group * p.Sq;
    const int vi = qi < slots ? qi : slots - 1;
    const int hh = vi / p.Sq;
    head = hk * p.group + hh;
    qi = qi < slots ? vi - hh * p.Sq : p.Sq;     // (>= Sq: "no such row")
    qi_ld = vi - hh * p.Sq;
  }
  const uint16_t* Q = (const uint16_t*)p.q + b * p.bsq + (int64_t)head * p.hsq;
  Frag16 qf[KS];          // (loaded in the prologue, behind the first tiles' DMAs)

  // ---- tile counts ----
  const int n_tiles_all = (p.Sk + AT_BKV - 1) / AT_BKV;
  int n_tiles = n_tiles_all;           // tiles the workgroup stages
  int n_wave = n_tiles_all;            // tiles THIS wave computes (a causal wave stops at its last query's tile)
  if (CAUSAL) {
    const int last_q = q0 + AT_BQ - 1 < p.Sq ? q0 + AT_BQ - 1 : p.Sq - 1;
    const int t = (p.q_pos0 + last_q) / AT_BKV + 1;
    n_tiles = t < n_tiles_all ? t : n_tiles_all;
    const int tw = (p.q_pos0 + q0 + wave * 32 + 31) / AT_BKV + 1;
    n_wave = tw < n_tiles ? tw : n_tiles;
    if (QPK) {                                   // a wave's slots wrap over heads: every wave walks to the last row's tile
      const int tq = (p.q_pos0 + p.Sq - 1) / AT_BKV + 1;
      n_tiles = n_wave = tq < n_tiles_all ? tq : n_tiles_all;
    }
  }

  // ---- KV staging by LDS-DMA: one instruction = 4 rows x 256 B; wave w stages rows [16w, 16w+16) of a tile.
  //      LDS: buffer j&1 of tile j = { K image 16 KiB, V image 16 KiB }.
  //      Row (16w + 4i + srow) of piece i has swizzle kv_swz = (srow << 2) | i, so its source chunk is chunk0 ^ i:
  //      two registers describe all four pieces.
  const int srow = lane >> 4;                 // row within the 4-row DMA piece
  const int st_row0 = wave * 16 + srow;
  const int st_chunk0 = (lane & 15) ^ (srow << 2);
  const unsigned ldk_b = (unsigned)p.ldk * 2u, ldv_b = (unsigned)p.ldv * 2u;
  auto stage = [&](const uint16_t* src, unsigned ld_b, char* dst, int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int key = t * AT_BKV + st_row0 + 4 * i;
      key = key < p.Sk ? key : p.Sk - 1;        // tail keys are masked in the scores
      int chunk = st_chunk0 ^ i;
      if (CH < 16) chunk = chunk < CH ? chunk : CH - 1;        // D=96: the 4 pad slots are never read
      glds16a((const char*)src + ((unsigned)key * ld_b + (unsigned)chunk * 16u), dst + i * 4 * AT_ROW);
    }
  };
  auto stage_k = [&](int buf, int t) { stage(QPK && t < p.sh_tiles ? Ksh : K, ldk_b, smem + buf * 2 * AT_TILE + (wave * 16) * AT_ROW, t); };
  auto stage_v = [&](int buf, int t) { stage(QPK && t < p.sh_tiles ? Vsh : V, ldv_b, smem + buf * 2 * AT_TILE + AT_TILE + (wave * 16) * AT_ROW, t); };
  // r03, steady state (every row of the tile exists): the per-lane part of the four piece addresses is constant, the tile's base
  // is wave-uniform and advances on the scalar unit - no v_mad_u64 / v_min per piece in the vector issue stream (8 pieces per
  // step cost 8 x (add, min, 64-bit mad) = ~190 issue cycles of a step's ~1000).
  unsigned koff[4], voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int chunk = st_chunk0 ^ i;
    if (CH < 16) chunk = chunk < CH ? chunk : CH - 1;
    koff[i] = (unsigned)(st_row0 + 4 * i) * ldk_b + (unsigned)chunk * 16u;
    voff[i] = (unsigned)(st_row0 + 4 * i) * ldv_b + (unsigned)chunk * 16u;
  }
  // (inline asm: with the builtin, hipcc hoists the zero-extension of the loop-invariant offsets out of the loop, then no longer
  //  matches the scalar-base + 32-bit-offset form and spends a 64-bit vector add and a register pair per piece)
  const unsigned lds_wave0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (unsigned)(wave * 16) * AT_ROW;
  auto stage_fast = [&](const uint16_t* src, unsigned ld_b, const unsigned (&off)[4], unsigned lds_dst, int t) {
    const char* base = (const char*)src + (size_t)t * (size_t)(AT_BKV * ld_b);      // wave-uniform: scalar registers
#pragma unroll
    for (int i = 0; i < 4; ++i)
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                   : : "v"(off[i]), "s"(base), "s"(lds_dst + (unsigned)(i * 4 * AT_ROW)) : "memory", "m0");
  };
  auto stage_k_fast = [&](int buf, int t) { stage_fast(QPK && t < p.sh_tiles ? Ksh : K, ldk_b, koff, lds_wave0 + buf * 2 * AT_TILE, t); };
  auto stage_v_fast = [&](int buf, int t) { stage_fast(QPK && t < p.sh_tiles ? Vsh : V, ldv_b, voff, lds_wave0 + buf * 2 * AT_TILE + AT_TILE, t); };

  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  // ---- per-lane LDS read addresses (buffer 0; buffer 1 and the key sub-blocks are immediate offsets) ----
  // K row read: key row (32kt + ql), logical chunk 2ks + h
  const int k_sw = kv_swz(ql);                 // kv_swz(32kt + ql) == kv_swz(ql)
  // V transposed read: lane = 16g + 4qq + pp supplies row (base + qq), 4 columns at d = 32dt + 16(g&1) + 4pp
  const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
  const int v_chunk_lo = 2 * (g & 1) + (pp >> 1);          // + 4dt
  const int v_byte = 8 * (pp & 1);
  const int v_row0 = 4 * h + qq;               // rows 16*s4 + 4h + qq (+8): the swizzle does not depend on s4
  const int v_sw0 = kv_swz(v_row0), v_sw1 = kv_swz(v_row0 + 8);
  // chunk (2ks + h) ^ k_sw = ((h ^ k_sw) ^ 2ks): the k-step toggles address bits 5..7, the d tile bits 6..7 of V's
  const unsigned kaddr0_ = lds_base + ql * AT_ROW + ((h ^ k_sw) << 4);
  const unsigned vaddr0_ = lds_base + AT_TILE + v_row0 * AT_ROW + v_byte + ((v_chunk_lo ^ v_sw0) << 4);
  const unsigned vaddr1_ = lds_base + AT_TILE + (v_row0 + 8) * AT_ROW + v_byte + ((v_chunk_lo ^ v_sw1) << 4);

  f32x16 o[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = 0.f, l_run = 0.f;              // m_run in scaled log2 units; fixed by the first tile
  const int q_pos = p.q_pos0 + (QPK ? qi_ld : qi);
  const int wave_first_pos = QPK ? p.q_pos0 : p.q_pos0 + q0 + wave * 32;      // (QPK: some slot of the wave may be a head's first row)
#ifdef V3D_ATTN_PROF
  unsigned long long prof_acc[6] = {0, 0, 0, 0, 0, 0};
#endif

#define V3D_KR(dst, ks, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(kaddr0 ^ ((ks) << 5)), "i"(imm))
#define V3D_KW(cnt, f) asm volatile("s_waitcnt lgkmcnt(" #cnt ")" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : : "memory")
#define V3D_KM(f, i, kt, ks) s[kt] = M::run(__builtin_bit_cast(typename M::frag, f[i]), as_frag<T>(qf[ks]), s[kt])
  // S'^T = K . (cQ)^T - m_run for the tile in ring buffer KB: K fragments by inline-asm ds_read_b128 in a 2 x 4 register
  // ring with hand-counted lgkmcnt waits (hipcc serialises read -> wait -> MFMA for compiler-visible LDS reads and drains
  // the LDS-DMA in front of them).  `fill` issues the first ring-full; the caller puts independent work behind it.
  v4i ka_[4], kc_[4];
  // (asm operands inside a generic lambda do not trigger implicit capture: each lambda first binds plain references)
  auto qk_fill = [&](auto kb_c) {
    constexpr int KB = decltype(kb_c)::value * 2 * AT_TILE;
    auto& ka = ka_; auto& kc = kc_; const unsigned kaddr0 = kaddr0_;
    if constexpr (KS == 8) {
      V3D_KR(ka[0], 0, KB); V3D_KR(ka[1], 1, KB); V3D_KR(ka[2], 2, KB); V3D_KR(ka[3], 3, KB);
      V3D_KR(kc[0], 4, KB); V3D_KR(kc[1], 5, KB); V3D_KR(kc[2], 6, KB); V3D_KR(kc[3], 7, KB);
    } else {
      V3D_KR(ka[0], 0, KB); V3D_KR(ka[1], 1, KB); V3D_KR(ka[2], 2, KB); V3D_KR(ka[3], 3, KB);
      V3D_KR(kc[0], 4, KB);
      if constexpr (KSV == 6) V3D_KR(kc[1], 5, KB);
      V3D_KR(kc[2], 0, KB + 8192); V3D_KR(kc[3], 1, KB + 8192);
    }
  };
  // `mid` (r04): called where the first half of the fragment ring has had its last use - the steady-state step issues the first V^T
  // fragment reads of the P.V phase there, into the registers that half frees, one MFMA group (and its LDS latency: 300-450 cycles
  // under this load, profiles/r03_attn_probe.txt "vreads") earlier than behind the score MFMAs.  VPF = LDS reads `mid` issued: the
  // last K wait leaves exactly those in flight (LDS data returns in order).
  auto qk_run = [&](auto kb_c, f32x16 (&s)[2], auto vpf_c, auto&& mid) {
    constexpr int KB = decltype(kb_c)::value * 2 * AT_TILE;
    constexpr int VPF = decltype(vpf_c)::value;
    static_assert(VPF == 0 || VPF == 8, "the mid hook issues none or eight reads");
    auto& ka = ka_; auto& kc = kc_; const unsigned kaddr0 = kaddr0_;
    const float init = -m_run;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kt][r] = init;
    if constexpr (KS == 8) {
      V3D_KW(4, ka); V3D_KM(ka, 0, 0, 0); V3D_KM(ka, 1, 0, 1); V3D_KM(ka, 2, 0, 2); V3D_KM(ka, 3, 0, 3);
      V3D_KR(ka[0], 0, KB + 8192); V3D_KR(ka[1], 1, KB + 8192); V3D_KR(ka[2], 2, KB + 8192); V3D_KR(ka[3], 3, KB + 8192);
      V3D_KW(4, kc); V3D_KM(kc, 0, 0, 4); V3D_KM(kc, 1, 0, 5); V3D_KM(kc, 2, 0, 6); V3D_KM(kc, 3, 0, 7);
      V3D_KR(kc[0], 4, KB + 8192); V3D_KR(kc[1], 5, KB + 8192); V3D_KR(kc[2], 6, KB + 8192); V3D_KR(kc[3], 7, KB + 8192);
      V3D_KW(4, ka); V3D_KM(ka, 0, 1, 0); V3D_KM(ka, 1, 1, 1); V3D_KM(ka, 2, 1, 2); V3D_KM(ka, 3, 1, 3);
      mid();
      if constexpr (VPF == 8) { V3D_KW(8, kc); } else { V3D_KW(0, kc); }
      V3D_KM(kc, 0, 1, 4); V3D_KM(kc, 1, 1, 5); V3D_KM(kc, 2, 1, 6); V3D_KM(kc, 3, 1, 7);
    } else if constexpr (KSV == 6) {   // KS == 6 (head dim 96): 4 + 2 k-steps per key half
      V3D_KW(4, ka); V3D_KM(ka, 0, 0, 0); V3D_KM(ka, 1, 0, 1); V3D_KM(ka, 2, 0, 2); V3D_KM(ka, 3, 0, 3);
      V3D_KR(ka[0], 2, KB + 8192); V3D_KR(ka[1], 3, KB + 8192); V3D_KR(ka[2], 4, KB + 8192); V3D_KR(ka[3], 5, KB + 8192);
      V3D_KW(4, kc); V3D_KM(kc, 0, 0, 4); V3D_KM(kc, 1, 0, 5); V3D_KM(kc, 2, 1, 0); V3D_KM(kc, 3, 1, 1);
      V3D_KW(0, ka); V3D_KM(ka, 0, 1, 2); V3D_KM(ka, 1, 1, 3); V3D_KM(ka, 2, 1, 4); V3D_KM(ka, 3, 1, 5);
    } else {                           // five k-steps (dims < 80): the same ring without k-step 5's two reads and two MFMAs
      V3D_KW(3, ka); V3D_KM(ka, 0, 0, 0); V3D_KM(ka, 1, 0, 1); V3D_KM(ka, 2, 0, 2); V3D_KM(ka, 3, 0, 3);
      V3D_KR(ka[0], 2, KB + 8192); V3D_KR(ka[1], 3, KB + 8192); V3D_KR(ka[2], 4, KB + 8192);
      V3D_KW(3, kc); V3D_KM(kc, 0, 0, 4); V3D_KM(kc, 2, 1, 0); V3D_KM(kc, 3, 1, 1);
      V3D_KW(0, ka); V3D_KM(ka, 0, 1, 2); V3D_KM(ka, 1, 1, 3); V3D_KM(ka, 2, 1, 4);
    }
  };

  // Softmax part 1 for tile t (lane = one query; keys of register r: (r&3) + 8(r>>2) + 4h + 32kt): the mask (diagonal / tail
  // tiles only, wave-uniform) ...
  auto mask_scores = [&](f32x16 (&s)[2], int t) {
    const int kv0 = t * AT_BKV;
    const bool need_mask = (CAUSAL && kv0 + AT_BKV - 1 > wave_first_pos) || (kv0 + AT_BKV > p.Sk);
    if (need_mask) {   // only diagonal / tail tiles pay for it (compare + select per score)
      int last = p.Sk - 1;
      if (CAUSAL) last = q_pos < last ? q_pos : last;
      const int limit = last - kv0 - 4 * h;            // visible iff tile-local key offset <= limit
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          s[kt][r] = (kt * 32 + (r & 3) + 8 * (r >> 2)) > limit ? -INFINITY : s[kt][r];
    }
  };
  // ... and the raise of the running maximum: called for tile 0 (which FIXES the maximum) and for the rare tile whose row-sum
  // share crossed AT_LS_LIMIT.  Subtracts the raise from the scores and returns the factor O and l must be multiplied by
  // (1 = this lane's reference stays).
  auto raise_max = [&](f32x16 (&s)[2], int t) -> float {
    float mx = fmaxf(s[0][0], s[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s[0][r]), s[1][r]);
    const float mp = fmaxf(mx, __shfl_xor(mx, 32));    // both key halves of the query
    // the first tile FIXES the maximum, later ones only raise it - and only for the queries that crossed the threshold
    // themselves: the decision is per lane, so a query's rounding history depends on its own scores alone and its output
    // bits do not change with the rows that happen to share its wave (batch composition, position of the row inside the
    // query tile: tests/test_gpu_scene_reuse.py)
    float d = t == 0 ? mp : (mp > AT_RAISE ? mp : 0.f);
    d = mp == -INFINITY ? 0.f : d;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kt][r] -= d;
    m_run += d;
    return t == 0 ? 1.0f : __builtin_amdgcn_exp2f(-d);       // nothing accumulated yet at t = 0
  };
  // Softmax part 2, one quarter: 8 scores -> exp2 -> one P^T fragment (k order of the V^T fragments) + row-sum share.
  auto softmax_quarter = [&](const f32x16 (&s)[2], int i, Frag16& pf, float& ls0, float& ls1) {
    float e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = __builtin_amdgcn_exp2f(s[i >> 1][8 * (i & 1) + j]);
    ls0 += (e[0] + e[1]) + (e[2] + e[3]);
    ls1 += (e[4] + e[5]) + (e[6] + e[7]);
    pf.u = make_uint4(pack2<T>(e[0], e[1]), pack2<T>(e[2], e[3]), pack2<T>(e[4], e[5]), pack2<T>(e[6], e[7]));
  };

#define V3D_VR(f, dt, VB) { \
  const unsigned a0 = vaddr0 ^ ((dt) << 6), a1 = vaddr1 ^ ((dt) << 6); \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[0]) : "v"(a0), "i"(VB)); \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[1]) : "v"(a1), "i"(VB)); \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[2]) : "v"(a0), "i"(VB + 4096)); \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[3]) : "v"(a1), "i"(VB + 4096)); \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[4]) : "v"(a0), "i"(VB + 8192)); \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[5]) : "v"(a1), "i"(VB + 8192)); \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[6]) : "v"(a0), "i"(VB + 12288)); \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[7]) : "v"(a1), "i"(VB + 12288)); }
#define V3D_VW(cnt, f) asm volatile("s_waitcnt lgkmcnt(" #cnt ")" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : : "memory")

  // One pipeline step: S' of tile t+1 (MFMA only), then O^T += V^T . P^T of tile t with the softmax of tile t+1
  // slotted between its MFMAs (v_exp / adds / packs issue in the shadow of the 32-cycle MFMAs of the same wave).
  //   PAR = t & 1: V of tile t is in ring buffer PAR, K of tile t+1 in buffer 1-PAR.
  //   FULL = 1: steady state (tiles t+1 and t+2 exist, this wave computes both halves) - one straight-line block, so
  //   the scheduler can weave the VALU between the MFMAs; FULL = 0: the guarded form for the last two steps.
  auto step = [&](auto par_c, auto full_c, int t, Frag16 (&pc)[4], Frag16 (&pn)[4]) {
    constexpr int PAR = decltype(par_c)::value;
    constexpr bool FULL = decltype(full_c)::value != 0;
    constexpr int VB = PAR * 2 * AT_TILE;
    const unsigned vaddr0 = vaddr0_, vaddr1 = vaddr1_;
    const bool do_qk = FULL || t + 1 < n_wave, do_pv = FULL || t < n_wave;          // wave-uniform
    f32x16 s[2];
    V3D_STAMP(ts0);
    if (do_qk) qk_fill(IntC<1 - PAR>{});
    if constexpr (FULL) {
      stage_k_fast(PAR, t + 2);                                // K[PAR] (tile t) was consumed one step ago
      stage_v_fast(1 - PAR, t + 1);                            // V[1-PAR] (tile t-1) was consumed one step ago
    } else {
      if (t + 2 < n_tiles) stage_k(PAR, t + 2);
      if (t + 1 < n_tiles) stage_v(1 - PAR, t + 1);
    }
    v2i va[8], vc[8];          // V^T fragments [2*s4 + half], ring of two d-tiles
    if constexpr (FULL && KS == 8 && V3D_ATTN_VPF) {
      qk_run(IntC<1 - PAR>{}, s, IntC<8>{}, [&] { V3D_VR(va, 0, VB) });
      V3D_VR(vc, 1, VB)
    } else {
      if (do_qk) qk_run(IntC<1 - PAR>{}, s, IntC<0>{}, [] {});
      if (do_pv) { V3D_VR(va, 0, VB) V3D_VR(vc, 1, VB) }
    }
    V3D_STAMP(ts1);
    float ls0 = 0.f, ls1 = 0.f;
    if (do_qk) mask_scores(s, t + 1);
    V3D_STAMP(ts2);
    auto mmav = [&](const v2i* f, int dt, int quarter) {
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const v4i vf = {f[2 * s4][0], f[2 * s4][1], f[2 * s4 + 1][0], f[2 * s4 + 1][1]};
        o[dt] = M::run(__builtin_bit_cast(typename M::frag, vf), as_frag<T>(pc[s4]), o[dt]);
      }
      if (quarter >= 0 && do_qk) {
        softmax_quarter(s, quarter, pn[quarter], ls0, ls1);
        // keep this quarter's VALU here, between the MFMAs (the fragment is only consumed one step later and the
        // compiler would otherwise sink the whole softmax out of the MFMA shadow)
        asm volatile("" : "+v"(pn[quarter].i4), "+v"(ls0), "+v"(ls1));
      }
    };
    if (do_pv) {
      V3D_VW(8, va); mmav(va, 0, 0); V3D_VR(va, 2, VB)
      if constexpr (DT == 4) {
        V3D_VW(8, vc); mmav(vc, 1, 1); V3D_VR(vc, 3, VB)
        V3D_VW(8, va); mmav(va, 2, 2);
        V3D_VW(0, vc); mmav(vc, 3, 3);
      } else {
        V3D_VW(8, vc); mmav(vc, 1, 1);
        V3D_VW(0, va); mmav(va, 2, 2);
        if (do_qk) softmax_quarter(s, 3, pn[3], ls0, ls1);
      }
    }
    if (do_qk) {
      if (__any(ls0 + ls1 > AT_LS_LIMIT)) {     // wave-uniform, rare: some probability of tile t+1 may exceed 2^AT_RAISE
        // the scores were consumed in place by the exponentials: form them again (K of tile t+1 stays in its ring buffer until
        // the next step; same MFMA chain from the same -m, so the same bits), then decide per lane as the max-first form did
        asm volatile("; V3D_RARE_BEGIN (tests/test_kernel_resources.py: register spills are tolerated only between these markers)");
        f32x16 s2[2];
        qk_fill(IntC<1 - PAR>{});
        qk_run(IntC<1 - PAR>{}, s2, IntC<0>{}, [] {});
        mask_scores(s2, t + 1);
        const float alpha = raise_max(s2, t + 1);
        if (__any(alpha != 1.0f)) {   // the maximum moved: P of tile t+1 again at the new reference, O and l brought to it
          ls0 = 0.f; ls1 = 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) softmax_quarter(s2, i, pn[i], ls0, ls1);
#pragma unroll
          for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
          l_run *= alpha;
        }
        asm volatile("; V3D_RARE_END");
      }
      l_run += ls0 + ls1;
    }
    V3D_STAMP(ts3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    V3D_STAMP(ts4);
    __builtin_amdgcn_s_barrier();      // tiles t+1 (V) / t+2 (K) landed and visible; everyone is done with tile t
    V3D_STAMP(ts5);
    V3D_ACC(0, ts0, ts1); V3D_ACC(1, ts1, ts2); V3D_ACC(2, ts2, ts3); V3D_ACC(3, ts3, ts4); V3D_ACC(4, ts4, ts5); V3D_ACC(5, ts0, ts5);
  };

#ifdef V3D_ATTN_PROF
  const unsigned long long blk_rt0 = __builtin_amdgcn_s_memrealtime(), blk_c0 = __builtin_amdgcn_s_memtime();
#endif
  // ---- prologue: tiles 0 (K, V) and 1 (K) in flight, S' and P of tile 0 ----
  stage_k(0, 0);
  stage_v(0, 0);
  if (n_tiles > 1) stage_k(1, 1);
  // Q after the DMAs were issued: its HBM round trip runs beside theirs instead of in front of them (r03)
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const uint4 raw = *reinterpret_cast<const uint4*>(Q + (int64_t)qi_ld * p.ldq + ks * 16 + h * 8);
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = vec_get<T>(raw, j) * p.scale_log2;
    if (D > 72 && ks * 16 + 8 >= 72) {     // only tiles that can reach past d_out (the SigLIP head: 72 of the 96-wide tile)
      // dims >= d_out do not belong to this head: with heads packed at their true stride (72) they hold the NEXT head's values.
      // Zeroing them in Q removes their products from Q.K^T whatever the K tile carries there (finite activations).
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = (ks * 16 + h * 8 + j) < p.d_out ? f[j] : 0.f;
    }
    qf[ks].u = vec_pack<T>(f);
  }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  Frag16 pa[4], pb[4];
  {
    f32x16 s[2];
    float ls0 = 0.f, ls1 = 0.f;
    qk_fill(IntC<0>{});
    qk_run(IntC<0>{}, s, IntC<0>{}, [] {});
    __builtin_amdgcn_s_barrier();      // every wave has read K of tile 0 before step 0 restages its buffer (tile 2)
    mask_scores(s, 0);
    raise_max(s, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) softmax_quarter(s, i, pa[i], ls0, ls1);
    l_run = ls0 + ls1;
  }
  {
    // steady-state steps, in pairs: tiles t+1 and t+2 exist, this wave computes both, and every row of tile t+2 exists (the
    // steady state stages with unclamped addresses)
    int n_full = n_tiles - 2 < n_wave - 1 ? n_tiles - 2 : n_wave - 1;
    n_full = (n_full < p.Sk / AT_BKV - 2 ? n_full : p.Sk / AT_BKV - 2) & ~1;
    n_full = __builtin_amdgcn_readfirstlane(n_full);
    int t = 0;
    for (; t < n_full; t += 2) {
      step(IntC<0>{}, IntC<1>{}, t, pa, pb);
      step(IntC<1>{}, IntC<1>{}, t + 1, pb, pa);
    }
    for (; t < n_tiles; t += 2) {
      step(IntC<0>{}, IntC<0>{}, t, pa, pb);
      if (t + 1 < n_tiles) step(IntC<1>{}, IntC<0>{}, t + 1, pb, pa);
    }
  }
  __syncthreads();
#undef V3D_KR
#undef V3D_KW
#undef V3D_KM
#undef V3D_VR
#undef V3D_VW
#ifdef V3D_ATTN_PROF
  if (tid == 0) {
    const int bid = (p.n_qt - 1 - qt) * p.Hq + head;       // (reversed query tile, head), whatever the grid mapping
    if (bid < 4096) {
      unsigned hwid, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      g_attn_blocks[4 * bid + 0] = blk_rt0;
      g_attn_blocks[4 * bid + 1] = __builtin_amdgcn_s_memrealtime();
      g_attn_blocks[4 * bid + 2] = __builtin_amdgcn_s_memtime() - blk_c0;
      g_attn_blocks[4 * bid + 3] = ((unsigned long long)xcc << 32) | hwid;
      g_attn_blocks[4 * 4095 + (bid & 3)] = (unsigned long long)n_tiles;
    }
  }
  if (lane == 0 && head == 3 && b == 0 && p.n_qt - 1 - qt < 16) {
    unsigned long long* pr = g_attn_prof + ((p.n_qt - 1 - qt) * 4 + wave) * 8;
    for (int i = 0; i < 6; ++i) pr[i] = prof_acc[i];
    pr[6] = (unsigned long long)n_tiles;
  }
#endif

  // ---- normalise, transpose through LDS, store whole rows ----
  float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  if constexpr (LSE) {      // what the backward needs to recompute the probabilities: p = exp2(c q.k - lse)
    if (h == 0 && qi < p.Sq) p.lse[((int64_t)b * p.Hq + head) * p.Sq + qi] = l_tot > 0.f ? m_run + __log2f(l_tot) : -INFINITY;
  }
  constexpr int OROW = D * 2 + 16;
  char* so = smem + wave * 32 * OROW;
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int d = 32 * dt + 8 * r4 + 4 * h;
      uint2 pk;
      pk.x = pack2<T>(o[dt][4 * r4 + 0] * inv, o[dt][4 * r4 + 1] * inv);
      pk.y = pack2<T>(o[dt][4 * r4 + 2] * inv, o[dt][4 * r4 + 3] * inv);
      *reinterpret_cast<uint2*>(so + ql * OROW + d * 2) = pk;
    }
  __syncthreads();
  uint16_t* O = (uint16_t*)p.o + b * p.bso + (int64_t)(QPK ? hk * p.group : head) * p.hso;
  constexpr int OCH = D / 8;                   // 16-byte chunks per row
#pragma unroll
  for (int i = 0; i < (32 * OCH + 63) / 64; ++i) {
    const int idx = i * 64 + lane;
    const int row = idx / OCH, ch = idx - row * OCH;
    int q = q0 + wave * 32 + row;
    int64_t off = 0;
    bool ok = row < 32 && q < p.Sq;
    if (QPK) {                                 // slot -> (head of the group, row)
      ok = row < 32 && q < p.group * p.Sq;
      const int hh = q / p.Sq;
      q -= hh * p.Sq;
      off = (int64_t)hh * p.hso;
    }
    if (ok && ch * 8 < p.d_out)
      *reinterpret_cast<uint4*>(O + off + (int64_t)q * p.ldo + ch * 8) = *reinterpret_cast<const uint4*>(so + row * OROW + ch * 16);
  }
}

// ------------------------------------------------------------------------------------------
// r04: the SigLIP tower's attention as a PERSISTENT kernel.  attn_prefill_kernel<96, non-causal, KSV = 5> spends
// about 9 k of the 41 k cycles of an item (one 128-query tile of one (frame, head): twelve 64-key steps) at its seam - workgroup launch,
// the first tiles' DMAs and Q's HBM round trip with nothing else to do, tile 0's scores and softmax outside the pipeline, the
// normalise / transpose / store tail - and because the 3072 items are all equal, the two workgroups of a CU reach their seams TOGETHER
// for the whole launch.  Here a workgroup walks its items itself (grid = 2 per CU, items in the old grid's order), and the LAST step of
// an item - which only has the final P.V product left - already stages the next item's first tiles (K0, V0, K1) into the three ring buffers that step no longer reads and
// loads the next Q fragments into the registers the finished score phase freed.  The output tile leaves through the fourth buffer
// (the last V tile's, free after the step-end barrier), 16 query rows per wave at a time, wave-private: no workgroup barrier in the tail.
// Same arithmetic in the same order as attn_prefill_kernel<96, false, 5>: outputs bit for bit (tests/test_gpu_llm_ops.py).
// Needs an even number of key tiles (the next item's buffers are free in the last step only then): 729 keys = 12 tiles.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256, 2) void attn_vit_persistent_kernel(AttnArgs p, int n_items) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  using M = Mfma32<T>;
  constexpr int D = 96, KS = 6, DT = 3, CH = 12;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ql = lane & 31, h = lane >> 5;
  const int n_pairs = n_items / p.n_qt;                     // (batch, head) pairs; item L = pair + n_pairs * query tile
  const int n_tiles = (p.Sk + AT_BKV - 1) / AT_BKV;         // even (checked by the launcher)

  const int srow = lane >> 4;
  const int st_row0 = wave * 16 + srow;
  const int st_chunk0 = (lane & 15) ^ (srow << 2);
  const unsigned ldk_b = (unsigned)p.ldk * 2u, ldv_b = (unsigned)p.ldv * 2u;
  // every LDS-DMA of this kernel is an asm statement: with the builtin anywhere in the item loop hipcc orders EVERY compiler-visible LDS
  // access (the output staging of the tail) behind all outstanding vector-memory operations - vmcnt(0), i.e. behind the next item's
  // tiles.  `stage`: the clamped form (a tile whose last rows do not exist: tail keys are masked in the scores).
  auto stage = [&](const uint16_t* src, unsigned ld_b, unsigned lds_dst, int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int key = t * AT_BKV + st_row0 + 4 * i;
      key = key < p.Sk ? key : p.Sk - 1;
      int chunk = st_chunk0 ^ i;
      chunk = chunk < CH ? chunk : CH - 1;
      const unsigned off = (unsigned)key * ld_b + (unsigned)chunk * 16u;
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                   : : "v"(off), "s"(src), "s"(lds_dst + (unsigned)(i * 4 * AT_ROW)) : "memory", "m0");
    }
  };
  unsigned koff[4], voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int chunk = st_chunk0 ^ i;
    chunk = chunk < CH ? chunk : CH - 1;
    koff[i] = (unsigned)(st_row0 + 4 * i) * ldk_b + (unsigned)chunk * 16u;
    voff[i] = (unsigned)(st_row0 + 4 * i) * ldv_b + (unsigned)chunk * 16u;
  }
  const unsigned lds_wave0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (unsigned)(wave * 16) * AT_ROW;
  auto stage_fast = [&](const uint16_t* src, unsigned ld_b, const unsigned (&off)[4], unsigned lds_dst, int t) {
    const char* base = (const char*)src + (size_t)t * (size_t)(AT_BKV * ld_b);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                   : : "v"(off[i]), "s"(base), "s"(lds_dst + (unsigned)(i * 4 * AT_ROW)) : "memory", "m0");
  };

  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int k_sw = kv_swz(ql);
  const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
  const int v_chunk_lo = 2 * (g & 1) + (pp >> 1);
  const int v_byte = 8 * (pp & 1);
  const int v_row0 = 4 * h + qq;
  const int v_sw0 = kv_swz(v_row0), v_sw1 = kv_swz(v_row0 + 8);
  const unsigned kaddr0_ = lds_base + ql * AT_ROW + ((h ^ k_sw) << 4);
  const unsigned vaddr0_ = lds_base + AT_TILE + v_row0 * AT_ROW + v_byte + ((v_chunk_lo ^ v_sw0) << 4);
  const unsigned vaddr1_ = lds_base + AT_TILE + (v_row0 + 8) * AT_ROW + v_byte + ((v_chunk_lo ^ v_sw1) << 4);

  Frag16 qf[KS];
  // item -> pointers
  auto item_ptrs = [&](int L, const uint16_t*& Q, const uint16_t*& K, const uint16_t*& V, int& q0, int& head, int& b) {
    // item order = the one-workgroup-per-item grid's (head fastest, then query tile, then frame): the six query tiles of a (frame,
    // head) pair are in flight TOGETHER on workgroups of one XCD (ids equal mod 8), so the pair's K / V come out of that XCD's L2.
    // (Giving a workgroup the six tiles of ONE pair, one after the other, was measured first: 240 us against 130 - every tile then
    // pulls the pair's 210 KB of K / V through an L2 that 64 other pairs have flushed in the meantime.)
    head = L % p.Hq;
    const int r_ = L / p.Hq, qt = r_ % p.n_qt;
    b = r_ / p.n_qt;
    (void)n_pairs;
    const int hk = head / p.group;
    q0 = qt * AT_BQ;
    Q = (const uint16_t*)p.q + b * p.bsq + (int64_t)head * p.hsq;
    K = (const uint16_t*)p.k + b * p.bsk + (int64_t)hk * p.hsk;
    V = (const uint16_t*)p.v + b * p.bsk + (int64_t)hk * p.hsk;
  };
  // Q in two parts: the loads (issued in the previous item's last step, beside its P.V product - their round trip is not waited for
  // there) and the scaling / packing into the B-operand fragments (in the prologue, after the loads have landed)
  uint4 qraw[KS];
  auto load_q = [&](const uint16_t* Q, int q0) {
    const int qi = q0 + wave * 32 + ql;
    const int qi_ld = qi < p.Sq ? qi : p.Sq - 1;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qraw[ks] = *reinterpret_cast<const uint4*>(Q + (int64_t)qi_ld * p.ldq + ks * 16 + h * 8);
  };
  auto pack_q = [&]() {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const uint4 raw = qraw[ks];
      float f[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = vec_get<T>(raw, j) * p.scale_log2;
      if (ks * 16 + 8 >= 72) {
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (ks * 16 + h * 8 + j) < p.d_out ? f[j] : 0.f;
      }
      qf[ks].u = vec_pack<T>(f);
    }
  };

#define V3D_KR(dst, ks, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(kaddr0 ^ ((ks) << 5)), "i"(imm))
#define V3D_KW(cnt, f) asm volatile("s_waitcnt lgkmcnt(" #cnt ")" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : : "memory")
#define V3D_KM(f, i, kt, ks) s[kt] = M::run(__builtin_bit_cast(typename M::frag, f[i]), as_frag<T>(qf[ks]), s[kt])
#define V3D_VR(f, dt, VB) { \
  const unsigned a0 = vaddr0 ^ ((dt) << 6), a1 = vaddr1 ^ ((dt) << 6); \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[0]) : "v"(a0), "i"(VB)); \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[1]) : "v"(a1), "i"(VB)); \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[2]) : "v"(a0), "i"(VB + 4096)); \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[3]) : "v"(a1), "i"(VB + 4096)); \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[4]) : "v"(a0), "i"(VB + 8192)); \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[5]) : "v"(a1), "i"(VB + 8192)); \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[6]) : "v"(a0), "i"(VB + 12288)); \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[7]) : "v"(a1), "i"(VB + 12288)); }
#define V3D_VW(cnt, f) asm volatile("s_waitcnt lgkmcnt(" #cnt ")" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : : "memory")

  bool prefetched = false;
  for (int L = blockIdx.x; L < n_items; L += gridDim.x) {
    const uint16_t *Q, *K, *V;
    int q0, head, b;
    item_ptrs(L, Q, K, V, q0, head, b);
    const int Ln = L + (int)gridDim.x;
    const bool has_next = Ln < n_items;                       // workgroup-uniform

    auto stage_k = [&](int buf, int t) { stage(K, ldk_b, lds_wave0 + buf * 2 * AT_TILE, t); };
    auto stage_v = [&](int buf, int t) { stage(V, ldv_b, lds_wave0 + buf * 2 * AT_TILE + AT_TILE, t); };
    auto stage_k_fast = [&](int buf, int t) { stage_fast(K, ldk_b, koff, lds_wave0 + buf * 2 * AT_TILE, t); };
    auto stage_v_fast = [&](int buf, int t) { stage_fast(V, ldv_b, voff, lds_wave0 + buf * 2 * AT_TILE + AT_TILE, t); };

    f32x16 o[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m_run = 0.f, l_run = 0.f;

    v4i ka_[4], kc_[4];
    auto qk_fill = [&](auto kb_c) {
      constexpr int KB = decltype(kb_c)::value * 2 * AT_TILE;
      auto& ka = ka_; auto& kc = kc_; const unsigned kaddr0 = kaddr0_;
      V3D_KR(ka[0], 0, KB); V3D_KR(ka[1], 1, KB); V3D_KR(ka[2], 2, KB); V3D_KR(ka[3], 3, KB);
      V3D_KR(kc[0], 4, KB);
      V3D_KR(kc[2], 0, KB + 8192); V3D_KR(kc[3], 1, KB + 8192);
    };
    auto qk_run = [&](auto kb_c, f32x16 (&s)[2]) {
      constexpr int KB = decltype(kb_c)::value * 2 * AT_TILE;
      auto& ka = ka_; auto& kc = kc_; const unsigned kaddr0 = kaddr0_;
      const float init = -m_run;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kt][r] = init;
      V3D_KW(3, ka); V3D_KM(ka, 0, 0, 0); V3D_KM(ka, 1, 0, 1); V3D_KM(ka, 2, 0, 2); V3D_KM(ka, 3, 0, 3);
      V3D_KR(ka[0], 2, KB + 8192); V3D_KR(ka[1], 3, KB + 8192); V3D_KR(ka[2], 4, KB + 8192);
      V3D_KW(3, kc); V3D_KM(kc, 0, 0, 4); V3D_KM(kc, 2, 1, 0); V3D_KM(kc, 3, 1, 1);
      V3D_KW(0, ka); V3D_KM(ka, 0, 1, 2); V3D_KM(ka, 1, 1, 3); V3D_KM(ka, 2, 1, 4);
    };
    auto mask_scores = [&](f32x16 (&s)[2], int t) {
      const int kv0 = t * AT_BKV;
      if (kv0 + AT_BKV > p.Sk) {
        const int limit = p.Sk - 1 - kv0 - 4 * h;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            s[kt][r] = (kt * 32 + (r & 3) + 8 * (r >> 2)) > limit ? -INFINITY : s[kt][r];
      }
    };
    auto raise_max = [&](f32x16 (&s)[2], int t) -> float {
      float mx = fmaxf(s[0][0], s[1][0]);
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s[0][r]), s[1][r]);
      const float mp = fmaxf(mx, __shfl_xor(mx, 32));
      float d = t == 0 ? mp : (mp > AT_RAISE ? mp : 0.f);
      d = mp == -INFINITY ? 0.f : d;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kt][r] -= d;
      m_run += d;
      return t == 0 ? 1.0f : __builtin_amdgcn_exp2f(-d);
    };
    auto softmax_quarter = [&](const f32x16 (&s)[2], int i, Frag16& pf, float& ls0, float& ls1) {
      float e[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = __builtin_amdgcn_exp2f(s[i >> 1][8 * (i & 1) + j]);
      ls0 += (e[0] + e[1]) + (e[2] + e[3]);
      ls1 += (e[4] + e[5]) + (e[6] + e[7]);
      pf.u = make_uint4(pack2<T>(e[0], e[1]), pack2<T>(e[2], e[3]), pack2<T>(e[4], e[5]), pack2<T>(e[6], e[7]));
    };

    // one pipeline step of attn_prefill_kernel (non-causal: every wave computes every tile); LAST = the item's final step, which
    // stages the next item's first tiles and loads its Q instead of this item's (there are none left)
    auto step = [&](auto par_c, auto full_c, auto last_c, int t, Frag16 (&pc)[4], Frag16 (&pn)[4]) {
      constexpr int PAR = decltype(par_c)::value;
      constexpr bool FULL = decltype(full_c)::value != 0;
      constexpr bool LAST = decltype(last_c)::value != 0;
      constexpr int VB = PAR * 2 * AT_TILE;
      const unsigned vaddr0 = vaddr0_, vaddr1 = vaddr1_;
      const bool do_qk = FULL || t + 1 < n_tiles;
      f32x16 s[2];
      if (do_qk) qk_fill(IntC<1 - PAR>{});
      if constexpr (FULL) {
        stage_k_fast(PAR, t + 2);
        stage_v_fast(1 - PAR, t + 1);
      } else if constexpr (LAST) {
        if (has_next) {          // PAR = 1 here (even tile count): K[0], K[1] and V[0] are free; V[1] is this step's
          const uint16_t *Qn, *Kn, *Vn;      // (formed here, not carried through the item: they would live in 6 scalar pairs for 12 steps)
          int q0n, headn, bn;
          item_ptrs(Ln, Qn, Kn, Vn, q0n, headn, bn);
          // (tiles 0 and 1 are whole: the launcher asks for two full key tiles)
          stage_fast(Kn, ldk_b, koff, lds_wave0, 0);
          stage_fast(Vn, ldv_b, voff, lds_wave0 + AT_TILE, 0);
          stage_fast(Kn, ldk_b, koff, lds_wave0 + 2 * AT_TILE, 1);
          load_q(Qn, q0n);       // the score phase of this item is over: its Q registers are free
        }
      } else {
        if (t + 2 < n_tiles) stage_k(PAR, t + 2);
        if (t + 1 < n_tiles) stage_v(1 - PAR, t + 1);
      }
      if (do_qk) qk_run(IntC<1 - PAR>{}, s);
      v2i va[8], vc[8];
      { V3D_VR(va, 0, VB) V3D_VR(vc, 1, VB) }
      float ls0 = 0.f, ls1 = 0.f;
      if (do_qk) mask_scores(s, t + 1);
      auto mmav = [&](const v2i* f, int dt, int quarter) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const v4i vf = {f[2 * s4][0], f[2 * s4][1], f[2 * s4 + 1][0], f[2 * s4 + 1][1]};
          o[dt] = M::run(__builtin_bit_cast(typename M::frag, vf), as_frag<T>(pc[s4]), o[dt]);
        }
        if (quarter >= 0 && do_qk) {
          softmax_quarter(s, quarter, pn[quarter], ls0, ls1);
          asm volatile("" : "+v"(pn[quarter].i4), "+v"(ls0), "+v"(ls1));
        }
      };
      V3D_VW(8, va); mmav(va, 0, 0); V3D_VR(va, 2, VB)
      V3D_VW(8, vc); mmav(vc, 1, 1);
      V3D_VW(0, va); mmav(va, 2, 2);
      if (do_qk) softmax_quarter(s, 3, pn[3], ls0, ls1);
      if (do_qk) {
        if (__any(ls0 + ls1 > AT_LS_LIMIT)) {
          asm volatile("; V3D_RARE_BEGIN (tests/test_kernel_resources.py: register spills are tolerated only between these markers)");
          f32x16 s2[2];
          qk_fill(IntC<1 - PAR>{});
          qk_run(IntC<1 - PAR>{}, s2);
          mask_scores(s2, t + 1);
          const float alpha = raise_max(s2, t + 1);
          if (__any(alpha != 1.0f)) {
            ls0 = 0.f; ls1 = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) softmax_quarter(s2, i, pn[i], ls0, ls1);
#pragma unroll
            for (int i = 0; i < DT; ++i)
#pragma unroll
              for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
            l_run *= alpha;
          }
          asm volatile("; V3D_RARE_END");
        }
        l_run += ls0 + ls1;
      }
      if constexpr (!LAST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the last step's DMAs are the NEXT item's: waited for in its prologue)
      __builtin_amdgcn_s_barrier();
    };

    // ---- prologue: tiles 0 (K, V) and 1 (K) staged (by the previous item's last step, or here), S' and P of tile 0 ----
    if (!prefetched) {
      stage_k(0, 0);
      stage_v(0, 0);
      if (n_tiles > 1) stage_k(1, 1);
      load_q(Q, q0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    pack_q();
    Frag16 pa[4], pb[4];
    {
      f32x16 s[2];
      float ls0 = 0.f, ls1 = 0.f;
      qk_fill(IntC<0>{});
      qk_run(IntC<0>{}, s);
      __builtin_amdgcn_s_barrier();      // every wave has read K of tile 0 before step 0 restages its buffer (tile 2)
      mask_scores(s, 0);
      raise_max(s, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) softmax_quarter(s, i, pa[i], ls0, ls1);
      l_run = ls0 + ls1;
    }
    {
      int n_full = n_tiles - 2;
      n_full = (n_full < p.Sk / AT_BKV - 2 ? n_full : p.Sk / AT_BKV - 2) & ~1;
      n_full = __builtin_amdgcn_readfirstlane(n_full < 0 ? 0 : n_full);
      int t = 0;
      for (; t < n_full; t += 2) {
        step(IntC<0>{}, IntC<1>{}, IntC<0>{}, t, pa, pb);
        step(IntC<1>{}, IntC<1>{}, IntC<0>{}, t + 1, pb, pa);
      }
      for (; t + 2 < n_tiles; t += 2) {
        step(IntC<0>{}, IntC<0>{}, IntC<0>{}, t, pa, pb);
        step(IntC<1>{}, IntC<0>{}, IntC<0>{}, t + 1, pb, pa);
      }
      step(IntC<0>{}, IntC<0>{}, IntC<0>{}, t, pa, pb);               // t = n_tiles - 2
      step(IntC<1>{}, IntC<0>{}, IntC<1>{}, t + 1, pb, pa);           // the item's last step: P.V only + the next item's staging
    }
    prefetched = has_next;

    // ---- normalise, transpose through the (now free) V[1] buffer 16 query rows per wave at a time, store whole rows ----
    float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    constexpr int OROW = D * 2 + 16;                                  // 208 B
    char* so = smem + 3 * AT_TILE + wave * 16 * 256;                  // 4 KiB per wave (16 rows x 208 B = 3.3 KiB)
    uint16_t* O = (uint16_t*)p.o + b * p.bso + (int64_t)head * p.hso;
    constexpr int OCH = D / 8;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if ((ql >> 4) == half) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const int d = 32 * dt + 8 * r4 + 4 * h;
            uint2 pk;
            pk.x = pack2<T>(o[dt][4 * r4 + 0] * inv, o[dt][4 * r4 + 1] * inv);
            pk.y = pack2<T>(o[dt][4 * r4 + 2] * inv, o[dt][4 * r4 + 3] * inv);
            *reinterpret_cast<uint2*>(so + (ql & 15) * OROW + d * 2) = pk;
          }
      }
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < (16 * OCH + 63) / 64; ++i) {
        const int idx = i * 64 + lane;
        const int row = idx / OCH, ch = idx - row * OCH;
        const int q = q0 + wave * 32 + 16 * half + row;
        if (row < 16 && q < p.Sq && ch * 8 < p.d_out)
          *reinterpret_cast<uint4*>(O + (int64_t)q * p.ldo + ch * 8) = *reinterpret_cast<const uint4*>(so + row * OROW + ch * 16);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // the reads are done before the second half overwrites the rows
      __builtin_amdgcn_wave_barrier();
    }
  }
#undef V3D_KR
#undef V3D_KW
#undef V3D_KM
#undef V3D_VR
#undef V3D_VW
}

// ------------------------------------------------------------------------------------------
// r04: attn_prefill_kernel<128> on v_mfma_f32_16x16x32 - the A/B MI355X_MICROARCH.md asks for ('DVFS give-back' item 7, rule 28: the
// chip may hold a higher clock on one bf16 MFMA shape than on the other, so build both at the same per-wave tile and keep the faster
// by wall on random data).  Same workgroup (4 waves x 32 queries), same 64-key tiles, same LDS image / staging / swizzle, same
// pipeline step (S' of tile t+1, then O^T += V^T.P^T of tile t with tile t+1's softmax between the MFMAs), same max-deferred exp2
// softmax with the raise detected from the row-sum share.  What the 16 x 16 tile changes:
//   * a wave's 32 queries are two blocks of 16 (qb); a K fragment (16 keys x 32 dims, one ds_read_b128) feeds both blocks' MFMAs and a
//     V^T fragment (16 dims x 32 keys, two ds_read_b64_tr_b16) likewise, so LDS bytes per MFMA FLOP are those of the 32 x 32 form;
//   * S^T[key][q] tile: lane (g = lane >> 4, n = lane & 15) owns query n of the block and the 4 keys of MFMA rows 4g .. 4g+3 of each
//     16-key block - a query's 64 keys sit in 4 lanes (row sums per lane, reduced once after the loop; the rare maximum raise reduces
//     over the 4 lanes);
//   * the score accumulators are 4 registers per tile, so -m (the start value of a score chain) lives in a 4-register block per query
//     block that the chain's FIRST MFMA takes as its C operand: no per-step re-initialisation of the score registers (32 v_mov per
//     lane and step in the 32 x 32 form);
//   * MFMA row m of a key block is LDS row rho(m) = 4 pi(m >> 2) + (m & 3), pi = (0, 2, 3, 1): with the image's swizzle this keeps both
//     the K row reads (ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...) and V's transposed
//     reads (lane groups of 32) free of bank conflicts; P^T comes straight out of the S^T registers in the matching k order.
// Outputs differ from the 32 x 32 kernel's by summation order only (k runs in chunks of 32, row sums over 4 lanes).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int at16_rho(int m) {      // MFMA row within a 16-key block -> LDS row within the block
  const int a = m >> 2;
  const int pi = a == 0 ? 0 : (a == 1 ? 2 : (a == 2 ? 3 : 1));
  return 4 * pi + (m & 3);
}

// PART (r04): the shared-prefix segment of the decode attention of M rows (questions about one scene) - see AttnArgs; non-causal.
template <typename T, bool CAUSAL, bool LSE = false, bool PART = false>
__global__ __launch_bounds__(256, 2) void attn_prefill16_kernel(AttnArgs p) {
  static_assert(!PART || (!CAUSAL && !LSE), "the partial form is non-causal and writes no log-sum-exp");
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  using M = Mfma16<T>;
  constexpr int D = 128, KS = 4, DB = 8;          // 32-wide k-steps of QK^T; 16-wide d blocks of O^T

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  int qt, head, b;                                  // grid mapping: as attn_prefill_kernel
  if (p.xcd_p > 0) {
    const int n_kv = p.Hq / p.group, per_b = ((p.Hq * p.n_qt + 7) >> 3) << 3;
    b = (int)blockIdx.x / per_b;
    const int l = (int)blockIdx.x - b * per_b;
    const int xcd = l & 7, kvh = xcd % n_kv, slot = xcd / n_kv;
    const int jj = (l >> 3) * p.xcd_p + slot;
    if (jj >= p.group * p.n_qt) return;
    head = kvh * p.group + jj % p.group;
    qt = p.n_qt - 1 - jj / p.group;
  } else {
    qt = (int)gridDim.y - 1 - (int)blockIdx.y;
    head = blockIdx.x; b = blockIdx.z;
  }
  const int hk = head / p.group;
  const int q0 = qt * AT_BQ;
  if (PART) {                                      // this workgroup's key chunk (the last one may be short)
    const int left = p.sk_total - b * p.Sk;
    p.Sk = left < p.Sk ? left : p.Sk;
  }

  const uint16_t* Q = (const uint16_t*)p.q + b * p.bsq + (int64_t)head * p.hsq;
  const uint16_t* K = (const uint16_t*)p.k + b * p.bsk + (int64_t)hk * p.hsk;
  const uint16_t* V = (const uint16_t*)p.v + b * p.bsk + (int64_t)hk * p.hsk;

  // ---- Q fragments: B operand, lane (g, n) holds c * Q[16qb + n][32ks + 8g .. +8) ----
  int qi[2];
  Frag16 qf[2][KS];
  qi[0] = q0 + wave * 32 + n;
  qi[1] = qi[0] + 16;

  const int n_tiles_all = (p.Sk + AT_BKV - 1) / AT_BKV;
  int n_tiles = n_tiles_all;
  int n_wave = n_tiles_all;
  if (CAUSAL) {
    const int last_q = q0 + AT_BQ - 1 < p.Sq ? q0 + AT_BQ - 1 : p.Sq - 1;
    const int t = (p.q_pos0 + last_q) / AT_BKV + 1;
    n_tiles = t < n_tiles_all ? t : n_tiles_all;
    const int tw = (p.q_pos0 + q0 + wave * 32 + 31) / AT_BKV + 1;
    n_wave = tw < n_tiles ? tw : n_tiles;
  }

  // ---- KV staging by LDS-DMA: exactly attn_prefill_kernel's ----
  const int srow = lane >> 4;
  const int st_row0 = wave * 16 + srow;
  const int st_chunk0 = (lane & 15) ^ (srow << 2);
  const unsigned ldk_b = (unsigned)p.ldk * 2u, ldv_b = (unsigned)p.ldv * 2u;
  auto stage = [&](const uint16_t* src, unsigned ld_b, char* dst, int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int key = t * AT_BKV + st_row0 + 4 * i;
      key = key < p.Sk ? key : p.Sk - 1;
      const int chunk = st_chunk0 ^ i;
      glds16a((const char*)src + ((unsigned)key * ld_b + (unsigned)chunk * 16u), dst + i * 4 * AT_ROW);
    }
  };
  auto stage_k = [&](int buf, int t) { stage(K, ldk_b, smem + buf * 2 * AT_TILE + (wave * 16) * AT_ROW, t); };
  auto stage_v = [&](int buf, int t) { stage(V, ldv_b, smem + buf * 2 * AT_TILE + AT_TILE + (wave * 16) * AT_ROW, t); };
  unsigned koff[4], voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int chunk = st_chunk0 ^ i;
    koff[i] = (unsigned)(st_row0 + 4 * i) * ldk_b + (unsigned)chunk * 16u;
    voff[i] = (unsigned)(st_row0 + 4 * i) * ldv_b + (unsigned)chunk * 16u;
  }
  const unsigned lds_wave0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (unsigned)(wave * 16) * AT_ROW;
  auto stage_fast = [&](const uint16_t* src, unsigned ld_b, const unsigned (&off)[4], unsigned lds_dst, int t) {
    const char* base = (const char*)src + (size_t)t * (size_t)(AT_BKV * ld_b);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                   : : "v"(off[i]), "s"(base), "s"(lds_dst + (unsigned)(i * 4 * AT_ROW)) : "memory", "m0");
  };
  auto stage_k_fast = [&](int buf, int t) { stage_fast(K, ldk_b, koff, lds_wave0 + buf * 2 * AT_TILE, t); };
  auto stage_v_fast = [&](int buf, int t) { stage_fast(V, ldv_b, voff, lds_wave0 + buf * 2 * AT_TILE + AT_TILE, t); };

  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  // K row read (A operand): MFMA row n of key block kb is LDS row 16kb + rho(n); logical chunk 4ks + g
  const int krow = at16_rho(n);
  const unsigned kaddr0_ = lds_base + krow * AT_ROW + ((g ^ kv_swz(krow)) << 4);
  // V transposed read (A operand of O^T += V^T.P^T): lane 16g + 4i + jj supplies LDS row 16kb + 4pi(g) + i, 4 columns at
  // d = 16db + 4jj; the hardware transpose hands lane 16g + (4a + b) the column d = 16db + 4a + b of those four rows
  const int vi = (lane >> 2) & 3, vjj = lane & 3;
  const int vrow = at16_rho(4 * g) + vi;
  const unsigned vaddr0_ = lds_base + AT_TILE + vrow * AT_ROW + 8 * (vjj & 1) + (((vjj >> 1) ^ kv_swz(vrow)) << 4);
  const int key_g = at16_rho(4 * g);             // tile-local key of score register r of key block kb: 16kb + key_g + r

  f32x4 o[2][DB];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int i = 0; i < DB; ++i) o[qb][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run[2] = {0.f, 0.f}, l_run[2] = {0.f, 0.f};
  f32x4 negm[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};      // -m_run splat: the C operand of a score chain's first MFMA
  const int q_pos[2] = {p.q_pos0 + qi[0], p.q_pos0 + qi[1]};
  const int wave_first_pos = p.q_pos0 + q0 + wave * 32;

#define V3D_KR(dst, ks, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(kaddr0 ^ ((ks) << 6)), "i"(imm))
#define V3D_KW(cnt, f) asm volatile("s_waitcnt lgkmcnt(" #cnt ")" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : : "memory")
  // both query blocks' MFMAs of one K fragment; the chain's first k-step starts from -m
#define V3D_KM0(f, kb) { s[0][kb] = M::run(__builtin_bit_cast(typename M::frag, f[0]), as_frag16<T>(qf[0][0]), negm[0]); \
                         s[1][kb] = M::run(__builtin_bit_cast(typename M::frag, f[0]), as_frag16<T>(qf[1][0]), negm[1]); }
#define V3D_KM(f, ks, kb) { s[0][kb] = M::run(__builtin_bit_cast(typename M::frag, f[ks]), as_frag16<T>(qf[0][ks]), s[0][kb]); \
                            s[1][kb] = M::run(__builtin_bit_cast(typename M::frag, f[ks]), as_frag16<T>(qf[1][ks]), s[1][kb]); }
#define V3D_KM4(f, kb) { V3D_KM0(f, kb) V3D_KM(f, 1, kb) V3D_KM(f, 2, kb) V3D_KM(f, 3, kb) }
  v4i ka_[4], kc_[4];                // K fragments of two key blocks: [ks]
  auto qk_fill = [&](auto kb_c) {
    constexpr int KB = decltype(kb_c)::value * 2 * AT_TILE;
    auto& ka = ka_; auto& kc = kc_; const unsigned kaddr0 = kaddr0_;
    V3D_KR(ka[0], 0, KB); V3D_KR(ka[1], 1, KB); V3D_KR(ka[2], 2, KB); V3D_KR(ka[3], 3, KB);
    V3D_KR(kc[0], 0, KB + 4096); V3D_KR(kc[1], 1, KB + 4096); V3D_KR(kc[2], 2, KB + 4096); V3D_KR(kc[3], 3, KB + 4096);
  };
  auto qk_run = [&](auto kb_c, f32x4 (&s)[2][4], auto vpf_c, auto&& mid) {      // (mid / VPF: see attn_prefill_kernel)
    constexpr int KB = decltype(kb_c)::value * 2 * AT_TILE;
    constexpr int VPF = decltype(vpf_c)::value;
    auto& ka = ka_; auto& kc = kc_; const unsigned kaddr0 = kaddr0_;
    V3D_KW(4, ka); V3D_KM4(ka, 0)
    V3D_KR(ka[0], 0, KB + 8192); V3D_KR(ka[1], 1, KB + 8192); V3D_KR(ka[2], 2, KB + 8192); V3D_KR(ka[3], 3, KB + 8192);
    V3D_KW(4, kc); V3D_KM4(kc, 1)
    V3D_KR(kc[0], 0, KB + 12288); V3D_KR(kc[1], 1, KB + 12288); V3D_KR(kc[2], 2, KB + 12288); V3D_KR(kc[3], 3, KB + 12288);
    V3D_KW(4, ka); V3D_KM4(ka, 2)
    mid();
    if constexpr (VPF == 8) { V3D_KW(8, kc); } else { V3D_KW(0, kc); }
    V3D_KM4(kc, 3)
  };

  auto mask_scores = [&](f32x4 (&s)[2][4], int t) {
    const int kv0 = t * AT_BKV;
    const bool need_mask = (CAUSAL && kv0 + AT_BKV - 1 > wave_first_pos) || (kv0 + AT_BKV > p.Sk);
    if (need_mask) {
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        int last = p.Sk - 1;
        if (CAUSAL) last = q_pos[qb] < last ? q_pos[qb] : last;
        const int limit = last - kv0 - key_g;            // visible iff 16kb + r <= limit
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[qb][kb][r] = (16 * kb + r) > limit ? -INFINITY : s[qb][kb][r];
      }
    }
  };
  // the raise of the running maximum of query block qb (tile 0 fixes it; later only the rare tile whose row-sum share crossed the
  // limit gets here): per QUERY decision, as attn_prefill_kernel's - the query's 64 scores of the tile sit in 4 lanes
  auto raise_max = [&](f32x4 (&s)[2][4], int qb, int t) -> float {
    float mx = s[qb][0][0];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[qb][kb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    const float mp = fmaxf(mx, __shfl_xor(mx, 32));
    float d = t == 0 ? mp : (mp > AT_RAISE ? mp : 0.f);
    d = mp == -INFINITY ? 0.f : d;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) s[qb][kb][r] -= d;
    m_run[qb] += d;
    const float nm = -m_run[qb];
    negm[qb] = f32x4{nm, nm, nm, nm};
    return t == 0 ? 1.0f : __builtin_amdgcn_exp2f(-d);
  };
  // quarter (qb, j): the 8 scores of key blocks 2j, 2j+1 -> exp2 -> the P^T fragment of 32-key step j (k index 8g + i <-> key
  // 16(2j + (i >> 2)) + key_g + (i & 3): the order V^T's fragments are read in) + the lane's row-sum share
  auto softmax_quarter = [&](const f32x4 (&s)[2][4], int qb, int j, Frag16& pf, float& ls) {
    float e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = __builtin_amdgcn_exp2f(s[qb][2 * j + (i >> 2)][i & 3]);
    ls += ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]));
    pf.u = make_uint4(pack2<T>(e[0], e[1]), pack2<T>(e[2], e[3]), pack2<T>(e[4], e[5]), pack2<T>(e[6], e[7]));
  };

  // V^T fragments of one d-block pair (2dp, 2dp+1): f[4dbl + kb] = rows of key block kb, d block 2dp + dbl
#define V3D_VR(f, dp, VB) { \
  const unsigned a0 = vaddr0 ^ ((2 * (dp)) << 5), a1 = vaddr0 ^ ((2 * (dp) + 1) << 5); \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[0]) : "v"(a0), "i"(VB)); \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[1]) : "v"(a0), "i"(VB + 4096)); \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[2]) : "v"(a0), "i"(VB + 8192)); \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[3]) : "v"(a0), "i"(VB + 12288)); \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[4]) : "v"(a1), "i"(VB)); \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[5]) : "v"(a1), "i"(VB + 4096)); \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[6]) : "v"(a1), "i"(VB + 8192)); \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[7]) : "v"(a1), "i"(VB + 12288)); }
#define V3D_VW(cnt, f) asm volatile("s_waitcnt lgkmcnt(" #cnt ")" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : : "memory")

  // One pipeline step (attn_prefill_kernel's): S' of tile t+1, then O^T += V^T . P^T of tile t with the softmax of tile t+1 slotted
  // between its MFMAs.  pc / pn: P^T fragments [2qb + j] of tile t / t+1.
  auto step = [&](auto par_c, auto full_c, int t, Frag16 (&pc)[4], Frag16 (&pn)[4]) {
    constexpr int PAR = decltype(par_c)::value;
    constexpr bool FULL = decltype(full_c)::value != 0;
    constexpr int VB = PAR * 2 * AT_TILE;
    const unsigned vaddr0 = vaddr0_;
    const bool do_qk = FULL || t + 1 < n_wave, do_pv = FULL || t < n_wave;          // wave-uniform
    f32x4 s[2][4];
    if (do_qk) qk_fill(IntC<1 - PAR>{});
    if constexpr (FULL) {
      stage_k_fast(PAR, t + 2);
      stage_v_fast(1 - PAR, t + 1);
    } else {
      if (t + 2 < n_tiles) stage_k(PAR, t + 2);
      if (t + 1 < n_tiles) stage_v(1 - PAR, t + 1);
    }
    v2i va[8], vc[8];
    if constexpr (FULL && V3D_ATTN_VPF) {
      qk_run(IntC<1 - PAR>{}, s, IntC<8>{}, [&] { V3D_VR(va, 0, VB) });
      V3D_VR(vc, 1, VB)
    } else {
      if (do_qk) qk_run(IntC<1 - PAR>{}, s, IntC<0>{}, [] {});
      if (do_pv) { V3D_VR(va, 0, VB) V3D_VR(vc, 1, VB) }
    }
    float ls[2] = {0.f, 0.f};
    if (do_qk) mask_scores(s, t + 1);
    auto mmav = [&](const v2i* f, int dp, int quarter) {
#pragma unroll
      for (int dbl = 0; dbl < 2; ++dbl)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const v4i vf = {f[4 * dbl + 2 * j][0], f[4 * dbl + 2 * j][1], f[4 * dbl + 2 * j + 1][0], f[4 * dbl + 2 * j + 1][1]};
          o[0][2 * dp + dbl] = M::run(__builtin_bit_cast(typename M::frag, vf), as_frag16<T>(pc[j]), o[0][2 * dp + dbl]);
          o[1][2 * dp + dbl] = M::run(__builtin_bit_cast(typename M::frag, vf), as_frag16<T>(pc[2 + j]), o[1][2 * dp + dbl]);
        }
      if (do_qk) {
        softmax_quarter(s, quarter >> 1, quarter & 1, pn[quarter], ls[quarter >> 1]);
        asm volatile("" : "+v"(pn[quarter].i4), "+v"(ls[0]), "+v"(ls[1]));      // keep the quarter's VALU here, between the MFMAs
      }
    };
    if (do_pv) {
      V3D_VW(8, va); mmav(va, 0, 0); V3D_VR(va, 2, VB)
      V3D_VW(8, vc); mmav(vc, 1, 1); V3D_VR(vc, 3, VB)
      V3D_VW(8, va); mmav(va, 2, 2);
      V3D_VW(0, vc); mmav(vc, 3, 3);
    }
    if (do_qk) {
      if (__any(ls[0] > AT_LS_LIMIT || ls[1] > AT_LS_LIMIT)) {     // wave-uniform, rare (a lane holds 16 of a query's probabilities)
        asm volatile("; V3D_RARE_BEGIN (tests/test_kernel_resources.py: register spills are tolerated only between these markers)");
        f32x4 s2[2][4];
        qk_fill(IntC<1 - PAR>{});
        qk_run(IntC<1 - PAR>{}, s2, IntC<0>{}, [] {});
        mask_scores(s2, t + 1);
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
          const float alpha = raise_max(s2, qb, t + 1);
          if (__any(alpha != 1.0f)) {
            ls[qb] = 0.f;
            softmax_quarter(s2, qb, 0, pn[2 * qb], ls[qb]);
            softmax_quarter(s2, qb, 1, pn[2 * qb + 1], ls[qb]);
#pragma unroll
            for (int i = 0; i < DB; ++i)
#pragma unroll
              for (int r = 0; r < 4; ++r) o[qb][i][r] *= alpha;
            l_run[qb] *= alpha;
          }
        }
        asm volatile("; V3D_RARE_END");
      }
      l_run[0] += ls[0];
      l_run[1] += ls[1];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };

  // ---- prologue ----
  stage_k(0, 0);
  stage_v(0, 0);
  if (n_tiles > 1) stage_k(1, 1);
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qi_ld = qi[qb] < p.Sq ? qi[qb] : p.Sq - 1;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int64_t qoff = PART ? (int64_t)(qi_ld / p.q_rpg) * p.ldq + (qi_ld % p.q_rpg) * D : (int64_t)qi_ld * p.ldq;
      const uint4 raw = *reinterpret_cast<const uint4*>(Q + qoff + ks * 32 + g * 8);
      float f[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = vec_get<T>(raw, j) * p.scale_log2;
      qf[qb][ks].u = vec_pack<T>(f);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  Frag16 pa[4], pb[4];
  {
    f32x4 s[2][4];
    qk_fill(IntC<0>{});
    qk_run(IntC<0>{}, s, IntC<0>{}, [] {});
    __builtin_amdgcn_s_barrier();      // every wave has read K of tile 0 before step 0 restages its buffer (tile 2)
    mask_scores(s, 0);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      raise_max(s, qb, 0);
      float ls = 0.f;
      softmax_quarter(s, qb, 0, pa[2 * qb], ls);
      softmax_quarter(s, qb, 1, pa[2 * qb + 1], ls);
      l_run[qb] = ls;
    }
  }
  {
    int n_full = n_tiles - 2 < n_wave - 1 ? n_tiles - 2 : n_wave - 1;
    n_full = (n_full < p.Sk / AT_BKV - 2 ? n_full : p.Sk / AT_BKV - 2) & ~1;
    n_full = __builtin_amdgcn_readfirstlane(n_full);
    int t = 0;
    for (; t < n_full; t += 2) {
      step(IntC<0>{}, IntC<1>{}, t, pa, pb);
      step(IntC<1>{}, IntC<1>{}, t + 1, pb, pa);
    }
    for (; t < n_tiles; t += 2) {
      step(IntC<0>{}, IntC<0>{}, t, pa, pb);
      if (t + 1 < n_tiles) step(IntC<1>{}, IntC<0>{}, t + 1, pb, pa);
    }
  }
  __syncthreads();
#undef V3D_KR
#undef V3D_KW
#undef V3D_KM0
#undef V3D_KM
#undef V3D_KM4
#undef V3D_VR
#undef V3D_VW

  if constexpr (PART) {     // the decode kernels' partial: unnormalised O (f32) + the reference maximum m (scaled log2 units) + row sum l
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 16);
      l_tot = l_tot + __shfl_xor(l_tot, 32);
      if (qi[qb] < p.Sq) {
        const int row = qi[qb] / p.q_rpg, hd_ = head * p.q_rpg + qi[qb] % p.q_rpg;
        float* w = (float*)p.o + (int64_t)row * p.part_ws_stride + ((int64_t)(p.part_split0 + b) * p.part_hq + hd_) * (D + 2);
#pragma unroll
        for (int db = 0; db < DB; ++db)
          *reinterpret_cast<float4*>(w + 16 * db + 4 * g) = make_float4(o[qb][db][0], o[qb][db][1], o[qb][db][2], o[qb][db][3]);
        if (g == 0) { w[D] = m_run[qb]; w[D + 1] = l_tot; }
      }
    }
    return;
  }
  // ---- normalise, transpose through LDS, store whole rows ----
  constexpr int OROW = D * 2 + 16;
  char* so = smem + wave * 32 * OROW;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 16);
    l_tot = l_tot + __shfl_xor(l_tot, 32);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if constexpr (LSE) {
      if (g == 0 && qi[qb] < p.Sq) p.lse[((int64_t)b * p.Hq + head) * p.Sq + qi[qb]] = l_tot > 0.f ? m_run[qb] + __log2f(l_tot) : -INFINITY;
    }
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      const int d = 16 * db + 4 * g;
      uint2 pk;
      pk.x = pack2<T>(o[qb][db][0] * inv, o[qb][db][1] * inv);
      pk.y = pack2<T>(o[qb][db][2] * inv, o[qb][db][3] * inv);
      *reinterpret_cast<uint2*>(so + (16 * qb + n) * OROW + d * 2) = pk;
    }
  }
  __syncthreads();
  uint16_t* O = (uint16_t*)p.o + b * p.bso + (int64_t)head * p.hso;
  constexpr int OCH = D / 8;
#pragma unroll
  for (int i = 0; i < (32 * OCH + 63) / 64; ++i) {
    const int idx = i * 64 + lane;
    const int row = idx / OCH, ch = idx - row * OCH;
    const int q = q0 + wave * 32 + row;
    if (row < 32 && q < p.Sq && ch * 8 < p.d_out)
      *reinterpret_cast<uint4*>(O + (int64_t)q * p.ldo + ch * 8) = *reinterpret_cast<const uint4*>(so + row * OROW + ch * 16);
  }
}

// ------------------------------------------------------------------------------------------
// Prefill, head dim 128, 64 queries per wave (r02): workgroup = 4 waves = 256 queries of one (batch, head), ONE workgroup per CU,
// each wave alone on its SIMD with the whole 512-entry register file (O^T of its two 32-query blocks: 128 accumulator
// registers).  The same LDS image, staging, swizzle, S^T = K.Q^T / O^T += V^T.P^T operand trick and max-deferred softmax as
// attn_prefill_kernel; what changes is who hides whom.  With two independent 4-wave workgroups per CU the two waves of a SIMD
// competed for its single vector-issue port and the younger one ran 3670 cycles per tile against the older one's 2566
// (profiles/r01_attn_probe.txt): 68 % of the matrix pipe inside the loop.  Here one wave carries BOTH chains of two query
// blocks and interleaves them itself:
//   phase 1  S'^T(t+1) of block 0 (16 MFMAs), then of block 1 (16 MFMAs) with the first half of block 0's softmax (mask, max,
//            16 exp2 in place) in their shadow;
//   phase 2  O^T += V^T . P^T(t) of both blocks (32 MFMAs; every V^T fragment feeds two MFMAs) with the rest of the softmax of
//            tile t+1 in their shadow (block 0: 16 exp2 + packs; block 1: everything).
// K and V fragments are read once per tile and wave for 64 queries (half the LDS fragment traffic per flop), the K/V tiles are
// staged once per 256 queries (half the DMA pieces per flop), and one barrier per tile serves twice the work.
// ------------------------------------------------------------------------------------------
constexpr int A2_BQ = 256;
constexpr int A2_LDS = 6 * AT_TILE;     // 3-deep ring of {K, V} tile pairs: 96 KiB

#ifndef V3D_A64_ABL
#define V3D_A64_ABL 0     // tools/probes/attn64_probe.hip: bit 0 no softmax pieces, 1 no score MFMAs, 2 no P.V MFMAs, 3 no K reads, 4 no V reads, 5 no DMA
#endif
template <typename T, bool CAUSAL>
__global__ __launch_bounds__(256, 1) void attn_prefill64_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  using M = Mfma32<T>;
  constexpr int D = 128, KS = 8, DT = 4;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ql = lane & 31, h = lane >> 5;
  const int qt = (int)gridDim.y - 1 - (int)blockIdx.y;      // heaviest (last, causal) query tiles first, as attn_prefill_kernel
  const int head = blockIdx.x, b = blockIdx.z;
  const int hk = head / p.group;
  const int q0 = qt * A2_BQ;

  const uint16_t* Q = (const uint16_t*)p.q + b * p.bsq + (int64_t)head * p.hsq;
  const uint16_t* K = (const uint16_t*)p.k + b * p.bsk + (int64_t)hk * p.hsk;
  const uint16_t* V = (const uint16_t*)p.v + b * p.bsk + (int64_t)hk * p.hsk;

  // ---- Q fragments of the wave's two 32-query blocks: B operand, lane (q, h) holds c * Q[q][16ks + 8h .. +8) ----
  int qi[2];
  Frag16 qf[2][KS];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    qi[blk] = q0 + wave * 64 + blk * 32 + ql;
    const int qi_ld = qi[blk] < p.Sq ? qi[blk] : p.Sq - 1;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const uint4 raw = *reinterpret_cast<const uint4*>(Q + (int64_t)qi_ld * p.ldq + ks * 16 + h * 8);
      float f[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = vec_get<T>(raw, j) * p.scale_log2;
      qf[blk][ks].u = vec_pack<T>(f);
      asm volatile("" : "+a"(qf[blk][ks].i4));       // from here on the fragment LIVES in the accumulator half (64 registers)
    }
  }

  auto qk_mfma = [&](f32x16& acc, const v4i& a, const v4i& b_) {
    if constexpr (sizeof(T) == 2 && __is_same(T, bf16_t))
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b_));
    else
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b_));
  };
  auto pv_mfma = [&](f32x16& acc, const v4i& a, const v4i& b_) {
    if constexpr (sizeof(T) == 2 && __is_same(T, bf16_t))
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b_));
    else
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b_));
  };
  // ---- tile counts ----
  const int n_tiles_all = (p.Sk + AT_BKV - 1) / AT_BKV;
  int n_tiles = n_tiles_all;           // tiles the workgroup stages
  int n_wave = n_tiles_all;            // tiles THIS wave computes (a causal wave stops at its last query's tile)
  if (CAUSAL) {
    const int last_q = q0 + A2_BQ - 1 < p.Sq ? q0 + A2_BQ - 1 : p.Sq - 1;
    const int t = (p.q_pos0 + last_q) / AT_BKV + 1;
    n_tiles = t < n_tiles_all ? t : n_tiles_all;
    const int tw = (p.q_pos0 + q0 + wave * 64 + 63) / AT_BKV + 1;
    n_wave = tw < n_tiles ? tw : n_tiles;
  }

  // ---- KV staging by LDS-DMA (as attn_prefill_kernel): wave w stages rows [16w, 16w+16) of a tile ----
  const int srow = lane >> 4;
  const int st_row0 = wave * 16 + srow;
  const int st_chunk0 = (lane & 15) ^ (srow << 2);
  const unsigned ldk_b = (unsigned)p.ldk * 2u, ldv_b = (unsigned)p.ldv * 2u;
  auto stage = [&](const uint16_t* src, unsigned ld_b, char* dst, int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int key = t * AT_BKV + st_row0 + 4 * i;
      key = key < p.Sk ? key : p.Sk - 1;        // tail keys are masked in the scores
      const int chunk = st_chunk0 ^ i;
      glds16a((const char*)src + ((unsigned)key * ld_b + (unsigned)chunk * 16u), dst + i * 4 * AT_ROW);
    }
  };
  auto stage_k = [&](int buf, int t) { stage(K, ldk_b, smem + buf * 2 * AT_TILE + (wave * 16) * AT_ROW, t); };
  auto stage_v = [&](int buf, int t) { stage(V, ldv_b, smem + buf * 2 * AT_TILE + AT_TILE + (wave * 16) * AT_ROW, t); };

  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int k_sw = kv_swz(ql);
  const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
  const int v_chunk_lo = 2 * (g & 1) + (pp >> 1);
  const int v_byte = 8 * (pp & 1);
  const int v_row0 = 4 * h + qq;
  const int v_sw0 = kv_swz(v_row0), v_sw1 = kv_swz(v_row0 + 8);
  const unsigned kaddr0_ = lds_base + ql * AT_ROW + ((h ^ k_sw) << 4);
  const unsigned vaddr0_ = lds_base + AT_TILE + v_row0 * AT_ROW + v_byte + ((v_chunk_lo ^ v_sw0) << 4);
  const unsigned vaddr1_ = lds_base + AT_TILE + (v_row0 + 8) * AT_ROW + v_byte + ((v_chunk_lo ^ v_sw1) << 4);

  f32x16 o[2][DT];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk)
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[blk][i][r] = 0.f;
  float m_run[2] = {0.f, 0.f}, l_run[2] = {0.f, 0.f};
  const int q_pos[2] = {p.q_pos0 + qi[0], p.q_pos0 + qi[1]};
  const int blk_first_pos[2] = {p.q_pos0 + q0 + wave * 64, p.q_pos0 + q0 + wave * 64 + 32};

  // 3-deep ring of {K tile, V tile} pairs (96 KiB): tile t lives in buffer t % 3.  One wave per SIMD has no partner to hide a
  // DMA wait, so K is staged THREE tiles ahead and V two, and the step-end wait is counted (vmcnt(8): what was issued this step
  // stays in flight across the barrier).  Buffer offsets are run-time values added to the per-lane fragment addresses.
#define V3D_KR(dst, ks, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"((kaddr0 ^ ((ks) << 5)) + kb), "i"(imm))
#define V3D_KW(cnt, f) asm volatile("s_waitcnt lgkmcnt(" #cnt ")" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : : "memory")
  // Every MFMA of this kernel is inline asm with explicit register classes: the scores S'^T in VGPRs (the softmax reads them), the Q
  // fragments and O^T in the accumulator half.  hipcc does not see an MFMA in an asm statement, so the hazards it would pad are
  // padded by hand: V3D_MFMA_TO_VALU before VALU reads of MFMA results, V3D_VALU_TO_MFMA after VALU writes of MFMA operands.
#define V3D_KM(B, f, i, kt, ks) qk_mfma(s[B][kt], f[i], qf[B][ks].i4)
#define V3D_MFMA_TO_VALU() asm volatile("s_nop 15\n\ts_nop 7" ::: "memory")
#define V3D_VALU_TO_MFMA() asm volatile("s_nop 3" ::: "memory")
  v4i ka_[4], kc_[4];
  auto qk_fill = [&](unsigned kb) {
    constexpr int KB = 0;
    auto& ka = ka_; auto& kc = kc_; const unsigned kaddr0 = kaddr0_;
    V3D_KR(ka[0], 0, KB); V3D_KR(ka[1], 1, KB); V3D_KR(ka[2], 2, KB); V3D_KR(ka[3], 3, KB);
    V3D_KR(kc[0], 4, KB); V3D_KR(kc[1], 5, KB); V3D_KR(kc[2], 6, KB); V3D_KR(kc[3], 7, KB);
  };
  // S'^T = K . (cQ)^T - m_run of ONE block for the tile in ring buffer KB (16 MFMAs); `refill` re-issues the first ring-full
  // of K fragments for the other block's pass over the same tile.
  auto qk_run = [&](unsigned kb, auto blk_c, auto refill_c, f32x16 (&s)[2][2]) {
    constexpr int KB = 0;
    constexpr int B = decltype(blk_c)::value;
    constexpr bool REFILL = decltype(refill_c)::value != 0;
    auto& ka = ka_; auto& kc = kc_; const unsigned kaddr0 = kaddr0_;
    const float init = -m_run[B];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[B][kt][r] = init;
    asm volatile("" : "+v"(s[B][0]), "+v"(s[B][1]));      // the initialisation is complete before the first asm MFMA reads it
    V3D_VALU_TO_MFMA();
    V3D_KW(4, ka); V3D_KM(B, ka, 0, 0, 0); V3D_KM(B, ka, 1, 0, 1); V3D_KM(B, ka, 2, 0, 2); V3D_KM(B, ka, 3, 0, 3);
    V3D_KR(ka[0], 0, KB + 8192); V3D_KR(ka[1], 1, KB + 8192); V3D_KR(ka[2], 2, KB + 8192); V3D_KR(ka[3], 3, KB + 8192);
    V3D_KW(4, kc); V3D_KM(B, kc, 0, 0, 4); V3D_KM(B, kc, 1, 0, 5); V3D_KM(B, kc, 2, 0, 6); V3D_KM(B, kc, 3, 0, 7);
    V3D_KR(kc[0], 4, KB + 8192); V3D_KR(kc[1], 5, KB + 8192); V3D_KR(kc[2], 6, KB + 8192); V3D_KR(kc[3], 7, KB + 8192);
    V3D_KW(4, ka); V3D_KM(B, ka, 0, 1, 0); V3D_KM(B, ka, 1, 1, 1); V3D_KM(B, ka, 2, 1, 2); V3D_KM(B, ka, 3, 1, 3);
    if constexpr (REFILL) { V3D_KR(ka[0], 0, KB); V3D_KR(ka[1], 1, KB); V3D_KR(ka[2], 2, KB); V3D_KR(ka[3], 3, KB); V3D_KW(4, kc); }
    else { V3D_KW(0, kc); }
    V3D_KM(B, kc, 0, 1, 4); V3D_KM(B, kc, 1, 1, 5); V3D_KM(B, kc, 2, 1, 6); V3D_KM(B, kc, 3, 1, 7);
    if constexpr (REFILL) { V3D_KR(kc[0], 4, KB); V3D_KR(kc[1], 5, KB); V3D_KR(kc[2], 6, KB); V3D_KR(kc[3], 7, KB); }
  };

  // softmax part 1 of block B for tile t: mask (diagonal / tail tiles only), max, the (rare) raise of the running maximum.
  auto softmax_prep = [&](auto blk_c, f32x16 (&s)[2][2], int t) -> float {
    constexpr int B = decltype(blk_c)::value;
    const int kv0 = t * AT_BKV;
    const bool need_mask = (CAUSAL && kv0 + AT_BKV - 1 > blk_first_pos[B]) || (kv0 + AT_BKV > p.Sk);
    if (need_mask) {
      int last = p.Sk - 1;
      if (CAUSAL) last = q_pos[B] < last ? q_pos[B] : last;
      const int limit = last - kv0 - 4 * h;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          s[B][kt][r] = (kt * 32 + (r & 3) + 8 * (r >> 2)) > limit ? -INFINITY : s[B][kt][r];
    }
    float mx = fmaxf(s[B][0][0], s[B][1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s[B][0][r]), s[B][1][r]);
    float alpha = 1.0f;
    if (t == 0 || __any(mx > AT_RAISE)) {
      const float mp = fmaxf(mx, __shfl_xor(mx, 32));
      float d = t == 0 ? mp : (mp > AT_RAISE ? mp : 0.f);       // per-lane decision: see attn_prefill_kernel
      d = mp == -INFINITY ? 0.f : d;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[B][kt][r] -= d;
      m_run[B] += d;
      alpha = t == 0 ? 1.0f : __builtin_amdgcn_exp2f(-d);
    }
    return alpha;
  };
  // quarter i of block B: 8 scores -> exp2 -> one P^T fragment + row-sum share
  auto softmax_quarter = [&](auto blk_c, const f32x16 (&s)[2][2], int i, Frag16& pf, float& ls0, float& ls1) {
    constexpr int B = decltype(blk_c)::value;
    float e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = __builtin_amdgcn_exp2f(s[B][i >> 1][8 * (i & 1) + j]);
    ls0 += (e[0] + e[1]) + (e[2] + e[3]);
    ls1 += (e[4] + e[5]) + (e[6] + e[7]);
    pf.u = make_uint4(pack2<T>(e[0], e[1]), pack2<T>(e[2], e[3]), pack2<T>(e[4], e[5]), pack2<T>(e[6], e[7]));
  };

#define V3D_VR(f, dt, VB) { \
  const unsigned a0 = (vaddr0 ^ ((dt) << 6)) + vb, a1 = (vaddr1 ^ ((dt) << 6)) + vb; \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[0]) : "v"(a0), "i"(VB)); \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[1]) : "v"(a1), "i"(VB)); \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[2]) : "v"(a0), "i"(VB + 4096)); \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[3]) : "v"(a1), "i"(VB + 4096)); \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[4]) : "v"(a0), "i"(VB + 8192)); \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[5]) : "v"(a1), "i"(VB + 8192)); \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[6]) : "v"(a0), "i"(VB + 12288)); \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[7]) : "v"(a1), "i"(VB + 12288)); }
#define V3D_VW(cnt, f) asm volatile("s_waitcnt lgkmcnt(" #cnt ")" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : : "memory")

  // COARSE step (the wave's first and last tiles, where some of {scores, P.V, staging} are absent): tile t's P.V of both blocks,
  // tile t+1's scores and softmax, every part guarded.  PAR = t & 1: V(t) in ring buffer PAR,
  // K(t+1) in buffer 1 - PAR.  pc = P^T fragments of tile t (both blocks), pn receives those of tile t+1.
  auto cstep = [&](int t, Frag16 (&pc0)[4], Frag16 (&pc1)[4], Frag16 (&pn0)[4], Frag16 (&pn1)[4]) {
    constexpr bool FULL = false;                               // the guarded form: first / last tiles of a wave only
    constexpr int VB = 0;
    const unsigned vb = (unsigned)(t % 3) * (2 * AT_TILE), kbn = (unsigned)((t + 1) % 3) * (2 * AT_TILE);
    const unsigned vaddr0 = vaddr0_, vaddr1 = vaddr1_;
    const bool do_qk = FULL || t + 1 < n_wave, do_pv = FULL || t < n_wave;          // wave-uniform
    f32x16 s[2][2];
    float alpha0 = 1.0f, alpha1 = 1.0f, ls00 = 0.f, ls01 = 0.f, ls10 = 0.f, ls11 = 0.f;
    if (do_qk) qk_fill(kbn);
    if (FULL || t + 3 < n_tiles) stage_k((t + 3) % 3, t + 3);  // K of tile t was consumed one step ago
    if (FULL || t + 2 < n_tiles) stage_v((t + 2) % 3, t + 2);  // V of tile t - 1 was consumed one step ago
    // ---- phase 1: scores of block 0, then of block 1 with the head of block 0's softmax in their shadow
    if (do_qk) {
      qk_run(kbn, IntC<0>{}, IntC<1>{}, s);
      V3D_MFMA_TO_VALU();
      alpha0 = softmax_prep(IntC<0>{}, s, t + 1);
      softmax_quarter(IntC<0>{}, s, 0, pn0[0], ls00, ls01);
      softmax_quarter(IntC<0>{}, s, 1, pn0[1], ls00, ls01);
      qk_run(kbn, IntC<1>{}, IntC<0>{}, s);
    }
    // (block 1's scores are first read by VALU after the first two P.V groups: 16 MFMAs later)
    // ---- phase 2: P.V of both blocks, the rest of the softmax between the MFMA groups
    v2i va[8], vc[8];
    if (do_pv) { V3D_VR(va, 0, VB) V3D_VR(vc, 1, VB) }
    // O^T lives in the ACCUMULATOR half of the register file for the whole kernel: the P.V MFMAs are inline asm with "+a"
    // operands (hipcc otherwise keeps every MFMA in the VGPR form and shuttles ~650 values per tile between the two halves)
    auto mmav = [&](const v2i* f, int dt) {
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        v4i vf = {f[2 * s4][0], f[2 * s4][1], f[2 * s4 + 1][0], f[2 * s4 + 1][1]};
        asm volatile("s_nop 1" : "+v"(vf));           // the fragment is assembled (v_mov) before the asm MFMA reads it
        pv_mfma(o[0][dt], vf, pc0[s4].i4);
        pv_mfma(o[1][dt], vf, pc1[s4].i4);
      }
    };
    if (do_pv) { V3D_VW(8, va); mmav(va, 0); V3D_VR(va, 2, VB) }
    if (do_qk) {
      softmax_quarter(IntC<0>{}, s, 2, pn0[2], ls00, ls01);
      softmax_quarter(IntC<0>{}, s, 3, pn0[3], ls00, ls01);
      asm volatile("" : "+v"(pn0[2].i4), "+v"(pn0[3].i4), "+v"(ls00), "+v"(ls01));
    }
    if (do_pv) { V3D_VW(8, vc); mmav(vc, 1); V3D_VR(vc, 3, VB) }
    if (do_qk) {
      alpha1 = softmax_prep(IntC<1>{}, s, t + 1);
      softmax_quarter(IntC<1>{}, s, 0, pn1[0], ls10, ls11);
      asm volatile("" : "+v"(pn1[0].i4), "+v"(ls10), "+v"(ls11));
    }
    if (do_pv) { V3D_VW(8, va); mmav(va, 2); }
    if (do_qk) {
      softmax_quarter(IntC<1>{}, s, 1, pn1[1], ls10, ls11);
      softmax_quarter(IntC<1>{}, s, 2, pn1[2], ls10, ls11);
      asm volatile("" : "+v"(pn1[1].i4), "+v"(pn1[2].i4), "+v"(ls10), "+v"(ls11));
    }
    if (do_pv) { V3D_VW(0, vc); mmav(vc, 3); }
    if (do_qk) {
      softmax_quarter(IntC<1>{}, s, 3, pn1[3], ls10, ls11);
      // wave-uniform, rare: the maximum moved, bring O and l to the new reference.  O sits in the accumulator half: the asm pins
      // keep its read-modify-write INSIDE the cold branch (hipcc otherwise hoists 128 v_accvgpr_read into every iteration)
      if (__any(alpha0 != 1.0f)) {
        V3D_MFMA_TO_VALU();
#pragma unroll
        for (int i = 0; i < DT; ++i) asm volatile("" : "+a"(o[0][i]));
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[0][i][r] *= alpha0;
#pragma unroll
        for (int i = 0; i < DT; ++i) asm volatile("" : "+a"(o[0][i]));
        V3D_VALU_TO_MFMA();
        l_run[0] *= alpha0;
      }
      if (__any(alpha1 != 1.0f)) {
        V3D_MFMA_TO_VALU();
#pragma unroll
        for (int i = 0; i < DT; ++i) asm volatile("" : "+a"(o[1][i]));
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[1][i][r] *= alpha1;
#pragma unroll
        for (int i = 0; i < DT; ++i) asm volatile("" : "+a"(o[1][i]));
        V3D_VALU_TO_MFMA();
        l_run[1] *= alpha1;
      }
      l_run[0] += ls00 + ls01;
      l_run[1] += ls10 + ls11;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();      // tiles t+1 (V) / t+2 (K) landed and visible; everyone is done with tile t
  };


  // ---- the steady-state step, hand-interleaved: 64 MFMAs, one softmax slice after every second MFMA ----
  // MFMA order: S'^T(t+1) of block 0 (16), of block 1 (16), then O^T += V^T.P^T(t) for both blocks (32, a V^T fragment feeds two).
  // Softmax of a block and tile = 16 slices: 0-1 mask / max / (rare) raise; 2-9 four exp2 each, in place; 10-13 pack one P^T
  // fragment each; 14-15 idle.  Placement (MFMA pairs 0..31 of step t):
  //   pairs  0.. 7 (scores of block 0)    block 1, tile t,   slices 8-15  -> P_1(t), consumed from pair 16 on
  //   pairs  8..15 (scores of block 1)    block 0, tile t+1, slices 0-7
  //   pairs 16..23 (P.V, d tiles 0-1)     block 0, tile t+1, slices 8-15  -> P_0(t+1) (second buffer: P_0(t) is being read)
  //   pairs 24..31 (P.V, d tiles 2-3)     block 1, tile t+1, slices 0-7   (its scores stay in registers into the next step)
  // so that every MFMA gap carries ~3 VALU instructions, one v_exp_f32 among them, and neither pipe waits for the other.
  f32x16 s0[2], s1[2];                       // scores of block 0 / block 1 (block 1's live across the step boundary)
  float a0_ = 1.0f, a1_ = 1.0f, ls0a = 0.f, ls0b = 0.f, ls1a = 0.f, ls1b = 0.f, mx0 = 0.f, mx1 = 0.f, cy0 = 0.f, cy1 = 0.f;
  // One wave per SIMD issues in order: a dependent MFMA blocks everything behind it, so the softmax has to sit in EVERY MFMA gap in
  // pieces of <= 24 issue cycles (an MFMA holds the vector issue for 8 of its 32; v_exp_f32 costs 8, plain VALU 4).  The softmax of a
  // block and tile is 32 half-slices: 0-3 mask / max / (rare) raise; 4-19 two exp2 each, in place, + their row sum; 20-27 pack half
  // a P^T fragment each; 28-31 re-arm eight score registers each with -m for the block's next tile.
  auto sm = [&](auto blk_c, auto h_c, auto masked_c, f32x16 (&sc)[2], Frag16 (&dst)[4], int tt, float& alpha, float& lsa, float& lsb, float& mx,
                float& carry) {
    constexpr int B = decltype(blk_c)::value, HH = decltype(h_c)::value;
    if constexpr ((V3D_A64_ABL & 1) != 0) { __builtin_amdgcn_sched_barrier(0); return; }
    if constexpr (HH < 4) {
      constexpr int kt = HH >> 1, base = 8 * (HH & 1);
      if constexpr (HH == 0) {
        const int kv0 = tt * AT_BKV;
        const bool need_mask = decltype(masked_c)::value && ((CAUSAL && kv0 + AT_BKV - 1 > blk_first_pos[B]) || (kv0 + AT_BKV > p.Sk));
        if (need_mask) {
          int last = p.Sk - 1;
          if (CAUSAL) last = q_pos[B] < last ? q_pos[B] : last;
          const int limit = last - kv0 - 4 * h;
#pragma unroll
          for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[k2][r] = (k2 * 32 + (r & 3) + 8 * (r >> 2)) > limit ? -INFINITY : sc[k2][r];
        }
        mx = sc[0][0];
      }
#pragma unroll
      for (int r = (HH == 0 ? 1 : 0); r < 8; ++r) mx = fmaxf(mx, sc[kt][base + r]);
      if constexpr (HH == 3) {
        alpha = 1.0f;
        if (tt == 0 || __any(mx > AT_RAISE)) {
          const float mp = fmaxf(mx, __shfl_xor(mx, 32));
          float d = tt == 0 ? mp : (mp > AT_RAISE ? mp : 0.f);       // per-lane decision: see attn_prefill_kernel
          d = mp == -INFINITY ? 0.f : d;
#pragma unroll
          for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[k2][r] -= d;
          m_run[B] += d;
          alpha = tt == 0 ? 1.0f : __builtin_amdgcn_exp2f(-d);
        }
        asm volatile("" : "+v"(alpha), "+v"(mx));
      } else {
        asm volatile("" : "+v"(mx));              // (the pins materialise a piece's results HERE, in the order of the asm MFMAs:
      }                                           //  IR-level sinking otherwise moves the arithmetic to its first use)
    } else if constexpr (HH < 20) {
      constexpr int e = HH - 4, q = e >> 2, hf = (e >> 1) & 1, half = e & 1, kt = q >> 1, base = 8 * (q & 1) + 4 * hf + 2 * half;
      const float e0 = __builtin_amdgcn_exp2f(sc[kt][base]), e1 = __builtin_amdgcn_exp2f(sc[kt][base + 1]);
      sc[kt][base] = e0; sc[kt][base + 1] = e1;
      if constexpr (half == 0) {
        carry = e0 + e1;
        asm volatile("" : "+v"(sc[kt]), "+v"(carry));
      } else {
        if (hf == 0) lsa += carry + (e0 + e1); else lsb += carry + (e0 + e1);
        asm volatile("" : "+v"(sc[kt]), "+v"(lsa), "+v"(lsb));
      }
    } else if constexpr (HH < 28) {
      constexpr int c = HH - 20, q = c >> 1, half = c & 1, kt = q >> 1, base = 8 * (q & 1) + 4 * half;
      const unsigned w0 = pack2<T>(sc[kt][base + 0], sc[kt][base + 1]), w1 = pack2<T>(sc[kt][base + 2], sc[kt][base + 3]);
      if constexpr (half == 0) { dst[q].u.x = w0; dst[q].u.y = w1; } else { dst[q].u.z = w0; dst[q].u.w = w1; }
      if constexpr (half == 1) asm volatile("" : "+v"(dst[q].i4));
      else asm volatile("" : "+v"(dst[q].u.x), "+v"(dst[q].u.y));
    } else {
      constexpr int i = HH - 28, kt = i >> 1, base = 8 * (i & 1);
      const float init = -m_run[B];
#pragma unroll
      for (int r = 0; r < 8; ++r) {              // asm moves: placed here, and written straight into the MFMA's accumulator tuple
        float x;
        asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(init));
        sc[kt][base + r] = x;
      }
    }
    __builtin_amdgcn_sched_barrier(0);       // the piece stays in this MFMA gap
  };
#define V3D_H1(HH, TT) sm(IntC<1>{}, IntC<HH>{}, IntC<0>{}, s1, p1, TT, a1_, ls1a, ls1b, mx1, cy1)
#define V3D_H0(HH, TT) sm(IntC<0>{}, IntC<HH>{}, IntC<0>{}, s0, pn0, TT, a0_, ls0a, ls0b, mx0, cy0)
#define V3D_H1M(HH, TT) sm(IntC<1>{}, IntC<HH>{}, IntC<1>{}, s1, p1, TT, a1_, ls1a, ls1b, mx1, cy1)
#define V3D_H0M(HH, TT) sm(IntC<0>{}, IntC<HH>{}, IntC<1>{}, s0, pn0, TT, a0_, ls0a, ls0b, mx0, cy0)
#if (V3D_A64_ABL & 2)
#define V3D_QM(S, f, i, kt, B, ks)
#else
#define V3D_QM(S, f, i, kt, B, ks) qk_mfma(S[kt], f[i], qf[B][ks].i4)
#endif
#if (V3D_A64_ABL & 4)
#define V3D_PVM(acc, vf, pf)
#else
#define V3D_PVM(acc, vf, pf) pv_mfma(acc, vf, pf)
#endif
  // four score MFMAs of one K fragment group, a piece of block OB's softmax after each
#define V3D_QM4(S, f, kt, B, ks0, HM, h0, TT) \
    V3D_QM(S, f, 0, kt, B, ks0 + 0); HM(h0 + 0, TT); V3D_QM(S, f, 1, kt, B, ks0 + 1); HM(h0 + 1, TT); \
    V3D_QM(S, f, 2, kt, B, ks0 + 2); HM(h0 + 2, TT); V3D_QM(S, f, 3, kt, B, ks0 + 3); HM(h0 + 3, TT);
  // the two P.V MFMAs (q-blocks 0 / 1) that share one V^T fragment, a piece after each
#define V3D_PV2(f, dt, s4, HM, h0, TT) { v4i vf = {f[2 * (s4)][0], f[2 * (s4)][1], f[2 * (s4) + 1][0], f[2 * (s4) + 1][1]}; \
    asm volatile("s_nop 0" : "+v"(vf)); V3D_PVM(o[0][dt], vf, pc0[s4].i4); HM(h0 + 0, TT); V3D_PVM(o[1][dt], vf, p1[s4].i4); HM(h0 + 1, TT); }
#define V3D_PV8(f, dt, HM, h0, TT) V3D_PV2(f, dt, 0, HM, h0, TT) V3D_PV2(f, dt, 1, HM, h0 + 2, TT) V3D_PV2(f, dt, 2, HM, h0 + 4, TT) V3D_PV2(f, dt, 3, HM, h0 + 6, TT)
  Frag16 p1[4];                              // P^T of block 1 (single buffer: written in gaps 4..11, read from MFMA 32 on)
  // ---- the steady-state step: 64 MFMAs, a softmax piece in every gap ----
  //   MFMAs  0..15 (scores of block 0, tile t+1)   block 1, tile t,   pieces 16..31  -> P_1(t), consumed from MFMA 32 on
  //   MFMAs 16..31 (scores of block 1, tile t+1)   block 0, tile t+1, pieces  0..15
  //   MFMAs 32..47 (P.V of tile t, d tiles 0-1)    block 0, tile t+1, pieces 16..31  -> P_0(t+1) (second buffer: P_0(t) is being read)
  //   MFMAs 48..63 (P.V of tile t, d tiles 2-3)    block 1, tile t+1, pieces  0..15  (its scores stay in registers into the next step)
  auto fstep = [&](int t, unsigned kb, unsigned vb, int bk3, int bv2, Frag16 (&pc0)[4], Frag16 (&pn0)[4]) {
    constexpr int VB = 0, KB = 0;             // kb / vb: byte offsets of the ring buffers of K(t+1) / V(t); bk3 / bv2: buffers of K(t+3) / V(t+2)
    const unsigned vaddr0 = vaddr0_, vaddr1 = vaddr1_, kaddr0 = kaddr0_;
    auto& ka = ka_; auto& kc = kc_;
#if (V3D_A64_ABL & 8)
#define V3D_FKR(dst, ks, imm)
#else
#define V3D_FKR(dst, ks, imm) V3D_KR(dst, ks, imm)
    qk_fill(kb);
#endif
#if (V3D_A64_ABL & 16)
#define V3D_FVR(f, dt, VB)
#else
#define V3D_FVR(f, dt, VB) V3D_VR(f, dt, VB)
#endif
#if !(V3D_A64_ABL & 32)
    stage_k(bk3, t + 3);
    stage_v(bv2, t + 2);
#endif
    __builtin_amdgcn_sched_barrier(0);
    V3D_KW(4, ka); V3D_QM4(s0, ka, 0, 0, 0, V3D_H1, 16, t)
    V3D_FKR(ka[0], 0, KB + 8192); V3D_FKR(ka[1], 1, KB + 8192); V3D_FKR(ka[2], 2, KB + 8192); V3D_FKR(ka[3], 3, KB + 8192);
    l_run[1] += ls1a + ls1b;                   // tile t's row sums of block 1 are complete (its rescale happened a step ago)
    ls1a = 0.f; ls1b = 0.f;
    V3D_KW(4, kc); V3D_QM4(s0, kc, 0, 0, 4, V3D_H1, 20, t)
    V3D_FKR(kc[0], 4, KB + 8192); V3D_FKR(kc[1], 5, KB + 8192); V3D_FKR(kc[2], 6, KB + 8192); V3D_FKR(kc[3], 7, KB + 8192);
    V3D_KW(4, ka); V3D_QM4(s0, ka, 1, 0, 0, V3D_H1, 24, t)
    V3D_FKR(ka[0], 0, KB); V3D_FKR(ka[1], 1, KB); V3D_FKR(ka[2], 2, KB); V3D_FKR(ka[3], 3, KB);
    V3D_KW(4, kc); V3D_QM4(s0, kc, 1, 0, 4, V3D_H1, 28, t)
    V3D_FKR(kc[0], 4, KB); V3D_FKR(kc[1], 5, KB); V3D_FKR(kc[2], 6, KB); V3D_FKR(kc[3], 7, KB);
    V3D_KW(4, ka); V3D_QM4(s1, ka, 0, 1, 0, V3D_H0, 0, t + 1)
    V3D_FKR(ka[0], 0, KB + 8192); V3D_FKR(ka[1], 1, KB + 8192); V3D_FKR(ka[2], 2, KB + 8192); V3D_FKR(ka[3], 3, KB + 8192);
    V3D_KW(4, kc); V3D_QM4(s1, kc, 0, 1, 4, V3D_H0, 4, t + 1)
    V3D_FKR(kc[0], 4, KB + 8192); V3D_FKR(kc[1], 5, KB + 8192); V3D_FKR(kc[2], 6, KB + 8192); V3D_FKR(kc[3], 7, KB + 8192);
    v2i va[8], vc[8];
    V3D_KW(4, ka); V3D_QM4(s1, ka, 1, 1, 0, V3D_H0, 8, t + 1)
    V3D_FVR(va, 0, VB)
    V3D_KW(8, kc); V3D_QM4(s1, kc, 1, 1, 4, V3D_H0, 12, t + 1)
    V3D_FVR(vc, 1, VB)
    V3D_VW(8, va); V3D_PV8(va, 0, V3D_H0, 16, t + 1)
    V3D_FVR(va, 2, VB)
    V3D_VW(8, vc); V3D_PV8(vc, 1, V3D_H0, 24, t + 1)
    V3D_FVR(vc, 3, VB)
    V3D_VW(8, va); V3D_PV8(va, 2, V3D_H1, 0, t + 1)
    V3D_VW(0, vc); V3D_PV8(vc, 3, V3D_H1, 8, t + 1)
    // ---- rare: the maximum of a block moved - bring its O and l to the new reference (O sits in the accumulator half: the
    //      asm pins keep the read-modify-write inside the cold branch)
    if (__any(a0_ != 1.0f)) {
      V3D_MFMA_TO_VALU();
#pragma unroll
      for (int i = 0; i < DT; ++i) asm volatile("" : "+a"(o[0][i]));
#pragma unroll
      for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[0][i][r] *= a0_;
#pragma unroll
      for (int i = 0; i < DT; ++i) asm volatile("" : "+a"(o[0][i]));
      V3D_VALU_TO_MFMA();
      l_run[0] *= a0_;
    }
    if (__any(a1_ != 1.0f)) {
      V3D_MFMA_TO_VALU();
#pragma unroll
      for (int i = 0; i < DT; ++i) asm volatile("" : "+a"(o[1][i]));
#pragma unroll
      for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[1][i][r] *= a1_;
#pragma unroll
      for (int i = 0; i < DT; ++i) asm volatile("" : "+a"(o[1][i]));
      V3D_VALU_TO_MFMA();
      l_run[1] *= a1_;
    }
    l_run[0] += ls0a + ls0b;
    ls0a = 0.f; ls0b = 0.f;
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // K(t+2), V(t+1) (issued a step ago) landed; K(t+3), V(t+2) stay in flight
    __builtin_amdgcn_s_barrier();
  };

  // ---- prologue: tiles 0 (K, V) and 1 (K) in flight; scores of tile 0 for both blocks; block 0's softmax complete (P_0(0)),
  //      block 1's slices 0..7 (the steady-state step finishes them) ----
#ifdef V3D_ATTN_PROF
  const unsigned long long blk_rt0 = __builtin_amdgcn_s_memrealtime(), blk_c0 = __builtin_amdgcn_s_memtime();
  unsigned long long blk_c1 = 0, blk_c2 = 0;
#endif
  stage_k(0, 0);
  stage_v(0, 0);
  if (n_tiles > 1) { stage_k(1, 1); stage_v(1, 1); }
  if (n_tiles > 2) stage_k(2, 2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  Frag16 pa0[4], pb0[4], p1b[4];
  {
    f32x16 sp[2][2];
    qk_fill(0u);
    qk_run(0u, IntC<0>{}, IntC<1>{}, sp);
    qk_run(0u, IntC<1>{}, IntC<0>{}, sp);
    __builtin_amdgcn_s_barrier();      // every wave has read K of tile 0 before step 0 restages its ring slot (tile 3)
    V3D_MFMA_TO_VALU();
    s0[0] = sp[0][0]; s0[1] = sp[0][1]; s1[0] = sp[1][0]; s1[1] = sp[1][1];
  }
  {
    auto& pn0 = pa0;
#define V3D_RUN4(HM, h0, TT) HM(h0, TT); HM(h0 + 1, TT); HM(h0 + 2, TT); HM(h0 + 3, TT);
    V3D_H0M(0, 0); V3D_H0(1, 0); V3D_H0(2, 0); V3D_H0(3, 0);
    V3D_RUN4(V3D_H0, 4, 0) V3D_RUN4(V3D_H0, 8, 0) V3D_RUN4(V3D_H0, 12, 0) V3D_RUN4(V3D_H0, 16, 0) V3D_RUN4(V3D_H0, 20, 0) V3D_RUN4(V3D_H0, 24, 0)
    V3D_RUN4(V3D_H0, 28, 0)
    l_run[0] = ls0a + ls0b;
    ls0a = 0.f; ls0b = 0.f;
    V3D_H1M(0, 0); V3D_H1(1, 0); V3D_H1(2, 0); V3D_H1(3, 0);
    V3D_RUN4(V3D_H1, 4, 0) V3D_RUN4(V3D_H1, 8, 0) V3D_RUN4(V3D_H1, 12, 0)
  }
  {
    // steady-state steps need tiles t+3 (K) and t+2 (V) to exist and the wave to compute tile t+1; in pairs (P_0 double buffer)
    // ... and tile t + 1 to need no mask for either q-block of the wave (no diagonal, no Sk tail)
    int n_plain = p.Sk / AT_BKV;
    if (CAUSAL) { const int d = (blk_first_pos[0] + 1) / AT_BKV; n_plain = d < n_plain ? d : n_plain; }
    int n_full = n_tiles - 3 < n_wave - 1 ? n_tiles - 3 : n_wave - 1;
    n_full = (n_full < n_plain - 1 ? n_full : n_plain - 1);
    n_full = n_full < 0 ? 0 : n_full & ~1;
    int t = 0;
    int bk = 1, bv = 0;                         // ring buffers of K(t+1), V(t)
#ifdef V3D_ATTN_PROF
    blk_c1 = __builtin_amdgcn_s_memtime();
#endif
    for (; t < n_full; t += 2) {
      fstep(t, (unsigned)bk * (2 * AT_TILE), (unsigned)bv * (2 * AT_TILE), bv, bk == 2 ? 0 : bk + 1, pa0, pb0);       // K(t+3) -> buffer of tile t; V(t+2) -> bk + 1
      bk = bk == 2 ? 0 : bk + 1; bv = bv == 2 ? 0 : bv + 1;
      fstep(t + 1, (unsigned)bk * (2 * AT_TILE), (unsigned)bv * (2 * AT_TILE), bv, bk == 2 ? 0 : bk + 1, pb0, pa0);
      bk = bk == 2 ? 0 : bk + 1; bv = bv == 2 ? 0 : bv + 1;
    }
#ifdef V3D_ATTN_PROF
    blk_c2 = __builtin_amdgcn_s_memtime();
    if (tid == 0) {
      const int bid = blockIdx.y * gridDim.x + blockIdx.x;
      if (bid < 4096) { g_attn_prof[0 * 4096 + bid] = blk_c1 - blk_c0; g_attn_prof[1 * 4096 + bid] = blk_c2 - blk_c1; g_attn_prof[2 * 4096 + bid] = (unsigned long long)n_full; }
    }
#endif
    // (a coarse step reads only K(t+1) / V(t), which the previous step's counted wait + barrier already made visible, and ends
    //  with vmcnt(0): waves may leave the steady state at different t - every step, fine or coarse, has exactly one barrier)
    // leave the pipeline: finish block 1's softmax of tile t, then the guarded coarse steps for the wave's last tiles
    {
      auto& pn0 = pb0;      // (unused by slices 8..15 of block 1)
      (void)pn0;
      V3D_RUN4(V3D_H1, 16, t) V3D_RUN4(V3D_H1, 20, t) V3D_RUN4(V3D_H1, 24, t)
      l_run[1] += ls1a + ls1b;
    }
    for (; t < n_tiles; t += 2) {
      cstep(t, pa0, p1, pb0, p1b);
      if (t + 1 < n_tiles) cstep(t + 1, pb0, p1b, pa0, p1);
    }
  }
#undef V3D_H0
#undef V3D_H1
#undef V3D_H0M
#undef V3D_H1M
#undef V3D_RUN4
#undef V3D_QM4
#undef V3D_PV8
#undef V3D_QM
#undef V3D_PV2
  V3D_MFMA_TO_VALU();
  __syncthreads();
#undef V3D_MFMA_TO_VALU
#undef V3D_VALU_TO_MFMA
#undef V3D_KR
#undef V3D_KW
#undef V3D_KM
#undef V3D_VR
#undef V3D_VW

  // ---- normalise, transpose through LDS, store whole rows (one 32-query block at a time per wave) ----
  constexpr int OROW = D * 2 + 16;
  char* so = smem + wave * 32 * OROW;
  uint16_t* O = (uint16_t*)p.o + b * p.bso + (int64_t)head * p.hso;
  constexpr int OCH = D / 8;
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    const float l_tot = l_run[blk] + __shfl_xor(l_run[blk], 32);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int d = 32 * dt + 8 * r4 + 4 * h;
        uint2 pk;
        pk.x = pack2<T>(o[blk][dt][4 * r4 + 0] * inv, o[blk][dt][4 * r4 + 1] * inv);
        pk.y = pack2<T>(o[blk][dt][4 * r4 + 2] * inv, o[blk][dt][4 * r4 + 3] * inv);
        *reinterpret_cast<uint2*>(so + ql * OROW + d * 2) = pk;
      }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the wave's own LDS writes are done (its region is private)
#pragma unroll
    for (int i = 0; i < (32 * OCH + 63) / 64; ++i) {
      const int idx = i * 64 + lane;
      const int row = idx / OCH, ch = idx - row * OCH;
      const int q = q0 + wave * 64 + blk * 32 + row;
      if (row < 32 && q < p.Sq && ch * 8 < p.d_out)
        *reinterpret_cast<uint4*>(O + (int64_t)q * p.ldo + ch * 8) = *reinterpret_cast<const uint4*>(so + row * OROW + ch * 16);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // reads done before the next block overwrites the region
    __builtin_amdgcn_wave_barrier();
  }
#ifdef V3D_ATTN_PROF
  if (tid == 0) {
    const int bid = (p.n_qt - 1 - qt) * p.Hq + head;       // (reversed query tile, head), whatever the grid mapping
    if (bid < 4096) {
      g_attn_blocks[4 * bid + 0] = blk_rt0;
      g_attn_blocks[4 * bid + 1] = __builtin_amdgcn_s_memrealtime();
      g_attn_blocks[4 * bid + 2] = __builtin_amdgcn_s_memtime() - blk_c0;
      g_attn_blocks[4 * bid + 3] = (unsigned long long)n_tiles;
    }
  }
#endif
}

// ------------------------------------------------------------------------------------------
// Decode attention (q_len small, K/V cache long): HBM-bound cache streaming.  One workgroup per
// (query row, head); keys split over the 4 waves, 16-byte loads straight to registers, f32
// online softmax per wave, combined through LDS.
// ------------------------------------------------------------------------------------------
template <typename T, int D>
__global__ __launch_bounds__(256) void attn_decode_kernel(AttnArgs p) {
  __shared__ float red_m[4], red_l[4];
  __shared__ float red_o[4][D];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qi = blockIdx.x, head = blockIdx.y;
  const int hk = head / p.group;
  const uint16_t* Q = (const uint16_t*)p.q + (int64_t)qi * p.ldq + (int64_t)head * p.hsq;
  const uint16_t* K = (const uint16_t*)p.k + (int64_t)hk * p.hsk;
  const uint16_t* V = (const uint16_t*)p.v + (int64_t)hk * p.hsk;
  constexpr int CH = D / 8;                    // chunks per row; 16 lanes cover one key row (D=128)
  constexpr int LPR = 16;                      // lanes per key row
  const int sub = lane / LPR, cl = lane % LPR; // 4 keys per wave step
  const int n_keys = p.q_pos0 + qi + 1 < p.Sk ? p.q_pos0 + qi + 1 : p.Sk;
  float qv[8];
  {
    const uint4 q4 = cl < CH ? *reinterpret_cast<const uint4*>(Q + cl * 8) : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) qv[j] = vec_get<T>(q4, j);
  }
  float m = -INFINITY, l = 0.f, acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (int k0 = wave * 4; k0 < n_keys; k0 += 16) {
    const int key = k0 + sub;
    const bool ok = key < n_keys && cl < CH;
    const int kc = key < n_keys ? key : n_keys - 1;
    const uint4 k4 = cl < CH ? *reinterpret_cast<const uint4*>(K + (int64_t)kc * p.ldk + cl * 8) : make_uint4(0, 0, 0, 0);
    const uint4 v4 = cl < CH ? *reinterpret_cast<const uint4*>(V + (int64_t)kc * p.ldv + cl * 8) : make_uint4(0, 0, 0, 0);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s = fmaf(qv[j], vec_get<T>(k4, j), s);
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off);     // within the 16-lane row group
    s = (key < n_keys) ? s * p.scale_log2 : -INFINITY;
    const float m_new = fmaxf(m, s);
    const float m_use = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = exp2f(m - m_use);
    const float e = exp2f(s - m_use);
    l = l * alpha + e;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = acc[j] * alpha + (ok ? e * vec_get<T>(v4, j) : 0.f);
    m = m_new;
  }
  // combine the 4 key sub-groups of the wave (lanes cl, cl+16, cl+32, cl+48), then the 4 waves
  float mw = m;
  mw = fmaxf(mw, __shfl_xor(mw, 16));
  mw = fmaxf(mw, __shfl_xor(mw, 32));
  const float mu = mw == -INFINITY ? 0.f : mw;
  const float f = exp2f(m - mu);
  float lw = l * f;
  lw += __shfl_xor(lw, 16);
  lw += __shfl_xor(lw, 32);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float a = acc[j] * f;
    a += __shfl_xor(a, 16);
    a += __shfl_xor(a, 32);
    acc[j] = a;
  }
  if (lane < LPR && lane < CH) {
#pragma unroll
    for (int j = 0; j < 8; ++j) red_o[wave][lane * 8 + j] = acc[j];
  }
  if (lane == 0) { red_m[wave] = mw; red_l[wave] = lw; }
  __syncthreads();
  if (tid < p.d_out) {
    float mm = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3]));
    mm = mm == -INFINITY ? 0.f : mm;
    float lt = 0.f, ot = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float fw = exp2f(red_m[w] - mm);
      lt += red_l[w] * fw;
      ot += red_o[w][tid] * fw;
    }
    T* O = (T*)p.o + (int64_t)qi * p.ldo + (int64_t)head * p.hso;
    O[tid] = from_f32<T>(lt > 0.f ? ot / lt : 0.f);
  }
}


// ------------------------------------------------------------------------------------------
// Split-KV decode attention (one query row, long cache): workgroup = (kv head, key split); the K/V
// rows of the split are streamed ONCE for all `group` query heads that share the kv head
// (16 lanes x 16 B per 256-byte row, 16 key slots per workgroup pass), f32 online softmax per
// (slot, head), slots merged through LDS, one partial (m, l, o[128]) per (split, head) written to
// the workspace; a second tiny kernel merges the splits.  HBM-bound: cache bytes read once.
// ------------------------------------------------------------------------------------------
constexpr int DEC_MAXG = 8;
constexpr int DEC_MAXROWS = 32;

// Scenes decoding together (blockIdx.z = scene): each has its own cache, length, query row and workspace slice.
struct DecRows {
  const void* k[DEC_MAXROWS];
  const void* v[DEC_MAXROWS];
  int sk[DEC_MAXROWS];
  int64_t q_stride, o_stride;       // elements between the scenes' query / output rows
  int64_t ws_stride;                // floats between the scenes' workspace slices
  int kps, cap;                     // keys per split / most splits per scene: a scene's OWN split count min(cap, ceil(n_keys / kps))
                                    // decides its key partition, so its output does not depend on the other scenes of the launch
  int skip;                         // r04: keys < skip are NOT this launch's (the shared-prefix segment went to attn_prefill16_kernel<PART>,
                                    // its partials sit in the split slots behind this launch's): a scene partitions keys [skip, n_keys)
  const void* kp;                   // shared prefix (scene-level reuse): keys < prefix are read HERE for every scene of the launch -
  const void* vp;                   // the scenes' own copies of those rows hold the same bytes, so outputs do not change, but the
  int prefix;                       // chip reads the prefix once (L2 / Infinity Cache hits for the other scenes) instead of M times
};

template <typename T, int G>
__global__ __launch_bounds__(256) void attn_decode_split_kernel(AttnArgs p, DecRows rw, int n_split, float* __restrict__ ws) {
  constexpr int D = 128;
  {
    const int sc = blockIdx.z;
    p.k = rw.k[sc]; p.v = rw.v[sc]; p.Sk = rw.sk[sc]; p.q_pos0 = rw.sk[sc] - 1;
    p.q = (const uint16_t*)p.q + sc * rw.q_stride;
    ws += sc * rw.ws_stride;
  }
  __shared__ float sm_m[16][G], sm_l[16][G];
  __shared__ float sm_o[16][G][D + 4];
  const int tid = threadIdx.x;
  const int slot = tid >> 4, cl = tid & 15;       // 16 key slots, 16 lanes per key row
  const int hk = blockIdx.x, split = blockIdx.y;
  const int n_keys = p.q_pos0 + 1 < p.Sk ? p.q_pos0 + 1 : p.Sk;
  const int n_own = n_keys - rw.skip;                                    // (rw.skip = 0 unless the shared prefix has its own launch)
  const int ns_own = min(rw.cap, (n_own + rw.kps - 1) / rw.kps);         // splits >= ns_own are empty: (m, l, o) = (-inf, 0, 0)
  const int per = ns_own > 0 ? (n_own + ns_own - 1) / ns_own : 1;
  const int k_begin = rw.skip + (split < ns_own ? split * per : n_own);
  int k_end = k_begin + per;
  k_end = k_end < n_keys ? k_end : n_keys;
  const uint16_t* K = (const uint16_t*)p.k + (int64_t)hk * p.hsk;
  const uint16_t* V = (const uint16_t*)p.v + (int64_t)hk * p.hsk;
  const uint16_t* Kp = rw.prefix > 0 ? (const uint16_t*)rw.kp + (int64_t)hk * p.hsk : K;
  const uint16_t* Vp = rw.prefix > 0 ? (const uint16_t*)rw.vp + (int64_t)hk * p.hsk : V;
  float qv[G][8];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const uint4 q4 = *reinterpret_cast<const uint4*>((const uint16_t*)p.q + (int64_t)(hk * G + g) * p.hsq + cl * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) qv[g][j] = vec_get<T>(q4, j) * p.scale_log2;
  }
  float m[G], l[G], acc[G][8];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    m[g] = -INFINITY; l[g] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[g][j] = 0.f;
  }
  for (int kb = k_begin; kb < k_end; kb += 64) {
    // four independent (K,V) row loads per lane in flight before any arithmetic (memory-level parallelism)
    uint4 k4[4], v4[4];
    bool ok[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int key = kb + slot + 16 * it;
      ok[it] = key < k_end;
      const int kc = ok[it] ? key : k_begin;
      k4[it] = *reinterpret_cast<const uint4*>((kc < rw.prefix ? Kp : K) + (int64_t)kc * p.ldk + cl * 8);
      v4[it] = *reinterpret_cast<const uint4*>((kc < rw.prefix ? Vp : V) + (int64_t)kc * p.ldv + cl * 8);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      float kf[8], vf[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { kf[j] = vec_get<T>(k4[it], j); vf[j] = vec_get<T>(v4[it], j); }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s = fmaf(qv[g][j], kf[j], s);
        s += __shfl_xor(s, 8); s += __shfl_xor(s, 4); s += __shfl_xor(s, 2); s += __shfl_xor(s, 1);
        s = ok[it] ? s : -INFINITY;
        if (s > m[g]) {                      // running max moved (rare after the first keys): rescale this head
          const float alpha = __builtin_amdgcn_exp2f(m[g] - s);     // m = -inf -> 0
          l[g] *= alpha;
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[g][j] *= alpha;
          m[g] = s;
        }
        const float e = ok[it] ? __builtin_amdgcn_exp2f(s - m[g]) : 0.f;
        l[g] += e;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[g][j] = fmaf(e, vf[j], acc[g][j]);
      }
    }
  }
  // merge the 16 key slots: partial accumulators -> LDS, 16 x G rescale factors computed ONCE, then 16 FMAs per output
#pragma unroll
  for (int g = 0; g < G; ++g) {
    if (cl == 0) { sm_m[slot][g] = m[g]; sm_l[slot][g] = l[g]; }
    float4* dst = reinterpret_cast<float4*>(&sm_o[slot][g][cl * 8]);
    dst[0] = make_float4(acc[g][0], acc[g][1], acc[g][2], acc[g][3]);
    dst[1] = make_float4(acc[g][4], acc[g][5], acc[g][6], acc[g][7]);
  }
  __syncthreads();
  if (tid < G) {
    const int g = tid;
    float mm = -INFINITY;
#pragma unroll
    for (int s_ = 0; s_ < 16; ++s_) mm = fmaxf(mm, sm_m[s_][g]);
    const float mu = mm == -INFINITY ? 0.f : mm;
    float lt = 0.f;
#pragma unroll
    for (int s_ = 0; s_ < 16; ++s_) {
      const float f = __builtin_amdgcn_exp2f(sm_m[s_][g] - mu);      // empty slots (m = -inf) -> 0
      lt += sm_l[s_][g] * f;
      sm_m[s_][g] = f;                                                 // reuse as the factor table
    }
    float* w = ws + ((size_t)split * p.Hq + hk * G + g) * (D + 2);
    w[D] = mm;
    w[D + 1] = lt;
  }
  __syncthreads();
  for (int idx = tid; idx < G * D; idx += 256) {
    const int g = idx / D, d = idx - g * D;
    float ot = 0.f;
#pragma unroll
    for (int s_ = 0; s_ < 16; ++s_) ot = fmaf(sm_o[s_][g][d], sm_m[s_][g], ot);
    ws[((size_t)split * p.Hq + hk * G + g) * (D + 2) + d] = ot;
  }
}

// Matrix-core variant of the split kernel: the scores of a wave's 16 keys against all G query heads are ONE chain of four
// v_mfma_f32_16x16x32 (A = the query heads as rows, zero above G; B = K rows straight from HBM, lane (key, g) holds
// K[key][32t + 8g .. +8)), instead of G x 4 shuffle reductions per key.  Softmax runs on the 16 x 16 score tile (lane =
// one key column, 4 heads), probabilities and rescale factors go through 2 KB of LDS into the V layout (16 lanes per key
// row, as in the kernel above), where O accumulates on the VALU.  Same partial format, same merge kernel.
template <typename T, int G>
__global__ __launch_bounds__(256) void attn_decode_split_mm_kernel(AttnArgs p, DecRows rw, int n_split, float* __restrict__ ws) {
  constexpr int D = 128;
  {
    const int sc = blockIdx.z;
    p.k = rw.k[sc]; p.v = rw.v[sc]; p.Sk = rw.sk[sc]; p.q_pos0 = rw.sk[sc] - 1;
    p.q = (const uint16_t*)p.q + sc * rw.q_stride;
    ws += sc * rw.ws_stride;
  }
  __shared__ float sm_m[16][G], sm_l[16][G];
  __shared__ float sm_o[16][G][D + 4];
  __shared__ __attribute__((aligned(16))) float sm_p[4][16][8];     // [wave][key][head]
  __shared__ __attribute__((aligned(16))) float sm_a[4][8];         // [wave][head] rescale factor of this chunk
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hr = lane & 15, g4 = lane >> 4;        // score tile: column (key) hr, heads 4 g4 + i;  V layout: slot g4, dims 8 hr..
  const int hk = blockIdx.x, split = blockIdx.y;
  const int n_keys = p.q_pos0 + 1 < p.Sk ? p.q_pos0 + 1 : p.Sk;
  const int n_own = n_keys - rw.skip;                                    // (rw.skip = 0 unless the shared prefix has its own launch)
  const int ns_own = min(rw.cap, (n_own + rw.kps - 1) / rw.kps);         // splits >= ns_own are empty: (m, l, o) = (-inf, 0, 0)
  const int per = ns_own > 0 ? (n_own + ns_own - 1) / ns_own : 1;
  const int k_begin = rw.skip + (split < ns_own ? split * per : n_own);
  int k_end = k_begin + per;
  k_end = k_end < n_keys ? k_end : n_keys;
  const uint16_t* K = (const uint16_t*)p.k + (int64_t)hk * p.hsk;
  const uint16_t* V = (const uint16_t*)p.v + (int64_t)hk * p.hsk;
  const uint16_t* Kp = rw.prefix > 0 ? (const uint16_t*)rw.kp + (int64_t)hk * p.hsk : K;
  const uint16_t* Vp = rw.prefix > 0 ? (const uint16_t*)rw.vp + (int64_t)hk * p.hsk : V;
  Frag16 qf[4];                                    // A operand: head hr (zero rows above G), dims 32t + 8 g4 ..
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    qf[t].u = make_uint4(0u, 0u, 0u, 0u);
    if (hr < G) qf[t].u = *reinterpret_cast<const uint4*>((const uint16_t*)p.q + (int64_t)(hk * G + hr) * p.hsq + 32 * t + 8 * g4);
  }
  float m[4], lp[4], acc[G][8];
#pragma unroll
  for (int i = 0; i < 4; ++i) { m[i] = -INFINITY; lp[i] = 0.f; }
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[g][j] = 0.f;
  for (int kb = k_begin; kb < k_end; kb += 64) {
    const int kw0 = kb + 16 * wave;                // this wave's 16 keys of the chunk
    Frag16 kf[4];
    uint4 v4[4];
    {
      const int key = kw0 + hr;
      const int kc = key < k_end ? key : k_begin;
#pragma unroll
      for (int t = 0; t < 4; ++t) kf[t].u = *reinterpret_cast<const uint4*>((kc < rw.prefix ? Kp : K) + (int64_t)kc * p.ldk + 32 * t + 8 * g4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kv = kw0 + 4 * r + g4;
        const int vc = kv < k_end ? kv : k_begin;
        v4[r] = *reinterpret_cast<const uint4*>((vc < rw.prefix ? Vp : V) + (int64_t)vc * p.ldv + hr * 8);
      }
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t) c = Mfma16<T>::run(as_frag16<T>(qf[t]), as_frag16<T>(kf[t]), c);
    const bool key_ok = kw0 + hr < k_end;
    float alpha[4], pr[4];
    bool moved = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = key_ok && 4 * g4 + i < G;
      const float sc = ok ? c[i] * p.scale_log2 : -INFINITY;
      float mx = sc;
      mx = fmaxf(mx, __shfl_xor(mx, 8)); mx = fmaxf(mx, __shfl_xor(mx, 4));
      mx = fmaxf(mx, __shfl_xor(mx, 2)); mx = fmaxf(mx, __shfl_xor(mx, 1));
      const float mn = fmaxf(m[i], mx);
      alpha[i] = mn == -INFINITY ? 1.0f : __builtin_amdgcn_exp2f(m[i] - mn);      // m = -inf -> 0
      pr[i] = ok ? __builtin_amdgcn_exp2f(sc - mn) : 0.f;
      lp[i] = lp[i] * alpha[i] + pr[i];
      m[i] = mn;
      moved |= alpha[i] != 1.0f;
    }
    if (g4 < 2) {
      *reinterpret_cast<float4*>(&sm_p[wave][hr][4 * g4]) = make_float4(pr[0], pr[1], pr[2], pr[3]);
      if (hr == 0) *reinterpret_cast<float4*>(&sm_a[wave][4 * g4]) = make_float4(alpha[0], alpha[1], alpha[2], alpha[3]);
    }
    const bool any_moved = __any(moved);            // wave-uniform
    __syncthreads();
    if (any_moved) {
      float a8[8];
      *reinterpret_cast<float4*>(&a8[0]) = *reinterpret_cast<const float4*>(&sm_a[wave][0]);
      *reinterpret_cast<float4*>(&a8[4]) = *reinterpret_cast<const float4*>(&sm_a[wave][4]);
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[g][j] *= a8[g];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {                   // V layout: key 4r + g4 of the wave's 16, dims 8 hr ..
      float ph[8], vf[8];
      *reinterpret_cast<float4*>(&ph[0]) = *reinterpret_cast<const float4*>(&sm_p[wave][4 * r + g4][0]);
      *reinterpret_cast<float4*>(&ph[4]) = *reinterpret_cast<const float4*>(&sm_p[wave][4 * r + g4][4]);
#pragma unroll
      for (int j = 0; j < 8; ++j) vf[j] = vec_get<T>(v4[r], j);
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[g][j] = fmaf(ph[g], vf[j], acc[g][j]);
    }
    __syncthreads();                                // sm_p / sm_a are rewritten by the next chunk
  }
  // per-wave (m, l) -> the four slots of the wave (l counted once); accumulators -> slot 4 wave + g4
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float l = lp[i];
    l += __shfl_xor(l, 8); l += __shfl_xor(l, 4); l += __shfl_xor(l, 2); l += __shfl_xor(l, 1);
    if (hr == 0 && 4 * g4 + i < G) {
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) { sm_m[4 * wave + s_][4 * g4 + i] = m[i]; sm_l[4 * wave + s_][4 * g4 + i] = s_ == 0 ? l : 0.f; }
    }
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float4* dst = reinterpret_cast<float4*>(&sm_o[4 * wave + g4][g][hr * 8]);
    dst[0] = make_float4(acc[g][0], acc[g][1], acc[g][2], acc[g][3]);
    dst[1] = make_float4(acc[g][4], acc[g][5], acc[g][6], acc[g][7]);
  }
  __syncthreads();
  if (tid < G) {
    const int g = tid;
    float mm = -INFINITY;
#pragma unroll
    for (int s_ = 0; s_ < 16; ++s_) mm = fmaxf(mm, sm_m[s_][g]);
    const float mu = mm == -INFINITY ? 0.f : mm;
    float lt = 0.f;
#pragma unroll
    for (int s_ = 0; s_ < 16; ++s_) {
      const float f = __builtin_amdgcn_exp2f(sm_m[s_][g] - mu);      // empty waves (m = -inf) -> 0
      lt += sm_l[s_][g] * f;
      sm_m[s_][g] = f;                                                 // reuse as the factor table
    }
    float* w = ws + ((size_t)split * p.Hq + hk * G + g) * (D + 2);
    w[D] = mm;
    w[D + 1] = lt;
  }
  __syncthreads();
  for (int idx = tid; idx < G * D; idx += 256) {
    const int g = idx / D, d = idx - g * D;
    float ot = 0.f;
#pragma unroll
    for (int s_ = 0; s_ < 16; ++s_) ot = fmaf(sm_o[s_][g][d], sm_m[s_][g], ot);
    ws[((size_t)split * p.Hq + hk * G + g) * (D + 2) + d] = ot;
  }
}

template <typename T>
__global__ __launch_bounds__(128) void attn_decode_merge_kernel(AttnArgs p, DecRows rw, int n_split, const float* __restrict__ ws) {
  constexpr int D = 128;
  p.o = (uint16_t*)p.o + blockIdx.z * rw.o_stride;
  ws += blockIdx.z * rw.ws_stride;
  __shared__ float sm_f[1024], sm_lsum;
  const int head = blockIdx.x, d = threadIdx.x;
  // per-split (max, sum) pairs -> LDS once, then every thread reuses the rescale factors
  float mm = -INFINITY;
  for (int s = d; s < n_split; s += 128) {
    const float m = ws[((size_t)s * p.Hq + head) * (D + 2) + D];
    sm_f[s] = m;
    mm = fmaxf(mm, m);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mm = fmaxf(mm, __shfl_xor(mm, off));
  __shared__ float sm_m2[2];
  if ((d & 63) == 0) sm_m2[d >> 6] = mm;
  __syncthreads();
  mm = fmaxf(sm_m2[0], sm_m2[1]);
  const float mu = mm == -INFINITY ? 0.f : mm;
  float lpart = 0.f;
  for (int s = d; s < n_split; s += 128) {
    const float f = exp2f(sm_f[s] - mu);
    lpart += ws[((size_t)s * p.Hq + head) * (D + 2) + D + 1] * f;
    sm_f[s] = f;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) lpart += __shfl_xor(lpart, off);
  __syncthreads();
  if ((d & 63) == 0) sm_m2[d >> 6] = lpart;
  __syncthreads();
  const float lt = sm_m2[0] + sm_m2[1];
  float ot = 0.f;
#pragma unroll 8
  for (int s = 0; s < n_split; ++s) ot = fmaf(ws[((size_t)s * p.Hq + head) * (D + 2) + d], sm_f[s], ot);   // independent, coalesced loads
  (void)sm_lsum;
  T* O = (T*)p.o + (int64_t)head * p.hso;
  O[d] = from_f32<T>(lt > 0.f ? ot / lt : 0.f);
}

// V3D_ATTN64=1 opts in to attn_prefill64_kernel (one wave per SIMD, 64 queries per wave) for head dim 128 prefill; default 0 =
// attn_prefill_kernel.  Measured (S = 6794, 28/4 heads, same box): 374 us vs 328 us - halving the LDS traffic does not pay while a
// single in-order wave has to issue the softmax, the LDS reads and the MFMAs (tools/probes/attn64_probe.hip).
static int attn64_mode() {
  const char* e = getenv("V3D_ATTN64");
  return e ? atoi(e) : 0;
}

// V3D_ATTN_MFMA=16 selects attn_prefill16_kernel (v_mfma_f32_16x16x32) for head dim 128; default 32 = attn_prefill_kernel
// (v_mfma_f32_32x32x16).  A/B: profiles/r04_attn_mfma_shape.txt.
static int attn_mfma_shape() {
  const char* e = getenv("V3D_ATTN_MFMA");
  return e ? atoi(e) : 32;
}

template <typename T>
static int launch_attn(const AttnArgs& p, int D, int causal, int B, hipStream_t st) {
  if (D == 128 && p.Sq >= 64 && attn64_mode() != 0 && !p.lse) {
    const dim3 grid64(p.Hq, (p.Sq + A2_BQ - 1) / A2_BQ, B);
#define V3D_ATTN64(CC)                                                                                            \
    {                                                                                                             \
      auto k = attn_prefill64_kernel<T, CC>;                                                                      \
      static bool done = false;                                                                                   \
      if (!done) {                                                                                                \
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, A2_LDS);   \
        if (e != hipSuccess) { set_error("v3d_attention: LDS attribute: %s", hipGetErrorString(e)); return V3D_E_LAUNCH; } \
        done = true;                                                                                              \
      }                                                                                                           \
      hipLaunchKernelGGL(k, grid64, dim3(256), A2_LDS, st, p);                                                    \
    }
    if (causal) V3D_ATTN64(true) else V3D_ATTN64(false)
#undef V3D_ATTN64
    return check_launch("v3d_attention (64 queries per wave)");
  }
  dim3 grid(p.Hq, (p.Sq + AT_BQ - 1) / AT_BQ, B), block(256);
  AttnArgs pm = p;
  pm.n_qt = (p.Sq + AT_BQ - 1) / AT_BQ;
  pm.xcd_p = 0;
  if (D == 128 && attn_mfma_shape() == 16) {      // r04 A/B: the same kernel on v_mfma_f32_16x16x32 (V3D_ATTN_MFMA=16)
#define V3D_ATTN16(CC, LL)                                                                                        \
    {                                                                                                             \
      auto k = attn_prefill16_kernel<T, CC, LL>;                                                                  \
      static bool done = false;                                                                                   \
      if (!done) {                                                                                                \
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, AT_LDS);   \
        if (e != hipSuccess) { set_error("v3d_attention: LDS attribute: %s", hipGetErrorString(e)); return V3D_E_LAUNCH; } \
        done = true;                                                                                              \
      }                                                                                                           \
      hipLaunchKernelGGL(k, grid, block, AT_LDS, st, pm);                                                         \
    }
    if (causal && p.lse) V3D_ATTN16(true, true)
    else if (p.lse) V3D_ATTN16(false, true)
    else if (causal) V3D_ATTN16(true, false)
    else V3D_ATTN16(false, false)
#undef V3D_ATTN16
    return check_launch("v3d_attention (16x16x32)");
  }
  {   // XCD-aware mapping (attn_prefill_kernel): causal prefill at head dim 128 with the kv heads dividing the 8 XCDs
    const int n_kv = p.Hq / p.group;
    static const int xcd_env = getenv("V3D_ATTN_XCD") ? atoi(getenv("V3D_ATTN_XCD")) : 1;
    if (xcd_env && D == 128 && causal && n_kv <= 8 && 8 % n_kv == 0 && pm.n_qt > 1) {
      pm.xcd_p = 8 / n_kv;
      grid = dim3((unsigned)(((p.Hq * pm.n_qt + 7) / 8 * 8) * B), 1, 1);
    }
  }
#define V3D_ATTN(DD, CC, KK, LL)                                                                                    \
  {                                                                                                               \
    auto k = attn_prefill_kernel<T, DD, CC, KK, LL>;                                                              \
    static bool done = false;                                                                                     \
    if (!done) {                                                                                                  \
      hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, AT_LDS);     \
      if (e != hipSuccess) { set_error("v3d_attention: LDS attribute: %s", hipGetErrorString(e)); return V3D_E_LAUNCH; } \
      done = true;                                                                                                \
    }                                                                                                             \
    hipLaunchKernelGGL(k, grid, block, AT_LDS, st, pm);                                                           \
  }
  if (D == 96 && !causal && p.d_out <= 80 && !p.lse && !getenv("V3D_ATTN_KS6")) {
    // SigLIP: V3D_ATTN_VIT_PERSIST=1 opts in to the persistent form (bit-identical).  Default: one workgroup per item, the r03 kernel -
    // measured equal within the boxes' spread (136 vs 131-140 us per layer, profiles/r04_vit_attention.txt): the seam was not the cost.
    const char* e = getenv("V3D_ATTN_VIT_PERSIST");
    const int n_tiles = (p.Sk + AT_BKV - 1) / AT_BKV;
    if (e && atoi(e) != 0 && n_tiles >= 2 && n_tiles % 2 == 0 && p.Sk / AT_BKV >= 2) {
      const int n_items = p.Hq * B * pm.n_qt;
      int cus = 256;
      { static int cached = 0; if (!cached) { int dev = 0; hipDeviceProp_t prop; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cached = prop.multiProcessorCount; if (cached <= 0) cached = 256; } cus = cached; }
      int nwg = 2 * cus;
      if (nwg > n_items) nwg = n_items;
      auto k = attn_vit_persistent_kernel<T>;
      static bool done = false;
      if (!done) {
        hipError_t e2 = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, AT_LDS);
        if (e2 != hipSuccess) { set_error("v3d_attention: LDS attribute: %s", hipGetErrorString(e2)); return V3D_E_LAUNCH; }
        done = true;
      }
      hipLaunchKernelGGL(k, dim3(nwg), dim3(256), AT_LDS, st, pm, n_items);
      return check_launch("v3d_attention (SigLIP, persistent)");
    }
  }
  if (p.lse && D != 128) { set_error("v3d_attention_train: head dim 128 only (zero-pad narrower heads)"); return V3D_E_UNSUPPORTED; }
  if (D == 128 && causal && p.lse) V3D_ATTN(128, true, 8, true)
  else if (D == 128 && p.lse) V3D_ATTN(128, false, 8, true)
  else if (D == 128 && causal) V3D_ATTN(128, true, 8, false)
  else if (D == 128) V3D_ATTN(128, false, 8, false)
  else if (D == 96 && causal) V3D_ATTN(96, true, 6, false)
  else if (D == 96 && p.d_out <= 80 && !getenv("V3D_ATTN_KS6")) V3D_ATTN(96, false, 5, false)       // SigLIP: 72-wide heads (V3D_ATTN_KS6: A/B against the 6-step form)
  else if (D == 96) V3D_ATTN(96, false, 6, false)
  else { set_error("v3d_attention: head dim %d unsupported (128 or 96)", D); return V3D_E_UNSUPPORTED; }
#undef V3D_ATTN
  return check_launch("v3d_attention");
}

}  // namespace v3d

using namespace v3d;

static int attention_entry(const void* q, const void* k, const void* v, void* o, int dtype, int B, int Sq, int Sk,
                           int Hq, int Hkv, int D, int d_out, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                           int64_t bsq, int64_t bsk, int64_t bso, int hsq, int hsk, int hso, int causal, int q_pos0,
                           float scale, void* stream, float* lse) {
  V3D_REQUIRE(q && k && v && o, "v3d_attention: null pointer");
  V3D_REQUIRE(dtype == V3D_F16 || dtype == V3D_BF16, "v3d_attention: dtype must be f16 or bf16");
  V3D_REQUIRE(B > 0 && Sq > 0 && Sk > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0, "v3d_attention: bad shape");
  V3D_REQUIRE(d_out > 0 && d_out <= D && d_out % 8 == 0, "v3d_attention: d_out=%d", d_out);
  V3D_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0 && hsq % 8 == 0 && hsk % 8 == 0 && hso % 8 == 0 &&
                  bsq % 8 == 0 && bsk % 8 == 0 && bso % 8 == 0,
              "v3d_attention: strides must be multiples of 8 elements");
  V3D_REQUIRE(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(o), "v3d_attention: pointers must be 16-byte aligned");
  V3D_REQUIRE(!causal || q_pos0 + Sq <= Sk, "v3d_attention: causal needs q_pos0 + Sq <= Sk");
  AttnArgs p;
  p.q = q; p.k = k; p.v = v; p.o = o;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.bsq = bsq; p.bsk = bsk; p.bso = bso;
  p.hsq = hsq; p.hsk = hsk; p.hso = hso; p.Sq = Sq; p.Sk = Sk; p.Hq = Hq; p.group = Hq / Hkv;
  p.d_out = d_out; p.q_pos0 = causal ? q_pos0 : 0;
  p.scale_log2 = scale * 1.44269504088896340736f;
  p.lse = lse;
  p.xcd_p = 0; p.n_qt = 0;
  hipStream_t st = (hipStream_t)stream;
  if (Sq <= 8 && B == 1 && D == 128 && causal && !p.lse) {   // decode: stream the cache
    if (dtype == V3D_BF16) hipLaunchKernelGGL((attn_decode_kernel<bf16_t, 128>), dim3(Sq, Hq), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((attn_decode_kernel<f16_t, 128>), dim3(Sq, Hq), dim3(256), 0, st, p);
    return check_launch("v3d_attention (decode)");
  }
  return dtype == V3D_BF16 ? launch_attn<bf16_t>(p, D, causal, B, st) : launch_attn<f16_t>(p, D, causal, B, st);
}

extern "C" int v3d_attention(const void* q, const void* k, const void* v, void* o, int dtype, int B, int Sq, int Sk,
                             int Hq, int Hkv, int D, int d_out, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                             int64_t bsq, int64_t bsk, int64_t bso, int hsq, int hsk, int hso, int causal, int q_pos0,
                             float scale, void* stream) {
  return attention_entry(q, k, v, o, dtype, B, Sq, Sk, Hq, Hkv, D, d_out, ldq, ldk, ldv, ldo, bsq, bsk, bso, hsq, hsk, hso, causal, q_pos0,
                         scale, stream, nullptr);
}

// r04: the question rows of an answer batch (B questions x Sq rows at positions q_pos0 .. about ONE prefilled scene): causal attention at
// head dim 128 as v3d_attention computes it - every output bit equal - but (1) key tiles below shared_len (a multiple of 64, <= q_pos0)
// are read from the scene's cache k_shared / v_shared (same ldk / ldv / hsk, no batch stride) instead of each question's copy of them,
// so the B copies need not exist, and (2) a workgroup's query slots hold the rows of all the query heads of a kv head (attn_prefill_kernel
// <QPK>).  Keys from shared_len on come from k / v + b bsk as before (each question's own cache, rows at their absolute positions).
extern "C" int v3d_attention_shared_prefix(const void* q, const void* k, const void* v, const void* k_shared, const void* v_shared,
                                           int shared_len, void* o, int dtype, int B, int Sq, int Sk, int Hq, int Hkv, int64_t ldq,
                                           int64_t ldk, int64_t ldv, int64_t ldo, int64_t bsq, int64_t bsk, int64_t bso, int hsq, int hsk,
                                           int hso, int q_pos0, float scale, void* stream) {
  const char* who = "v3d_attention_shared_prefix";
  V3D_REQUIRE(q && k && v && o && k_shared && v_shared, "%s: null pointer", who);
  V3D_REQUIRE(dtype == V3D_F16 || dtype == V3D_BF16, "%s: dtype must be f16 or bf16", who);
  V3D_REQUIRE(B > 0 && Sq > 0 && Sk > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0, "%s: bad shape", who);
  V3D_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0 && hsq % 8 == 0 && hsk % 8 == 0 && hso % 8 == 0 &&
                  bsq % 8 == 0 && bsk % 8 == 0 && bso % 8 == 0, "%s: strides must be multiples of 8 elements", who);
  V3D_REQUIRE(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(o) && aligned16(k_shared) && aligned16(v_shared),
              "%s: pointers must be 16-byte aligned", who);
  V3D_REQUIRE(q_pos0 >= 0 && q_pos0 + Sq <= Sk, "%s: causal needs q_pos0 + Sq <= Sk", who);
  V3D_REQUIRE(shared_len >= 0 && shared_len % AT_BKV == 0 && shared_len <= q_pos0, "%s: shared_len=%d must be a multiple of %d and <= q_pos0", who,
              shared_len, AT_BKV);
  AttnArgs p{};
  p.q = q; p.k = k; p.v = v; p.o = o;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.bsq = bsq; p.bsk = bsk; p.bso = bso;
  p.hsq = hsq; p.hsk = hsk; p.hso = hso; p.Sq = Sq; p.Sk = Sk; p.Hq = Hq; p.group = Hq / Hkv;
  p.d_out = 128; p.q_pos0 = q_pos0;
  p.scale_log2 = scale * 1.44269504088896340736f;
  p.lse = nullptr; p.xcd_p = 0;
  p.k_sh = k_shared; p.v_sh = v_shared; p.sh_tiles = shared_len / AT_BKV;
  p.n_qt = (p.group * Sq + AT_BQ - 1) / AT_BQ;
  const dim3 grid(Hkv, p.n_qt, B);
  hipStream_t st = (hipStream_t)stream;
#define V3D_ATTN_SP(TT)                                                                                           \
  {                                                                                                               \
    auto kern = attn_prefill_kernel<TT, 128, true, 8, false, true>;                                               \
    static bool done = false;                                                                                     \
    if (!done) {                                                                                                  \
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, AT_LDS);  \
      if (e != hipSuccess) { set_error("%s: LDS attribute: %s", who, hipGetErrorString(e)); return V3D_E_LAUNCH; } \
      done = true;                                                                                                \
    }                                                                                                             \
    hipLaunchKernelGGL(kern, grid, dim3(256), AT_LDS, st, p);                                                     \
  }
  if (dtype == V3D_BF16) V3D_ATTN_SP(bf16_t) else V3D_ATTN_SP(f16_t)
#undef V3D_ATTN_SP
  return check_launch(who);
}

// The training forward: v3d_attention (head dim 128) that also writes the row log-sum-exp the backward recomputes the
// probabilities from (same kernel with one more store after the tile loop; outputs bit-identical to v3d_attention's).
extern "C" int v3d_attention_train(const void* q, const void* k, const void* v, void* o, float* lse, int dtype, int B, int Sq, int Sk,
                                   int Hq, int Hkv, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t bsq, int64_t bsk,
                                   int64_t bso, int hsq, int hsk, int hso, int causal, int q_pos0, float scale, void* stream) {
  V3D_REQUIRE(lse, "v3d_attention_train: null lse");
  return attention_entry(q, k, v, o, dtype, B, Sq, Sk, Hq, Hkv, 128, 128, ldq, ldk, ldv, ldo, bsq, bsk, bso, hsq, hsk, hso, causal, q_pos0, scale,
                         stream, lse);
}

extern "C" int64_t v3d_attention_decode_workspace_bytes(int Hq, int max_splits) {
  return (int64_t)max_splits * Hq * (128 + 2) * (int64_t)sizeof(float);
}

static int attention_decode_rows(const void* q, int64_t q_stride, int M, const void* const* k_caches, const void* const* v_caches,
                                 const int* Sk, void* o, int64_t o_stride, int dtype, int Hq, int Hkv, int64_t ldk, int64_t ldv,
                                 int hsq, int hsk, int hso, float scale, void* workspace, int64_t workspace_bytes, void* stream,
                                 const char* who, const void* k_prefix = nullptr, const void* v_prefix = nullptr, int prefix = 0) {
  V3D_REQUIRE(prefix >= 0 && (prefix == 0 || (k_prefix && v_prefix && aligned16(k_prefix) && aligned16(v_prefix))), "%s: bad shared prefix", who);
  V3D_REQUIRE(q && k_caches && v_caches && Sk && o && workspace, "%s: null pointer", who);
  V3D_REQUIRE(M >= 1 && M <= DEC_MAXROWS, "%s: 1 to %d scenes (got %d)", who, DEC_MAXROWS, M);
  V3D_REQUIRE(dtype == V3D_F16 || dtype == V3D_BF16, "%s: dtype must be f16 or bf16", who);
  V3D_REQUIRE(Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && Hq / Hkv <= DEC_MAXG, "%s: bad head counts", who);
  V3D_REQUIRE(ldk % 8 == 0 && ldv % 8 == 0 && hsq % 8 == 0 && hsk % 8 == 0 && q_stride % 8 == 0 && aligned16(q), "%s: alignment", who);
  DecRows rw{};
  int sk_max = 0;
  for (int m = 0; m < M; ++m) {
    V3D_REQUIRE(k_caches[m] && v_caches[m] && aligned16(k_caches[m]) && aligned16(v_caches[m]) && Sk[m] > 0, "%s: scene %d cache", who, m);
    rw.k[m] = k_caches[m]; rw.v[m] = v_caches[m]; rw.sk[m] = Sk[m];
    sk_max = Sk[m] > sk_max ? Sk[m] : sk_max;
  }
  static int kps_env = -1;
  if (kps_env < 0) { const char* e = getenv("V3D_DEC_KEYS_PER_SPLIT"); kps_env = e ? atoi(e) : 0; }
  // keys per split: enough workgroups to cover the chip, few enough that the per-split merge stays small.  The launch's
  // split dimension follows the longest scene, but every scene partitions its keys by its OWN count min(cap, ceil(n/kps))
  // (shorter scenes leave the trailing splits empty; the merge adds their exact zeros), so a scene's split boundaries - and
  // with them every bit of its output - depend neither on M nor on the other scenes' lengths (measured at
  // S = 6.8k: 128 is 10 % faster for a single scene, 256 is 20-25 % faster from four scenes on)
  const int kps = kps_env > 0 ? kps_env : 256;
  const int cap = 1024 / Hkv;                        // ~4 workgroups per CU and scene
  for (int m = 0; m < M && prefix > 0; ++m) V3D_REQUIRE(Sk[m] >= prefix, "%s: scene %d is shorter than the shared prefix", who, m);
  // r04, shared prefix on the matrix cores (V3D_DEC_PREFIX_MM=0: the r03 form, every row walks the prefix keys itself out of one copy):
  // the prefix keys are ONE launch of attn_prefill16_kernel<PART> for all rows - per kv head the M x G query heads are the "queries", a
  // workgroup takes a chunk of the prefix through LDS once for all of them, scores and P.V on v_mfma_f32_16x16x32 - and the rows' own
  // keys (question + generated tokens) stay with the split kernels (rw.skip); the merge folds both sets of partials.  A row's chunks are
  // fixed by the prefix length and its own key count, so its bits depend neither on M nor on the other rows.
  int pmm_env = 1, pmm_chunk = 256;                  // (read per call: the tests and the A/B switch it inside one process)
  if (prefix > 0) {
    const char* e = getenv("V3D_DEC_PREFIX_MM"); if (e) pmm_env = atoi(e);
    const char* c = getenv("V3D_DEC_PREFIX_CHUNK"); if (c && atoi(c) >= 64) pmm_chunk = (atoi(c) + 63) / 64 * 64;
  }
  const int G = Hq / Hkv;
  const bool pmm = prefix > 0 && pmm_env != 0 && ldk == ldv;
  const int n_chunks = pmm ? (prefix + pmm_chunk - 1) / pmm_chunk : 0;
  const int own_max = pmm ? sk_max - prefix : sk_max;
  int n_split = (own_max + kps - 1) / kps;
  if (n_split > cap) n_split = cap;
  if (n_split < 1) n_split = 1;
  const int n_slots = n_split + n_chunks;
  const int64_t ws_one = v3d_attention_decode_workspace_bytes(Hq, n_slots);
  V3D_REQUIRE(workspace_bytes >= ws_one * M, "%s: workspace too small for %d scenes x %d splits", who, M, n_slots);
  rw.q_stride = q_stride; rw.o_stride = o_stride; rw.ws_stride = ws_one / (int64_t)sizeof(float);
  rw.kps = kps; rw.cap = cap;
  rw.kp = k_prefix; rw.vp = v_prefix; rw.prefix = prefix;
  rw.skip = pmm ? prefix : 0;
  AttnArgs p{};
  p.q = q; p.o = o;
  p.ldk = ldk; p.ldv = ldv; p.hsq = hsq; p.hsk = hsk; p.hso = hso;
  p.Sq = 1; p.Hq = Hq; p.group = Hq / Hkv; p.d_out = 128;
  p.scale_log2 = scale * 1.44269504088896340736f;
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)workspace;
  if (pmm) {
    AttnArgs pp{};
    pp.q = q; pp.k = k_prefix; pp.v = v_prefix; pp.o = ws;
    pp.ldq = q_stride; pp.ldk = ldk; pp.ldv = ldv; pp.hsq = G * 128; pp.hsk = hsk;
    pp.bsq = 0; pp.bsk = (int64_t)pmm_chunk * ldk;
    pp.Sq = G * M; pp.Sk = pmm_chunk; pp.sk_total = prefix; pp.Hq = Hkv; pp.group = 1; pp.d_out = 128;
    pp.scale_log2 = p.scale_log2;
    pp.q_rpg = G; pp.part_split0 = n_split; pp.part_hq = Hq; pp.part_ws_stride = rw.ws_stride;
    pp.n_qt = (pp.Sq + AT_BQ - 1) / AT_BQ;
    const dim3 pgrid(Hkv, pp.n_qt, n_chunks);
#define V3D_PART(TT)                                                                                                             \
    {                                                                                                                            \
      auto kfn = attn_prefill16_kernel<TT, false, false, true>;                                                                  \
      static bool done = false;                                                                                                  \
      if (!done) {                                                                                                               \
        hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, AT_LDS);                \
        if (e != hipSuccess) { set_error("%s: LDS attribute: %s", who, hipGetErrorString(e)); return V3D_E_LAUNCH; }             \
        done = true;                                                                                                             \
      }                                                                                                                          \
      hipLaunchKernelGGL(kfn, pgrid, dim3(256), AT_LDS, st, pp);                                                                 \
    }
    if (dtype == V3D_BF16) V3D_PART(bf16_t) else V3D_PART(f16_t)
#undef V3D_PART
    if (int e = check_launch(who)) return e;
  }
  static int use_mm = -1;
  if (use_mm < 0) { const char* e = getenv("V3D_DEC_ATTN"); use_mm = e && e[0] == 'v' ? 0 : 1; }     // "valu" selects the shuffle-reduction kernel
#define V3D_DEC(TT, GG)                                                                                                          \
  { if (use_mm) hipLaunchKernelGGL((attn_decode_split_mm_kernel<TT, GG>), dim3(Hkv, n_split, M), dim3(256), 0, st, p, rw, n_split, ws); \
    else hipLaunchKernelGGL((attn_decode_split_kernel<TT, GG>), dim3(Hkv, n_split, M), dim3(256), 0, st, p, rw, n_split, ws); }
#define V3D_DEC_G(TT)                                                                          \
  switch (G) {                                                                                 \
    case 1: V3D_DEC(TT, 1); break; case 2: V3D_DEC(TT, 2); break; case 4: V3D_DEC(TT, 4); break; \
    case 7: V3D_DEC(TT, 7); break; case 8: V3D_DEC(TT, 8); break;                              \
    default: set_error("%s: group size %d unsupported (1,2,4,7,8)", who, G); return V3D_E_UNSUPPORTED; \
  }
  if (dtype == V3D_BF16) { V3D_DEC_G(bf16_t) } else { V3D_DEC_G(f16_t) }
#undef V3D_DEC_G
#undef V3D_DEC
  if (int e = check_launch(who)) return e;
  if (dtype == V3D_BF16) hipLaunchKernelGGL((attn_decode_merge_kernel<bf16_t>), dim3(Hq, 1, M), dim3(128), 0, st, p, rw, n_slots, ws);
  else hipLaunchKernelGGL((attn_decode_merge_kernel<f16_t>), dim3(Hq, 1, M), dim3(128), 0, st, p, rw, n_slots, ws);
  return check_launch(who);
}

extern "C" int v3d_attention_decode(const void* q, const void* k_cache, const void* v_cache, void* o, int dtype, int Sk,
                                    int Hq, int Hkv, int64_t ldk, int64_t ldv, int hsq, int hsk, int hso, float scale,
                                    void* workspace, int64_t workspace_bytes, void* stream) {
  return attention_decode_rows(q, 0, 1, &k_cache, &v_cache, &Sk, o, 0, dtype, Hq, Hkv, ldk, ldv, hsq, hsk, hso, scale, workspace,
                               workspace_bytes, stream, "v3d_attention_decode");
}

extern "C" int v3d_attention_decode_rows_prefix(const void* q, int64_t q_stride, int M, const void* k_prefix, const void* v_prefix,
                                                int prefix_len, const void* const* k_caches, const void* const* v_caches, const int* Sk,
                                                void* o, int64_t o_stride, int dtype, int Hq, int Hkv, int64_t ldk, int64_t ldv, int hsq,
                                                int hsk, int hso, float scale, void* workspace, int64_t workspace_bytes, void* stream) {
  return attention_decode_rows(q, q_stride, M, k_caches, v_caches, Sk, o, o_stride, dtype, Hq, Hkv, ldk, ldv, hsq, hsk, hso, scale,
                               workspace, workspace_bytes, stream, "v3d_attention_decode_rows_prefix", k_prefix, v_prefix, prefix_len);
}

extern "C" int v3d_attention_decode_rows(const void* q, int64_t q_stride, int M, const void* const* k_caches,
                                         const void* const* v_caches, const int* Sk, void* o, int64_t o_stride, int dtype, int Hq,
                                         int Hkv, int64_t ldk, int64_t ldv, int hsq, int hsk, int hso, float scale, void* workspace,
                                         int64_t workspace_bytes, void* stream) {
  return attention_decode_rows(q, q_stride, M, k_caches, v_caches, Sk, o, o_stride, dtype, Hq, Hkv, ldk, ldv, hsq, hsk, hso, scale,
                               workspace, workspace_bytes, stream, "v3d_attention_decode_rows");
}
