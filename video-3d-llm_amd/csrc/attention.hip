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

union Frag16 {   // 16 bytes viewed as MFMA fragment / raw words
  uint4 u;
  bf16x8 b;
  f16x8 h;
  i16x4 s[2];
};
template <typename T> __device__ __forceinline__ typename Mfma32<T>::frag as_frag(const Frag16& f);
template <> __device__ __forceinline__ bf16x8 as_frag<bf16_t>(const Frag16& f) { return f.b; }
template <> __device__ __forceinline__ f16x8 as_frag<f16_t>(const Frag16& f) { return f.h; }

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
};

__device__ __forceinline__ int kv_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

__device__ __forceinline__ void glds16a(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <typename T, int D, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_prefill_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using M = Mfma32<T>;
  constexpr int KS = D / 16;       // k-steps of QK^T
  constexpr int DT = D / 32;       // 32-wide d tiles of O^T
  constexpr int CH = D / 8;        // 16-byte chunks per K/V row actually present

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ql = lane & 31, h = lane >> 5;
  const int qt = (int)gridDim.x - 1 - (int)blockIdx.x;   // heaviest (last) query tiles first
  const int head = blockIdx.y, b = blockIdx.z;
  const int hk = head / p.group;
  const int q0 = qt * AT_BQ;

  const uint16_t* Q = (const uint16_t*)p.q + b * p.bsq + (int64_t)head * p.hsq;
  const uint16_t* K = (const uint16_t*)p.k + b * p.bsk + (int64_t)hk * p.hsk;
  const uint16_t* V = (const uint16_t*)p.v + b * p.bsk + (int64_t)hk * p.hsk;

  // ---- Q fragments: B operand, lane (q, h) holds Q[q][16ks + 8h .. +8) ----
  int qi = q0 + wave * 32 + ql;
  const int qi_ld = qi < p.Sq ? qi : p.Sq - 1;
  Frag16 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
    qf[ks].u = *reinterpret_cast<const uint4*>(Q + (int64_t)qi_ld * p.ldq + ks * 16 + h * 8);

  // ---- KV staging: one DMA = 4 rows x 256 B; wave w stages rows [16w, 16w+16) of K and of V ----
  const int n_tiles_all = (p.Sk + AT_BKV - 1) / AT_BKV;
  int n_tiles = n_tiles_all;
  if (CAUSAL) {
    const int last_key = p.q_pos0 + (q0 + AT_BQ - 1 < p.Sq ? q0 + AT_BQ - 1 : p.Sq - 1);
    const int t = last_key / AT_BKV + 1;
    n_tiles = t < n_tiles_all ? t : n_tiles_all;
  }
  // DMA source addressing: loop-invariant per-lane parts hoisted; per tile 3 VALU per 4-row piece and the
  // SGPR-base + 32-bit VGPR-offset form of global_load_lds (K and V bases are workgroup-uniform).
  const int srow = lane >> 4;                 // row within the 4-row DMA piece
  int st_row[4];
  unsigned st_ch[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    st_row[i] = wave * 16 + i * 4 + srow;
    int chunk = (lane & 15) ^ kv_swz(st_row[i]);
    chunk = chunk < CH ? chunk : CH - 1;        // D=96: the 4 pad slots are never read
    st_ch[i] = (unsigned)chunk * 16u;
  }
  const unsigned ldk_b = (unsigned)p.ldk * 2u, ldv_b = (unsigned)p.ldv * 2u;
  auto stage = [&](int buf, int t) {
    char* kb = smem + buf * 2 * AT_TILE + (wave * 16) * AT_ROW;
    char* vb = kb + AT_TILE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int key = t * AT_BKV + st_row[i];
      key = key < p.Sk ? key : p.Sk - 1;        // tail keys are masked in the scores
      glds16a((const char*)K + ((unsigned)key * ldk_b + st_ch[i]), kb + i * 4 * AT_ROW);
      glds16a((const char*)V + ((unsigned)key * ldv_b + st_ch[i]), vb + i * 4 * AT_ROW);
    }
  };

  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  // ---- per-lane LDS offsets ----
  // K row read: key row (32kt + ql), logical chunk 2ks + h
  const int k_row_off = ql * AT_ROW;
  const int k_sw = kv_swz(ql);                 // kv_swz(32kt + ql) == kv_swz(ql)
  // V transposed read: lane = 16g + 4qq + pp supplies row (base + qq), 4 columns at d = 32dt + 16(g&1) + 4pp
  const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
  const int v_chunk_lo = 2 * (g & 1) + (pp >> 1);          // + 4dt
  const int v_byte = 8 * (pp & 1);
  // rows: 16*s4 + 4h + qq (+8); (row & 15) = 4h + qq (+8)  -> swizzle independent of s4
  const int v_row0 = 4 * h + qq;
  const int v_sw0 = kv_swz(v_row0), v_sw1 = kv_swz(v_row0 + 8);
  unsigned kaddr[KS], vaddr[2 * DT];          // fragment read addresses in ring buffer 0 (flipped every tile)
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) kaddr[ks] = lds_base + k_row_off + (((2 * ks + h) ^ k_sw) << 4);
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    vaddr[2 * dt] = lds_base + AT_TILE + v_row0 * AT_ROW + v_byte + (((4 * dt + v_chunk_lo) ^ v_sw0) << 4);
    vaddr[2 * dt + 1] = lds_base + AT_TILE + (v_row0 + 8) * AT_ROW + v_byte + (((4 * dt + v_chunk_lo) ^ v_sw1) << 4);
  }

  f32x16 o[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float sc = p.scale_log2;
  const int q_pos = p.q_pos0 + qi;
  const int wave_last_pos = p.q_pos0 + q0 + wave * 32 + 31;

  stage(0, 0);
  __syncthreads();
  for (int t = 0; t < n_tiles; ++t) {
    const int cur = t & 1;
    if (t + 1 < n_tiles) stage(cur ^ 1, t + 1);
    const int kv0 = t * AT_BKV;
    const bool active = !CAUSAL || kv0 <= wave_last_pos;      // wave-uniform
    if (active) {
      // ---- S^T = K . Q^T ----
      // K fragments by inline-asm ds_read_b128 in a 2 x 4 register ring with hand-counted lgkmcnt waits (hipcc
      // serialises read -> wait -> MFMA for compiler-visible LDS reads and drains the next tile's LDS-DMA in front
      // of them).  Addresses are loop-carried registers (kaddr[ks], current ring buffer); the 32-key half is an
      // immediate offset, so a read costs no VALU.
      f32x16 s[2];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
      {
        v4i ka[4], kc[4];
#define V3D_KR(dst, ks, imm) asm volatile("ds_read_b128 %0, %1 offset:" #imm : "=v"(dst) : "v"(kaddr[ks]))
#define V3D_KW(cnt, f) asm volatile("s_waitcnt lgkmcnt(" #cnt ")" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : : "memory")
#define V3D_KM(f, i, kt, ks) s[kt] = M::run(__builtin_bit_cast(typename M::frag, f[i]), as_frag<T>(qf[ks]), s[kt])
        if constexpr (KS == 8) {
          V3D_KR(ka[0], 0, 0); V3D_KR(ka[1], 1, 0); V3D_KR(ka[2], 2, 0); V3D_KR(ka[3], 3, 0);
          V3D_KR(kc[0], 4, 0); V3D_KR(kc[1], 5, 0); V3D_KR(kc[2], 6, 0); V3D_KR(kc[3], 7, 0);
          V3D_KW(4, ka); V3D_KM(ka, 0, 0, 0); V3D_KM(ka, 1, 0, 1); V3D_KM(ka, 2, 0, 2); V3D_KM(ka, 3, 0, 3);
          V3D_KR(ka[0], 0, 8192); V3D_KR(ka[1], 1, 8192); V3D_KR(ka[2], 2, 8192); V3D_KR(ka[3], 3, 8192);
          V3D_KW(4, kc); V3D_KM(kc, 0, 0, 4); V3D_KM(kc, 1, 0, 5); V3D_KM(kc, 2, 0, 6); V3D_KM(kc, 3, 0, 7);
          V3D_KR(kc[0], 4, 8192); V3D_KR(kc[1], 5, 8192); V3D_KR(kc[2], 6, 8192); V3D_KR(kc[3], 7, 8192);
          V3D_KW(4, ka); V3D_KM(ka, 0, 1, 0); V3D_KM(ka, 1, 1, 1); V3D_KM(ka, 2, 1, 2); V3D_KM(ka, 3, 1, 3);
          V3D_KW(0, kc); V3D_KM(kc, 0, 1, 4); V3D_KM(kc, 1, 1, 5); V3D_KM(kc, 2, 1, 6); V3D_KM(kc, 3, 1, 7);
        } else {   // KS == 6 (head dim 96): 4 + 2 k-steps per key half
          V3D_KR(ka[0], 0, 0); V3D_KR(ka[1], 1, 0); V3D_KR(ka[2], 2, 0); V3D_KR(ka[3], 3, 0);
          V3D_KR(kc[0], 4, 0); V3D_KR(kc[1], 5, 0); V3D_KR(kc[2], 0, 8192); V3D_KR(kc[3], 1, 8192);
          V3D_KW(4, ka); V3D_KM(ka, 0, 0, 0); V3D_KM(ka, 1, 0, 1); V3D_KM(ka, 2, 0, 2); V3D_KM(ka, 3, 0, 3);
          V3D_KR(ka[0], 2, 8192); V3D_KR(ka[1], 3, 8192); V3D_KR(ka[2], 4, 8192); V3D_KR(ka[3], 5, 8192);
          V3D_KW(4, kc); V3D_KM(kc, 0, 0, 4); V3D_KM(kc, 1, 0, 5); V3D_KM(kc, 2, 1, 0); V3D_KM(kc, 3, 1, 1);
          V3D_KW(0, ka); V3D_KM(ka, 0, 1, 2); V3D_KM(ka, 1, 1, 3); V3D_KM(ka, 2, 1, 4); V3D_KM(ka, 3, 1, 5);
        }
#undef V3D_KR
#undef V3D_KW
#undef V3D_KM
      }
      // ---- mask + online softmax (lane = one query; keys of reg r: (r&3) + 8(r>>2) + 4h) ----
      const bool need_mask = (CAUSAL && kv0 + AT_BKV - 1 > p.q_pos0 + q0 + wave * 32) || (kv0 + AT_BKV > p.Sk);
      if (need_mask) {   // wave-uniform; only diagonal / tail tiles pay for it (compare + select per score)
        int last = p.Sk - 1;
        if (CAUSAL) last = q_pos < last ? q_pos : last;
        const int limit = last - kv0 - 4 * h;            // visible iff tile-local key offset <= limit
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            s[kt][r] = (kt * 32 + (r & 3) + 8 * (r >> 2)) > limit ? -INFINITY : s[kt][r];
      }
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m_run, mx);
      const float m_use = m_new == -INFINITY ? 0.f : m_new;
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_use) * sc);   // v_exp_f32; m_run = -inf -> 0
      const float mb = m_use * sc;
      float ls = 0.f;
      Frag16 pf[4];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          float e[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            e[j] = __builtin_amdgcn_exp2f(fmaf(s[kt][8 * s2 + j], sc, -mb));
            ls += e[j];
          }
          pf[2 * kt + s2].u = make_uint4(pack2<T>(e[0], e[1]), pack2<T>(e[2], e[3]), pack2<T>(e[4], e[5]), pack2<T>(e[6], e[7]));
        }
      l_run = l_run * alpha + ls;
      m_run = m_new;
      if (__any(alpha != 1.0f)) {     // wave-uniform: once the running max has settled the rescale is skipped
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
      }
      // ---- O^T += V^T . P^T ----  (V^T fragments by asm ds_read_b64_tr_b16, ring of two d-tiles; the 16-key
      //                              step is an immediate offset, vaddr[2*dt + half] the loop-carried address)
      {
        v2i va[8], vc[8];          // [2*s4 + half]
#define V3D_VR(f, dt) \
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f[0]) : "v"(vaddr[2 * (dt)])); \
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f[1]) : "v"(vaddr[2 * (dt) + 1])); \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(f[2]) : "v"(vaddr[2 * (dt)])); \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(f[3]) : "v"(vaddr[2 * (dt) + 1])); \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8192" : "=v"(f[4]) : "v"(vaddr[2 * (dt)])); \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8192" : "=v"(f[5]) : "v"(vaddr[2 * (dt) + 1])); \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:12288" : "=v"(f[6]) : "v"(vaddr[2 * (dt)])); \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:12288" : "=v"(f[7]) : "v"(vaddr[2 * (dt) + 1]));
        auto mmav = [&](const v2i* f, int dt) {
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            const v4i vf = {f[2 * s4][0], f[2 * s4][1], f[2 * s4 + 1][0], f[2 * s4 + 1][1]};
            o[dt] = M::run(__builtin_bit_cast(typename M::frag, vf), as_frag<T>(pf[s4]), o[dt]);
          }
        };
#define V3D_VW(cnt, f) asm volatile("s_waitcnt lgkmcnt(" #cnt ")" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : : "memory")
        V3D_VR(va, 0) V3D_VR(vc, 1)
        V3D_VW(8, va); mmav(va, 0); V3D_VR(va, 2)
        if constexpr (DT == 4) {
          V3D_VW(8, vc); mmav(vc, 1); V3D_VR(vc, 3)
          V3D_VW(8, va); mmav(va, 2);
          V3D_VW(0, vc); mmav(vc, 3);
        } else {
          V3D_VW(8, vc); mmav(vc, 1);
          V3D_VW(0, va); mmav(va, 2);
        }
#undef V3D_VR
#undef V3D_VW
      }
    }
    {   // flip the loop-carried fragment addresses to the other ring buffer (one VALU add each)
      const int delta = cur ? -2 * AT_TILE : 2 * AT_TILE;
#pragma unroll
      for (int i = 0; i < KS; ++i) kaddr[i] += delta;
#pragma unroll
      for (int i = 0; i < 2 * DT; ++i) vaddr[i] += delta;
    }
    __syncthreads();   // next tile landed; everyone is done with `cur`
  }

  // ---- normalise, transpose through LDS, store whole rows ----
  float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
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
  uint16_t* O = (uint16_t*)p.o + b * p.bso + (int64_t)head * p.hso;
  constexpr int OCH = D / 8;                   // 16-byte chunks per row
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

template <typename T, int G>
__global__ __launch_bounds__(256) void attn_decode_split_kernel(AttnArgs p, int n_split, float* __restrict__ ws) {
  constexpr int D = 128;
  __shared__ float sm_m[16][G], sm_l[16][G];
  __shared__ float sm_o[16][G][D + 4];
  const int tid = threadIdx.x;
  const int slot = tid >> 4, cl = tid & 15;       // 16 key slots, 16 lanes per key row
  const int hk = blockIdx.x, split = blockIdx.y;
  const int n_keys = p.q_pos0 + 1 < p.Sk ? p.q_pos0 + 1 : p.Sk;
  const int per = (n_keys + n_split - 1) / n_split;
  const int k_begin = split * per;
  int k_end = k_begin + per;
  k_end = k_end < n_keys ? k_end : n_keys;
  const uint16_t* K = (const uint16_t*)p.k + (int64_t)hk * p.hsk;
  const uint16_t* V = (const uint16_t*)p.v + (int64_t)hk * p.hsk;
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
      k4[it] = *reinterpret_cast<const uint4*>(K + (int64_t)kc * p.ldk + cl * 8);
      v4[it] = *reinterpret_cast<const uint4*>(V + (int64_t)kc * p.ldv + cl * 8);
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

template <typename T>
__global__ __launch_bounds__(128) void attn_decode_merge_kernel(AttnArgs p, int n_split, const float* __restrict__ ws) {
  constexpr int D = 128;
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

template <typename T>
static int launch_attn(const AttnArgs& p, int D, int causal, int B, hipStream_t st) {
  const dim3 grid((p.Sq + AT_BQ - 1) / AT_BQ, p.Hq, B), block(256);
#define V3D_ATTN(DD, CC)                                                                                          \
  {                                                                                                               \
    auto k = attn_prefill_kernel<T, DD, CC>;                                                                      \
    static bool done = false;                                                                                     \
    if (!done) {                                                                                                  \
      hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, AT_LDS);     \
      if (e != hipSuccess) { set_error("v3d_attention: LDS attribute: %s", hipGetErrorString(e)); return V3D_E_LAUNCH; } \
      done = true;                                                                                                \
    }                                                                                                             \
    hipLaunchKernelGGL(k, grid, block, AT_LDS, st, p);                                                            \
  }
  if (D == 128 && causal) V3D_ATTN(128, true)
  else if (D == 128) V3D_ATTN(128, false)
  else if (D == 96 && causal) V3D_ATTN(96, true)
  else if (D == 96) V3D_ATTN(96, false)
  else { set_error("v3d_attention: head dim %d unsupported (128 or 96)", D); return V3D_E_UNSUPPORTED; }
#undef V3D_ATTN
  return check_launch("v3d_attention");
}

}  // namespace v3d

using namespace v3d;

extern "C" int v3d_attention(const void* q, const void* k, const void* v, void* o, int dtype, int B, int Sq, int Sk,
                             int Hq, int Hkv, int D, int d_out, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                             int64_t bsq, int64_t bsk, int64_t bso, int hsq, int hsk, int hso, int causal, int q_pos0,
                             float scale, void* stream) {
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
  hipStream_t st = (hipStream_t)stream;
  if (Sq <= 8 && B == 1 && D == 128 && causal) {   // decode: stream the cache
    if (dtype == V3D_BF16) hipLaunchKernelGGL((attn_decode_kernel<bf16_t, 128>), dim3(Sq, Hq), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((attn_decode_kernel<f16_t, 128>), dim3(Sq, Hq), dim3(256), 0, st, p);
    return check_launch("v3d_attention (decode)");
  }
  return dtype == V3D_BF16 ? launch_attn<bf16_t>(p, D, causal, B, st) : launch_attn<f16_t>(p, D, causal, B, st);
}


extern "C" int64_t v3d_attention_decode_workspace_bytes(int Hq, int max_splits) {
  return (int64_t)max_splits * Hq * (128 + 2) * (int64_t)sizeof(float);
}

extern "C" int v3d_attention_decode(const void* q, const void* k_cache, const void* v_cache, void* o, int dtype, int Sk,
                                    int Hq, int Hkv, int64_t ldk, int64_t ldv, int hsq, int hsk, int hso, float scale,
                                    void* workspace, int64_t workspace_bytes, void* stream) {
  V3D_REQUIRE(q && k_cache && v_cache && o && workspace, "v3d_attention_decode: null pointer");
  V3D_REQUIRE(dtype == V3D_F16 || dtype == V3D_BF16, "v3d_attention_decode: dtype must be f16 or bf16");
  V3D_REQUIRE(Sk > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && Hq / Hkv <= DEC_MAXG, "v3d_attention_decode: bad head counts");
  V3D_REQUIRE(ldk % 8 == 0 && ldv % 8 == 0 && hsq % 8 == 0 && hsk % 8 == 0 && aligned16(q) && aligned16(k_cache) && aligned16(v_cache),
              "v3d_attention_decode: alignment");
  static int kps = 0;
  if (!kps) { const char* e = getenv("V3D_DEC_KEYS_PER_SPLIT"); kps = e ? atoi(e) : 64; }
  int n_split = (Sk + kps - 1) / kps;                // >= 64 keys per split: 4 x 107 workgroups at S = 6.8k
  const int cap = 1024 / Hkv;                        // ~4 workgroups per CU
  if (n_split > cap) n_split = cap;
  if (n_split < 1) n_split = 1;
  V3D_REQUIRE(workspace_bytes >= v3d_attention_decode_workspace_bytes(Hq, n_split), "v3d_attention_decode: workspace too small for %d splits", n_split);
  AttnArgs p{};
  p.q = q; p.k = k_cache; p.v = v_cache; p.o = o;
  p.ldk = ldk; p.ldv = ldv; p.hsq = hsq; p.hsk = hsk; p.hso = hso;
  p.Sq = 1; p.Sk = Sk; p.Hq = Hq; p.group = Hq / Hkv; p.d_out = 128; p.q_pos0 = Sk - 1;
  p.scale_log2 = scale * 1.44269504088896340736f;
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)workspace;
  const int G = Hq / Hkv;
#define V3D_DEC(TT, GG) hipLaunchKernelGGL((attn_decode_split_kernel<TT, GG>), dim3(Hkv, n_split), dim3(256), 0, st, p, n_split, ws)
#define V3D_DEC_G(TT)                                                                          \
  switch (G) {                                                                                 \
    case 1: V3D_DEC(TT, 1); break; case 2: V3D_DEC(TT, 2); break; case 4: V3D_DEC(TT, 4); break; \
    case 7: V3D_DEC(TT, 7); break; case 8: V3D_DEC(TT, 8); break;                              \
    default: set_error("v3d_attention_decode: group size %d unsupported (1,2,4,7,8)", G); return V3D_E_UNSUPPORTED; \
  }
  if (dtype == V3D_BF16) { V3D_DEC_G(bf16_t) } else { V3D_DEC_G(f16_t) }
#undef V3D_DEC_G
#undef V3D_DEC
  if (int e = check_launch("v3d_attention_decode (split)")) return e;
  if (dtype == V3D_BF16) hipLaunchKernelGGL((attn_decode_merge_kernel<bf16_t>), dim3(Hq), dim3(128), 0, st, p, n_split, ws);
  else hipLaunchKernelGGL((attn_decode_merge_kernel<f16_t>), dim3(Hq), dim3(128), 0, st, p, n_split, ws);
  return check_launch("v3d_attention_decode (merge)");
}
