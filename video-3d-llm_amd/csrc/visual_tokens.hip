// 3-D sinusoidal position encoding and the fused visual-token kernel (K5-K9).
//
//   v3d_sin3d_table_build : PositionEmbeddingSine3D for every integer voxel id, once per model
//   v3d_sin3d_pe          : the same module for arbitrary coordinates (on-the-fly trig)
//   v3d_visual_tokens     : bilinear 27x27->14x14 pool + PE(voxel id) add + newline insertion,
//                           one HBM pass over the projector output  (the north-star kernel)
//   v3d_embed_gather      : text-token embedding rows
//
// All of it is HBM-bound streaming: 16-byte loads/stores per lane, no LDS, no MFMA.
// Compiled with -ffp-contract=off (the bilinear blend must round like the reference's f32 ops;
// the one fused multiply-add the reference does use is written explicitly as fmaf).
#include "v3d_common.h"

namespace v3d {

__host__ __device__ inline int table_row_elems(int nf, int vec) { return ((nf + vec - 1) / vec) * vec + vec; }

// position_encoding.py:27-40: feature j of an axis is sin(p/dim_t[j]) for even j, cos for odd j
// (the odd-num_feats branch pads one column and drops it again: same rule).
// p/dim_t is an IEEE f32 division as in the reference; the trig is evaluated in double and
// rounded once to f32, i.e. the correctly rounded f32 value (torch's vectorised sin/cos is
// within 1 ulp of it).
__device__ __forceinline__ float pe_value(float p, float dim_t, int j) {
  const float arg = __fdiv_rn(p, dim_t);
  return (float)((j & 1) ? cos((double)arg) : sin((double)arg));
}

template <typename T>
__global__ __launch_bounds__(256) void sin3d_table_kernel(const float* __restrict__ dim_t, int nf, int n_ids,
                                                          int row_elems, T* __restrict__ table,
                                                          float* __restrict__ table_f32) {
  constexpr int VEC = 16 / sizeof(T);
  const int64_t total = (int64_t)n_ids * nf;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int id = (int)(i / nf), j = (int)(i - (int64_t)id * nf);
    const float val = pe_value((float)id, dim_t[j], j);
    if (table_f32) table_f32[i] = val;
    if (table) {
      const T t = from_f32<T>(val);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const int shift = (a * nf) % VEC;
        table[((size_t)a * n_ids + id) * row_elems + shift + j] = t;
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void sin3d_pe_kernel(const T* __restrict__ xyz, int64_t N,
                                                       const float* __restrict__ dim_t, int E, int nf,
                                                       T* __restrict__ out) {
  const int64_t total = N * E;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t tok = i / E;
    const int c = (int)(i - tok * E);
    const int a = c / nf;
    float val = 0.0f;
    if (a < 3) {
      const int j = c - a * nf;
      val = pe_value(to_f32(xyz[tok * 3 + a]), dim_t[j], j);
    }
    out[i] = from_f32<T>(val);
  }
}

// ----------------------------------------------------------------------------------------
// Fused visual-token kernel.  One workgroup = one OUTPUT token row of C channels; lane = one
// 16-byte vector of channels.  Everything that depends only on the token (taps, blend weights,
// voxel ids, table rows) is workgroup-uniform and lives in SGPRs.
//
//   POOL   : x = h0*(w0*x00 + w1*x01) + h1*(w0*x10 + w1*x11) in f32, rounded to T
//            (ATen upsample_bilinear2d, align_corners=False; source index = fma(scale, o+.5, -.5))
//   PE     : x = T(x + table[axis][id][channel])    (llava_arch.py:515, add in the model dtype)
//   NEWLINE: row (v, oh, n) of the output is `newline`
//
// Per output token the kernel reads 4 input rows (2 when a tap pair collapses at an edge), one
// table row (L2-resident: 3 x n_ids x ~2.4 KB) and writes one row: algorithmic HBM traffic is
// feat + out, every input row being touched exactly once except tap row/col 13 (twice).
// ----------------------------------------------------------------------------------------
struct Taps { int i0, i1; float l0, l1; };

__device__ __forceinline__ Taps bilinear_tap(int o, int n_in, int n_out) {
  const float scale = __fdiv_rn((float)n_in, (float)n_out);
  float src = fmaf(scale, (float)o + 0.5f, -0.5f);   // single rounding, as ATen's builds do
  src = src < 0.0f ? 0.0f : src;
  Taps t;
  t.i0 = (int)src;
  t.i1 = t.i0 + (t.i0 < n_in - 1 ? 1 : 0);
  t.l1 = __fsub_rn(src, (float)t.i0);
  t.l0 = __fsub_rn(1.0f, t.l1);
  return t;
}

template <typename T, bool POOL, bool PE>
__global__ __launch_bounds__(512) void visual_tokens_kernel(const T* __restrict__ feat, const int32_t* __restrict__ ids,
                                                            const T* __restrict__ table, int n_ids, int row_elems,
                                                            const T* __restrict__ newline, T* __restrict__ out,
                                                            int64_t out_stride, int side, int n, int C, int newline_on) {
  constexpr int VEC = 16 / sizeof(T);
  const int vpt = C / VEC;                      // vectors per token
  const int cols = newline_on ? n + 1 : n;      // output rows per grid row
  const int tok = blockIdx.x;
  const int v = tok / (n * cols);
  const int rem = tok - v * (n * cols);
  const int oh = rem / cols, ow = rem - oh * cols;
  uint4* dst = reinterpret_cast<uint4*>(out + (size_t)tok * out_stride);

  if (ow == n) {  // newline row
    const uint4* nl = reinterpret_cast<const uint4*>(newline);
    for (int k = threadIdx.x; k < vpt; k += blockDim.x) dst[k] = nl[k];
    return;
  }

  const int in_side = POOL ? side : n;
  const T* frame = feat + (size_t)v * in_side * in_side * C;
  Taps th, tw;
  const uint4 *r00, *r01, *r10, *r11;
  if (POOL) {
    th = bilinear_tap(oh, side, n);
    tw = bilinear_tap(ow, side, n);
    r00 = reinterpret_cast<const uint4*>(frame + ((size_t)th.i0 * side + tw.i0) * C);
    r01 = reinterpret_cast<const uint4*>(frame + ((size_t)th.i0 * side + tw.i1) * C);
    r10 = reinterpret_cast<const uint4*>(frame + ((size_t)th.i1 * side + tw.i0) * C);
    r11 = reinterpret_cast<const uint4*>(frame + ((size_t)th.i1 * side + tw.i1) * C);
  } else {
    r00 = reinterpret_cast<const uint4*>(frame + ((size_t)oh * n + ow) * C);
    r01 = r10 = r11 = r00;
  }

  const int nf = C / 3;
  const uint4* trow[3] = {nullptr, nullptr, nullptr};
  int tbase[3] = {0, 0, 0};
  if (PE) {
    const int32_t* id3 = ids + ((size_t)(v * n + oh) * n + ow) * 3;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      int id = id3[a];
      id = id < 0 ? 0 : (id >= n_ids ? n_ids - 1 : id);   // ids come from the clamped voxeliser; stay in bounds regardless
      trow[a] = reinterpret_cast<const uint4*>(table + ((size_t)a * n_ids + id) * row_elems);
      tbase[a] = (a * nf) / VEC;                          // first vector of the token row that overlaps axis a
    }
  }

  for (int k = threadIdx.x; k < vpt; k += blockDim.x) {
    float x[VEC];
    if (POOL) {
      const uint4 a = r00[k], b = r01[k], c = r10[k], d = r11[k];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float top = __fadd_rn(__fmul_rn(tw.l0, vec_get<T>(a, j)), __fmul_rn(tw.l1, vec_get<T>(b, j)));
        const float bot = __fadd_rn(__fmul_rn(tw.l0, vec_get<T>(c, j)), __fmul_rn(tw.l1, vec_get<T>(d, j)));
        x[j] = round_to<T>(__fadd_rn(__fmul_rn(th.l0, top), __fmul_rn(th.l1, bot)));
      }
    } else {
      const uint4 a = r00[k];
#pragma unroll
      for (int j = 0; j < VEC; ++j) x[j] = vec_get<T>(a, j);
    }
    if (PE) {
      const int c0 = k * VEC;
      const int a0 = c0 / nf, a1 = (c0 + VEC - 1) / nf;
      uint4 p = make_uint4(0, 0, 0, 0);
      if (a0 < 3) {
        p = trow[a0 == 0 ? 0 : (a0 == 1 ? 1 : 2)][k - tbase[a0 == 0 ? 0 : (a0 == 1 ? 1 : 2)]];
        if (a1 != a0 && a1 < 3) {
          const uint4 q = trow[a1 == 1 ? 1 : 2][k - tbase[a1 == 1 ? 1 : 2]];
          p.x |= q.x; p.y |= q.y; p.z |= q.z; p.w |= q.w;   // disjoint support: padding is zero bits
        }
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) x[j] = __fadd_rn(x[j], vec_get<T>(p, j));
    }
    dst[k] = vec_pack<T>(x);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void embed_gather_kernel(const T* __restrict__ table, int64_t vocab, int C,
                                                           const int64_t* __restrict__ ids, T* __restrict__ out,
                                                           int64_t out_stride) {
  constexpr int VEC = 16 / sizeof(T);
  const int64_t row = blockIdx.x;
  int64_t id = ids[row];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const uint4* src = reinterpret_cast<const uint4*>(table + id * C);
  uint4* dst = reinterpret_cast<uint4*>(out + row * out_stride);
  for (int k = threadIdx.x; k < C / VEC; k += blockDim.x) dst[k] = src[k];
}

}  // namespace v3d

using namespace v3d;

extern "C" int64_t v3d_sin3d_table_row_elems(int embedding_size, int dtype) {
  const int vec = dtype == V3D_F32 ? 4 : 8;
  return table_row_elems(embedding_size / 3, vec);
}

extern "C" int v3d_sin3d_table_build(const float* dim_t, int embedding_size, int n_ids, int dtype, void* table,
                                     float* table_f32, void* stream) {
  V3D_REQUIRE(dim_t && (table || table_f32), "v3d_sin3d_table_build: null pointer");
  V3D_REQUIRE(embedding_size >= 3 && n_ids > 0, "v3d_sin3d_table_build: bad shape E=%d n_ids=%d", embedding_size, n_ids);
  const int nf = embedding_size / 3;
  hipStream_t st = (hipStream_t)stream;
  const int row = (int)v3d_sin3d_table_row_elems(embedding_size, dtype);
  if (table) {
    const size_t esz = dtype == V3D_F32 ? 4 : 2;
    hipError_t e = hipMemsetAsync(table, 0, (size_t)3 * n_ids * row * esz, st);
    if (e != hipSuccess) { set_error("v3d_sin3d_table_build: memset: %s", hipGetErrorString(e)); return V3D_E_LAUNCH; }
  }
  int64_t blocks = ((int64_t)n_ids * nf + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  V3D_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(sin3d_table_kernel<T>, dim3((int)blocks), dim3(256), 0, st, dim_t, nf,
                                               n_ids, row, (T*)table, table_f32));
  return check_launch("v3d_sin3d_table_build");
}

extern "C" int v3d_sin3d_pe(const void* xyz, int dtype, int64_t N, const float* dim_t, int embedding_size, void* out,
                            void* stream) {
  V3D_REQUIRE(xyz && dim_t && out, "v3d_sin3d_pe: null pointer");
  V3D_REQUIRE(N >= 0 && embedding_size >= 3, "v3d_sin3d_pe: bad shape");
  if (N == 0) return V3D_OK;
  int64_t blocks = (N * embedding_size + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  V3D_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(sin3d_pe_kernel<T>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream,
                                               (const T*)xyz, N, dim_t, embedding_size, embedding_size / 3, (T*)out));
  return check_launch("v3d_sin3d_pe");
}

extern "C" int v3d_visual_tokens(const void* feat, const int32_t* ids, const void* table, int n_ids,
                                 const void* newline, void* out, int64_t out_stride, int dtype, int V, int side, int n,
                                 int C, int flags, void* stream) {
  const bool pool = flags & V3D_VT_POOL, pe = flags & V3D_VT_PE, nl = flags & V3D_VT_NEWLINE;
  V3D_REQUIRE(feat && out, "v3d_visual_tokens: null pointer");
  V3D_REQUIRE(!pe || (ids && table && n_ids > 0), "v3d_visual_tokens: PE needs ids and a table");
  V3D_REQUIRE(!nl || newline, "v3d_visual_tokens: NEWLINE needs a newline row");
  V3D_REQUIRE(V > 0 && n > 0 && C > 0 && (!pool || side >= n), "v3d_visual_tokens: bad shape V=%d side=%d n=%d C=%d", V, side, n, C);
  const int vec = dtype == V3D_F32 ? 4 : 8;
  V3D_REQUIRE(C % vec == 0, "v3d_visual_tokens: C=%d must be a multiple of %d", C, vec);
  V3D_REQUIRE(!pe || C / 3 >= vec, "v3d_visual_tokens: C=%d too narrow for the table layout", C);
  V3D_REQUIRE(out_stride >= C && out_stride % vec == 0, "v3d_visual_tokens: out_stride=%lld", (long long)out_stride);
  V3D_REQUIRE(aligned16(feat) && aligned16(out) && (!pe || aligned16(table)) && (!nl || aligned16(newline)),
              "v3d_visual_tokens: pointers must be 16-byte aligned");
  const int64_t tokens = (int64_t)V * n * (nl ? n + 1 : n);
  V3D_REQUIRE(tokens < (1ll << 31), "v3d_visual_tokens: too many tokens");
  const int vpt = C / vec;
  int threads = ((vpt + 63) / 64) * 64;
  if (threads > 512) threads = 512;
  const int row = (int)v3d_sin3d_table_row_elems(C, dtype);
  hipStream_t st = (hipStream_t)stream;
#define V3D_VT_LAUNCH(P, E)                                                                                       \
  hipLaunchKernelGGL((visual_tokens_kernel<T, P, E>), dim3((unsigned)tokens), dim3(threads), 0, st, (const T*)feat, \
                     ids, (const T*)table, n_ids, row, (const T*)newline, (T*)out, out_stride, side, n, C, nl ? 1 : 0)
  V3D_DISPATCH_DTYPE(dtype, {
    if (pool && pe) V3D_VT_LAUNCH(true, true);
    else if (pool) V3D_VT_LAUNCH(true, false);
    else if (pe) V3D_VT_LAUNCH(false, true);
    else V3D_VT_LAUNCH(false, false);
  });
#undef V3D_VT_LAUNCH
  return check_launch("v3d_visual_tokens");
}

extern "C" int v3d_embed_gather(const void* table, int64_t vocab, int C, const int64_t* ids, int64_t n, void* out,
                                int64_t out_stride, int dtype, void* stream) {
  V3D_REQUIRE(table && ids && out, "v3d_embed_gather: null pointer");
  V3D_REQUIRE(vocab > 0 && C > 0 && n >= 0, "v3d_embed_gather: bad shape");
  if (n == 0) return V3D_OK;
  const int vec = dtype == V3D_F32 ? 4 : 8;
  V3D_REQUIRE(C % vec == 0 && out_stride % vec == 0 && out_stride >= C, "v3d_embed_gather: C/out_stride not a multiple of %d", vec);
  V3D_REQUIRE(aligned16(table) && aligned16(out), "v3d_embed_gather: pointers must be 16-byte aligned");
  V3D_REQUIRE(n < (1ll << 31), "v3d_embed_gather: too many rows");
  V3D_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(embed_gather_kernel<T>, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream,
                                               (const T*)table, vocab, C, ids, (T*)out, out_stride));
  return check_launch("v3d_embed_gather");
}
