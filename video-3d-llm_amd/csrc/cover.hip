// a3 on the device: greedy max-coverage frame selection (scripts/3d/preprocessing/max_coverage_sampling.py:44-94).
//
// Integer / bit work, HBM- and atomic-bound - no matrix cores.  Everything is expressed over the SCENE voxel set:
// only |frame & scene| and |used & frame & scene| enter the reference's gain, so a frame is a bitset over the m scene
// voxels and the covered set another one:
//   1. hash the m scene voxel keys (21 bits per axis packed into 63) -> open-addressing table key -> index;
//   2. one thread per frame point: key -> probe -> set bit index in the frame's bitset (atomicOr, skipped when set);
//   3. up to max_frames rounds, two launches each: gain[i] = popcount(B_i & ~U) for every remaining frame (one
//      workgroup per frame), then one workgroup picks the largest gain (lowest frame position on ties - the declared
//      replacement of the reference's unseeded random.choice, :84), records it and ORs its bitset into U;
//   4. totals: |union of all frames & scene| and |U|.
// No host synchronisation inside; results stay on the device until the caller reads them.
#include "v3d_common.h"

namespace v3d {

constexpr uint64_t COVER_EMPTY = ~0ull;

__device__ __forceinline__ bool cover_pack(int x, int y, int z, uint64_t& key) {
  const int o = 1 << 20;
  const unsigned ux = (unsigned)(x + o), uy = (unsigned)(y + o), uz = (unsigned)(z + o);
  if ((ux | uy | uz) >> 21) return false;                   // outside the packable cube: can never equal a scene key
  key = ((uint64_t)ux << 42) | ((uint64_t)uy << 21) | (uint64_t)uz;
  return true;
}
__device__ __forceinline__ unsigned cover_slot(uint64_t key, int log2cap) {
  return (unsigned)((key * 0x9E3779B97F4A7C15ull) >> (64 - log2cap));
}

__global__ __launch_bounds__(256) void cover_fill_kernel(uint64_t* __restrict__ tkeys, int64_t cap, unsigned* __restrict__ words, int64_t n_words) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < cap) tkeys[i] = COVER_EMPTY;
  if (i < n_words) words[i] = 0u;
}

// scene key j -> table (first insertion of a key wins; a repeated key simply never gets its own bit set)
__global__ __launch_bounds__(256) void cover_insert_kernel(const int32_t* __restrict__ scene, int64_t m, uint64_t* __restrict__ tkeys,
                                                           int32_t* __restrict__ tvals, int log2cap, int* __restrict__ err) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  uint64_t key;
  if (!cover_pack(scene[3 * j], scene[3 * j + 1], scene[3 * j + 2], key)) { atomicExch(err, 1); return; }
  const unsigned mask = (1u << log2cap) - 1u;
  unsigned s = cover_slot(key, log2cap);
  for (;;) {
    const unsigned long long prev = atomicCAS((unsigned long long*)&tkeys[s], (unsigned long long)COVER_EMPTY, (unsigned long long)key);
    if (prev == COVER_EMPTY) { tvals[s] = (int32_t)j; return; }
    if (prev == key) return;                                 // duplicate scene voxel
    s = (s + 1) & mask;
  }
}

// frame points -> bits.  tvals of a slot may be written after its key becomes visible (another thread's insert), but the
// insert kernel has completed before this one starts (stream order), so plain loads are enough.
__global__ __launch_bounds__(256) void cover_mark_kernel(const int32_t* __restrict__ keys, int64_t pts_per_frame, int64_t n_points,
                                                         const uint64_t* __restrict__ tkeys, const int32_t* __restrict__ tvals, int log2cap,
                                                         unsigned* __restrict__ bits, int64_t words_per_frame) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= n_points) return;
  uint64_t key;
  if (!cover_pack(keys[3 * p], keys[3 * p + 1], keys[3 * p + 2], key)) return;
  const unsigned mask = (1u << log2cap) - 1u;
  unsigned s = cover_slot(key, log2cap);
  for (;;) {
    const uint64_t k = tkeys[s];
    if (k == COVER_EMPTY) return;                            // not a scene voxel
    if (k == key) break;
    s = (s + 1) & mask;
  }
  const int idx = tvals[s];
  const int64_t frame = p / pts_per_frame;
  unsigned* w = bits + frame * words_per_frame + (idx >> 5);
  const unsigned bit = 1u << (idx & 31);
  if (!(*w & bit)) atomicOr(w, bit);
}

__device__ __forceinline__ long long cover_block_sum(long long v, long long* sm) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  const long long t = sm[0] + sm[1] + sm[2] + sm[3];
  __syncthreads();
  return t;
}

// gain[i] = |B_i & ~U| for frames not yet chosen, -1 for chosen ones
__global__ __launch_bounds__(256) void cover_gain_kernel(const unsigned* __restrict__ bits, const unsigned* __restrict__ used,
                                                         const int* __restrict__ chosen, int64_t words_per_frame, long long* __restrict__ gain) {
  __shared__ long long sm[4];
  const int f = blockIdx.x;
  if (chosen[f]) { if (threadIdx.x == 0) gain[f] = -1; return; }
  const unsigned* b = bits + (int64_t)f * words_per_frame;
  long long c = 0;
  for (int64_t w = threadIdx.x; w < words_per_frame; w += 256) c += __popc(b[w] & ~used[w]);
  c = cover_block_sum(c, sm);
  if (threadIdx.x == 0) gain[f] = c;
}

// one workgroup: argmax (lowest frame position on ties), record, U |= B_best
__global__ __launch_bounds__(256) void cover_pick_kernel(const unsigned* __restrict__ bits, unsigned* __restrict__ used, int* __restrict__ chosen,
                                                         const long long* __restrict__ gain, int n_frames, int64_t words_per_frame, int round,
                                                         int32_t* __restrict__ sel, int64_t* __restrict__ gain_out, int32_t* __restrict__ n_sel) {
  __shared__ long long sv[4];
  __shared__ int si[4];
  long long best = -1;
  int bi = 0x7fffffff;
  for (int i = threadIdx.x; i < n_frames; i += 256) {
    const long long g = gain[i];
    if (g > best || (g == best && g >= 0 && i < bi)) { best = g; bi = i; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const long long ob = __shfl_xor(best, off);
    const int oi = __shfl_xor(bi, off);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; }
  __syncthreads();
  best = sv[0]; bi = si[0];
  for (int w = 1; w < 4; ++w)
    if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
  if (best < 0) return;                                      // no frame left (uniform over the workgroup)
  if (threadIdx.x == 0) { sel[round] = bi; gain_out[round] = best; chosen[bi] = 1; *n_sel = round + 1; }
  const unsigned* b = bits + (int64_t)bi * words_per_frame;
  for (int64_t w = threadIdx.x; w < words_per_frame; w += 256) used[w] |= b[w];
}

// totals[0] = |union_i B_i|, totals[1] = |U|
__global__ __launch_bounds__(256) void cover_totals_kernel(const unsigned* __restrict__ bits, const unsigned* __restrict__ used, int n_frames,
                                                           int64_t words_per_frame, unsigned long long* __restrict__ totals) {
  __shared__ long long sm[4];
  long long all = 0, us = 0;
  for (int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x; w < words_per_frame; w += (int64_t)gridDim.x * 256) {
    unsigned o = 0;
    for (int f = 0; f < n_frames; ++f) o |= bits[(int64_t)f * words_per_frame + w];
    all += __popc(o);
    us += __popc(used[w]);
  }
  all = cover_block_sum(all, sm);
  us = cover_block_sum(us, sm);
  if (threadIdx.x == 0) { atomicAdd(&totals[0], (unsigned long long)all); atomicAdd(&totals[1], (unsigned long long)us); }
}

// round(xyz / voxel) as int32 (max_coverage_sampling.py:44-45: f32 division, numpy round = half to even)
__global__ __launch_bounds__(256) void voxel_keys_kernel(const float* __restrict__ xyz, int64_t n, float voxel, int32_t* __restrict__ keys) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) keys[i] = (int32_t)rintf(xyz[i] / voxel);
}

struct CoverLayout {
  int log2cap;
  int64_t cap, words_per_frame, n_words;
  int64_t off_tvals, off_bits, off_used, off_chosen, off_gain, off_totals, off_err, total;
};

static CoverLayout cover_layout(int n_frames, int64_t m) {
  CoverLayout L;
  L.log2cap = 4;
  while ((1ll << L.log2cap) < 2 * m) ++L.log2cap;
  L.cap = 1ll << L.log2cap;
  L.words_per_frame = (m + 31) / 32;
  if (L.words_per_frame < 1) L.words_per_frame = 1;
  L.n_words = L.words_per_frame * (n_frames + 1);            // frame bitsets, then U
  auto up = [](int64_t v) { return (v + 255) / 256 * 256; };
  int64_t o = up(L.cap * 8);
  L.off_tvals = o; o += up(L.cap * 4);
  L.off_bits = o; o += up(L.words_per_frame * 4 * n_frames);
  L.off_used = L.off_bits + L.words_per_frame * 4 * n_frames;    // contiguous with the frame bitsets (one clear)
  o = up(L.off_used + L.words_per_frame * 4);
  L.off_chosen = o; o += up((int64_t)n_frames * 4);
  L.off_gain = o; o += up((int64_t)n_frames * 8);
  L.off_totals = o; o += 256;
  L.off_err = o; o += 256;
  L.total = o;
  return L;
}

}  // namespace v3d

using namespace v3d;

extern "C" int v3d_voxel_keys_f32(const float* xyz, int64_t n_values, float voxel_size, int32_t* keys, void* stream) {
  V3D_REQUIRE(xyz && keys && n_values >= 0 && voxel_size > 0.f, "v3d_voxel_keys_f32: bad arguments");
  if (n_values == 0) return V3D_OK;
  hipLaunchKernelGGL(voxel_keys_kernel, dim3((unsigned)((n_values + 255) / 256)), dim3(256), 0, (hipStream_t)stream, xyz, n_values, voxel_size, keys);
  return check_launch("v3d_voxel_keys_f32");
}

extern "C" int64_t v3d_greedy_cover_workspace_bytes(int n_frames, int64_t m) {
  if (n_frames <= 0 || m < 0) return 0;
  return cover_layout(n_frames, m).total;
}

extern "C" int v3d_greedy_cover(const int32_t* keys, int n_frames, int64_t pts_per_frame, const int32_t* scene, int64_t m,
                                int max_frames, int32_t* sel, int64_t* gain, int64_t* totals, int32_t* n_sel, void* workspace,
                                int64_t workspace_bytes, void* stream) {
  V3D_REQUIRE(keys && sel && gain && totals && n_sel && workspace, "v3d_greedy_cover: null pointer");
  V3D_REQUIRE(n_frames > 0 && pts_per_frame > 0 && m >= 0 && max_frames > 0 && (m == 0 || scene), "v3d_greedy_cover: bad shape");
  V3D_REQUIRE(m < (1ll << 30) && (int64_t)n_frames * pts_per_frame < (1ll << 40), "v3d_greedy_cover: problem too large");
  const CoverLayout L = cover_layout(n_frames, m);
  V3D_REQUIRE(workspace_bytes >= L.total && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "v3d_greedy_cover: workspace too small (%lld needed)", (long long)L.total);
  char* ws = (char*)workspace;
  uint64_t* tkeys = (uint64_t*)ws;
  int32_t* tvals = (int32_t*)(ws + L.off_tvals);
  unsigned* bits = (unsigned*)(ws + L.off_bits);
  unsigned* used = (unsigned*)(ws + L.off_used);
  int* chosen = (int*)(ws + L.off_chosen);
  long long* g = (long long*)(ws + L.off_gain);
  unsigned long long* tot = (unsigned long long*)(ws + L.off_totals);
  int* err = (int*)(ws + L.off_err);
  hipStream_t st = (hipStream_t)stream;
  const int64_t clr = L.cap > L.n_words ? L.cap : L.n_words;
  hipLaunchKernelGGL(cover_fill_kernel, dim3((unsigned)((clr + 255) / 256)), dim3(256), 0, st, tkeys, L.cap, bits, L.n_words);
  if (hipMemsetAsync(ws + L.off_chosen, 0, (size_t)(L.total - L.off_chosen), st) != hipSuccess) { set_error("v3d_greedy_cover: memset"); return V3D_E_LAUNCH; }
  if (hipMemsetAsync(n_sel, 0, sizeof(int32_t), st) != hipSuccess) { set_error("v3d_greedy_cover: memset"); return V3D_E_LAUNCH; }
  if (m > 0)
    hipLaunchKernelGGL(cover_insert_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, scene, m, tkeys, tvals, L.log2cap, err);
  const int64_t n_points = (int64_t)n_frames * pts_per_frame;
  hipLaunchKernelGGL(cover_mark_kernel, dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0, st, keys, pts_per_frame, n_points, tkeys, tvals,
                     L.log2cap, bits, L.words_per_frame);
  const int rounds = max_frames < n_frames ? max_frames : n_frames;
  for (int r = 0; r < rounds; ++r) {
    hipLaunchKernelGGL(cover_gain_kernel, dim3(n_frames), dim3(256), 0, st, bits, used, chosen, L.words_per_frame, g);
    hipLaunchKernelGGL(cover_pick_kernel, dim3(1), dim3(256), 0, st, bits, used, chosen, g, n_frames, L.words_per_frame, r, sel, gain, n_sel);
  }
  int tb = (int)((L.words_per_frame + 255) / 256);
  tb = tb > 1024 ? 1024 : tb;
  hipLaunchKernelGGL(cover_totals_kernel, dim3(tb), dim3(256), 0, st, bits, used, n_frames, L.words_per_frame, tot);
  if (hipMemcpyAsync(totals, tot, 2 * sizeof(int64_t), hipMemcpyDeviceToDevice, st) != hipSuccess) { set_error("v3d_greedy_cover: copy"); return V3D_E_LAUNCH; }
  return check_launch("v3d_greedy_cover");
}
