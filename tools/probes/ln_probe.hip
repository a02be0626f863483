// Probe (not product): variants of the SigLIP LayerNorm pass (23328 rows x 1152 of stride 1280, bf16).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probes/ln_probe.hip -o tools/probes/_build/ln_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__device__ __forceinline__ float bf(uint32_t w, int hi) { return __uint_as_float(hi ? (w & 0xffff0000u) : (w << 16)); }
__device__ __forceinline__ uint32_t pk(float a, float b) {
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  const b2 v = {static_cast<__bf16>(a), static_cast<__bf16>(b)};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float wsum(float x) { for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o); return x; }
// dpp-based reduction
__device__ __forceinline__ float wsum_dpp(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x111, 0xf, 0xf, true));  // row_shr:1
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x112, 0xf, 0xf, true));  // row_shr:2
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x114, 0xf, 0xf, true));  // row_shr:4
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x118, 0xf, 0xf, true));  // row_shr:8
  // now lane 15 of each row of 16 has the row sum: combine the 4 rows through readlane
  float t = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 15)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 31)) +
            __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 47)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
  return t;
}

template <int MODE>   // 0: product form (wave per row, shuffles); 1: dpp reductions; 2: two rows per wave, loads of both in flight together (dpp)
__global__ __launch_bounds__(256) void ln(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w, const uint16_t* __restrict__ b, uint16_t* __restrict__ out,
                                          int rows, int cols, int ld, float eps) {
  const int lane = threadIdx.x & 63;
  const int nv = cols / 8;
  constexpr int R = MODE == 2 ? 2 : 1;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  uint4 v[R][3];
  float s[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    s[r] = 0.f;
    const int row = row0 + r < rows ? row0 + r : rows - 1;
    const uint4* xr = reinterpret_cast<const uint4*>(x + (size_t)row * ld);
#pragma unroll
    for (int i = 0; i < 3; ++i) { const int k = i * 64 + lane; v[r][i] = k < nv ? xr[k] : make_uint4(0, 0, 0, 0); }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int i = 0; i < 3; ++i) { const uint32_t ws[4] = {v[r][i].x, v[r][i].y, v[r][i].z, v[r][i].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) s[r] += bf(ws[j], 0) + bf(ws[j], 1); }
  }
  float mean[R], rstd[R];
#pragma unroll
  for (int r = 0; r < R; ++r) mean[r] = (MODE == 0 ? wsum(s[r]) : wsum_dpp(s[r])) / (float)cols;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) { const int k = i * 64 + lane; if (k < nv) { const uint32_t ws[4] = {v[r][i].x, v[r][i].y, v[r][i].z, v[r][i].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float d0 = bf(ws[j], 0) - mean[r], d1 = bf(ws[j], 1) - mean[r]; q = fmaf(d0, d0, q); q = fmaf(d1, d1, q); } } }
    rstd[r] = 1.0f / sqrtf((MODE == 0 ? wsum(q) : wsum_dpp(q)) / (float)cols + eps);
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (row0 + r >= rows) break;
    uint4* orow = reinterpret_cast<uint4*>(out + (size_t)(row0 + r) * ld);
#pragma unroll
    for (int i = 0; i < 3; ++i) { const int k = i * 64 + lane; if (k < nv) {
      const uint4 wv = reinterpret_cast<const uint4*>(w)[k], bv = reinterpret_cast<const uint4*>(b)[k];
      const uint32_t xs[4] = {v[r][i].x, v[r][i].y, v[r][i].z, v[r][i].w}, wz[4] = {wv.x, wv.y, wv.z, wv.w}, bz[4] = {bv.x, bv.y, bv.z, bv.w};
      uint32_t o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = pk((bf(xs[j], 0) - mean[r]) * rstd[r] * bf(wz[j], 0) + bf(bz[j], 0), (bf(xs[j], 1) - mean[r]) * rstd[r] * bf(wz[j], 1) + bf(bz[j], 1));
      orow[k] = make_uint4(o[0], o[1], o[2], o[3]); } }
  }
}

int main() {
  const int rows = 23328, cols = 1152, ld = 1280;
  uint16_t *x, *w, *b, *o;
  (void)hipMalloc(&x, (size_t)rows * ld * 2); (void)hipMalloc(&w, cols * 2); (void)hipMalloc(&b, cols * 2); (void)hipMalloc(&o, (size_t)rows * ld * 2);
  (void)hipMemset(x, 0x3f, (size_t)rows * ld * 2); (void)hipMemset(w, 0x3f, cols * 2); (void)hipMemset(b, 0x3f, cols * 2);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto time = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) launch();
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %6.1f us  %5.2f TB/s\n", name, ms * 50, 2.0 * rows * cols * 2 / (ms / 20 * 1e-3) / 1e12);
  };
  time("wave per row, shuffle reductions (product)", [&] { hipLaunchKernelGGL((ln<0>), dim3((rows + 3) / 4), dim3(256), 0, 0, x, w, b, o, rows, cols, ld, 1e-6f); });
  time("wave per row, DPP + readlane reductions", [&] { hipLaunchKernelGGL((ln<1>), dim3((rows + 3) / 4), dim3(256), 0, 0, x, w, b, o, rows, cols, ld, 1e-6f); });
  time("two rows per wave in flight together, DPP", [&] { hipLaunchKernelGGL((ln<2>), dim3((rows + 7) / 8), dim3(256), 0, 0, x, w, b, o, rows, cols, ld, 1e-6f); });
  return 0;
}
