// Probe (not product): streaming variants of the prefill RMSNorm (6794 x 3584 bf16: 48.7 MB read + 48.7 MB written).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probes/norm_probe.hip -o tools/probes/_build/norm_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__device__ __forceinline__ float bf(uint32_t w, int hi) { return __uint_as_float(hi ? (w & 0xffff0000u) : (w << 16)); }
__device__ __forceinline__ uint32_t pk(float a, float b) {
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  const b2 v = {static_cast<__bf16>(a), static_cast<__bf16>(b)};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float rt(float x) { return __uint_as_float(((uint32_t)__builtin_bit_cast(uint16_t, static_cast<__bf16>(x))) << 16); }
__device__ __forceinline__ float wsum(float x) { for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o); return x; }

template <int MODE, int ROWS_PER_WAVE>
__global__ __launch_bounds__(256) void rms(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w, uint16_t* __restrict__ out, int rows, int cols, float eps) {
  const int lane = threadIdx.x & 63;
  const int nv = cols / 8;
  const uint4* wr = reinterpret_cast<const uint4*>(w);
  for (int rr = 0; rr < ROWS_PER_WAVE; ++rr) {
    const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS_PER_WAVE + rr;
    if (row >= rows) return;
    const uint4* xr = reinterpret_cast<const uint4*>(x + (size_t)row * cols);
    uint4 v[7];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int k = i * 64 + lane;
      if (k < nv) {
        if (MODE == 1) { const uint32_t* p = reinterpret_cast<const uint32_t*>(xr + k); v[i] = make_uint4(__builtin_nontemporal_load(p), __builtin_nontemporal_load(p + 1), __builtin_nontemporal_load(p + 2), __builtin_nontemporal_load(p + 3)); }
        else v[i] = xr[k];
        const uint32_t ws[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float a = bf(ws[j], 0), b = bf(ws[j], 1); ss = fmaf(a, a, ss); ss = fmaf(b, b, ss); }
      }
    }
    ss = wsum(ss);
    const float r = 1.0f / sqrtf(ss / (float)cols + eps);
    uint4* orow = reinterpret_cast<uint4*>(out + (size_t)row * cols);
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int k = i * 64 + lane;
      if (k < nv) {
        const uint4 wv = wr[k];
        const uint32_t xs[4] = {v[i].x, v[i].y, v[i].z, v[i].w}, wz[4] = {wv.x, wv.y, wv.z, wv.w};
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = pk(bf(wz[j], 0) * rt(bf(xs[j], 0) * r), bf(wz[j], 1) * rt(bf(xs[j], 1) * r));
        if (MODE == 1) { uint32_t* p = reinterpret_cast<uint32_t*>(orow + k); __builtin_nontemporal_store(o[0], p); __builtin_nontemporal_store(o[1], p + 1); __builtin_nontemporal_store(o[2], p + 2); __builtin_nontemporal_store(o[3], p + 3); }
        else orow[k] = make_uint4(o[0], o[1], o[2], o[3]);
      }
    }
  }
}

// block per row: 448 threads = one 16-byte vector each (3584 / 8), LDS reduction
__global__ __launch_bounds__(448) void rms_block(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w, uint16_t* __restrict__ out, int rows, int cols, float eps) {
  __shared__ float sm[8];
  const int row = blockIdx.x, k = threadIdx.x;
  const uint4 v = reinterpret_cast<const uint4*>(x + (size_t)row * cols)[k];
  const uint32_t xs[4] = {v.x, v.y, v.z, v.w};
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) { const float a = bf(xs[j], 0), b = bf(xs[j], 1); ss = fmaf(a, a, ss); ss = fmaf(b, b, ss); }
  ss = wsum(ss);
  if ((k & 63) == 0) sm[k >> 6] = ss;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < 7; ++i) t += sm[i];
  const float r = 1.0f / sqrtf(t / (float)cols + eps);
  const uint4 wv = reinterpret_cast<const uint4*>(w)[k];
  const uint32_t wz[4] = {wv.x, wv.y, wv.z, wv.w};
  uint32_t o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = pk(bf(wz[j], 0) * rt(bf(xs[j], 0) * r), bf(wz[j], 1) * rt(bf(xs[j], 1) * r));
  reinterpret_cast<uint4*>(out + (size_t)row * cols)[k] = make_uint4(o[0], o[1], o[2], o[3]);
}

int main() {
  const int rows = 6794, cols = 3584;
  uint16_t *x, *w, *o;
  (void)hipMalloc(&x, (size_t)rows * cols * 2); (void)hipMalloc(&w, cols * 2); (void)hipMalloc(&o, (size_t)rows * cols * 2);
  (void)hipMemset(x, 0x3f, (size_t)rows * cols * 2); (void)hipMemset(w, 0x3f, cols * 2);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto time = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) launch();
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %6.1f us  %5.2f TB/s\n", name, ms * 50, 2.0 * rows * cols * 2 / (ms / 20 * 1e-3) / 1e12);
  };
  time("wave per row (product)", [&] { hipLaunchKernelGGL((rms<0, 1>), dim3((rows + 3) / 4), dim3(256), 0, 0, x, w, o, rows, cols, 1e-6f); });
  time("wave per row, non-temporal", [&] { hipLaunchKernelGGL((rms<1, 1>), dim3((rows + 3) / 4), dim3(256), 0, 0, x, w, o, rows, cols, 1e-6f); });
  time("wave per 2 rows", [&] { hipLaunchKernelGGL((rms<0, 2>), dim3((rows + 7) / 8), dim3(256), 0, 0, x, w, o, rows, cols, 1e-6f); });
  time("wave per 4 rows", [&] { hipLaunchKernelGGL((rms<0, 4>), dim3((rows + 15) / 16), dim3(256), 0, 0, x, w, o, rows, cols, 1e-6f); });
  time("block (448 threads) per row", [&] { hipLaunchKernelGGL(rms_block, dim3(rows), dim3(448), 0, 0, x, w, o, rows, cols, 1e-6f); });
  return 0;
}
