// Probe: sustained bf16 MFMA rate and clock for 16x16x32 vs 32x32x16 with register-resident operands (no memory), 2 waves/SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(16))) float f16v;
template <int SHAPE>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, unsigned long long* clk) {
  bf8 a[4], b[4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) { a[i][j] = (__bf16)(0.001f * (threadIdx.x + i + j)); b[i][j] = (__bf16)(0.002f * (i + j)); }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  if (SHAPE == 16) {
    f4 acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = f4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][3];
  } else {
    f16v acc[8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + r) & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 7) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
int main(int argc, char** argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 512;
  float* out; unsigned long long* clk; hipMalloc(&out, 512 * 256 * 4); hipMalloc(&clk, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int rep = 0; rep < 2; ++rep)
    for (int shape : {16, 32}) {
      hipEventRecord(e0, 0);
      for (int l = 0; l < 5; ++l) {
        if (shape == 16) hipLaunchKernelGGL(k<16>, dim3(nb), dim3(256), 0, 0, out, iters, clk);
        else hipLaunchKernelGGL(k<32>, dim3(nb), dim3(256), 0, 0, out, iters, clk);
      }
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
      const double flops = 5.0 * nb * 4 * (double)iters * (shape == 16 ? 32 * 16384.0 : 16 * 32768.0);
      printf("mfma %dx%d: %.1f ms  %.0f TF/s  in-kernel clock %.0f MHz\n", shape, shape, ms, flops / ms / 1e9, (double)h[0] / h[1] * 100.0);
    }
  return 0;
}
