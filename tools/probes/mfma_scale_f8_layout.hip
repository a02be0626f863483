#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
// each lane supplies 32 bytes of A and of B (fp8 e4m3).  We pass raw bytes from global.
__global__ void k(const uint8_t* A, const uint8_t* B, float* D, int scale_a, int scale_b) {
  const int l = threadIdx.x;
  v8i a, b;
  for (int i = 0; i < 8; ++i) { a[i] = ((const int*)A)[l * 8 + i]; b[i] = ((const int*)B)[l * 8 + i]; }
  v4f c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);
  for (int i = 0; i < 4; ++i) D[l * 4 + i] = c[i];
}
int main() {
  // fp8 e4m3: 1.0 = 0x38, 2.0 = 0x40
  uint8_t hA[64 * 32], hB[64 * 32]; float hD[256];
  uint8_t *dA, *dB; float* dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
  // experiment 1: A all ones, B all ones, scales 127 -> expect 128 everywhere
  for (int sc : {127, 128, 0x7f7f7f7f}) {
    memset(hA, 0x38, sizeof hA); memset(hB, 0x38, sizeof hB);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, sc, sc);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    printf("scale %#x: D[0]=%g D[37]=%g D[255]=%g\n", sc, hD[0], hD[37], hD[255]);
  }
  // experiment 2: A one-hot: lane la byte ba = 1.0, everything else 0; B: all lanes byte j = value encoding (lane, byte)?
  // B all ones -> D row sums: which D entries (lane, reg) are nonzero tells A's (row) mapping for (la, ba).
  for (int la : {0, 1, 15, 16, 17, 33, 63}) {
    memset(hA, 0, sizeof hA); memset(hB, 0x38, sizeof hB);
    hA[la * 32 + 5] = 0x38;
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, 127, 127);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    printf("A one-hot lane %d byte 5 -> nonzero D (lane,reg):", la);
    int n = 0; for (int i = 0; i < 256; ++i) if (hD[i] != 0 && n++ < 20) printf(" (%d,%d)=%g", i / 4, i % 4, hD[i]);
    printf("  [%d nonzero]\n", n);
  }
  // experiment 3: k mapping: A lane la byte ba = 1, B lane lb byte bb = 1: nonzero iff same k.  Sweep (lb,bb) for fixed (la=0,ba).
  for (int ba : {0, 1, 15, 16, 31}) for (int la : {0, 16, 32, 48}) {
    memset(hA, 0, sizeof hA); hA[la * 32 + ba] = 0x38;
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
    int found_lb = -1, found_bb = -1, cnt = 0;
    for (int lb = 0; lb < 64; lb += 16) for (int bb = 0; bb < 32; ++bb) {
      memset(hB, 0, sizeof hB); hB[lb * 32 + bb] = 0x38;
      hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, 127, 127);
      hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
      for (int i = 0; i < 256; ++i) if (hD[i] != 0) { found_lb = lb; found_bb = bb; ++cnt; }
    }
    printf("A(lane %d, byte %d) pairs with B(lane %d, byte %d) [%d hits]\n", la, ba, found_lb, found_bb, cnt);
  }
  return 0;
}
