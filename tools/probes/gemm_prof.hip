// Probe (not product): builds gemm.hip with -DV3D_GEMM_PROF and prints per-wave cycles per K-step of the 256x256 kernel:
// {phases 0-2 (48 MFMAs + fragment reads), wait for own DMA + last reads, barrier, restage + prefetch + phase 3 issue}.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DV3D_GEMM_PROF -Iinclude -Ivideo-3d-llm_amd/csrc tools/probes/gemm_prof.hip \
//         video-3d-llm_amd/csrc/host.cpp -o tools/probes/_build/gemm_prof
#include "../../video-3d-llm_amd/csrc/gemm.hip"
#include <vector>
#include <random>

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 6794, N = argc > 2 ? atoi(argv[2]) : 37888, K = argc > 3 ? atoi(argv[3]) : 3584;
  std::vector<uint16_t> ha((size_t)M * K), hw((size_t)N * K);
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  auto bf = [](float x) { union { float f; uint32_t u; } c; c.f = x; return (uint16_t)((c.u + 0x7fff + ((c.u >> 16) & 1)) >> 16); };
  for (auto& x : ha) x = bf(0.5f * nd(rng));
  for (auto& x : hw) x = bf(0.05f * nd(rng));
  void *a, *w, *o;
  hipMalloc(&a, ha.size() * 2); hipMalloc(&w, hw.size() * 2); hipMalloc(&o, (size_t)M * N * 2);
  hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int it = 0; it < 4; ++it) {
    hipEventRecord(e0, 0);
    int rc = v3d_gemm(a, K, w, K, nullptr, nullptr, 0, 0, o, N, M, N, K, V3D_BF16, 0, nullptr);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    if (rc) { printf("error %d: %s\n", rc, v3d_last_error()); return 1; }
  }
  printf("M=%d N=%d K=%d: %.1f us  %.0f TF/s\n", M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
  unsigned long long hp[8 * 64];
  hipMemcpyFromSymbol(hp, HIP_SYMBOL(v3d::g_gemm_prof), sizeof(hp));
  printf("block wave ksteps | phases0-2   wait   barrier   restage+phase3 | per K-step   (shader cycles; 64 MFMAs = 1024 cycles per wave)\n");
  for (int b = 0; b < 8; b += 3)
    for (int wv = 0; wv < 8; ++wv) {
      const unsigned long long* r = hp + (b * 8 + wv) * 8;
      const double n = (double)r[5];
      if (n > 0) printf("%5d %4d %6.0f | %8.0f %6.0f %8.0f %10.0f | %8.0f\n", b, wv, n, r[0] / n, r[1] / n, r[2] / n, r[3] / n, r[4] / n);
    }
  return 0;
}
