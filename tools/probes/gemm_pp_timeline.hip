// Probe (not product): where does a tile of the ping-pong GEMM spend its time BETWEEN K loops?  Shader-clock stamps (s_memtime) per
// wave around the segments of a tile; prints the mean cycles per tile of waves 0 and 4 (the two wave rows) over all workgroups.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DV3D_PP_TIMELINE -Iinclude -Ivideo-3d-llm_amd/csrc tools/probes/gemm_pp_timeline.hip \
//         video-3d-llm_amd/csrc/host.cpp -o tools/probes/_build/gemm_pp_timeline
#include "../../video-3d-llm_amd/csrc/gemm.hip"
#include <vector>
#include <random>

int main(int argc, char** argv) {
  struct Shape { const char* name; int M, N, K, epi; };
  const Shape shapes[] = {{"gate_up", 6794, 37888, 3584, 6}, {"qkv", 6794, 4608, 3584, 1}, {"vit qkv", 23328, 3584, 1152, 1}, {"vit fc2", 23328, 1280, 4352, 4}};
  for (const Shape& sh : shapes) {
    const int M = sh.M, N = sh.N, K = sh.K;
    std::vector<uint16_t> ha((size_t)M * K), hw((size_t)N * K);
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    auto bf = [](float x) { union { float f; uint32_t u; } c; c.f = x; return (uint16_t)((c.u + 0x7fff + ((c.u >> 16) & 1)) >> 16); };
    for (auto& x : ha) x = bf(0.5f * nd(rng));
    for (auto& x : hw) x = bf(0.05f * nd(rng));
    void *a, *w, *o, *b, *r;
    hipMalloc(&a, ha.size() * 2); hipMalloc(&w, hw.size() * 2); hipMalloc(&o, (size_t)M * N * 2); hipMalloc(&b, (size_t)N * 2); hipMalloc(&r, (size_t)M * N * 2);
    hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMemset(b, 0, (size_t)N * 2); hipMemset(r, 0, (size_t)M * N * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int it = 0; it < 4; ++it) {
      hipEventRecord(e0, 0);
      int rc = v3d_gemm(a, K, w, K, b, r, N, 0, o, N, M, N, K, V3D_BF16, sh.epi, nullptr);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      if (rc) { printf("error %d: %s\n", rc, v3d_last_error()); return 1; }
    }
    std::vector<unsigned long long> hp(256 * 8 * 8);
    hipMemcpyFromSymbol(hp.data(), HIP_SYMBOL(v3d::g_pp_tl), hp.size() * 8);
    printf("%-8s M=%d N=%d K=%d: %.1f us  %.0f TF/s (with stamps)\n", sh.name, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
    for (int wv : {0, 4}) {
      double s[7] = {0, 0, 0, 0, 0, 0, 0};
      for (int blk = 0; blk < 256; ++blk) for (int i = 0; i < 7; ++i) s[i] += (double)hp[(blk * 8 + wv) * 8 + i];
      const double n = s[6] > 0 ? s[6] : 1;
      printf("   wave %d, cycles per tile: zero+first wait %6.0f | K loop %7.0f (%.0f per K-step) | ring-free barrier %6.0f | next-tile issue %6.0f | epilogue %6.0f | drain %6.0f | tiles/wg %.1f\n",
             wv, s[0] / n, s[1] / n, s[1] / n / (K / 64), s[2] / n, s[3] / n, s[4] / n, s[5] / n, n / 256);
    }
    hipFree(a); hipFree(w); hipFree(o); hipFree(b); hipFree(r);
  }
  return 0;
}
