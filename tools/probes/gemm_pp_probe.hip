// Probe (not product): where does a tile of the ping-pong GEMM spend its time?  Builds gemm.hip three ways and times one shape:
//   V3D_PP_PROBE=0  the product kernel
//   V3D_PP_PROBE=1  K loop alone (epilogue replaced by a never-taken store; results wrong)
//   V3D_PP_PROBE=1 -DV3D_PP_SAMETILE  the same with every workgroup reading tile (0, 0)'s operands (all L2 hits): gate/up 1236 -> 1166 us,
//                   i.e. memory latency costs the K loop ~6 %; the rest of the gap to the MFMA peak is issue / LDS and the sustained clock
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DV3D_PP_PROBE=1 -Iinclude -Ivideo-3d-llm_amd/csrc tools/probes/gemm_pp_probe.hip \
//         video-3d-llm_amd/csrc/host.cpp -o tools/probes/_build/gemm_pp_probe1
#include "../../video-3d-llm_amd/csrc/gemm.hip"
#include <vector>

int main(int argc, char** argv) {
  struct Shape { const char* name; int M, N, K, epi; };
  const Shape shapes[] = {{"gate_up", 6794, 37888, 3584, 6}, {"qkv", 6794, 4608, 3584, 1}, {"down", 6794, 3584, 18944, 5},
                          {"vit qkv", 23328, 4608, 1152, 1}, {"vit fc1", 23328, 4352, 1152, 3}, {"vit fc2", 23328, 1280, 4352, 4}};
  for (const Shape& sh : shapes) {
    const int M = sh.M, N = sh.N, K = sh.K;
    void *a, *w, *o, *b, *r;
    hipMalloc(&a, (size_t)M * K * 2); hipMalloc(&w, (size_t)N * K * 2); hipMalloc(&o, (size_t)M * N * 2); hipMalloc(&b, (size_t)N * 2); hipMalloc(&r, (size_t)M * N * 2);
    hipMemset(a, 0x3c, (size_t)M * K * 2); hipMemset(w, 0x3c, (size_t)N * K * 2); hipMemset(b, 0, (size_t)N * 2); hipMemset(r, 0, (size_t)M * N * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0, best = 1e9f;
    for (int it = 0; it < 8; ++it) {
      hipEventRecord(e0, 0);
      int rc = v3d_gemm(a, K, w, K, b, r, N, 0, o, N, M, N, K, V3D_BF16, sh.epi, nullptr);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      if (rc) { printf("error %d: %s\n", rc, v3d_last_error()); return 1; }
      if (it >= 2 && ms < best) best = ms;
    }
    printf("probe %d  %-8s M=%d N=%d K=%d: %.1f us  %.0f TF/s\n", V3D_PP_PROBE, sh.name, M, N, K, best * 1e3, 2.0 * M * N * K / best / 1e9);
    hipFree(a); hipFree(w); hipFree(o); hipFree(b); hipFree(r);
  }
  return 0;
}
