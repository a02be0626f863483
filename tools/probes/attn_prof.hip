// Probe (not product): builds attention.hip with -DV3D_ATTN_PROF and prints, for the 16 heaviest query tiles of one
// head, each wave's cycles in {K reads + QK^T issue, softmax, V reads + PV issue, barrier wait, whole tile}.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DV3D_ATTN_PROF -Iinclude -Ivideo-3d-llm_amd/csrc tools/probes/attn_prof.hip \
//         video-3d-llm_amd/csrc/host.cpp -o tools/probes/_build/attn_prof
#include "../../video-3d-llm_amd/csrc/attention.hip"
#include <string.h>
#include <vector>
#include <random>

int main(int argc, char** argv) {
  const int S = argc > 1 ? atoi(argv[1]) : 6794, H = 28, KV = 4, D = 128;
  const int causal = argc > 2 ? atoi(argv[2]) : 1;
  std::vector<uint16_t> hq((size_t)S * H * D), hk((size_t)S * KV * D), hv((size_t)S * KV * D);
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  auto bf = [](float x) { union { float f; uint32_t u; } c; c.f = x; const uint32_t u = c.u; return (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); };
  for (auto& x : hq) x = bf(nd(rng));
  for (auto& x : hk) x = bf(nd(rng));
  for (auto& x : hv) x = bf(nd(rng));
  void *q, *k, *v, *o;
  hipMalloc(&q, hq.size() * 2); hipMalloc(&k, hk.size() * 2); hipMalloc(&v, hv.size() * 2); hipMalloc(&o, hq.size() * 2);
  hipMemcpy(q, hq.data(), hq.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(k, hk.data(), hk.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(v, hv.data(), hv.size() * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(e0, 0);
    int rc = v3d_attention(q, k, v, o, V3D_BF16, 1, S, S, H, KV, D, D, H * D, KV * D, KV * D, H * D, 0, 0, 0, D, D, D, causal, 0,
                           0.08838834764f, nullptr);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    if (rc) { printf("error %d: %s\n", rc, v3d_last_error()); return 1; }
  }
  printf("S=%d causal=%d: %.1f us\n", S, causal, ms * 1e3);
  {   // per-phase split of the pipeline step (stamps cost ~10 % and drain the LDS reads in flight at each stamp)
    unsigned long long hp[8 * 64];
    hipMemcpyFromSymbol(hp, HIP_SYMBOL(v3d::g_attn_prof), sizeof(hp));
    printf("qtile wave tiles | fill+dma+qk  vreads+prep  pv+softmax  vmcnt  barrier | step   (cycles per tile)\n");
    for (int b = 0; b < 16; b += 5)
      for (int w = 0; w < 4; ++w) {
        const unsigned long long* r = hp + (b * 4 + w) * 8;
        const double n = (double)r[6];
        printf("%5d %4d %5.0f | %8.0f %10.0f %10.0f %8.0f %7.0f | %6.0f\n", b, w, n, r[0] / n, r[1] / n, r[2] / n, r[3] / n, r[4] / n, r[5] / n);
      }
  }
  // schedule: per workgroup realtime window (100 MHz ticks), shader clock, and which CU it ran on
  const int nblk = ((S + 127) / 128) * H;
  std::vector<unsigned long long> hb(4 * 4096);
  hipMemcpyFromSymbol(hb.data(), HIP_SYMBOL(v3d::g_attn_blocks), hb.size() * 8);
  unsigned long long tmin = ~0ull, tmax = 0;
  for (int i = 0; i < nblk && i < 4096; ++i) { if (hb[4 * i] < tmin) tmin = hb[4 * i]; if (hb[4 * i + 1] > tmax) tmax = hb[4 * i + 1]; }
  printf("blocks %d  span %.1f us\n", nblk, (tmax - tmin) / 100.0);
  double clk = 0; int ncl = 0;
  for (int i = 0; i < nblk && i < 4096; ++i) { const double rt = (double)(hb[4 * i + 1] - hb[4 * i]); if (rt > 500) { clk += hb[4 * i + 2] / rt * 100.0; ++ncl; } }
  printf("in-kernel shader clock %.0f MHz (mean over %d long workgroups)\n", clk / (ncl ? ncl : 1), ncl);
  // concurrency per CU at a few instants
  for (int frac = 1; frac < 10; frac += 2) {
    const unsigned long long at = tmin + (tmax - tmin) * frac / 10;
    int live = 0; std::vector<int> percu(8 * 64, 0);
    for (int i = 0; i < nblk && i < 4096; ++i)
      if (hb[4 * i] <= at && hb[4 * i + 1] > at) {
        ++live;
        const unsigned hw = (unsigned)hb[4 * i + 3]; const unsigned xcc = (unsigned)(hb[4 * i + 3] >> 32) & 15;
        const unsigned cu = (hw >> 8) & 15, se = (hw >> 13) & 7;     // gfx9 HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
        percu[(xcc & 7) * 64 + se * 16 + cu]++;
      }
    int c1 = 0, c2 = 0, c3 = 0;
    for (int v : percu) { if (v == 1) ++c1; else if (v == 2) ++c2; else if (v > 2) ++c3; }
    printf("t=%d0%%: live workgroups %d; CUs holding 1: %d, 2: %d, >2: %d\n", frac, live, c1, c2, c3);
  }
  {   // cycles per staged tile, heaviest and mid-weight workgroups (causal: workgroup of query tile j stages 2(j+1) tiles)
    const int nq = (S + 127) / 128;
    for (int j : {0, nq / 4, nq / 2, 3 * nq / 4}) {
      const int bid = j * H + 3;       // grid = (head, reversed query tile)
      const int tiles = causal ? 2 * (nq - j) : (S + 63) / 64;
      printf("query tile %d (from the end): %llu shader cycles, ~%d tiles -> %.0f cycles/tile\n", j, hb[4 * bid + 2], tiles, (double)hb[4 * bid + 2] / tiles);
    }
  }
  printf("first 6 workgroups: ");
  for (int i = 0; i < 6; ++i) printf("[%.1f..%.1f us] ", (hb[4 * i] - tmin) / 100.0, (hb[4 * i + 1] - tmin) / 100.0);
  printf("\n");
  return 0;
}
