// Probe (not product): the 64-queries-per-wave prefill kernel with in-kernel stamps and, through -DV3D_A64_ABL=<bits>, with parts
// of the steady-state step removed (results are then garbage: timing only).  Prints cycles per steady-state step.
//   for abl in 0 1 2 4 6 8 16 32 63; do hipcc -O3 -std=c++17 -fno-slp-vectorize --offload-arch=gfx950 -DV3D_ATTN_PROF -DV3D_A64_ABL=$abl \
//     -Iinclude -Ivideo-3d-llm_amd/csrc tools/probes/attn64_probe.hip video-3d-llm_amd/csrc/host.cpp -o tools/probes/_build/attn64_probe_$abl; done
#include "../../video-3d-llm_amd/csrc/attention.hip"
#include <string.h>
#include <vector>
#include <random>
#include <algorithm>

int main(int argc, char** argv) {
  const int S = argc > 1 ? atoi(argv[1]) : 6794, H = 28, KV = 4, D = 128;
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
  float ms = 0, best = 1e9f;
  for (int it = 0; it < 5; ++it) {
    hipEventRecord(e0, 0);
    int rc = v3d_attention(q, k, v, o, V3D_BF16, 1, S, S, H, KV, D, D, H * D, KV * D, KV * D, H * D, 0, 0, 0, D, D, D, 1, 0, 0.08838834764f, nullptr);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    if (rc) { printf("error %d: %s\n", rc, v3d_last_error()); return 1; }
    best = std::min(best, ms);
  }
  const int nblk = std::min(4096, ((S + 255) / 256) * H);
  std::vector<unsigned long long> hp(4 * 4096), hb(4 * 4096);
  hipMemcpyFromSymbol(hp.data(), HIP_SYMBOL(v3d::g_attn_prof), hp.size() * 8);
  hipMemcpyFromSymbol(hb.data(), HIP_SYMBOL(v3d::g_attn_blocks), hb.size() * 8);
  double steps = 0, cyc = 0, pro = 0, tot = 0, tiles = 0, clk = 0; int ncl = 0;
  unsigned long long tmin = ~0ull, tmax = 0;
  for (int i = 0; i < nblk; ++i) {
    steps += (double)hp[2 * 4096 + i]; cyc += (double)hp[4096 + i]; pro += (double)hp[i]; tot += (double)hb[4 * i + 2]; tiles += (double)hb[4 * i + 3];
    tmin = std::min(tmin, hb[4 * i]); tmax = std::max(tmax, hb[4 * i + 1]);
    const double rt = (double)(hb[4 * i + 1] - hb[4 * i]);
    if (rt > 500) { clk += hb[4 * i + 2] / rt * 100.0; ++ncl; }
  }
  printf("ABL=%d S=%d: %.1f us | steady: %.0f cycles/step over %.0f steps | prologue %.0f cycles/WG | whole WG %.0f cycles/tile | clock %.0f MHz | span %.1f us\n",
         V3D_A64_ABL, S, best * 1e3, cyc / std::max(1.0, steps), steps, pro / nblk, tot / tiles, clk / std::max(1, ncl), (tmax - tmin) / 100.0);
  return 0;
}
