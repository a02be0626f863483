// Sanitizer self-test of the HOST half of the C ABI (csrc/host.cpp): built by `make -C video-3d-llm_amd/csrc sanitize` with
// g++ -fsanitize=address,undefined (GPU AddressSanitizer is not available on this pool; the device entry points validate their
// arguments on the host before any launch and are exercised by tests/test_abi_host.py and the -m gpu tests).
// Exercises the valid paths on ragged sizes and every argument-validation branch; exits non-zero on a wrong answer.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <tuple>
#include <vector>

#include "v3d.h"

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "host_selftest: %s failed at line %d (%s)\n", #c, __LINE__, v3d_last_error()); return 1; } } while (0)

int main() {
  CHECK(v3d_abi_version() == V3D_ABI_VERSION);
  // a1: np.linspace(0, n-1, F, dtype=int)
  for (int total : {1, 2, 7, 31, 32, 33, 100, 517}) {
    for (int n : {1, 8, 32, 600}) {
      std::vector<int32_t> out(n, -1);
      CHECK(v3d_uniform_frame_indices_host(total, n, out.data()) == V3D_OK);
      CHECK(out[0] == 0 && (n == 1 || out[n - 1] == total - 1));
      for (int i = 1; i < n; ++i) CHECK(out[i] >= out[i - 1] && out[i] < total);
    }
  }
  int32_t one = 0;
  CHECK(v3d_uniform_frame_indices_host(0, 4, &one) == V3D_E_INVALID && std::strlen(v3d_last_error()) > 0);
  CHECK(v3d_uniform_frame_indices_host(4, 0, &one) == V3D_E_INVALID);
  CHECK(v3d_uniform_frame_indices_host(4, 4, nullptr) == V3D_E_INVALID);
  // a3: greedy max-coverage against a brute-force set implementation
  std::srand(7);
  for (int trial = 0; trial < 6; ++trial) {
    const int n_frames = 3 + trial * 5, pts = 40 + 13 * trial;
    std::vector<int32_t> keys((size_t)n_frames * pts * 3);
    for (auto& k : keys) k = std::rand() % 9 - 4;
    std::set<std::tuple<int, int, int>> scene;
    std::vector<int32_t> sc;
    for (int i = 0; i < 120; ++i) {
      const int a = std::rand() % 9 - 4, b = std::rand() % 9 - 4, c = std::rand() % 9 - 4;
      sc.push_back(a); sc.push_back(b); sc.push_back(c);          // duplicates on purpose
      scene.insert({a, b, c});
    }
    const int max_frames = 4 + trial;
    std::vector<int32_t> sel(max_frames);
    std::vector<int64_t> gain(max_frames);
    int64_t n_all = -1, n_sel = -1;
    const int picks = v3d_greedy_cover_host(keys.data(), n_frames, pts, sc.data(), (int64_t)sc.size() / 3, max_frames, sel.data(), gain.data(), &n_all, &n_sel);
    CHECK(picks >= 0 && picks <= max_frames);
    std::vector<std::set<std::tuple<int, int, int>>> fs(n_frames);
    std::set<std::tuple<int, int, int>> all;
    for (int f = 0; f < n_frames; ++f)
      for (int p = 0; p < pts; ++p) {
        const int32_t* k = &keys[((size_t)f * pts + p) * 3];
        std::tuple<int, int, int> t{k[0], k[1], k[2]};
        if (scene.count(t)) { fs[f].insert(t); all.insert(t); }
      }
    std::set<std::tuple<int, int, int>> used;
    std::vector<char> taken(n_frames, 0);
    int want_picks = 0;
    for (int step = 0; step < max_frames; ++step) {
      int best = -1; size_t best_gain = 0;
      for (int f = 0; f < n_frames; ++f) {
        if (taken[f]) continue;
        size_t g = 0;
        for (auto& t : fs[f]) g += !used.count(t);
        if (best < 0 || g > best_gain) { best = f; best_gain = g; }       // first maximum = lowest frame position
      }
      if (best < 0 || best_gain == 0) break;
      CHECK(want_picks < picks && sel[want_picks] == best && gain[want_picks] == (int64_t)best_gain);
      taken[best] = 1;
      for (auto& t : fs[best]) used.insert(t);
      ++want_picks;
    }
    CHECK(want_picks == picks && n_all == (int64_t)all.size() && n_sel == (int64_t)used.size());
  }
  int32_t k3[3] = {0, 0, 0};
  int64_t g1 = 0, na = 0, ns = 0;
  CHECK(v3d_greedy_cover_host(nullptr, 1, 1, k3, 1, 1, &one, &g1, &na, &ns) == V3D_E_INVALID);
  CHECK(v3d_greedy_cover_host(k3, 0, 1, k3, 1, 1, &one, &g1, &na, &ns) == V3D_E_INVALID);
  CHECK(v3d_greedy_cover_host(k3, 1, 1, k3, 1, 0, &one, &g1, &na, &ns) == V3D_E_INVALID);
  CHECK(v3d_greedy_cover_host(k3, 1, 1, nullptr, 0, 1, &one, &g1, &na, &ns) >= 0);      // an empty scene is legal: nothing to cover
  std::puts("host_selftest ok");
  return 0;
}
