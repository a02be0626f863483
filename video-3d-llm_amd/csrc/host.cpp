// Host side of the C ABI: error reporting and the host-only helpers (frame sampling,
// greedy max-coverage frame selection).
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "v3d.h"

namespace v3d {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

}  // namespace v3d

extern "C" int v3d_abi_version(void) { return V3D_ABI_VERSION; }
extern "C" const char* v3d_last_error(void) { return v3d::g_err; }

// a1  llava/video_utils.py:187  np.linspace(0, total-1, n, dtype=int)
// numpy: y = arange(n) * (delta/div) + start in f64, y[-1] = stop, floor, cast.
extern "C" int v3d_uniform_frame_indices_host(int total_frames, int n, int32_t* out_host) {
  if (!out_host || total_frames <= 0 || n <= 0) {
    v3d::set_error("v3d_uniform_frame_indices_host: bad arguments total=%d n=%d", total_frames, n);
    return V3D_E_INVALID;
  }
  const double delta = (double)(total_frames - 1);
  const int div = n - 1;
  for (int i = 0; i < n; ++i) {
    double y;
    if (div > 0) {
      const double step = delta / div;
      y = step == 0.0 ? ((double)i / div) * delta : (double)i * step;
    } else {
      y = (double)i * delta;
    }
    out_host[i] = (int32_t)std::floor(y);
  }
  if (n > 1) out_host[n - 1] = total_frames - 1;
  return V3D_OK;
}

// a3  scripts/3d/preprocessing/max_coverage_sampling.py:44-94
static inline uint64_t pack_key(const int32_t* k) {
  // 21 bits per axis, biased: voxel keys of indoor scenes are within +-2^20
  return ((uint64_t)(uint32_t)(k[0] + (1 << 20)) << 42) | ((uint64_t)(uint32_t)(k[1] + (1 << 20)) << 21) |
         (uint64_t)(uint32_t)(k[2] + (1 << 20));
}

extern "C" int v3d_greedy_cover_host(const int32_t* keys_host, int n_frames, int64_t pts_per_frame,
                                     const int32_t* scene_host, int64_t m, int max_frames, int32_t* sel_host,
                                     int64_t* gain_host, int64_t* num_all_host, int64_t* num_sel_host) {
  if (!keys_host || (!scene_host && m > 0) || !sel_host || !gain_host || n_frames <= 0 || pts_per_frame <= 0 || m < 0 ||
      max_frames <= 0) {      // (an empty scene - m == 0, scene pointer unused - is legal, as for the device entry)
    v3d::set_error("v3d_greedy_cover_host: bad arguments");
    return V3D_E_INVALID;
  }
  for (int64_t i = 0; i < (int64_t)n_frames * pts_per_frame * 3; ++i)
    if (keys_host[i] < -(1 << 20) || keys_host[i] >= (1 << 20)) {
      v3d::set_error("v3d_greedy_cover_host: voxel key %d out of the +-2^20 range", keys_host[i]);
      return V3D_E_INVALID;
    }
  // scene voxel set -> sorted unique packed keys; a voxel's rank is its dense index
  std::vector<uint64_t> scene((size_t)m);
  for (int64_t i = 0; i < m; ++i) {
    const int32_t* k = scene_host + i * 3;
    if (k[0] < -(1 << 20) || k[0] >= (1 << 20) || k[1] < -(1 << 20) || k[1] >= (1 << 20) || k[2] < -(1 << 20) ||
        k[2] >= (1 << 20)) {
      v3d::set_error("v3d_greedy_cover_host: scene voxel out of range");
      return V3D_E_INVALID;
    }
    scene[(size_t)i] = pack_key(k);
  }
  std::sort(scene.begin(), scene.end());
  scene.erase(std::unique(scene.begin(), scene.end()), scene.end());
  const size_t ms = scene.size();

  // per frame: dense indices of its unique in-scene voxels (`set(frame) & pc_voxel`, :74)
  std::vector<std::vector<uint32_t>> frame_idx((size_t)n_frames);
  std::vector<uint64_t> tmp((size_t)pts_per_frame);
  std::vector<uint8_t> seen(ms, 0);  // union over frames, for num_all_voxels (:96)
  for (int f = 0; f < n_frames; ++f) {
    const int32_t* k = keys_host + (size_t)f * pts_per_frame * 3;
    for (int64_t p = 0; p < pts_per_frame; ++p) tmp[(size_t)p] = pack_key(k + p * 3);
    std::sort(tmp.begin(), tmp.end());
    const size_t nu = std::unique(tmp.begin(), tmp.end()) - tmp.begin();
    auto& out = frame_idx[(size_t)f];
    size_t j = 0;
    for (size_t i = 0; i < nu && j < ms; ++i) {   // merge of two sorted lists
      while (j < ms && scene[j] < tmp[i]) ++j;
      if (j < ms && scene[j] == tmp[i]) { out.push_back((uint32_t)j); seen[j] = 1; }
    }
  }
  int64_t num_all = 0;
  for (size_t i = 0; i < ms; ++i) num_all += seen[i];

  std::vector<uint8_t> used(ms, 0), taken((size_t)n_frames, 0);
  int picks = 0;
  int64_t num_sel = 0;
  const int limit = std::min(n_frames, max_frames);
  while (picks < limit) {
    int64_t best = -1;
    int best_f = -1;
    for (int f = 0; f < n_frames; ++f) {
      if (taken[(size_t)f]) continue;
      int64_t gain = 0;
      for (uint32_t i : frame_idx[(size_t)f]) gain += !used[i];
      if (gain > best) { best = gain; best_f = f; }   // strict '>' : first (lowest position) wins ties
    }
    for (uint32_t i : frame_idx[(size_t)best_f])
      if (!used[i]) { used[i] = 1; ++num_sel; }
    taken[(size_t)best_f] = 1;
    sel_host[picks] = best_f;
    gain_host[picks] = best;
    ++picks;
  }
  if (num_all_host) *num_all_host = num_all;
  if (num_sel_host) *num_sel_host = num_sel;
  return picks;
}
