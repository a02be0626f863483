// Geometry kernels: depth back-projection (K1, K1+K2), per-patch coordinate mean and
// voxelisation (K3+K4).  All HBM-bound byte/float streaming; no MFMA.
//
// Compiled with -ffp-contract=off: every f32 operation below is a single IEEE operation in the
// order written, so results are reproducible against the CPU oracle.
#include "v3d_common.h"

namespace v3d {

// ----------------------------------------------------------------------------------------
// K1  unproject  (llava/video_utils.py:38-68)
//   z = d/1000; x = (u-cx)*z/fx; y = (v-cy)*z/fy; w = P @ [x,y,z,1]; out = w[:3]/w[3]
// One thread = 4 consecutive pixels of a row: one 16-B depth load, three 16-B stores.
// ----------------------------------------------------------------------------------------
struct Cam {
  float fx, fy, cx, cy;
  float p[16];
};

__device__ __forceinline__ Cam load_cam(const float* K, const float* P, int v) {
  Cam c;
  const float* k = K + (size_t)v * 16;
  c.fx = k[0]; c.fy = k[5]; c.cx = k[2]; c.cy = k[6];
#pragma unroll
  for (int i = 0; i < 16; ++i) c.p[i] = P[(size_t)v * 16 + i];
  return c;
}

__device__ __forceinline__ void backproject(const Cam& c, float u, float v, float d, float* o) {
  const float z = __fdiv_rn(d, 1000.0f);
  const float x = __fdiv_rn(__fmul_rn(__fsub_rn(u, c.cx), z), c.fx);
  const float y = __fdiv_rn(__fmul_rn(__fsub_rn(v, c.cy), z), c.fy);
  float w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // sequential 4-term dot, one rounding per operation
    float s = __fmul_rn(c.p[i * 4 + 0], x);
    s = __fadd_rn(s, __fmul_rn(c.p[i * 4 + 1], y));
    s = __fadd_rn(s, __fmul_rn(c.p[i * 4 + 2], z));
    s = __fadd_rn(s, c.p[i * 4 + 3]);
    w[i] = s;
  }
  o[0] = __fdiv_rn(w[0], w[3]);
  o[1] = __fdiv_rn(w[1], w[3]);
  o[2] = __fdiv_rn(w[2], w[3]);
}

__global__ __launch_bounds__(256) void unproject_f32_kernel(const float* __restrict__ depth,
                                                            const float* __restrict__ K,
                                                            const float* __restrict__ P,
                                                            float* __restrict__ world, int H, int W,
                                                            int quads_per_frame) {
  const int v = blockIdx.y;
  const Cam c = load_cam(K, P, v);
  const size_t frame = (size_t)v * H * W;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < quads_per_frame; q += gridDim.x * blockDim.x) {
    const int pix = q * 4;
    const int row = pix / W, col = pix - row * W;
    const float4 d = *reinterpret_cast<const float4*>(depth + frame + pix);
    float o[12];
    backproject(c, (float)(col + 0), (float)row, d.x, o + 0);
    backproject(c, (float)(col + 1), (float)row, d.y, o + 3);
    backproject(c, (float)(col + 2), (float)row, d.z, o + 6);
    backproject(c, (float)(col + 3), (float)row, d.w, o + 9);
    float4* dst = reinterpret_cast<float4*>(world + (frame + pix) * 3);
    dst[0] = make_float4(o[0], o[1], o[2], o[3]);
    dst[1] = make_float4(o[4], o[5], o[6], o[7]);
    dst[2] = make_float4(o[8], o[9], o[10], o[11]);
  }
}

__global__ __launch_bounds__(256) void unproject_f32_scalar_kernel(const float* __restrict__ depth,
                                                                   const float* __restrict__ K,
                                                                   const float* __restrict__ P,
                                                                   float* __restrict__ world, int H, int W) {
  const int v = blockIdx.y;
  const Cam c = load_cam(K, P, v);
  const size_t frame = (size_t)v * H * W;
  const int n = H * W;
  for (int pix = blockIdx.x * blockDim.x + threadIdx.x; pix < n; pix += gridDim.x * blockDim.x) {
    const int row = pix / W, col = pix - row * W;
    float o[3];
    backproject(c, (float)col, (float)row, depth[frame + pix], o);
    float* dst = world + (frame + pix) * 3;
    dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2];
  }
}

// ----------------------------------------------------------------------------------------
// K1+K2  back-projection at the pixels kept by resize(INTER_NEAREST)+centre-crop
// (llava/video_utils.py:296-308).  OpenCV rule: src = min(floor(dst * src/dst_size), src-1),
// computed in double like cv::resize does.  One thread = one output pixel (3 values).
// ----------------------------------------------------------------------------------------
// One thread = 8 consecutive output pixels of a row (crop % 8 == 0): eight gathered u16 depth reads that fall into one or
// two cache lines, 24 values = three (16-bit) or six (f32) 16-byte stores.  The scalar form below serves other crops.
template <typename T>
__global__ __launch_bounds__(256) void unproject_sampled_x8_kernel(const uint16_t* __restrict__ depth,
                                                                   const float* __restrict__ K,
                                                                   const float* __restrict__ P,
                                                                   T* __restrict__ out, int H, int W, int crop,
                                                                   int left, double inv_fx, double inv_fy, int groups_per_frame) {
  const int v = blockIdx.y;
  const int grp = blockIdx.x * blockDim.x + threadIdx.x;
  if (grp >= groups_per_frame) return;
  const Cam c = load_cam(K, P, v);
  const int gpr = crop / 8;
  const int r = grp / gpr, col0 = (grp - r * gpr) * 8;
  int sr = (int)floor((double)r * inv_fy);
  sr = sr < H - 1 ? sr : H - 1;
  const uint16_t* drow = depth + ((size_t)v * H + sr) * W;
  float o[24];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int sc = (int)floor((double)(col0 + j + left) * inv_fx);
    sc = sc < W - 1 ? sc : W - 1;
    backproject(c, (float)sc, (float)sr, (float)(int)drow[sc], o + 3 * j);
  }
  T* dst = out + ((size_t)v * crop * crop + (size_t)r * crop + col0) * 3;
  constexpr int VEC = 16 / sizeof(T);
#pragma unroll
  for (int k = 0; k < 24 / VEC; ++k) reinterpret_cast<uint4*>(dst)[k] = vec_pack<T>(o + k * VEC);
}

template <typename T>
__global__ __launch_bounds__(256) void unproject_sampled_kernel(const uint16_t* __restrict__ depth,
                                                                const float* __restrict__ K,
                                                                const float* __restrict__ P,
                                                                T* __restrict__ out, int H, int W, int crop,
                                                                int new_w, int left, double inv_fx, double inv_fy) {
  const int v = blockIdx.y;
  const Cam c = load_cam(K, P, v);
  const int n = crop * crop;
  for (int pix = blockIdx.x * blockDim.x + threadIdx.x; pix < n; pix += gridDim.x * blockDim.x) {
    const int r = pix / crop, col = pix - r * crop;
    int sr = (int)floor((double)r * inv_fy);
    int sc = (int)floor((double)(col + left) * inv_fx);
    sr = sr < H - 1 ? sr : H - 1;
    sc = sc < W - 1 ? sc : W - 1;
    const float d = (float)(int)depth[((size_t)v * H + sr) * W + sc];
    float o[3];
    backproject(c, (float)sc, (float)sr, d, o);
    T* dst = out + ((size_t)v * n + pix) * 3;
    dst[0] = from_f32<T>(o[0]); dst[1] = from_f32<T>(o[1]); dst[2] = from_f32<T>(o[2]);
  }
}

// Bounds of the FULL-resolution back-projection (llava/video_utils.py:268-273: min / max of world_coords over all V*H*W
// pixels, taken before the resize and crop) without materialising the 118 MB tensor: every pixel is back-projected with the
// same arithmetic as unproject_f32_kernel and only the running min / max survive.  Stage 1: one partial (6 floats) per block;
// stage 2: one block folds the partials.  min / max are exact, so the two-stage order does not matter.
__global__ __launch_bounds__(256) void unproject_bounds_kernel(const uint16_t* __restrict__ depth, const float* __restrict__ K,
                                                               const float* __restrict__ P, int H, int W, float* __restrict__ partial) {
  const int v = blockIdx.y;
  const Cam c = load_cam(K, P, v);
  const int n = H * W;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int pix = blockIdx.x * blockDim.x + threadIdx.x; pix < n; pix += gridDim.x * blockDim.x) {
    const int row = pix / W, col = pix - row * W;
    float o[3];
    backproject(c, (float)col, (float)row, (float)(int)depth[(size_t)v * n + pix], o);
#pragma unroll
    for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], o[a]); hi[a] = fmaxf(hi[a], o[a]); }
  }
  __shared__ float sm[4][6];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
    }
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { sm[wave][2 * a] = lo[a]; sm[wave][2 * a + 1] = hi[a]; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int a = threadIdx.x;
    float r = sm[0][a];
    for (int w = 1; w < 4; ++w) r = (a & 1) ? fmaxf(r, sm[w][a]) : fminf(r, sm[w][a]);
    partial[((size_t)v * gridDim.x + blockIdx.x) * 6 + a] = r;
  }
}

__global__ __launch_bounds__(256) void bounds_fold_kernel(const float* __restrict__ partial, int n_partials, float* __restrict__ bounds) {
  __shared__ float sm[256][6];
  float r[6] = {INFINITY, -INFINITY, INFINITY, -INFINITY, INFINITY, -INFINITY};
  for (int i = threadIdx.x; i < n_partials; i += 256)
#pragma unroll
    for (int a = 0; a < 6; ++a) r[a] = (a & 1) ? fmaxf(r[a], partial[(size_t)i * 6 + a]) : fminf(r[a], partial[(size_t)i * 6 + a]);
#pragma unroll
  for (int a = 0; a < 6; ++a) sm[threadIdx.x][a] = r[a];
  __syncthreads();
  if (threadIdx.x < 6) {
    const int a = threadIdx.x;
    float x = sm[0][a];
    for (int i = 1; i < 256; ++i) x = (a & 1) ? fmaxf(x, sm[i][a]) : fminf(x, sm[i][a]);
    bounds[a] = x;
  }
}

// ----------------------------------------------------------------------------------------
// K4  discrete_coords (llava/model/llava_arch.py:259-272), torch-on-CPU dtype rules
// (pinned exhaustively for fp16 by tests/golden/discrete_coords.npz):
//   clamp; t = T(x - lo); q = T(float(t) / float(voxel)); r = rint(q)
// ----------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ float voxelise(float x, float lo, float hi, float voxel) {
  // torch.maximum / minimum propagate NaN
  if (x == x) {
    x = x < lo ? lo : x;
    x = x > hi ? hi : x;
  }
  const float t = round_to<T>(__fsub_rn(x, lo));
  const float q = round_to<T>(__fdiv_rn(t, voxel));
  return rintf(q);
}

struct Range { float lo[3], hi[3], voxel; };

template <typename T>
__global__ __launch_bounds__(256) void discrete_coords_kernel(const T* __restrict__ xyz, int64_t n3, Range rg,
                                                              T* __restrict__ vox, int32_t* __restrict__ ids) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n3; i += (int64_t)gridDim.x * blockDim.x) {
    const int a = (int)(i % 3);
    const float r = voxelise<T>(to_f32(xyz[i]), round_to<T>(rg.lo[a]), round_to<T>(rg.hi[a]), rg.voxel);
    if (vox) vox[i] = from_f32<T>(r);
    if (ids) ids[i] = (int32_t)r;
  }
}

// ----------------------------------------------------------------------------------------
// K3+K4  average_coordinate_in_patch + discrete_coords (llava_arch.py:213-223, 259-272)
//
// One workgroup = one (frame, patch-row): n patches x 3 channels = 3n running sums.  The
// reference (ATen avg_pool2d, CPU and CUDA alike) adds the patch*patch values of a window one
// by one in (row, column) order in f32; voxel ids are only bit-exact if we add in the same
// order, so the sum itself is a 729-long dependent chain per output.  To keep HBM busy anyway
// the whole workgroup streams rows of the strip into LDS with 16-byte loads (coalesced, full
// 128-B lines) a chunk of rows at a time, and lanes 0..3n-1 walk the chunk out of LDS.
// ----------------------------------------------------------------------------------------
constexpr int kPoolChunkRows = 9;

template <typename T, int PATCH>
__global__ __launch_bounds__(256) void coord_pool_voxel_kernel(const T* __restrict__ coords, int S, int patch_rt,
                                                               int n, Range rg, T* __restrict__ avg,
                                                               T* __restrict__ vox, int32_t* __restrict__ ids,
                                                               int vec_ok) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* buf = reinterpret_cast<float*>(smem_raw);
  const int patch = PATCH > 0 ? PATCH : patch_rt;
  const int v = blockIdx.y, py = blockIdx.x;
  const int row_elems = S * 3;              // full image row, the unused tail rides along
  const int tid = threadIdx.x;
  constexpr int VEC = 16 / sizeof(T);
  const T* strip = coords + ((size_t)v * S + (size_t)py * patch) * row_elems;

  const int lanes = n * 3;
  const int px = tid / 3, ch = tid - px * 3;
  float s = 0.0f;

  for (int r0 = 0; r0 < patch; r0 += kPoolChunkRows) {
    const int rows = (patch - r0) < kPoolChunkRows ? (patch - r0) : kPoolChunkRows;
    const T* src = strip + (size_t)r0 * row_elems;
    const int total = rows * row_elems;
    if (vec_ok) {
      const uint4* src4 = reinterpret_cast<const uint4*>(src);
      for (int i = tid; i < total / VEC; i += blockDim.x) {
        const uint4 raw = src4[i];
        float f[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) f[j] = vec_get<T>(raw, j);
        float4* d4 = reinterpret_cast<float4*>(buf + (size_t)i * VEC);
#pragma unroll
        for (int j = 0; j < VEC / 4; ++j) d4[j] = make_float4(f[4 * j], f[4 * j + 1], f[4 * j + 2], f[4 * j + 3]);
      }
    } else {
      for (int i = tid; i < total; i += blockDim.x) buf[i] = to_f32(src[i]);
    }
    __syncthreads();
    if (tid < lanes) {
      const float* p = buf + (px * patch) * 3 + ch;
      for (int r = 0; r < rows; ++r) {
        if (PATCH > 0) {
#pragma unroll
          for (int iw = 0; iw < (PATCH > 0 ? PATCH : 1); ++iw) s = __fadd_rn(s, p[iw * 3]);
        } else {
          for (int iw = 0; iw < patch; ++iw) s = __fadd_rn(s, p[iw * 3]);
        }
        p += row_elems;
      }
    }
    __syncthreads();
  }
  if (tid < lanes) {
    const float mean = round_to<T>(__fdiv_rn(s, (float)(patch * patch)));
    const size_t o = (((size_t)v * n + py) * n + px) * 3 + ch;
    if (avg) avg[o] = from_f32<T>(mean);
    if (vox || ids) {
      const float r = voxelise<T>(mean, round_to<T>(rg.lo[ch]), round_to<T>(rg.hi[ch]), rg.voxel);
      if (vox) vox[o] = from_f32<T>(r);
      if (ids) ids[o] = (int32_t)r;
    }
  }
}

}  // namespace v3d

using namespace v3d;

extern "C" int v3d_unproject_f32(const float* depth_mm, const float* intrinsics, const float* poses, float* world,
                                 int V, int H, int W, void* stream) {
  V3D_REQUIRE(depth_mm && intrinsics && poses && world, "v3d_unproject_f32: null pointer");
  V3D_REQUIRE(V > 0 && H > 0 && W > 0, "v3d_unproject_f32: bad shape V=%d H=%d W=%d", V, H, W);
  V3D_REQUIRE((int64_t)H * W < (1ll << 30), "v3d_unproject_f32: frame too large");
  hipStream_t st = (hipStream_t)stream;
  const bool vec = (W % 4 == 0) && aligned16(depth_mm) && aligned16(world);
  if (vec) {
    const int quads = H * W / 4;
    int bx = (quads + 255) / 256;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(unproject_f32_kernel, dim3(bx, V), dim3(256), 0, st, depth_mm, intrinsics, poses, world, H, W, quads);
  } else {
    int bx = (H * W + 255) / 256;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(unproject_f32_scalar_kernel, dim3(bx, V), dim3(256), 0, st, depth_mm, intrinsics, poses, world, H, W);
  }
  return check_launch("v3d_unproject_f32");
}

// a7: SigLipImageProcessor.preprocess (siglip_encoder.py:47-67) once the frame is 384 x 384 (VideoProcessor's crop,
// video_utils.py:292-308): rescale = f32(f64(u8) * (1/255)), normalize = (v - mean) / std in f32, HWC -> CHW.
// One thread = 4 pixels of one row (12 input bytes, three 4-element output runs).
template <typename T>
__global__ __launch_bounds__(256) void preprocess_rgb_kernel(const uint8_t* __restrict__ in, T* __restrict__ out, int64_t n_quads,
                                                             int HW, float m0, float m1, float m2, float s0, float s1, float s2,
                                                             double rescale) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= n_quads) return;
  const int64_t px = q * 4;                       // first pixel (HW % 4 == 0: the quad stays inside one frame)
  const int64_t f = px / HW, rem = px - f * HW;
  const uint32_t* src = reinterpret_cast<const uint32_t*>(in + px * 3);
  const uint32_t w0 = src[0], w1 = src[1], w2 = src[2];
  uint8_t b[12];
#pragma unroll
  for (int i = 0; i < 4; ++i) { b[i] = (w0 >> (8 * i)) & 0xff; b[4 + i] = (w1 >> (8 * i)) & 0xff; b[8 + i] = (w2 >> (8 * i)) & 0xff; }
  const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    T* dst = out + (f * 3 + c) * (int64_t)HW + rem;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float v = (float)((double)b[3 * i + c] * rescale);
      dst[i] = from_f32<T>((v - mean[c]) / sd[c]);
    }
  }
}

extern "C" int v3d_unproject_sampled_u16(const uint16_t* depth, const float* intrinsics, const float* poses,
                                         void* out, int out_dtype, int V, int H, int W, int crop, void* stream) {
  V3D_REQUIRE(depth && intrinsics && poses && out, "v3d_unproject_sampled_u16: null pointer");
  V3D_REQUIRE(V > 0 && H > 0 && W > 0 && crop > 0, "v3d_unproject_sampled_u16: bad shape");
  // llava/video_utils.py:297-304: new_height = crop; new_width = int(W * (crop / H)); centre crop
  const int new_h = crop;
  const int new_w = (int)((double)W * ((double)crop / (double)H));
  V3D_REQUIRE(new_w >= crop, "v3d_unproject_sampled_u16: resized width %d < crop %d", new_w, crop);
  const int left = (new_w - crop) / 2;
  const double inv_fx = (double)W / (double)new_w;   // cv::resize: scale_x = src.cols / dst.cols
  const double inv_fy = (double)H / (double)new_h;
  hipStream_t st = (hipStream_t)stream;
  if (crop % 8 == 0 && aligned16(out)) {
    const int groups = crop * crop / 8;
    V3D_DISPATCH_DTYPE(out_dtype,
                       hipLaunchKernelGGL(unproject_sampled_x8_kernel<T>, dim3((groups + 255) / 256, V), dim3(256), 0, st, depth, intrinsics,
                                          poses, (T*)out, H, W, crop, left, inv_fx, inv_fy, groups));
    return check_launch("v3d_unproject_sampled_u16");
  }
  int bx = (crop * crop + 255) / 256;
  if (bx > 1024) bx = 1024;
  V3D_DISPATCH_DTYPE(out_dtype,
                     hipLaunchKernelGGL(unproject_sampled_kernel<T>, dim3(bx, V), dim3(256), 0, st, depth, intrinsics,
                                        poses, (T*)out, H, W, crop, new_w, left, inv_fx, inv_fy));
  return check_launch("v3d_unproject_sampled_u16");
}

// VideoProcessor.preprocess, strategy "resize" (video_utils.py:293-296): cv2.resize(coords, (S, S), INTER_NEAREST) of the full-resolution
// coordinates without a crop - the same gather as the centre-crop form with both axes scaled independently (src = min(floor(dst * src_size /
// dst_size), src_size - 1); parity unpinned for the index rule itself, as for a6: cv2 is absent).
extern "C" int v3d_unproject_resized_u16(const uint16_t* depth, const float* intrinsics, const float* poses, void* out, int out_dtype, int V, int H,
                                         int W, int size, void* stream) {
  V3D_REQUIRE(depth && intrinsics && poses && out, "v3d_unproject_resized_u16: null pointer");
  V3D_REQUIRE(V > 0 && H > 0 && W > 0 && size > 0, "v3d_unproject_resized_u16: bad shape");
  const double inv_fx = (double)W / (double)size, inv_fy = (double)H / (double)size;
  hipStream_t st = (hipStream_t)stream;
  if (size % 8 == 0 && aligned16(out)) {
    const int groups = size * size / 8;
    V3D_DISPATCH_DTYPE(out_dtype, hipLaunchKernelGGL(unproject_sampled_x8_kernel<T>, dim3((groups + 255) / 256, V), dim3(256), 0, st, depth, intrinsics,
                                                     poses, (T*)out, H, W, size, 0, inv_fx, inv_fy, groups));
    return check_launch("v3d_unproject_resized_u16");
  }
  int bx = (size * size + 255) / 256;
  if (bx > 1024) bx = 1024;
  V3D_DISPATCH_DTYPE(out_dtype, hipLaunchKernelGGL(unproject_sampled_kernel<T>, dim3(bx, V), dim3(256), 0, st, depth, intrinsics, poses, (T*)out, H, W,
                                                   size, size, 0, inv_fx, inv_fy));
  return check_launch("v3d_unproject_resized_u16");
}

// calculate_world_coords(do_normalize=True) (video_utils.py:232-236, the "norm" frame-sampling strategies): every point clamped to the
// scene's box, x = min(max(x, lo), hi) per axis, in place.
template <typename T>
__global__ __launch_bounds__(256) void clamp_xyz_kernel(T* __restrict__ xyz, int64_t n_points, float lx, float ly, float lz, float hx, float hy, float hz) {
  const float lo[3] = {lx, ly, lz}, hi[3] = {hx, hy, hz};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_points * 3; i += (int64_t)gridDim.x * 256) {
    const int a = (int)(i % 3);
    const float v = to_f32(xyz[i]);
    xyz[i] = from_f32<T>(fminf(fmaxf(v, lo[a]), hi[a]));
  }
}

extern "C" int v3d_clamp_xyz(void* xyz, int64_t n_points, const float* lo_host, const float* hi_host, int dtype, void* stream) {
  V3D_REQUIRE(xyz && lo_host && hi_host && n_points >= 0, "v3d_clamp_xyz: bad arguments");
  if (n_points == 0) return V3D_OK;
  int64_t blocks = (n_points * 3 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  V3D_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(clamp_xyz_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (T*)xyz, n_points, lo_host[0],
                                               lo_host[1], lo_host[2], hi_host[0], hi_host[1], hi_host[2]));
  return check_launch("v3d_clamp_xyz");
}

constexpr int BOUNDS_BLOCKS = 64;       // per frame: 64 x 256 threads, ~19 pixels each at 480 x 640

extern "C" int64_t v3d_unproject_bounds_workspace_bytes(int V) { return (int64_t)(V > 0 ? V : 0) * BOUNDS_BLOCKS * 6 * (int64_t)sizeof(float); }

extern "C" int v3d_unproject_bounds_u16(const uint16_t* depth, const float* intrinsics, const float* poses, int V, int H, int W,
                                        float* bounds, void* workspace, int64_t workspace_bytes, void* stream) {
  V3D_REQUIRE(depth && intrinsics && poses && bounds && workspace, "v3d_unproject_bounds_u16: null pointer");
  V3D_REQUIRE(V > 0 && H > 0 && W > 0, "v3d_unproject_bounds_u16: bad shape");
  V3D_REQUIRE(workspace_bytes >= v3d_unproject_bounds_workspace_bytes(V), "v3d_unproject_bounds_u16: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(unproject_bounds_kernel, dim3(BOUNDS_BLOCKS, V), dim3(256), 0, st, depth, intrinsics, poses, H, W, (float*)workspace);
  if (int e = check_launch("v3d_unproject_bounds_u16")) return e;
  hipLaunchKernelGGL(bounds_fold_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, V * BOUNDS_BLOCKS, bounds);
  return check_launch("v3d_unproject_bounds_u16");
}

static int fill_range(Range& rg, const float* lo, const float* hi, float voxel, const char* who) {
  V3D_REQUIRE(lo && hi, "%s: min/max range pointers are null", who);
  V3D_REQUIRE(voxel > 0.0f, "%s: voxel_size must be > 0", who);
  for (int i = 0; i < 3; ++i) { rg.lo[i] = lo[i]; rg.hi[i] = hi[i]; }
  rg.voxel = voxel;
  return V3D_OK;
}

extern "C" int v3d_discrete_coords(const void* xyz, int dtype, int64_t N, const float* min_xyz_host,
                                   const float* max_xyz_host, float voxel_size, void* vox, int32_t* ids,
                                   void* stream) {
  V3D_REQUIRE(xyz && (vox || ids), "v3d_discrete_coords: null pointer");
  V3D_REQUIRE(N >= 0, "v3d_discrete_coords: negative N");
  if (N == 0) return V3D_OK;
  Range rg;
  if (int e = fill_range(rg, min_xyz_host, max_xyz_host, voxel_size, "v3d_discrete_coords")) return e;
  int64_t blocks = (N * 3 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  V3D_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(discrete_coords_kernel<T>, dim3((int)blocks), dim3(256), 0,
                                               (hipStream_t)stream, (const T*)xyz, N * 3, rg, (T*)vox, ids));
  return check_launch("v3d_discrete_coords");
}

extern "C" int v3d_coord_pool_voxel(const void* coords, int dtype, int V, int S, int patch,
                                    const float* min_xyz_host, const float* max_xyz_host, float voxel_size,
                                    void* avg, void* vox, int32_t* ids, void* stream) {
  V3D_REQUIRE(coords && (avg || vox || ids), "v3d_coord_pool_voxel: null pointer");
  V3D_REQUIRE(V > 0 && S > 0 && patch > 0 && patch <= S, "v3d_coord_pool_voxel: bad shape V=%d S=%d patch=%d", V, S, patch);
  // llava_arch.py:217: [:, :-6, :-6, :] then kernel=stride=patch -> n = floor((S-6)/patch)
  const int n = (S - 6) / patch;
  V3D_REQUIRE(n > 0 && n * 3 <= 256, "v3d_coord_pool_voxel: n=%d patches per row unsupported", n);
  Range rg;
  if (int e = fill_range(rg, min_xyz_host, max_xyz_host, voxel_size, "v3d_coord_pool_voxel")) return e;
  const size_t lds = (size_t)kPoolChunkRows * S * 3 * sizeof(float);
  V3D_REQUIRE(lds <= 160 * 1024, "v3d_coord_pool_voxel: S=%d needs %zu B of LDS", S, lds);
  const size_t esz = dtype == V3D_F32 ? 4 : 2;
  const int vec_ok = aligned16(coords) && ((size_t)S * 3 * esz) % 16 == 0;
  hipStream_t st = (hipStream_t)stream;
  V3D_DISPATCH_DTYPE(dtype, {
    auto k27 = coord_pool_voxel_kernel<T, 27>;
    auto kg = coord_pool_voxel_kernel<T, 0>;
    auto k = patch == 27 ? k27 : kg;
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) { set_error("v3d_coord_pool_voxel: LDS attribute: %s", hipGetErrorString(e)); return V3D_E_LAUNCH; }
    }
    hipLaunchKernelGGL(k, dim3(n, V), dim3(256), lds, st, (const T*)coords, S, patch, n, rg, (T*)avg, (T*)vox, ids, vec_ok);
  });
  return check_launch("v3d_coord_pool_voxel");
}

extern "C" int v3d_preprocess_rgb_u8(const uint8_t* frames, int F, int H, int W, const float* mean_host, const float* std_host,
                                     double rescale, void* out, int dtype, void* stream) {
  V3D_REQUIRE(frames && mean_host && std_host && out, "v3d_preprocess_rgb_u8: null pointer");
  V3D_REQUIRE(F > 0 && H > 0 && W > 0 && ((int64_t)H * W) % 4 == 0, "v3d_preprocess_rgb_u8: H*W must be a multiple of 4");
  V3D_REQUIRE((reinterpret_cast<uintptr_t>(frames) & 3) == 0, "v3d_preprocess_rgb_u8: frames must be 4-byte aligned");
  const int64_t n_quads = (int64_t)F * H * W / 4;
  const unsigned blocks = (unsigned)((n_quads + 255) / 256);
  V3D_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(preprocess_rgb_kernel<T>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, frames,
                                               (T*)out, n_quads, H * W, mean_host[0], mean_host[1], mean_host[2], std_host[0],
                                               std_host[1], std_host[2], rescale));
  return check_launch("v3d_preprocess_rgb_u8");
}

// ------------------------------------------------------------------------------------------
// a6 / a7, RGB half (r03): VideoProcessor.preprocess's `frame.resize((new_w, crop))` + centre crop (video_utils.py:285-306) on the
// device, bit for bit what Pillow's Image.resize does for 8-bit RGB with its default filter (BICUBIC): two passes, horizontal then
// vertical, over 22-bit fixed-point coefficients, the intermediate image rounded to 8 bits (libImaging Resample.c:
// ImagingResampleHorizontal_8bpc / Vertical_8bpc, clip8 = (ss >> 22) clamped to [0, 255], accumulators start at 1 << 21).  The
// coefficient tables are the caller's (v3d.ops.pil_resample_tables evaluates precompute_coeffs / normalize_coeffs_8bpc in
// double, as Pillow does) - integer arithmetic from there on, so there is nothing to round differently.  Optionally fused with
// SigLipImageProcessor's rescale / normalise / HWC -> CHW (v3d_preprocess_rgb_u8's arithmetic) so that the 8-bit crops never
// reach HBM.  One workgroup = a tile of RS_TW x th output pixels: the input rows x columns the tile's taps touch are staged in
// LDS with coalesced byte loads, pass 1 writes the 8-bit intermediate to LDS, pass 2 reads it back.
constexpr int RS_TW = 64, RS_RMAX = 64, RS_INB = 640, RS_KMAX = 32;

struct ResizeArgs {
  const uint8_t* src; void* out;
  const int32_t *bh, *kh, *bv, *kv;        // device tables: bounds [n][2] = (first tap, tap count), coefficients [n][ksize]
  int F, H, W, OH, OW, ksh, ksv;           // source size, resized size, table row lengths
  int top, left, ch_h, cw;                 // crop window inside the resized image
  int th;                                  // tile height
  float m0, m1, m2, s0, s1, s2;
  double rescale;
};

template <typename T, bool NORM>
__global__ __launch_bounds__(256) void resize_bicubic_kernel(ResizeArgs a) {
  __shared__ __attribute__((aligned(16))) uint8_t s_in[RS_RMAX * RS_INB];
  __shared__ uint8_t s_tmp[RS_RMAX * RS_TW * 3];
  __shared__ int32_t s_kh[RS_TW * RS_KMAX], s_kv[RS_RMAX * RS_KMAX];
  __shared__ int32_t s_bh[RS_TW * 2], s_bv[RS_RMAX * 2];
  const int tid = threadIdx.x, f = blockIdx.z;
  const int x0 = a.left + blockIdx.x * RS_TW, y0 = a.top + blockIdx.y * a.th;
  const int tw = min(RS_TW, a.left + a.cw - x0), th = min(a.th, a.top + a.ch_h - y0);
  // tables of the tile
  for (int i = tid; i < tw * 2; i += 256) s_bh[i] = a.bh[x0 * 2 + i];
  for (int i = tid; i < th * 2; i += 256) s_bv[i] = a.bv[y0 * 2 + i];
  for (int i = tid; i < tw * a.ksh; i += 256) s_kh[i] = a.kh[x0 * a.ksh + i];
  for (int i = tid; i < th * a.ksv; i += 256) s_kv[i] = a.kv[y0 * a.ksv + i];
  __syncthreads();
  const int r_lo = s_bv[0], r_hi = s_bv[2 * (th - 1)] + s_bv[2 * (th - 1) + 1];
  const int c_lo = s_bh[0], c_hi = s_bh[2 * (tw - 1)] + s_bh[2 * (tw - 1) + 1];
  const int R = r_hi - r_lo, nb = (c_hi - c_lo) * 3;
  if (R > RS_RMAX || nb + 4 > RS_INB || R <= 0 || nb <= 0) return;      // (the launcher sized the tile so that this cannot happen)
  // stage the input window: wave w takes rows w, w + 4, ...; lanes walk the row in 4-byte words from the word that holds the window's
  // first byte (rows of W * 3 bytes keep their 4-byte phase when W * 3 % 4 == 0: every ScanNet size; other widths go byte by byte)
  int shift = 0;
  {
    const int wave = tid >> 6, lane = tid & 63;
    const bool words = ((a.W * 3) & 3) == 0 && ((uintptr_t)a.src & 3) == 0;
    const int64_t total = (int64_t)a.F * a.H * a.W * 3;
    if (words) shift = (c_lo * 3) & 3;
    for (int r = wave; r < R; r += 4) {
      const int64_t off = (((int64_t)f * a.H + r_lo + r) * a.W + c_lo) * 3 - shift;      // multiple of 4 in the word form
      const uint8_t* srow = a.src + off;
      if (words) {
        const int ndw = (shift + nb + 3) >> 2;
        for (int dw = lane; dw < ndw; dw += 64) {
          uint32_t w;
          if (off + 4 * dw + 4 <= total) w = reinterpret_cast<const uint32_t*>(srow)[dw];
          else { w = 0; for (int k = 0; k < 4 && off + 4 * dw + k < total; ++k) w |= (uint32_t)srow[4 * dw + k] << (8 * k); }
          reinterpret_cast<uint32_t*>(s_in + r * RS_INB)[dw] = w;
        }
      } else {
        for (int bb = lane; bb < nb; bb += 64) s_in[r * RS_INB + bb] = srow[bb];
      }
    }
  }
  __syncthreads();
  // pass 1 (horizontal): one (row, output column) per thread and turn
  for (int idx = tid; idx < R * RS_TW; idx += 256) {
    const int r = idx >> 6, x = idx & (RS_TW - 1);
    if (x >= tw) continue;
    const int xmin = s_bh[2 * x] - c_lo, n = s_bh[2 * x + 1];
    const int32_t* k = s_kh + x * a.ksh;
    const uint8_t* px = s_in + r * RS_INB + shift + xmin * 3;
    int ss0 = 1 << 21, ss1 = 1 << 21, ss2 = 1 << 21;
    for (int t = 0; t < n; ++t) {
      const int c = k[t];
      ss0 += __mul24((int)px[3 * t], c); ss1 += __mul24((int)px[3 * t + 1], c); ss2 += __mul24((int)px[3 * t + 2], c);      // |c| < 2^23: full-rate 24-bit multiply
    }
    uint8_t* d = s_tmp + (r * RS_TW + x) * 3;
    d[0] = (uint8_t)min(255, max(0, ss0 >> 22)); d[1] = (uint8_t)min(255, max(0, ss1 >> 22)); d[2] = (uint8_t)min(255, max(0, ss2 >> 22));
  }
  __syncthreads();
  // pass 2 (vertical) + crop (+ rescale / normalise / CHW)
  const float mean[3] = {a.m0, a.m1, a.m2}, sd[3] = {a.s0, a.s1, a.s2};
  for (int idx = tid; idx < th * RS_TW; idx += 256) {
    const int y = idx >> 6, x = idx & (RS_TW - 1);
    if (x >= tw) continue;
    const int ymin = s_bv[2 * y] - r_lo, n = s_bv[2 * y + 1];
    const int32_t* k = s_kv + y * a.ksv;
    int ss[3] = {1 << 21, 1 << 21, 1 << 21};
    for (int t = 0; t < n; ++t) {
      const uint8_t* px = s_tmp + ((ymin + t) * RS_TW + x) * 3;
      const int c = k[t];
      ss[0] += __mul24((int)px[0], c); ss[1] += __mul24((int)px[1], c); ss[2] += __mul24((int)px[2], c);
    }
    const int oy = y0 + y - a.top, ox = x0 + x - a.left;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int u = min(255, max(0, ss[c] >> 22));
      if constexpr (NORM) {
        const float v = (float)((double)u * a.rescale);
        ((T*)a.out)[(((int64_t)f * 3 + c) * a.ch_h + oy) * a.cw + ox] = from_f32<T>((v - mean[c]) / sd[c]);
      } else {
        ((uint8_t*)a.out)[(((int64_t)f * a.ch_h + oy) * a.cw + ox) * 3 + c] = (uint8_t)u;
      }
    }
  }
}

extern "C" int v3d_resize_bicubic_u8(const uint8_t* frames, int F, int H, int W, int OH, int OW, const int32_t* bounds_h,
                                     const int32_t* coeffs_h, int ksize_h, const int32_t* bounds_v, const int32_t* coeffs_v, int ksize_v,
                                     int crop_top, int crop_left, int crop_h, int crop_w, const float* mean_host, const float* std_host,
                                     double rescale, void* out, int out_dtype, void* stream) {
  V3D_REQUIRE(frames && bounds_h && coeffs_h && bounds_v && coeffs_v && out, "v3d_resize_bicubic_u8: null pointer");
  V3D_REQUIRE(F > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, "v3d_resize_bicubic_u8: bad shape");
  V3D_REQUIRE(crop_top >= 0 && crop_left >= 0 && crop_h > 0 && crop_w > 0 && crop_top + crop_h <= OH && crop_left + crop_w <= OW,
              "v3d_resize_bicubic_u8: crop window (%d, %d, %d, %d) outside the %d x %d resized image", crop_top, crop_left, crop_h, crop_w, OH, OW);
  V3D_REQUIRE(ksize_h > 0 && ksize_h <= RS_KMAX && ksize_v > 0 && ksize_v <= RS_KMAX, "v3d_resize_bicubic_u8: at most %d taps per pass", RS_KMAX);
  V3D_REQUIRE(out_dtype == V3D_U8_HWC || mean_host && std_host, "v3d_resize_bicubic_u8: mean / std needed for a normalised output");
  // input window of a tile: (tile extent) x scale + taps; the tile shrinks until the window fits the LDS staging
  auto span = [](int n_out, int in, int out, int ks) { return (int)(((int64_t)n_out * in + out - 1) / out) + ks + 2; };
  V3D_REQUIRE(span(RS_TW, W, OW, ksize_h) * 3 + 4 <= RS_INB, "v3d_resize_bicubic_u8: horizontal reduction %d -> %d too strong for this kernel", W, OW);
  int th = 16;
  while (th > 1 && span(th, H, OH, ksize_v) > RS_RMAX) th >>= 1;
  V3D_REQUIRE(span(th, H, OH, ksize_v) <= RS_RMAX, "v3d_resize_bicubic_u8: vertical reduction %d -> %d too strong for this kernel", H, OH);
  ResizeArgs a;
  a.src = frames; a.out = out; a.bh = bounds_h; a.kh = coeffs_h; a.bv = bounds_v; a.kv = coeffs_v;
  a.F = F; a.H = H; a.W = W; a.OH = OH; a.OW = OW; a.ksh = ksize_h; a.ksv = ksize_v;
  a.top = crop_top; a.left = crop_left; a.ch_h = crop_h; a.cw = crop_w; a.th = th;
  a.m0 = a.m1 = a.m2 = 0.f; a.s0 = a.s1 = a.s2 = 1.f; a.rescale = rescale;
  if (mean_host && std_host) { a.m0 = mean_host[0]; a.m1 = mean_host[1]; a.m2 = mean_host[2]; a.s0 = std_host[0]; a.s1 = std_host[1]; a.s2 = std_host[2]; }
  const dim3 grid((crop_w + RS_TW - 1) / RS_TW, (crop_h + th - 1) / th, F);
  hipStream_t st = (hipStream_t)stream;
  switch (out_dtype) {
    case V3D_U8_HWC: hipLaunchKernelGGL((resize_bicubic_kernel<float, false>), grid, dim3(256), 0, st, a); break;
    case V3D_F32: hipLaunchKernelGGL((resize_bicubic_kernel<float, true>), grid, dim3(256), 0, st, a); break;
    case V3D_F16: hipLaunchKernelGGL((resize_bicubic_kernel<f16_t, true>), grid, dim3(256), 0, st, a); break;
    case V3D_BF16: hipLaunchKernelGGL((resize_bicubic_kernel<bf16_t, true>), grid, dim3(256), 0, st, a); break;
    default: set_error("v3d_resize_bicubic_u8: out_dtype %d", out_dtype); return V3D_E_INVALID;
  }
  return check_launch("v3d_resize_bicubic_u8");
}
