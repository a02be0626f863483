// Shared device/host helpers for libv3d_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "v3d.h"

namespace v3d {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return V3D_E_LAUNCH;
  }
  return V3D_OK;
}

#define V3D_REQUIRE(cond, ...)          \
  do {                                  \
    if (!(cond)) {                      \
      v3d::set_error(__VA_ARGS__);      \
      return V3D_E_INVALID;             \
    }                                   \
  } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- 16-bit storage types as raw bits --------------------------------------------------
struct f16_t { uint16_t v; };
struct bf16_t { uint16_t v; };

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(f16_t x) { return __half2float(__ushort_as_half(x.v)); }
__device__ __forceinline__ float to_f32(bf16_t x) { return __uint_as_float(((uint32_t)x.v) << 16); }

template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float x) {
  f16_t r; r.v = __half_as_ushort(__float2half_rn(x)); return r;
}
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) {
  // plain cast: round-to-nearest-even, NaN stays NaN; lowers to ONE v_cvt_pk_bf16_f32 on gfx950
  bf16_t r;
  r.v = __builtin_bit_cast(uint16_t, static_cast<__bf16>(x));
  return r;
}

// round an f32 value to storage type T and widen it again
template <typename T> __device__ __forceinline__ float round_to(float x) { return to_f32(from_f32<T>(x)); }

template <typename T> struct Vec16 {  // one 16-byte vector of T
  static constexpr int N = 16 / sizeof(T);
  uint4 raw;
};

template <typename T> __device__ __forceinline__ float vec_get(const uint4& r, int i);
template <> __device__ __forceinline__ float vec_get<float>(const uint4& r, int i) {
  const uint32_t w = i == 0 ? r.x : i == 1 ? r.y : i == 2 ? r.z : r.w;
  return __uint_as_float(w);
}
template <> __device__ __forceinline__ float vec_get<f16_t>(const uint4& r, int i) {
  const uint32_t w = (i >> 1) == 0 ? r.x : (i >> 1) == 1 ? r.y : (i >> 1) == 2 ? r.z : r.w;
  f16_t h; h.v = (uint16_t)((i & 1) ? (w >> 16) : (w & 0xffffu));
  return to_f32(h);
}
template <> __device__ __forceinline__ float vec_get<bf16_t>(const uint4& r, int i) {
  const uint32_t w = (i >> 1) == 0 ? r.x : (i >> 1) == 1 ? r.y : (i >> 1) == 2 ? r.z : r.w;
  return __uint_as_float((i & 1) ? (w & 0xffff0000u) : (w << 16));
}

template <typename T> __device__ __forceinline__ uint32_t pack2(float lo, float hi);
template <> __device__ __forceinline__ uint32_t pack2<f16_t>(float lo, float hi) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const h2 v = {static_cast<_Float16>(lo), static_cast<_Float16>(hi)};
  return __builtin_bit_cast(uint32_t, v);
}
template <> __device__ __forceinline__ uint32_t pack2<bf16_t>(float lo, float hi) {
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  const b2 v = {static_cast<__bf16>(lo), static_cast<__bf16>(hi)};     // one v_cvt_pk_bf16_f32
  return __builtin_bit_cast(uint32_t, v);
}

// the two 16-bit values of one packed 32-bit word, widened to f32
template <typename T> __device__ __forceinline__ float pair_lo(uint32_t w);
template <typename T> __device__ __forceinline__ float pair_hi(uint32_t w);
template <> __device__ __forceinline__ float pair_lo<bf16_t>(uint32_t w) { return __uint_as_float(w << 16); }
template <> __device__ __forceinline__ float pair_hi<bf16_t>(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
template <> __device__ __forceinline__ float pair_lo<f16_t>(uint32_t w) { f16_t h; h.v = (uint16_t)(w & 0xffffu); return to_f32(h); }
template <> __device__ __forceinline__ float pair_hi<f16_t>(uint32_t w) { f16_t h; h.v = (uint16_t)(w >> 16); return to_f32(h); }

template <typename T> __device__ __forceinline__ uint4 vec_pack(const float* v);
template <> __device__ __forceinline__ uint4 vec_pack<float>(const float* v) {
  return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
}
template <> __device__ __forceinline__ uint4 vec_pack<f16_t>(const float* v) {
  return make_uint4(pack2<f16_t>(v[0], v[1]), pack2<f16_t>(v[2], v[3]), pack2<f16_t>(v[4], v[5]), pack2<f16_t>(v[6], v[7]));
}
template <> __device__ __forceinline__ uint4 vec_pack<bf16_t>(const float* v) {
  return make_uint4(pack2<bf16_t>(v[0], v[1]), pack2<bf16_t>(v[2], v[3]), pack2<bf16_t>(v[4], v[5]), pack2<bf16_t>(v[6], v[7]));
}

}  // namespace v3d

#define V3D_DISPATCH_DTYPE(dtype, ...)                                        \
  switch (dtype) {                                                            \
    case V3D_F32: { using T = float; __VA_ARGS__; } break;                    \
    case V3D_F16: { using T = v3d::f16_t; __VA_ARGS__; } break;               \
    case V3D_BF16: { using T = v3d::bf16_t; __VA_ARGS__; } break;             \
    default: v3d::set_error("unknown dtype code %d", (int)(dtype)); return V3D_E_INVALID; \
  }
