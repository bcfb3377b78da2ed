// Shared device helpers for the DeepCAM gfx950 kernels.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "../../include/deepcam_hip.h"

namespace dc {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// 16 bytes of storage, the unit every vectorised global / LDS access moves.
struct alignas(16) vec16 {
  uint32_t w[4];
};

template <typename T>
struct Elem;
template <>
struct Elem<float> {
  static constexpr int kPerVec = 4;  // elements per 16-byte vector
  static constexpr int kDtype = DC_F32;
  __device__ static inline float load(const float* p) { return *p; }
  __device__ static inline void store(float* p, float v) { *p = v; }
};
template <>
struct Elem<bf16> {
  static constexpr int kPerVec = 8;
  static constexpr int kDtype = DC_BF16;
  __device__ static inline float load(const bf16* p) { return (float)*p; }
  __device__ static inline void store(bf16* p, float v) { *p = (bf16)v; }
};

// unpack / pack one 16-byte vector <-> floats
__device__ inline void unpack(const vec16& v, float (&f)[4], float) {
#pragma unroll
  for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(v.w[i]);
}
__device__ inline void unpack(const vec16& v, float (&f)[8], bf16) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(v.w[i] << 16);
    f[2 * i + 1] = __uint_as_float(v.w[i] & 0xffff0000u);
  }
}
__device__ inline uint32_t pack2_bf16(float lo, float hi) {
  // plain casts: hipcc emits v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN preserving)
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  bf16x2 r;
  r[0] = (bf16)lo;
  r[1] = (bf16)hi;
  return __builtin_bit_cast(uint32_t, r);
}
__device__ inline void pack(vec16& v, const float (&f)[4], float) {
#pragma unroll
  for (int i = 0; i < 4; ++i) v.w[i] = __float_as_uint(f[i]);
}
__device__ inline void pack(vec16& v, const float (&f)[8], bf16) {
#pragma unroll
  for (int i = 0; i < 4; ++i) v.w[i] = pack2_bf16(f[2 * i], f[2 * i + 1]);
}

__device__ inline vec16 zero16() {
  vec16 z;
  z.w[0] = z.w[1] = z.w[2] = z.w[3] = 0u;
  return z;
}
__device__ inline vec16 ldg16(const void* p) { return *reinterpret_cast<const vec16*>(p); }
__device__ inline void stg16(void* p, const vec16& v) { *reinterpret_cast<vec16*>(p) = v; }

// wave-level sum over 64 lanes
__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace dc

// One-time, thread-safe set-up at a call site (the dynamic-LDS attribute of a kernel, a symbol address): the C ABI may be called from
// any host thread (autograd's worker thread beside the main thread), and a plain `static bool` could read true before the other
// thread's hipFuncSetAttribute had returned.
#define DC_ONCE(...)                               \
  do {                                             \
    static std::once_flag once_flag__;             \
    std::call_once(once_flag__, [&] { __VA_ARGS__; }); \
  } while (0)

#define DC_CHECK_LAUNCH()                                  \
  do {                                                     \
    hipError_t e__ = hipGetLastError();                    \
    if (e__ != hipSuccess) return dc_set_error(e__, __FILE__, __LINE__); \
  } while (0)

extern "C" int dc_set_error(int code, const char* file, int line);
extern "C" int dc_fail(const char* msg, const char* file, int line);
extern "C" int dc_check_view(const void* ptr, int ld, int c, int dtype, const char* what);
#define DC_REQUIRE(cond, msg)                                  \
  do {                                                         \
    if (!(cond)) return dc_fail(msg, __FILE__, __LINE__);      \
  } while (0)
