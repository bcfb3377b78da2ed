// Device helpers shared by the LDS-tiled depthwise kernels (dwtile.hip: stride 1, dwtile_s2.hip: stride 2).
#pragma once
#include "common.h"

namespace dc {

constexpr int DT_PX = 4;   // pixels per strip

// XCD-aware bijective remap of the 1-D grid (see igemm.hip), then tile id -> (channel block fastest, tx, ty, n)
__device__ inline int xcd_remap(int bid, int nwg) {
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, slot = bid >> 3;
  return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
}

// A thread computes on HALF a channel group (8 bytes: 4 bf16 / 2 f32 channels): its nine taps then take 36 (18) registers
// instead of 72, which is what decides the occupancy of these kernels; LDS is read with ds_read_b64 (same bytes per clock as
// b128), global memory is still filled / written in full 512-byte runs per pixel.
struct alignas(8) vec8 {
  uint32_t w[2];
};
__device__ inline void unpack8(const vec8& v, float (&f)[2], float) {
  f[0] = __uint_as_float(v.w[0]);
  f[1] = __uint_as_float(v.w[1]);
}
__device__ inline void unpack8(const vec8& v, float (&f)[4], bf16) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    f[2 * i] = __uint_as_float(v.w[i] << 16);
    f[2 * i + 1] = __uint_as_float(v.w[i] & 0xffff0000u);
  }
}
__device__ inline void pack8(vec8& v, const float (&f)[2], float) {
  v.w[0] = __float_as_uint(f[0]);
  v.w[1] = __float_as_uint(f[1]);
}
__device__ inline void pack8(vec8& v, const float (&f)[4], bf16) {
  v.w[0] = pack2_bf16(f[0], f[1]);
  v.w[1] = pack2_bf16(f[2], f[3]);
}

template <int KH>
__device__ inline void load_taps(const float* __restrict__ wp, int ch0, int C, bool flip, float (&wk)[9][KH]) {
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float* src = wp + (size_t)(flip ? 8 - t : t) * C + ch0;
    if constexpr (KH == 4) {
      const float4 v = *reinterpret_cast<const float4*>(src);
      wk[t][0] = v.x; wk[t][1] = v.y; wk[t][2] = v.z; wk[t][3] = v.w;
    } else {
      const float2 v = *reinterpret_cast<const float2*>(src);
      wk[t][0] = v.x; wk[t][1] = v.y;
    }
  }
}


// Fused producer for the tiled kernels: the staged tile holds the RAW (pre-BatchNorm) conv output; every in-image element becomes
// v = x*scale[c] + shift[c] (and ReLU) in place, the zero padding stays zero.  One pass over the LDS tile (each element once,
// where the register-window kernels recomputed it per window position); a thread's channel group is fixed, so its 8 (4)
// coefficients live in registers.  Call between two workgroup barriers.
template <typename T, int HH, int HW, int CG>
__device__ inline void bn_transform_tile(char* smem, int iy0, int ix0, int H, int W, const float* __restrict__ pscale,
                                         const float* __restrict__ pshift, int prelu, int cg0, int ngroups) {
  constexpr int KPV = Elem<T>::kPerVec;
  constexpr int HP = HH * HW, ITER = (HP * CG + 255) / 256;
  const int tid = threadIdx.x, g = tid % CG;
  if (cg0 + g >= ngroups) return;
  float sc[KPV], sh[KPV];
#pragma unroll
  for (int e = 0; e < KPV; ++e) {
    sc[e] = pscale[(cg0 + g) * KPV + e];
    sh[e] = pshift[(cg0 + g) * KPV + e];
  }
#pragma unroll 2
  for (int it = 0; it < ITER; ++it) {
    const int slot = it * 256 + tid;
    const int hp = slot / CG;
    const int hy = hp / HW, hx = hp - hy * HW;
    const int iy = iy0 + hy, ix = ix0 + hx;
    if (hp < HP && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
      vec16* q = reinterpret_cast<vec16*>(smem) + slot;
      float f[KPV];
      unpack(*q, f, T());
#pragma unroll
      for (int e = 0; e < KPV; ++e) {
        const float v = fmaf(f[e], sc[e], sh[e]);
        f[e] = prelu ? fmaxf(v, 0.f) : v;
      }
      vec16 o;
      pack(o, f, T());
      *q = o;
    }
  }
}

}  // namespace dc
