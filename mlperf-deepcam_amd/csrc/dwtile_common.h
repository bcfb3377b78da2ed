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


// Fused consumer of a data-gradient kernel: the BatchNorm whose (never stored) output fed this depthwise conv needs
// sum(g) and sum(g * xhat) per channel, g = dx masked by the ReLU (recomputed from the BatchNorm input y).  The kernel that
// produces dx takes the two sums on the way out: one slab row per pixel tile, [2][rows][C].
struct DwBnStats {
  const void* y;        // BatchNorm input (raw conv output), same shape as dx
  int ldy;
  const float* mean;
  const float* invstd;
  const float* mscale;  // forward scale / shift: the mask is (y*mscale + mshift > 0)
  const float* mshift;
  int relu;
  float* slab;          // null: statistics off
  int rows;
  // stride-1 tiled kernel only: the depthwise WEIGHT gradient rides along too (dW[t] = sum_q xhat[q] * dy[q - t]: the dy window is the
  // one the data gradient reads, xhat = act(y*mscale + mshift) is recomputed from the y this epilogue loads anyway), one row of
  // [9][C] partial sums per pixel tile into wslab; null: off
  float* wslab;
};

template <int KH>
struct BnAcc {
  float mu[KH], is[KH], ms[KH], mh[KH], a[KH], b[KH];
  __device__ inline void init(const DwBnStats& st, int ch0) {
    const bool affine = st.relu || st.wslab != nullptr;     // the fused weight gradient needs act(y*ms + mh) even without a ReLU
#pragma unroll
    for (int e = 0; e < KH; ++e) {
      mu[e] = st.mean[ch0 + e];
      is[e] = st.invstd[ch0 + e];
      ms[e] = affine ? st.mscale[ch0 + e] : 0.f;
      mh[e] = affine ? st.mshift[ch0 + e] : 0.f;
      a[e] = b[e] = 0.f;
    }
  }
  // weight gradient only (no statistics): xhat = act(y*ms + mh), the identity when the vectors are absent (a stored input tensor)
  __device__ inline void init_affine(const DwBnStats& st, int ch0) {
#pragma unroll
    for (int e = 0; e < KH; ++e) {
      mu[e] = is[e] = a[e] = b[e] = 0.f;
      ms[e] = st.mscale != nullptr ? st.mscale[ch0 + e] : 1.f;
      mh[e] = st.mshift != nullptr ? st.mshift[ch0 + e] : 0.f;
    }
  }
  // g: the gradient values as STORED (already rounded), yv: the BatchNorm input at the same pixel
  __device__ inline void add(const float (&g)[KH], const float (&yv)[KH], int relu) {
#pragma unroll
    for (int e = 0; e < KH; ++e) {
      const float gm = (!relu || fmaf(yv[e], ms[e], mh[e]) > 0.f) ? g[e] : 0.f;
      a[e] += gm;
      b[e] = fmaf(gm, (yv[e] - mu[e]) * is[e], b[e]);
    }
  }
};

// fold the strip lanes of a workgroup through LDS and write the tile's slab row; call with ALL threads, after a barrier that
// ends the reads of the staged tile.  red needs NSL*2*CW floats.
template <int KH, int NSL, int CW>
__device__ inline void bn_acc_store(float* red, const float (&a)[KH], const float (&b)[KH], bool cok, int h, int sl, const DwBnStats& st,
                                    int tile, int c_base, int C) {
  if (cok) {
#pragma unroll
    for (int e = 0; e < KH; ++e) {
      red[(sl * 2 + 0) * CW + h * KH + e] = a[e];
      red[(sl * 2 + 1) * CW + h * KH + e] = b[e];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * CW; i += 256) {
    const int which = i / CW, cl = i % CW;
    if (c_base + cl < C) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < NSL; ++q) s += red[(q * 2 + which) * CW + cl];
      st.slab[((size_t)which * st.rows + tile) * C + c_base + cl] = s;
    }
  }
}

// Fused producer for the tiled kernels: the staged tile holds the RAW (pre-BatchNorm) conv output; every in-image element becomes
// v = x*scale[c] + shift[c] (and ReLU) in place, the zero padding stays zero.  One pass over the LDS tile (each element once,
// where the register-window kernels recomputed it per window position); a thread's channel group is fixed, so its 8 (4)
// coefficients live in registers.  Call between two workgroup barriers.
template <typename T, int HH, int HW, int CG>
__device__ inline void bn_transform_tile(char* smem, int iy0, int ix0, int H, int W, const float* __restrict__ pscale,
                                         const float* __restrict__ pshift, int prelu, int cg0, int ngroups, int cbase = 0) {
  constexpr int KPV = Elem<T>::kPerVec;
  constexpr int HP = HH * HW, ITER = (HP * CG + 255) / 256;
  const int tid = threadIdx.x, g = tid % CG;
  if (cg0 + g >= ngroups) return;
  float sc[KPV], sh[KPV];        // (cbase: the vectors start at channel cbase -- a workgroup's own coefficients in LDS)
#pragma unroll
  for (int e = 0; e < KPV; ++e) {
    sc[e] = pscale[(cg0 + g) * KPV + e - cbase];
    sh[e] = pshift[(cg0 + g) * KPV + e - cbase];
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
