// Depthwise 3x3 convolution of SeparableConv2d_same (fixed_padding + groups=C conv), NHWC, HBM-bound.
// One lane owns one 16-byte channel group (8 bf16 / 4 f32 channels) so every global access is a full vector and
// a wave covers up to 1 KiB of contiguous channels; each thread walks PX output pixels along W with the nine
// per-channel weights held in registers.  Forward / data-gradient read the weights repacked as [9][C] fp32.
#include <string.h>

#include "common.h"
#include "dwtile.h"
#include "bn_fin.h"
#include "dwtile_common.h"

namespace dc {

static int g_dw_tile = 1;   // 1: LDS-tiled stride-1 kernels (dwtile.hip); 0: the register-window kernels below

constexpr int DW_PX = 4;  // output pixels per thread (along W)

// weights arrive packed as [9][C] fp32 (dc_dwconv_pack_weights): one thread's 8 (4) channels of a tap are 1-2 vector loads
template <typename T>
__device__ inline void load_w9(const float* __restrict__ wp, int c0, int C, float (&wr)[9][Elem<T>::kPerVec]) {
  constexpr int KPV = Elem<T>::kPerVec;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float4* src = reinterpret_cast<const float4*>(wp + (size_t)t * C + c0);
#pragma unroll
    for (int q = 0; q < KPV / 4; ++q) {
      const float4 v = src[q];
      wr[t][4 * q] = v.x; wr[t][4 * q + 1] = v.y; wr[t][4 * q + 2] = v.z; wr[t][4 * q + 3] = v.w;
    }
  }
}

// Fused producer: when pscale != nullptr the depthwise input is the RAW (pre-BatchNorm) conv output and the kernel applies
// v = x*scale[c] + shift[c] (and ReLU) to every in-bounds element as it is loaded; the zero padding stays zero.  The
// normalised activation is then never materialised in HBM.
template <int KPV>
__device__ inline void dw_prologue(float (&f)[KPV], const float* __restrict__ pscale, const float* __restrict__ pshift, int prelu, int c0) {
#pragma unroll
  for (int e = 0; e < KPV; ++e) {
    float v = fmaf(f[e], pscale[c0 + e], pshift[c0 + e]);
    f[e] = prelu ? fmaxf(v, 0.f) : v;
  }
}

__global__ void dw_pack_kernel(const float* __restrict__ master, float* __restrict__ packed, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 9 * C) return;
  const int c = i / 9, t = i % 9;
  packed[(size_t)t * C + c] = master[i];
}

// mode 0: forward   out[n,oy,ox] = sum_t in[n, oy*s - d + ky*d, ox*s - d + kx*d] * w[t]
// mode 1: dgrad     out[n,iy,ix] = sum_t in[n, (iy + d - ky*d)/s, (ix + d - kx*d)/s] * w[t]   (when divisible) [+ addend]
template <typename T, int MODE>
__global__ __launch_bounds__(256) void dw_kernel(const T* __restrict__ in, int ldin, const float* __restrict__ w,
                                                 const T* __restrict__ addend, int ldadd, T* __restrict__ out,
                                                 int ldout, int N, int Hin, int Win, int Hout, int Wout, int C,
                                                 int stride, int dil, const float* __restrict__ pscale,
                                                 const float* __restrict__ pshift, int prelu) {
  constexpr int KPV = Elem<T>::kPerVec;
  const int ngroups = C / KPV;
  const int wq = (Wout + DW_PX - 1) / DW_PX;
  const long total = (long)N * Hout * wq * ngroups;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % ngroups);
    long r = idx / ngroups;
    const int xq = (int)(r % wq);
    r /= wq;
    const int oy = (int)(r % Hout);
    const int n = (int)(r / Hout);
    const int c0 = cg * KPV;
    float wr[9][KPV];
    load_w9<T>(w, c0, C, wr);
#pragma unroll
    for (int j = 0; j < DW_PX; ++j) {
      const int ox = xq * DW_PX + j;
      if (ox >= Wout) break;
      float acc[KPV];
#pragma unroll
      for (int e = 0; e < KPV; ++e) acc[e] = 0.f;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        int iy;
        bool yok;
        if (MODE == 0) {
          iy = oy * stride - dil + ky * dil;
          yok = (unsigned)iy < (unsigned)Hin;
        } else {
          const int a = oy + dil - ky * dil;
          yok = a >= 0 && (a % stride) == 0;
          iy = a / stride;
          yok = yok && iy < Hin;
        }
        if (!yok) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          int ix;
          bool xok;
          if (MODE == 0) {
            ix = ox * stride - dil + kx * dil;
            xok = (unsigned)ix < (unsigned)Win;
          } else {
            const int a = ox + dil - kx * dil;
            xok = a >= 0 && (a % stride) == 0;
            ix = a / stride;
            xok = xok && ix < Win;
          }
          if (!xok) continue;
          float f[KPV];
          unpack(ldg16(in + ((size_t)(n * Hin + iy) * Win + ix) * ldin + c0), f, T());
          if (MODE == 0 && pscale != nullptr) dw_prologue<KPV>(f, pscale, pshift, prelu, c0);
#pragma unroll
          for (int e = 0; e < KPV; ++e) acc[e] = fmaf(f[e], wr[ky * 3 + kx][e], acc[e]);
        }
      }
      const size_t opix = (size_t)(n * Hout + oy) * Wout + ox;
      if (MODE == 1 && addend != nullptr) {
        float f[KPV];
        unpack(ldg16(addend + opix * ldadd + c0), f, T());
#pragma unroll
        for (int e = 0; e < KPV; ++e) acc[e] += f[e];
      }
      vec16 v;
      pack(v, acc, T());
      stg16(out + opix * ldout + c0, v);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Stride-1 fast path (60 of the 63 depthwise layers).  The generic kernel above is latency-bound: it issues nine dependent
// batches of loads per pixel.  Here a thread owns DW_PX=4 consecutive output pixels of one channel group and first issues
// ALL loads of their 3 x (4+2*DIL) input window (18 or 24 independent 16-byte loads in flight), then does the arithmetic.
// The data gradient of a stride-1 depthwise conv is the same stencil with the taps reversed (FLIP) plus the optional addend.
template <typename T, int DIL, bool FLIP>
__global__ __launch_bounds__(256) void dw_s1_kernel(const T* __restrict__ in, int ldin, const float* __restrict__ wp,
                                                    const T* __restrict__ addend, int ldadd, T* __restrict__ out, int ldout,
                                                    int N, int H, int W, int C, const float* __restrict__ pscale,
                                                    const float* __restrict__ pshift, int prelu) {
  constexpr int KPV = Elem<T>::kPerVec;
  constexpr int WC = DW_PX + 2 * DIL;  // window columns
  const int ngroups = C / KPV;
  const int wq = (W + DW_PX - 1) / DW_PX;
  const int total = N * H * wq * ngroups;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int cg = idx % ngroups;
  int r = idx / ngroups;
  const int xq = r % wq;
  r /= wq;
  const int oy = r % H;
  const int n = r / H;
  const int c0 = cg * KPV;
  const int x0 = xq * DW_PX;
  vec16 win[3][WC];
  unsigned inb = 0;   // bit (ky*WC + c): window element is inside the image
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = oy - DIL + ky * DIL;
    const bool yok = (unsigned)iy < (unsigned)H;
    const T* rowp = in + ((size_t)(n * H + (yok ? iy : 0)) * W) * ldin + c0;
#pragma unroll
    for (int c = 0; c < WC; ++c) {
      const int ix = x0 - DIL + c;
      const bool ok = yok && (unsigned)ix < (unsigned)W;
      win[ky][c] = ok ? ldg16(rowp + (size_t)ix * ldin) : zero16();
      inb |= (ok ? 1u : 0u) << (ky * WC + c);
    }
  }
  float acc[DW_PX][KPV];
#pragma unroll
  for (int j = 0; j < DW_PX; ++j)
#pragma unroll
    for (int e = 0; e < KPV; ++e) acc[j][e] = 0.f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    float wk[3][KPV];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int t = FLIP ? 8 - (ky * 3 + kx) : ky * 3 + kx;
      const float4* src = reinterpret_cast<const float4*>(wp + (size_t)t * C + c0);
#pragma unroll
      for (int q = 0; q < KPV / 4; ++q) {
        const float4 v = src[q];
        wk[kx][4 * q] = v.x; wk[kx][4 * q + 1] = v.y; wk[kx][4 * q + 2] = v.z; wk[kx][4 * q + 3] = v.w;
      }
    }
#pragma unroll
    for (int c = 0; c < WC; ++c) {
      float f[KPV];
      unpack(win[ky][c], f, T());
      if (!FLIP && pscale != nullptr && ((inb >> (ky * WC + c)) & 1u)) dw_prologue<KPV>(f, pscale, pshift, prelu, c0);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int j = c - kx * DIL;          // output pixel that sees window column c through tap kx
        if (j >= 0 && j < DW_PX) {
#pragma unroll
          for (int e = 0; e < KPV; ++e) acc[j][e] = fmaf(f[e], wk[kx][e], acc[j][e]);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < DW_PX; ++j) {
    const int ox = x0 + j;
    if (ox < W) {
      const size_t opix = (size_t)(n * H + oy) * W + ox;
      if (FLIP && addend != nullptr) {
        float f[KPV];
        unpack(ldg16(addend + opix * ldadd + c0), f, T());
#pragma unroll
        for (int e = 0; e < KPV; ++e) acc[j][e] += f[e];
      }
      vec16 v;
      pack(v, acc[j], T());
      stg16(out + opix * ldout + c0, v);
    }
  }
}

// Stride-1 weight gradient: a thread owns 4 consecutive pixels x one channel group, loads their dy and the 3 x (4+2*DIL)
// window of x at once, and adds the 9 x KPV products; lanes of a wave = consecutive channel groups (x pixel strips), the
// strips of a block are folded through LDS, one partial row per block goes to the slab.
template <typename T, int DIL>
__global__ __launch_bounds__(256) void dw_s1_wgrad_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ dy, int lddy,
                                                          float* __restrict__ slab, int N, int H, int W, int C, int strips_per_block,
                                                          int cgw, const float* __restrict__ pscale,
                                                          const float* __restrict__ pshift, int prelu) {
  constexpr int KPV = Elem<T>::kPerVec;
  constexpr int WC = DW_PX + 2 * DIL;
  extern __shared__ __attribute__((aligned(16))) float red[];   // [256/cgw][9][cgw]
  const int ngroups = C / KPV;
  const int cgl = threadIdx.x % cgw, sl = threadIdx.x / cgw;   // lane -> (channel group, strip lane)
  const int nsl = 256 / cgw;
  const int cg = blockIdx.x * cgw + cgl;
  const int c0 = cg * KPV;
  const bool cok = cg < ngroups;
  const int wq = (W + DW_PX - 1) / DW_PX;
  const int nstrips = N * H * wq;
  const int sbeg = blockIdx.y * strips_per_block;
  const int send = min(nstrips, sbeg + strips_per_block);
  float acc[9][KPV];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < KPV; ++e) acc[t][e] = 0.f;
  if (cok) {
    for (int sidx = sbeg + sl; sidx < send; sidx += nsl) {
      const int xq = sidx % wq;
      int r = sidx / wq;
      const int oy = r % H;
      const int n = r / H;
      const int x0 = xq * DW_PX;
      vec16 g[DW_PX], win[3][WC];
      unsigned inb = 0;
#pragma unroll
      for (int j = 0; j < DW_PX; ++j)
        g[j] = (x0 + j < W) ? ldg16(dy + ((size_t)(n * H + oy) * W + x0 + j) * lddy + c0) : zero16();
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy - DIL + ky * DIL;
        const bool yok = (unsigned)iy < (unsigned)H;
        const T* rowp = x + ((size_t)(n * H + (yok ? iy : 0)) * W) * ldx + c0;
#pragma unroll
        for (int c = 0; c < WC; ++c) {
          const int ix = x0 - DIL + c;
          const bool ok = yok && (unsigned)ix < (unsigned)W;
          win[ky][c] = ok ? ldg16(rowp + (size_t)ix * ldx) : zero16();
          inb |= (ok ? 1u : 0u) << (ky * WC + c);
        }
      }
      float gf[DW_PX][KPV];
#pragma unroll
      for (int j = 0; j < DW_PX; ++j) unpack(g[j], gf[j], T());
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int c = 0; c < WC; ++c) {
          float f[KPV];
          unpack(win[ky][c], f, T());
          if (pscale != nullptr && ((inb >> (ky * WC + c)) & 1u)) dw_prologue<KPV>(f, pscale, pshift, prelu, c0);
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int j = c - kx * DIL;
            if (j >= 0 && j < DW_PX) {
#pragma unroll
              for (int e = 0; e < KPV; ++e) acc[ky * 3 + kx][e] = fmaf(gf[j][e], f[e], acc[ky * 3 + kx][e]);
            }
          }
        }
    }
  }
  // fold the strip lanes of the block, one channel element at a time
#pragma unroll
  for (int e = 0; e < KPV; ++e) {
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 9; ++t) red[(sl * 9 + t) * cgw + cgl] = acc[t][e];
    __syncthreads();
    for (int i = threadIdx.x; i < 9 * cgw; i += 256) {
      const int t = i / cgw, l = i % cgw;
      const int c = (blockIdx.x * cgw + l) * KPV + e;
      if (c < C) {
        float a = 0.f;
        for (int q = 0; q < nsl; ++q) a += red[(q * 9 + t) * cgw + l];
        slab[((size_t)blockIdx.y * 9 + t) * C + c] = a;
      }
    }
  }
}

// Weight gradient: dw[c][t] = sum over output pixels of dy[n,oy,ox,c] * x[n, oy*s - d + ky*d, ox*s - d + kx*d, c].
// A wave is tiled as CGW channel groups x (64/CGW) pixels so that thin layers (128 channels = 16 groups) still use all 64
// lanes; the pixel sub-lanes are folded with shuffles, the block's 4 waves through LDS, and every block leaves one
// partial row in slab[pixel block][9][C] (reduced in a fixed order afterwards: deterministic, no atomics).
template <typename T, int CGW>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ dy,
                                                       int lddy, float* __restrict__ slab, int N, int Hi, int Wi,
                                                       int Ho, int Wo, int C, int stride, int dil, int pix_per_block,
                                                       const float* __restrict__ pscale, const float* __restrict__ pshift,
                                                       int prelu) {
  constexpr int KPV = Elem<T>::kPerVec;
  constexpr int PPW = 64 / CGW;
  __shared__ float red[4][9][CGW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cgl = lane % CGW, psub = lane / CGW;
  const int cg = blockIdx.x * CGW + cgl;
  const int c0 = cg * KPV;
  const bool cok = c0 < C;
  const long P = (long)N * Ho * Wo;
  const long pbeg = (long)blockIdx.y * pix_per_block;
  const long pend = pbeg + pix_per_block < P ? pbeg + pix_per_block : P;
  float acc[9][KPV];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < KPV; ++e) acc[t][e] = 0.f;
  if (cok) {
    for (long pix = pbeg + wave * PPW + psub; pix < pend; pix += 4 * PPW) {
      const int ox = (int)(pix % Wo);
      const long r = pix / Wo;
      const int oy = (int)(r % Ho);
      const int n = (int)(r / Ho);
      float g[KPV];
      unpack(ldg16(dy + (size_t)pix * lddy + c0), g, T());
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * stride - dil + ky * dil;
        if ((unsigned)iy >= (unsigned)Hi) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int ix = ox * stride - dil + kx * dil;
          if ((unsigned)ix >= (unsigned)Wi) continue;
          float f[KPV];
          unpack(ldg16(x + ((size_t)(n * Hi + iy) * Wi + ix) * ldx + c0), f, T());
          if (pscale != nullptr) dw_prologue<KPV>(f, pscale, pshift, prelu, c0);
#pragma unroll
          for (int e = 0; e < KPV; ++e) acc[ky * 3 + kx][e] = fmaf(g[e], f[e], acc[ky * 3 + kx][e]);
        }
      }
    }
  }
  // fold the pixel sub-lanes of the wave
#pragma unroll
  for (int off = CGW; off < 64; off <<= 1)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int e = 0; e < KPV; ++e) acc[t][e] += __shfl_xor(acc[t][e], off, 64);
  // combine the 4 waves, one channel element at a time
#pragma unroll
  for (int e = 0; e < KPV; ++e) {
    __syncthreads();
    if (psub == 0)
#pragma unroll
      for (int t = 0; t < 9; ++t) red[wave][t][cgl] = acc[t][e];
    __syncthreads();
    for (int i = threadIdx.x; i < 9 * CGW; i += 256) {
      const int t = i / CGW, l = i % CGW;
      const int c = (blockIdx.x * CGW + l) * KPV + e;
      if (c < C) slab[((size_t)blockIdx.y * 9 + t) * C + c] = red[0][t][l] + red[1][t][l] + red[2][t][l] + red[3][t][l];
    }
  }
}

// grad[c][t] = sum over rows of slab[row][t][c]: 8 columns x 32 row-lanes per block, fp64, fixed order
__global__ __launch_bounds__(256) void dw_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ grad, int rows, int C) {
  __shared__ double red[32][8];
  const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int n = 9 * C;
  const int i = blockIdx.x * 8 + cl;
  double a = 0.0;
  if (i < n)
    for (int r = rl; r < rows; r += 32) a += (double)slab[(size_t)r * n + i];
  red[rl][cl] = a;
  __syncthreads();
  if (threadIdx.x < 8 && blockIdx.x * 8 + threadIdx.x < n) {
    const int j = blockIdx.x * 8 + threadIdx.x;
    double s2 = 0.0;
#pragma unroll 8
    for (int k = 0; k < 32; ++k) s2 += red[k][threadIdx.x];
    const int t = j / C, c = j % C;
    grad[(size_t)c * 9 + t] = (float)s2;
  }
}

// channel groups per wave row: the power of two that wastes the fewest lanes (ties -> wider, better coalescing)
static int dw_pick_cgw(int ngroups) {
  int best = 64, waste = cdiv(ngroups, 64) * 64 - ngroups;
  for (int w = 32; w >= 8; w >>= 1) {
    const int ws = cdiv(ngroups, w) * w - ngroups;
    if (ws < waste) { best = w; waste = ws; }
  }
  return best;
}

static int dw_out(int Hi, int stride, int dil) { return (Hi + 2 * dil - 2 * dil - 1) / stride + 1; }
static int dw_pix_per_block(long P) {
  long ppb = (P + 255) / 256;  // at most 256 pixel blocks (= slab rows)
  if (ppb < 32) ppb = 32;
  return (int)ppb;
}

}  // namespace dc

using namespace dc;

template <typename T, int MODE>
static int launch_dw(const void* in, int ldin, const float* w, const void* addend, int ldadd, void* out, int ldout,
                     int N, int Hin, int Win, int Hout, int Wout, int C, int stride, int dil, hipStream_t st,
                     const float* pscale = nullptr, const float* pshift = nullptr, int prelu = 0) {
  const long total = (long)N * Hout * ((Wout + DW_PX - 1) / DW_PX) * (C / Elem<T>::kPerVec);
  long blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL((dw_kernel<T, MODE>), dim3((int)blocks), dim3(256), 0, st, (const T*)in, ldin, w,
                     (const T*)addend, ldadd, (T*)out, ldout, N, Hin, Win, Hout, Wout, C, stride, dil, pscale, pshift, prelu);
  DC_CHECK_LAUNCH();
  return 0;
}

static int launch_dw_s1(int dtype, int dil, bool flip, const void* in, int ldin, const float* wp, const void* addend, int ldadd,
                        void* out, int ldout, int N, int H, int W, int C, hipStream_t st, const float* pscale = nullptr,
                        const float* pshift = nullptr, int prelu = 0) {
  if (g_dw_tile) return launch_dw_tile(dtype, dil, flip, in, ldin, wp, addend, ldadd, out, ldout, N, H, W, C, st, pscale, pshift, prelu, nullptr);
  const int kpv = dtype == DC_BF16 ? 8 : 4;
  const long total = (long)N * H * ((W + DW_PX - 1) / DW_PX) * (C / kpv);
  DC_REQUIRE(total < (1L << 31), "dc_dwconv: tensor too large for the stride-1 fast path");
  dim3 grid((unsigned)((total + 255) / 256));
#define DW_S1(TT, D, F) hipLaunchKernelGGL((dw_s1_kernel<TT, D, F>), grid, dim3(256), 0, st, (const TT*)in, ldin, wp, (const TT*)addend, ldadd, (TT*)out, ldout, N, H, W, C, pscale, pshift, prelu)
  if (dtype == DC_BF16) {
    if (dil == 1) { if (flip) DW_S1(bf16, 1, true); else DW_S1(bf16, 1, false); }
    else          { if (flip) DW_S1(bf16, 2, true); else DW_S1(bf16, 2, false); }
  } else {
    if (dil == 1) { if (flip) DW_S1(float, 1, true); else DW_S1(float, 1, false); }
    else          { if (flip) DW_S1(float, 2, true); else DW_S1(float, 2, false); }
  }
#undef DW_S1
  DC_CHECK_LAUNCH();
  return 0;
}

static int dw_check(int dtype, int C, int stride, int dil, int N, int Hi, int Wi) {
  DC_REQUIRE(dtype == DC_F32 || dtype == DC_BF16, "dc_dwconv: bad dtype");
  DC_REQUIRE(stride == 1 || stride == 2, "dc_dwconv: stride must be 1 or 2");
  DC_REQUIRE(dil >= 1 && C > 0 && N > 0 && Hi > 0 && Wi > 0, "dc_dwconv: bad shape");
  return 0;
}

extern "C" int dc_dw_set_option(const char* name, int value) {
  if (strcmp(name, "dw_tile") == 0) { g_dw_tile = value != 0; return 0; }
  if (strcmp(name, "dw_wgrad_tpb") == 0) { dw_tile_set_tpb(value); return 0; }
  if (strcmp(name, "dw_cg") == 0) { dw_tile_set_cg(value); return 0; }
  if (strcmp(name, "dw_pipe") == 0) { dw_pipe_set(value); return 0; }
  if (strcmp(name, "pw_bn_bwd") == 0) { pw_bn_bwd_set(value); return 0; }
  if (strcmp(name, "sep_fwd") == 0) { sep_fwd_set(value); return 0; }
  return -1;
}

extern "C" int dc_dwconv_pack_weights(int C, const float* master, float* packed, void* stream) {
  DC_REQUIRE(C > 0 && C % 4 == 0 && master && packed, "dc_dwconv_pack_weights: bad argument");
  DC_REQUIRE(((uintptr_t)packed & 15) == 0, "dc_dwconv_pack_weights: packed buffer must be 16-byte aligned");
  hipLaunchKernelGGL(dw_pack_kernel, dim3(cdiv(9 * C, 256)), dim3(256), 0, (hipStream_t)stream, master, packed, C);
  DC_CHECK_LAUNCH();
  return 0;
}

extern "C" int dc_dwconv_fwd(int dtype, int C, int stride, int dil, int N, int Hi, int Wi, const void* x, int ldx,
                             const float* w, void* y, int ldy, const float* pscale, const float* pshift, int prelu,
                             void* stream) {
  if (int e = dw_check(dtype, C, stride, dil, N, Hi, Wi)) return e;
  if (int e = dc_check_view(x, ldx, C, dtype, "dc_dwconv_fwd x")) return e;
  if (int e = dc_check_view(y, ldy, C, dtype, "dc_dwconv_fwd y")) return e;
  DC_REQUIRE(w != nullptr, "dc_dwconv_fwd: null weights");
  const int Ho = dw_out(Hi, stride, dil), Wo = dw_out(Wi, stride, dil);
  hipStream_t st = (hipStream_t)stream;
  DC_REQUIRE((pscale == nullptr) == (pshift == nullptr), "dc_dwconv_fwd: pscale and pshift go together");
  if (stride == 1 && (dil == 1 || dil == 2)) return launch_dw_s1(dtype, dil, false, x, ldx, w, nullptr, 0, y, ldy, N, Hi, Wi, C, st, pscale, pshift, prelu);
  if (stride == 2 && dil == 1 && g_dw_tile)
    return launch_dw_tile_s2(dtype, 0, N, Hi, Wi, C, x, ldx, w, nullptr, 0, y, ldy, nullptr, nullptr, st, pscale, pshift, prelu);
  return dtype == DC_BF16 ? launch_dw<bf16, 0>(x, ldx, w, nullptr, 0, y, ldy, N, Hi, Wi, Ho, Wo, C, stride, dil, st, pscale, pshift, prelu)
                          : launch_dw<float, 0>(x, ldx, w, nullptr, 0, y, ldy, N, Hi, Wi, Ho, Wo, C, stride, dil, st, pscale, pshift, prelu);
}

// dc_dwconv_fwd reading through a BatchNorm(+ReLU) whose finalize it runs itself: the producer left a SHORT slab of partial sums
// (rows <= dc_bn_bwd_apply_fin_max_rows()), every workgroup sums it for its own channels while its halo tile travels (same order, same bits as
// dc_bn_finalize), the workgroups of pixel tile 0 store scale / shift / save_mean / save_invstd and the running statistics.
extern "C" int dc_dwconv_fwd_fin_ok(int dtype, int C, int stride, int dil, int N, int Hi, int Wi) {
  return g_dw_tile && !dw_pipe_forward() && stride == 1 && (dil == 1 || dil == 2) && (dtype == DC_BF16 || dtype == DC_F32) &&
         C % (dtype == DC_BF16 ? 8 : 4) == 0;
}
extern "C" int dc_dwconv_fwd_fin(int dtype, int C, int stride, int dil, int N, int Hi, int Wi, const void* x, int ldx, const float* w,
                                 void* y, int ldy, int prelu, long count, const float* slab, int rows, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                                 float* scale, float* shift, float* save_mean, float* save_invstd, void* stream) {
  if (int e = dw_check(dtype, C, stride, dil, N, Hi, Wi)) return e;
  if (int e = dc_check_view(x, ldx, C, dtype, "dc_dwconv_fwd_fin x")) return e;
  if (int e = dc_check_view(y, ldy, C, dtype, "dc_dwconv_fwd_fin y")) return e;
  DC_REQUIRE(w && slab && gamma && beta && scale && shift && (rows > 0 || rows == SUM_ROW), "dc_dwconv_fwd_fin: null argument");
  DC_REQUIRE(((uintptr_t)slab & 15) == 0, "dc_dwconv_fwd_fin: the slab is read with 16-byte loads (16-byte aligned)");
  DC_REQUIRE(dc_dwconv_fwd_fin_ok(dtype, C, stride, dil, N, Hi, Wi), "dc_dwconv_fwd_fin: shape not served (dc_dwconv_fwd_fin_ok)");
  if (count <= 1) return dc_fail("Expected more than 1 value per channel when training", __FILE__, __LINE__);
  const BnFinArgs a = bn_fin_args(C, count, slab, rows, 0, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, scale,
                                  shift, save_mean, save_invstd);
  return launch_dw_tile(dtype, dil, false, x, ldx, w, nullptr, 0, y, ldy, N, Hi, Wi, C, (hipStream_t)stream, nullptr, nullptr, prelu, nullptr, &a);
}

extern "C" int dc_dwconv_dgrad(int dtype, int C, int stride, int dil, int N, int Hi, int Wi, const void* dy, int lddy,
                               const float* w, const void* addend, int ldadd, void* dx, int lddx, void* stream) {
  if (int e = dw_check(dtype, C, stride, dil, N, Hi, Wi)) return e;
  if (int e = dc_check_view(dy, lddy, C, dtype, "dc_dwconv_dgrad dy")) return e;
  if (int e = dc_check_view(dx, lddx, C, dtype, "dc_dwconv_dgrad dx")) return e;
  if (addend != nullptr)
    if (int e = dc_check_view(addend, ldadd, C, dtype, "dc_dwconv_dgrad addend")) return e;
  DC_REQUIRE(w != nullptr, "dc_dwconv_dgrad: null weights");
  const int Ho = dw_out(Hi, stride, dil), Wo = dw_out(Wi, stride, dil);
  hipStream_t st = (hipStream_t)stream;
  if (stride == 1 && (dil == 1 || dil == 2)) return launch_dw_s1(dtype, dil, true, dy, lddy, w, addend, ldadd, dx, lddx, N, Hi, Wi, C, st);
  if (stride == 2 && dil == 1 && g_dw_tile)
    return launch_dw_tile_s2(dtype, 1, N, Hi, Wi, C, dy, lddy, w, addend, ldadd, dx, lddx, nullptr, nullptr, st);
  // gather from dy [Ho,Wo] into dx [Hi,Wi]
  return dtype == DC_BF16 ? launch_dw<bf16, 1>(dy, lddy, w, addend, ldadd, dx, lddx, N, Ho, Wo, Hi, Wi, C, stride, dil, st)
                          : launch_dw<float, 1>(dy, lddy, w, addend, ldadd, dx, lddx, N, Ho, Wo, Hi, Wi, C, stride, dil, st);
}

extern "C" int dc_dwconv_dgrad_bnstats_rows(int dtype, int C, int stride, int dil, int N, int Hi, int Wi) {
  if (!g_dw_tile || (dtype != DC_BF16 && dtype != DC_F32) || C <= 0 || N <= 0) return 0;
  if (stride == 1 && (dil == 1 || dil == 2)) {
    const int pr = dw_pipe_rows(dtype, C, dil, N, Hi, Wi);    // persistent kernel (dwpipe.hip): one slab row per workgroup
    return pr > 0 ? pr : dw_tile_rows(dtype, C, N, Hi, Wi);
  }
  if (stride == 2 && dil == 1) return dw_tile_s2_dgrad_rows(dtype, C, N, Hi, Wi);
  return 0;
}

extern "C" int dc_dwconv_dgrad_bnstats(int dtype, int C, int stride, int dil, int N, int Hi, int Wi, const void* dy, int lddy,
                                       const float* w, void* dx, int lddx, const void* ybn, int ldybn, const float* save_mean,
                                       const float* save_invstd, const float* mscale, const float* mshift, int relu, float* slab,
                                       void* stream) {
  if (int e = dw_check(dtype, C, stride, dil, N, Hi, Wi)) return e;
  if (int e = dc_check_view(dy, lddy, C, dtype, "dc_dwconv_dgrad_bnstats dy")) return e;
  if (int e = dc_check_view(dx, lddx, C, dtype, "dc_dwconv_dgrad_bnstats dx")) return e;
  if (int e = dc_check_view(ybn, ldybn, C, dtype, "dc_dwconv_dgrad_bnstats y")) return e;
  DC_REQUIRE(w && save_mean && save_invstd && slab, "dc_dwconv_dgrad_bnstats: null argument");
  DC_REQUIRE(!relu || (mscale && mshift), "dc_dwconv_dgrad_bnstats: the ReLU mask needs the forward scale / shift vectors");
  DC_REQUIRE(dc_dwconv_dgrad_bnstats_rows(dtype, C, stride, dil, N, Hi, Wi) > 0, "dc_dwconv_dgrad_bnstats: shape not served by the tiled kernels");
  DwBnStats bs;
  bs.y = ybn; bs.ldy = ldybn; bs.mean = save_mean; bs.invstd = save_invstd; bs.mscale = mscale; bs.mshift = mshift; bs.relu = relu ? 1 : 0;
  bs.slab = slab; bs.rows = 0; bs.wslab = nullptr;
  hipStream_t st = (hipStream_t)stream;
  if (stride == 1) return launch_dw_tile(dtype, dil, true, dy, lddy, w, nullptr, 0, dx, lddx, N, Hi, Wi, C, st, nullptr, nullptr, 0, &bs);
  return launch_dw_tile_s2(dtype, 1, N, Hi, Wi, C, dy, lddy, w, nullptr, 0, dx, lddx, nullptr, nullptr, st, nullptr, nullptr, 0, &bs);
}

// The same with THIS depthwise layer's weight gradient taken from the same dy window (stride 1, tiled path, at most DWT_MAX_ROWS pixel
// tiles): wslab receives dc_dwconv_dgrad_wgrad_rows rows of [9][C] partial sums; dc_dwconv_wgrad_reduce adds them into grad_w.  The
// layer's forward input is act(ybn*mscale + mshift) (the never-stored BatchNorm output), so nothing is read that the data gradient
// with statistics does not read already.
extern "C" int dc_dwconv_dgrad_wgrad_rows(int dtype, int C, int stride, int dil, int N, int Hi, int Wi) {
  if (!g_dw_tile) return 0;
  if (stride == 2 && dil == 1) {          // the stride-2 data-gradient kernel walks its tiles: one row per workgroup
    const int rows = dc_dwconv_dgrad_bnstats_rows(dtype, C, stride, dil, N, Hi, Wi);
    return rows > 0 && rows <= DWT_MAX_ROWS ? rows : 0;
  }
  if (stride != 1 || (dil != 1 && dil != 2)) return 0;
  if (dc_dwconv_dgrad_bnstats_rows(dtype, C, stride, dil, N, Hi, Wi) <= 0) return 0;
  const int pr = dw_pipe_rows(dtype, C, dil, N, Hi, Wi);
  if (pr > 0) return pr;
  const int rows = dw_tile_rows(dtype, C, N, Hi, Wi);
  return rows <= DWT_MAX_ROWS ? rows : 0;
}

// 1: the data gradients that take BatchNorm sums (dc_dwconv_dgrad_bnstats_wgrad_sum, dc_dwconv_dgrad_wgrad_bnres_sum) serve this shape: the
// persistent pipelined kernel (dwpipe.hip), whose workgroups add their sums to ONE fp64 row (bn_fin.h: SUM_ROW)
extern "C" int dc_dwconv_dgrad_sum_row_ok(int dtype, int C, int stride, int dil, int N, int Hi, int Wi) {
  if (!g_dw_tile || stride != 1 || (dil != 1 && dil != 2) || C <= 0 || N <= 0 || (long)N * Hi * Wi >= (1L << 31)) return 0;
  return dw_pipe_rows(dtype, C, dil, N, Hi, Wi) > 0 && dc_dwconv_dgrad_wgrad_rows(dtype, C, stride, dil, N, Hi, Wi) > 0 ? 1 : 0;
}

static int dgrad_bnstats_wgrad_impl(int dtype, int C, int stride, int dil, int N, int Hi, int Wi, const void* dy, int lddy,
                                    const float* w, void* dx, int lddx, const void* ybn, int ldybn, const float* save_mean,
                                    const float* save_invstd, const float* mscale, const float* mshift, int relu, float* slab,
                                    float* wslab, void* stream, bool sum_row) {
  if (int e = dw_check(dtype, C, stride, dil, N, Hi, Wi)) return e;
  if (int e = dc_check_view(dy, lddy, C, dtype, "dc_dwconv_dgrad_bnstats_wgrad dy")) return e;
  if (int e = dc_check_view(dx, lddx, C, dtype, "dc_dwconv_dgrad_bnstats_wgrad dx")) return e;
  if (int e = dc_check_view(ybn, ldybn, C, dtype, "dc_dwconv_dgrad_bnstats_wgrad y")) return e;
  DC_REQUIRE(w && save_mean && save_invstd && slab && wslab && mscale && mshift, "dc_dwconv_dgrad_bnstats_wgrad: null argument");
  DC_REQUIRE(dc_dwconv_dgrad_wgrad_rows(dtype, C, stride, dil, N, Hi, Wi) > 0, "dc_dwconv_dgrad_bnstats_wgrad: shape not served");
  DC_REQUIRE(!sum_row || (dc_dwconv_dgrad_sum_row_ok(dtype, C, stride, dil, N, Hi, Wi) && ((uintptr_t)slab & 7) == 0),
             "dc_dwconv_dgrad_bnstats_wgrad_sum: shape not served (dc_dwconv_dgrad_sum_row_ok), or the sum row is not double[2][C]");
  DwBnStats bs;
  bs.y = ybn; bs.ldy = ldybn; bs.mean = save_mean; bs.invstd = save_invstd; bs.mscale = mscale; bs.mshift = mshift; bs.relu = relu ? 1 : 0;
  bs.slab = slab; bs.rows = sum_row ? SUM_ROW : 0; bs.wslab = wslab;
  if (stride == 2) return launch_dw_tile_s2(dtype, 1, N, Hi, Wi, C, dy, lddy, w, nullptr, 0, dx, lddx, nullptr, nullptr, (hipStream_t)stream, nullptr, nullptr, 0, &bs);
  return launch_dw_tile(dtype, dil, true, dy, lddy, w, nullptr, 0, dx, lddx, N, Hi, Wi, C, (hipStream_t)stream, nullptr, nullptr, 0, &bs);
}
extern "C" int dc_dwconv_dgrad_bnstats_wgrad(int dtype, int C, int stride, int dil, int N, int Hi, int Wi, const void* dy, int lddy,
                                             const float* w, void* dx, int lddx, const void* ybn, int ldybn, const float* save_mean,
                                             const float* save_invstd, const float* mscale, const float* mshift, int relu, float* slab,
                                             float* wslab, void* stream) {
  return dgrad_bnstats_wgrad_impl(dtype, C, stride, dil, N, Hi, Wi, dy, lddy, w, dx, lddx, ybn, ldybn, save_mean, save_invstd, mscale, mshift, relu, slab,
                                  wslab, stream, false);
}
// ... with the two BatchNorm sums added to a SUM ROW: `slab` is double[2][C] (sum g, sum g*xhat), zeroed by the caller; dc_bn_bwd_finalize and
// dc_bn_bwd_apply_fin take it with rows = -1
extern "C" int dc_dwconv_dgrad_bnstats_wgrad_sum(int dtype, int C, int stride, int dil, int N, int Hi, int Wi, const void* dy, int lddy,
                                                 const float* w, void* dx, int lddx, const void* ybn, int ldybn, const float* save_mean,
                                                 const float* save_invstd, const float* mscale, const float* mshift, int relu, float* slab,
                                                 float* wslab, void* stream) {
  return dgrad_bnstats_wgrad_impl(dtype, C, stride, dil, N, Hi, Wi, dy, lddy, w, dx, lddx, ybn, ldybn, save_mean, save_invstd, mscale, mshift, relu, slab,
                                  wslab, stream, true);
}

// dc_dwconv_dgrad_bnstats_wgrad when this layer is not the only reader of the BatchNorm's output: the data gradient is added onto what the
// other readers left in `addend` (may alias dx) and the sums are those of the COMPLETE gradient -- the caller makes this layer the last
// writer (block 1's first depthwise layer behind the shortcut conv: the BatchNorm in front of both is bn2, whose 226 MB output gradient and
// input a separate dc_bn_bwd_reduce pass would read once more).  Stride 1.
extern "C" int dc_dwconv_dgrad_bnstats_wgrad_add(int dtype, int C, int stride, int dil, int N, int Hi, int Wi, const void* dy, int lddy,
                                                 const float* w, const void* addend, int ldadd, void* dx, int lddx, const void* ybn, int ldybn,
                                                 const float* save_mean, const float* save_invstd, const float* mscale, const float* mshift, int relu,
                                                 float* slab, float* wslab, void* stream) {
  if (int e = dw_check(dtype, C, stride, dil, N, Hi, Wi)) return e;
  DC_REQUIRE(stride == 1, "dc_dwconv_dgrad_bnstats_wgrad_add: stride 1");
  if (int e = dc_check_view(dy, lddy, C, dtype, "dc_dwconv_dgrad_bnstats_wgrad_add dy")) return e;
  if (int e = dc_check_view(dx, lddx, C, dtype, "dc_dwconv_dgrad_bnstats_wgrad_add dx")) return e;
  if (int e = dc_check_view(ybn, ldybn, C, dtype, "dc_dwconv_dgrad_bnstats_wgrad_add y")) return e;
  if (addend != nullptr)
    if (int e = dc_check_view(addend, ldadd, C, dtype, "dc_dwconv_dgrad_bnstats_wgrad_add addend")) return e;
  DC_REQUIRE(w && save_mean && save_invstd && slab && wslab && mscale && mshift, "dc_dwconv_dgrad_bnstats_wgrad_add: null argument");
  DC_REQUIRE(dc_dwconv_dgrad_wgrad_rows(dtype, C, stride, dil, N, Hi, Wi) > 0, "dc_dwconv_dgrad_bnstats_wgrad_add: shape not served");
  DwBnStats bs;
  bs.y = ybn; bs.ldy = ldybn; bs.mean = save_mean; bs.invstd = save_invstd; bs.mscale = mscale; bs.mshift = mshift; bs.relu = relu ? 1 : 0;
  bs.slab = slab; bs.rows = 0; bs.wslab = wslab;
  return launch_dw_tile(dtype, dil, true, dy, lddy, w, addend, ldadd, dx, lddx, N, Hi, Wi, C, (hipStream_t)stream, nullptr, nullptr, 0, &bs);
}

// Data gradient (optionally accumulated onto `addend`) plus this layer's weight-gradient rows when the layer's forward input is a
// STORED tensor x (pscale == null) or act(x*pscale + pshift) of one: the first separable conv of an Xception block, whose input is the
// block input and whose data gradient joins the shortcut's.
extern "C" int dc_dwconv_dgrad_wgrad(int dtype, int C, int stride, int dil, int N, int Hi, int Wi, const void* dy, int lddy,
                                     const float* w, const void* addend, int ldadd, void* dx, int lddx, const void* x, int ldx,
                                     const float* pscale, const float* pshift, int prelu, float* wslab, void* stream) {
  if (int e = dw_check(dtype, C, stride, dil, N, Hi, Wi)) return e;
  if (int e = dc_check_view(dy, lddy, C, dtype, "dc_dwconv_dgrad_wgrad dy")) return e;
  if (int e = dc_check_view(dx, lddx, C, dtype, "dc_dwconv_dgrad_wgrad dx")) return e;
  if (int e = dc_check_view(x, ldx, C, dtype, "dc_dwconv_dgrad_wgrad x")) return e;
  if (addend != nullptr)
    if (int e = dc_check_view(addend, ldadd, C, dtype, "dc_dwconv_dgrad_wgrad addend")) return e;
  DC_REQUIRE(w && wslab && (pscale == nullptr) == (pshift == nullptr), "dc_dwconv_dgrad_wgrad: bad argument");
  DC_REQUIRE(dc_dwconv_dgrad_wgrad_rows(dtype, C, stride, dil, N, Hi, Wi) > 0, "dc_dwconv_dgrad_wgrad: shape not served");
  DwBnStats bs;
  bs.y = x; bs.ldy = ldx; bs.mean = bs.invstd = nullptr; bs.mscale = pscale; bs.mshift = pshift; bs.relu = (pscale != nullptr && prelu) ? 1 : 0;
  bs.slab = nullptr; bs.rows = 0; bs.wslab = wslab;
  if (stride == 2) return launch_dw_tile_s2(dtype, 1, N, Hi, Wi, C, dy, lddy, w, addend, ldadd, dx, lddx, nullptr, nullptr, (hipStream_t)stream, nullptr, nullptr, 0, &bs);
  return launch_dw_tile(dtype, dil, true, dy, lddy, w, addend, ldadd, dx, lddx, N, Hi, Wi, C, (hipStream_t)stream, nullptr, nullptr, 0, &bs);
}

// dc_dwconv_dgrad_wgrad on a layer whose stored input x is a BatchNorm output with a residual, relu(bn(ybn) + r) -- the first separable conv
// of an Xception block reading the previous block's output -- and whose data gradient is the LAST contribution to d(x): the kernel also
// takes that BatchNorm's backward sums (g = dx masked by x > 0 when relu; slab[2][rows][C]), so dc_bn_bwd_reduce's pass over dx, ybn and x
// disappears.  Served by the persistent kernel only (rows = 0: use dc_dwconv_dgrad_wgrad + dc_bn_bwd_reduce).
extern "C" int dc_dwconv_dgrad_wgrad_bnres_rows(int dtype, int C, int stride, int dil, int N, int Hi, int Wi) {
  if (stride != 1 || dil != 1 || !g_dw_tile) return 0;
  return dw_pipe_rows(dtype, C, dil, N, Hi, Wi);
}

static int dgrad_wgrad_bnres_impl(int dtype, int C, int stride, int dil, int N, int Hi, int Wi, const void* dy, int lddy,
                                  const float* w, const void* addend, int ldadd, void* dx, int lddx, const void* x, int ldx,
                                  float* wslab, const void* ybn, int ldybn, const float* save_mean, const float* save_invstd,
                                  int relu, float* slab, void* stream, bool sum_row) {
  if (int e = dw_check(dtype, C, stride, dil, N, Hi, Wi)) return e;
  if (int e = dc_check_view(dy, lddy, C, dtype, "dc_dwconv_dgrad_wgrad_bnres dy")) return e;
  if (int e = dc_check_view(dx, lddx, C, dtype, "dc_dwconv_dgrad_wgrad_bnres dx")) return e;
  if (int e = dc_check_view(x, ldx, C, dtype, "dc_dwconv_dgrad_wgrad_bnres x")) return e;
  if (int e = dc_check_view(ybn, ldybn, C, dtype, "dc_dwconv_dgrad_wgrad_bnres ybn")) return e;
  if (addend != nullptr)
    if (int e = dc_check_view(addend, ldadd, C, dtype, "dc_dwconv_dgrad_wgrad_bnres addend")) return e;
  DC_REQUIRE(w && wslab && save_mean && save_invstd && slab, "dc_dwconv_dgrad_wgrad_bnres: null argument");
  DC_REQUIRE(dc_dwconv_dgrad_wgrad_bnres_rows(dtype, C, stride, dil, N, Hi, Wi) > 0, "dc_dwconv_dgrad_wgrad_bnres: shape not served");
  DwBnStats bs;
  bs.y = x; bs.ldy = ldx; bs.mean = bs.invstd = nullptr; bs.mscale = nullptr; bs.mshift = nullptr; bs.relu = 0;
  bs.slab = nullptr; bs.rows = 0; bs.wslab = wslab;
  DwResStats rs;
  DC_REQUIRE(!sum_row || ((uintptr_t)slab & 7) == 0, "dc_dwconv_dgrad_wgrad_bnres_sum: a sum row is double[2][C]");
  rs.y = ybn; rs.ldy = ldybn; rs.mean = save_mean; rs.invstd = save_invstd; rs.relu = relu ? 1 : 0; rs.slab = slab; rs.sum_row = sum_row ? 1 : 0;
  return launch_dw_pipe(dil, true, dy, lddy, w, addend, ldadd, dx, lddx, N, Hi, Wi, C, (hipStream_t)stream, nullptr, nullptr, 0, &bs, &rs);
}
extern "C" int dc_dwconv_dgrad_wgrad_bnres(int dtype, int C, int stride, int dil, int N, int Hi, int Wi, const void* dy, int lddy,
                                           const float* w, const void* addend, int ldadd, void* dx, int lddx, const void* x, int ldx,
                                           float* wslab, const void* ybn, int ldybn, const float* save_mean, const float* save_invstd,
                                           int relu, float* slab, void* stream) {
  return dgrad_wgrad_bnres_impl(dtype, C, stride, dil, N, Hi, Wi, dy, lddy, w, addend, ldadd, dx, lddx, x, ldx, wslab, ybn, ldybn, save_mean, save_invstd,
                                relu, slab, stream, false);
}
// ... with the residual BatchNorm's two sums added to a SUM ROW (double[2][C], zeroed by the caller; served wherever the plain form is)
extern "C" int dc_dwconv_dgrad_wgrad_bnres_sum(int dtype, int C, int stride, int dil, int N, int Hi, int Wi, const void* dy, int lddy,
                                               const float* w, const void* addend, int ldadd, void* dx, int lddx, const void* x, int ldx,
                                               float* wslab, const void* ybn, int ldybn, const float* save_mean, const float* save_invstd,
                                               int relu, float* slab, void* stream) {
  return dgrad_wgrad_bnres_impl(dtype, C, stride, dil, N, Hi, Wi, dy, lddy, w, addend, ldadd, dx, lddx, x, ldx, wslab, ybn, ldybn, save_mean, save_invstd,
                                relu, slab, stream, true);
}

extern "C" int dc_dwconv_wgrad_reduce(int C, int rows, const float* wslab, float* grad_w, void* stream) {
  DC_REQUIRE(C > 0 && rows > 0 && wslab && grad_w, "dc_dwconv_wgrad_reduce: bad argument");
  return dw_tile_reduce(wslab, grad_w, rows, C, (hipStream_t)stream);
}

extern "C" size_t dc_dwconv_wgrad_workspace(int C, int N, int Hi, int Wi, int stride) {
  const int Ho = dw_out(Hi, stride, 1), Wo = dw_out(Wi, stride, 1);
  const long P = (long)N * Ho * Wo;
  const int ppb = dw_pix_per_block(P);
  int rows = (int)((P + ppb - 1) / ppb);
  if (rows < 256) rows = 256;            // the stride-1 register-window path uses up to 256 slab rows
  size_t bytes = (size_t)rows * 9 * C * sizeof(float);
  if (stride == 1) {
    const size_t tb = dw_tile_wgrad_workspace(C, N, Hi, Wi);
    if (tb > bytes) bytes = tb;
  } else {
    long tiles = (long)N * cdiv(Ho, 4) * cdiv(Wo, 8);           // stride-2 tiled path: 4 x (8..32) output tiles, one slab row each at most
    if (tiles > DWT_MAX_ROWS) tiles = DWT_MAX_ROWS;
    const size_t tb = (size_t)tiles * 9 * C * sizeof(float);
    if (tb > bytes) bytes = tb;
  }
  return bytes;
}

extern "C" int dc_dwconv_wgrad(int dtype, int C, int stride, int dil, int N, int Hi, int Wi, const void* x, int ldx,
                               const void* dy, int lddy, void* workspace, float* grad_w, const float* pscale,
                               const float* pshift, int prelu, void* stream) {
  if (int e = dw_check(dtype, C, stride, dil, N, Hi, Wi)) return e;
  if (int e = dc_check_view(x, ldx, C, dtype, "dc_dwconv_wgrad x")) return e;
  if (int e = dc_check_view(dy, lddy, C, dtype, "dc_dwconv_wgrad dy")) return e;
  DC_REQUIRE(workspace != nullptr && grad_w != nullptr, "dc_dwconv_wgrad: null argument");
  const int Ho = dw_out(Hi, stride, dil), Wo = dw_out(Wi, stride, dil);
  const long P = (long)N * Ho * Wo;
  const int ppb = dw_pix_per_block(P);
  const int rows = (int)((P + ppb - 1) / ppb);
  hipStream_t st = (hipStream_t)stream;
  const int kpv = dtype == DC_BF16 ? 8 : 4;
  const int cgw = dw_pick_cgw(C / kpv);
  if (stride == 1 && (dil == 1 || dil == 2) && g_dw_tile)
    return launch_dw_tile_wgrad(dtype, dil, x, ldx, dy, lddy, (float*)workspace, grad_w, N, Hi, Wi, C, st, pscale, pshift, prelu);
  if (stride == 2 && dil == 1 && g_dw_tile) {
    int trows = 0;
    if (int e = launch_dw_tile_s2(dtype, 2, N, Hi, Wi, C, x, ldx, nullptr, dy, lddy, nullptr, 0, (float*)workspace, &trows, st, pscale, pshift, prelu)) return e;
    return dw_tile_reduce((const float*)workspace, grad_w, trows, C, st);
  }
  if (stride == 1 && (dil == 1 || dil == 2)) {
    const int nstrips = N * Hi * ((Wi + DW_PX - 1) / DW_PX);
    const int nsl = 256 / cgw;
    int spb = cdiv(nstrips, 256);
    if (spb < 2 * nsl) spb = 2 * nsl;
    spb = cdiv(spb, nsl) * nsl;
    const int frows = cdiv(nstrips, spb);
    dim3 fgrid(cdiv(C / kpv, cgw), frows);
    const size_t lds = (size_t)256 * 9 * sizeof(float);
#define DW_WS1(TT, D) hipLaunchKernelGGL((dw_s1_wgrad_kernel<TT, D>), fgrid, dim3(256), lds, st, (const TT*)x, ldx, (const TT*)dy, lddy, (float*)workspace, N, Hi, Wi, C, spb, cgw, pscale, pshift, prelu)
    if (dtype == DC_BF16) { if (dil == 1) DW_WS1(bf16, 1); else DW_WS1(bf16, 2); }
    else                  { if (dil == 1) DW_WS1(float, 1); else DW_WS1(float, 2); }
#undef DW_WS1
    DC_CHECK_LAUNCH();
    hipLaunchKernelGGL(dw_wgrad_reduce_kernel, dim3(cdiv(9 * C, 8)), dim3(256), 0, st, (const float*)workspace, grad_w, frows, C);
    DC_CHECK_LAUNCH();
    return 0;
  }
  dim3 grid(cdiv(C / kpv, cgw), rows);
#define DW_WG(TT, W) hipLaunchKernelGGL((dw_wgrad_kernel<TT, W>), grid, dim3(256), 0, st, (const TT*)x, ldx, (const TT*)dy, lddy, (float*)workspace, N, Hi, Wi, Ho, Wo, C, stride, dil, ppb, pscale, pshift, prelu)
  if (dtype == DC_BF16) {
    if (cgw == 64) DW_WG(bf16, 64); else if (cgw == 32) DW_WG(bf16, 32); else if (cgw == 16) DW_WG(bf16, 16); else DW_WG(bf16, 8);
  } else {
    if (cgw == 64) DW_WG(float, 64); else if (cgw == 32) DW_WG(float, 32); else if (cgw == 16) DW_WG(float, 16); else DW_WG(float, 8);
  }
#undef DW_WG
  DC_CHECK_LAUNCH();
  hipLaunchKernelGGL(dw_wgrad_reduce_kernel, dim3(cdiv(9 * C, 8)), dim3(256), 0, st, (const float*)workspace, grad_w, rows, C);
  DC_CHECK_LAUNCH();
  return 0;
}
