// LDS-tiled STRIDE-2 depthwise 3x3 (dilation 1): the third separable conv of entry-flow blocks 1-3 (128 ch @384x576, 256 ch
// @192x288, 728 ch @96x144).  Same scheme as dwtile.hip (halo tile by LDS-DMA with a zero page for the padding, half a channel
// group per thread, taps in registers, XCD-aware tile order); only the index maps differ:
//   forward   out[oy,ox]   = sum_t in[2oy-1+ky, 2ox-1+kx] w[t]      output tile 4 x TW, input halo 9 x (2TW+1)
//   wgrad     dW[t]       += dy[oy,ox] * x[2oy-1+ky, 2ox-1+kx]       same tiles, dy strips straight from global memory
//   dgrad     dx[iy,ix]    = sum_{t: parity fits} dy[(iy+1-ky)/2, (ix+1-kx)/2] w[t]  (+ addend)
//                                                                    dx tile 8 x TW, dy tile 5 x (TW/2+1): 1, 2 or 4 taps per pixel
// The generic kernels they replace issued nine dependent loads per output pixel and ran at 2-5x the time of a copy of the
// same tensors (the weight gradient of the 128-channel layer: 516 us against ~115 us of traffic).
#include "dwtile.h"
#include "dwtile_common.h"

namespace dc {

namespace {

static __device__ __attribute__((aligned(256))) unsigned char dws2_zero_page[256];
typedef __attribute__((address_space(1))) const void* gas_ptr;
typedef __attribute__((address_space(3))) void* lds_ptr;

template <int CG, int TH_>
struct S2Cfg {
  static constexpr int TH = TH_, TW = 8 * (32 / CG);
  static constexpr int NSL = 128 / CG;              // strip lanes (a thread owns half a channel group)
  static constexpr int SPR = TW / DT_PX;            // strips per tile row
  static constexpr int SPT = TH * SPR / NSL;        // strips per thread
  static_assert(SPT * NSL == TH * SPR, "tile does not divide into strips");
};
constexpr int lds_bytes(int hh, int hw, int cg) { return (hh * hw * cg + 255) / 256 * 256 * 16; }

// HH x HW pixels x CG channel groups, origin (iy0, ix0) of image n, into LDS [pixel][group]
template <typename T, int HH, int HW, int CG>
__device__ inline void stage_tile(char* smem, const T* __restrict__ in, int ldin, int n, int iy0, int ix0, int cg0, int ngroups, int H,
                                  int W) {
  constexpr int KPV = Elem<T>::kPerVec;
  constexpr int HP = HH * HW, ITER = (HP * CG + 255) / 256;
  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = tid % CG;
  const bool gok = cg0 + g < ngroups;
  const T* base = in + (size_t)n * H * W * ldin + (size_t)(cg0 + g) * KPV;
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int hp = (it * 256 + tid) / CG;
    const int hy = hp / HW, hx = hp - hy * HW;
    const int iy = iy0 + hy, ix = ix0 + hx;
    const bool ok = gok && hp < HP && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    const void* src = ok ? (const void*)(base + ((size_t)iy * W + ix) * ldin) : (const void*)dws2_zero_page;
    __builtin_amdgcn_global_load_lds((gas_ptr)src, (lds_ptr)(smem + (it * 256 + wv * 64) * 16), 16, 0, 0);
  }
}

struct S2Args {
  int Hi, Wi, Ho, Wo, C, ncgb, ntx, nty;
  const float* pscale;   // fused BatchNorm(+ReLU) of the producer (forward / weight gradient), or null
  const float* pshift;
  int prelu;
};

// ---- forward ------------------------------------------------------------------------------------------------------------
template <typename T, int CG>
__global__ __launch_bounds__(256) void dws2_fwd_kernel(const T* __restrict__ in, int ldin, const float* __restrict__ wp,
                                                       T* __restrict__ out, int ldout, S2Args a) {
  typedef S2Cfg<CG, 4> K;
  constexpr int KPV = Elem<T>::kPerVec, KH = KPV / 2;
  constexpr int HH = 2 * K::TH + 1, HW = 2 * K::TW + 1, WC = 2 * DT_PX + 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = xcd_remap(blockIdx.x, gridDim.x);
  const int cgb = t % a.ncgb;
  int r = t / a.ncgb;
  const int tx = r % a.ntx;
  r /= a.ntx;
  const int ty = r % a.nty, n = r / a.nty;
  const int ngroups = a.C / KPV;
  const int cg0 = cgb * CG, y0 = ty * K::TH, x0 = tx * K::TW;
  stage_tile<T, HH, HW, CG>(smem, in, ldin, n, 2 * y0 - 1, 2 * x0 - 1, cg0, ngroups, a.Hi, a.Wi);
  const int h = threadIdx.x % (2 * CG), sl = threadIdx.x / (2 * CG);
  const bool cok = cg0 + (h >> 1) < ngroups;
  const int ch0 = cok ? cg0 * KPV + h * KH : 0;
  float wk[9][KH];
  load_taps<KH>(wp, ch0, a.C, false, wk);
  __syncthreads();
  if (a.pscale != nullptr) {
    bn_transform_tile<T, HH, HW, CG>(smem, 2 * y0 - 1, 2 * x0 - 1, a.Hi, a.Wi, a.pscale, a.pshift, a.prelu, cg0, ngroups);
    __syncthreads();
  }
  if (!cok) return;
  const char* tile = smem + h * 8;
#pragma unroll 1
  for (int k = 0; k < K::SPT; ++k) {
    const int q = sl + K::NSL * k;
    const int row = q / K::SPR, xs = (q % K::SPR) * DT_PX;
    float acc[DT_PX][KH];
#pragma unroll
    for (int j = 0; j < DT_PX; ++j)
#pragma unroll
      for (int e = 0; e < KH; ++e) acc[j][e] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int c = 0; c < WC; ++c) {
        float f[KH];
        unpack8(*reinterpret_cast<const vec8*>(tile + ((2 * row + ky) * HW + 2 * xs + c) * (CG * 16)), f, T());
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          if ((c - kx) >= 0 && ((c - kx) & 1) == 0 && (c - kx) / 2 < DT_PX) {   // window column c = 2j + kx
#pragma unroll
            for (int e = 0; e < KH; ++e) acc[(c - kx) / 2][e] = fmaf(f[e], wk[ky * 3 + kx][e], acc[(c - kx) / 2][e]);
          }
        }
      }
    const int oy = y0 + row;
    if (oy < a.Ho) {
#pragma unroll
      for (int j = 0; j < DT_PX; ++j) {
        const int ox = x0 + xs + j;
        if (ox < a.Wo) {
          vec8 v;
          pack8(v, acc[j], T());
          *reinterpret_cast<vec8*>(out + (((size_t)n * a.Ho + oy) * a.Wo + ox) * ldout + ch0) = v;
        }
      }
    }
  }
}

// ---- data gradient -------------------------------------------------------------------------------------------------------
// A workgroup walks `tpb` consecutive dx tiles of one channel block and keeps what rides along -- the BatchNorm sums of the producer
// (DwBnStats::slab) and, WG, THIS layer's weight-gradient products (DwBnStats::wslab: dW[t] += dy[(p + 1 - t) / 2] * xhat[p] over the same
// (dy element, tap) pairs the data gradient multiplies, xhat = act(y*ms + mh) recomputed from the BatchNorm input the sums read anyway) --
// in registers across them: ONE slab row per workgroup (~1 000 rows instead of one per tile: 13 824 on the 384 x 576 layer), and the
// separate weight-gradient kernel's second pass over x (453 MB there) and dy disappears.
template <typename T, int CG, bool WG>
__global__ __launch_bounds__(256) void dws2_dgrad_kernel(const T* __restrict__ dy, int lddy, const float* __restrict__ wp,
                                                         const T* __restrict__ addend, int ldadd, T* __restrict__ dx, int lddx, S2Args a,
                                                         const DwBnStats st, int ntiles, int tpb) {
  typedef S2Cfg<CG, 8> K;   // tile over dx (input resolution): 8 x TW
  constexpr int KPV = Elem<T>::kPerVec, KH = KPV / 2;
  constexpr int HH = K::TH / 2 + 1, HW = K::TW / 2 + 1;   // dy tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = xcd_remap(blockIdx.x, gridDim.x);
  const int cgb = t % a.ncgb;
  const int srow = t / a.ncgb;
  const int ngroups = a.C / KPV;
  const int cg0 = cgb * CG;
  const int h = threadIdx.x % (2 * CG), sl = threadIdx.x / (2 * CG);
  const bool cok = cg0 + (h >> 1) < ngroups;
  const int ch0 = cok ? cg0 * KPV + h * KH : 0;
  float wk[9][KH];
  load_taps<KH>(wp, ch0, a.C, false, wk);
  const bool stats = st.slab != nullptr;
  BnAcc<KH> bn;
  if (stats) bn.init(st, ch0);
  else if (WG) bn.init_affine(st, ch0);
  float dwa[WG ? 9 : 1][KH];
#pragma unroll
  for (int t9 = 0; t9 < (WG ? 9 : 1); ++t9)
#pragma unroll
    for (int e = 0; e < KH; ++e) dwa[t9][e] = 0.f;
  const char* tile = smem + h * 8;
  const int tbeg = srow * tpb, tend = min(ntiles, tbeg + tpb);
  for (int ti = tbeg; ti < tend; ++ti) {
    const int tx = ti % a.ntx;
    int r = ti / a.ntx;
    const int ty = r % a.nty, n = r / a.nty;
    const int y0 = ty * K::TH, x0 = tx * K::TW;
    if (ti != tbeg) __syncthreads();   // every strip of the previous tile has left the LDS
    stage_tile<T, HH, HW, CG>(smem, dy, lddy, n, y0 / 2, x0 / 2, cg0, ngroups, a.Ho, a.Wo);
    // the BatchNorm input at a strip's own pixels is requested one strip ahead: strip 0's beside the tile's LDS-DMA (the barrier below waits
    // for both at once), strip k+1's before the stencil of strip k (requested at the top of its own strip, every strip began with an
    // exposed round trip)
    auto load_y = [&](int k, vec8 (&yv)[DT_PX]) {
      const int q = sl + K::NSL * k;
      const int iy = y0 + q / K::SPR, xs = (q % K::SPR) * DT_PX;
#pragma unroll
      for (int j = 0; j < DT_PX; ++j) {
        const int ix = x0 + xs + j;
        vec8 z;
        z.w[0] = z.w[1] = 0u;
        yv[j] = (iy < a.Hi && ix < a.Wi) ? *reinterpret_cast<const vec8*>(reinterpret_cast<const T*>(st.y) + (((size_t)n * a.Hi + iy) * a.Wi + ix) * st.ldy + ch0) : z;
      }
    };
    vec8 yv[DT_PX], ynext[DT_PX];
    if (cok && (stats || WG)) load_y(0, yv);
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < (cok ? K::SPT : 0); ++k) {
      const int q = sl + K::NSL * k;
      const int row = q / K::SPR, xs = (q % K::SPR) * DT_PX;   // tile-local dx row, first column of the strip (a multiple of 4)
      const int iy = y0 + row;
      if ((stats || WG) && k + 1 < K::SPT) load_y(k + 1, ynext);
      vec8 av[DT_PX];
      if (addend != nullptr) {
#pragma unroll
        for (int j = 0; j < DT_PX; ++j) {
          const int ix = x0 + xs + j;
          vec8 z;
          z.w[0] = z.w[1] = 0u;
          av[j] = (iy < a.Hi && ix < a.Wi) ? *reinterpret_cast<const vec8*>(addend + (((size_t)n * a.Hi + iy) * a.Wi + ix) * ldadd + ch0) : z;
        }
      }
      float acc[DT_PX][KH];
#pragma unroll
      for (int j = 0; j < DT_PX; ++j)
#pragma unroll
        for (int e = 0; e < KH; ++e) acc[j][e] = 0.f;
      // WG: the layer's forward input at the strip's own pixels, act(y*ms + mh) rounded to T exactly as the forward kernel's prologue
      // did (bn_transform_tile); zero outside the image
      float xh[WG ? DT_PX : 1][KH];
      if constexpr (WG) {
#pragma unroll
        for (int j = 0; j < DT_PX; ++j) {
          const bool in = iy < a.Hi && x0 + xs + j < a.Wi;
          float yf[KH];
          unpack8(yv[j], yf, T());
#pragma unroll
          for (int e = 0; e < KH; ++e) {
            const float v = fmaf(yf[e], bn.ms[e], bn.mh[e]);
            yf[e] = st.relu ? fmaxf(v, 0.f) : v;
          }
          vec8 rr;
          pack8(rr, yf, T());
          unpack8(rr, xh[j], T());
          if (!in) {
#pragma unroll
            for (int e = 0; e < KH; ++e) xh[j][e] = 0.f;
          }
        }
      }
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        // dy row (iy + 1 - ky)/2 when that is an integer; the row parity differs between the lanes of a wave, so the tap is
        // switched off by zeroing the loaded values instead of branching
        const int num = row + 1 - ky;
        const bool yfit = (num & 1) == 0;
        const int trow = yfit ? num >> 1 : 0;
#pragma unroll
        for (int j = 0; j < DT_PX; ++j)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            if (((j + 1 - kx) & 1) == 0) {   // column parity is known at compile time (xs is a multiple of 4)
              const int tcol = (xs + j + 1 - kx) / 2;          // xs + j + 1 - kx >= 0 whenever the parity fits
              float f[KH];
              unpack8(*reinterpret_cast<const vec8*>(tile + (trow * HW + tcol) * (CG * 16)), f, T());
#pragma unroll
              for (int e = 0; e < KH; ++e) {
                const float fv = yfit ? f[e] : 0.f;
                acc[j][e] = fmaf(fv, wk[ky * 3 + kx][e], acc[j][e]);
                if constexpr (WG) dwa[ky * 3 + kx][e] = fmaf(fv, xh[j][e], dwa[ky * 3 + kx][e]);
              }
            }
          }
      }
      if (iy < a.Hi) {
#pragma unroll
        for (int j = 0; j < DT_PX; ++j) {
          const int ix = x0 + xs + j;
          if (ix < a.Wi) {
            if (addend != nullptr) {
              float ad[KH];
              unpack8(av[j], ad, T());
#pragma unroll
              for (int e = 0; e < KH; ++e) acc[j][e] += ad[e];
            }
            vec8 v;
            pack8(v, acc[j], T());
            *reinterpret_cast<vec8*>(dx + (((size_t)n * a.Hi + iy) * a.Wi + ix) * lddx + ch0) = v;
            if (stats) {
              float gs[KH], yf[KH];
              unpack8(v, gs, T());
              unpack8(yv[j], yf, T());
              bn.add(gs, yf, st.relu);
            }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < DT_PX; ++j) yv[j] = ynext[j];
    }
  }
  if (stats) {
    __syncthreads();
    bn_acc_store<KH, K::NSL, CG * KPV>(reinterpret_cast<float*>(smem), bn.a, bn.b, cok, h, sl, st, srow, cg0 * KPV, a.C);
  }
  if constexpr (WG) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    constexpr int CW = CG * KPV;
    if (cok) {
#pragma unroll
      for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int e = 0; e < KH; ++e) red[(sl * 9 + tp) * CW + h * KH + e] = dwa[tp][e];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 9 * CW; i += 256) {
      const int tp = i / CW, cl = i % CW;
      const int c = cg0 * KPV + cl;
      if (c < a.C) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < K::NSL; ++q) s += red[(q * 9 + tp) * CW + cl];
        st.wslab[((size_t)srow * 9 + tp) * a.C + c] = s;
      }
    }
  }
}

// ---- weight gradient ---------------------------------------------------------------------------------------------------
template <typename T, int CG>
__global__ __launch_bounds__(256) void dws2_wgrad_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ dy, int lddy,
                                                         float* __restrict__ slab, S2Args a, int ntiles, int tpb) {
  typedef S2Cfg<CG, 4> K;
  constexpr int KPV = Elem<T>::kPerVec, KH = KPV / 2;
  constexpr int HH = 2 * K::TH + 1, HW = 2 * K::TW + 1, WC = 2 * DT_PX + 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = xcd_remap(blockIdx.x, gridDim.x);
  const int cgb = t % a.ncgb, srow = t / a.ncgb;
  const int ngroups = a.C / KPV;
  const int cg0 = cgb * CG;
  const int h = threadIdx.x % (2 * CG), sl = threadIdx.x / (2 * CG);
  const bool cok = cg0 + (h >> 1) < ngroups;
  const int ch0 = cok ? cg0 * KPV + h * KH : 0;
  float acc[9][KH];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int e = 0; e < KH; ++e) acc[tp][e] = 0.f;
  const int tbeg = srow * tpb, tend = min(ntiles, tbeg + tpb);
  const char* tile = smem + h * 8;
  for (int ti = tbeg; ti < tend; ++ti) {
    const int tx = ti % a.ntx;
    int r = ti / a.ntx;
    const int ty = r % a.nty, n = r / a.nty;
    const int y0 = ty * K::TH, x0 = tx * K::TW;
    if (ti != tbeg) __syncthreads();
    stage_tile<T, HH, HW, CG>(smem, x, ldx, n, 2 * y0 - 1, 2 * x0 - 1, cg0, ngroups, a.Hi, a.Wi);
    auto load_dy = [&](int k, vec8 (&gv)[DT_PX]) {
      const int q = sl + K::NSL * k;
      const int row = q / K::SPR, xs = (q % K::SPR) * DT_PX;
      const int oy = y0 + row;
#pragma unroll
      for (int j = 0; j < DT_PX; ++j) {
        const int ox = x0 + xs + j;
        const bool ok = cok && oy < a.Ho && ox < a.Wo;
        vec8 z;
        z.w[0] = z.w[1] = 0u;
        gv[j] = ok ? *reinterpret_cast<const vec8*>(dy + (((size_t)n * a.Ho + oy) * a.Wo + ox) * lddy + ch0) : z;
      }
    };
    vec8 gcur[DT_PX], gnext[DT_PX];
    load_dy(0, gcur);
    __syncthreads();
    if (a.pscale != nullptr) {
      bn_transform_tile<T, HH, HW, CG>(smem, 2 * y0 - 1, 2 * x0 - 1, a.Hi, a.Wi, a.pscale, a.pshift, a.prelu, cg0, ngroups);
      __syncthreads();
    }
    if (cok) {
#pragma unroll 1
      for (int k = 0; k < K::SPT; ++k) {
        if (k + 1 < K::SPT) load_dy(k + 1, gnext);
        const int q = sl + K::NSL * k;
        const int row = q / K::SPR, xs = (q % K::SPR) * DT_PX;
        float gf[DT_PX][KH];
#pragma unroll
        for (int j = 0; j < DT_PX; ++j) unpack8(gcur[j], gf[j], T());
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int c = 0; c < WC; ++c) {
            float f[KH];
            unpack8(*reinterpret_cast<const vec8*>(tile + ((2 * row + ky) * HW + 2 * xs + c) * (CG * 16)), f, T());
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
              if ((c - kx) >= 0 && ((c - kx) & 1) == 0 && (c - kx) / 2 < DT_PX) {
#pragma unroll
                for (int e = 0; e < KH; ++e) acc[ky * 3 + kx][e] = fmaf(gf[(c - kx) / 2][e], f[e], acc[ky * 3 + kx][e]);
              }
            }
          }
#pragma unroll
        for (int j = 0; j < DT_PX; ++j) gcur[j] = gnext[j];
      }
    }
  }
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);
  constexpr int CW = CG * KPV;
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int e = 0; e < KH; ++e) red[(sl * 9 + tp) * CW + h * KH + e] = acc[tp][e];
  __syncthreads();
  for (int i = threadIdx.x; i < 9 * CW; i += 256) {
    const int tp = i / CW, cl = i % CW;
    const int c = cg0 * KPV + cl;
    if (c < a.C) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < K::NSL; ++q) s += red[(q * 9 + tp) * CW + cl];
      slab[((size_t)srow * 9 + tp) * a.C + c] = s;
    }
  }
}

static int pick_cg(int ngroups) { return ngroups <= 8 ? 8 : ngroups <= 16 ? 16 : 32; }

// The data-gradient launch: about S2_DGRAD_WGS workgroups (four per CU, all resident: a workgroup's tile loop has no software pipeline, its
// neighbours on the CU cover its staging latency), each walking `tpb` consecutive 8 x TW tiles of one channel block; rows = workgroups per
// channel block = slab rows of the sums that ride along.
constexpr int S2_DGRAD_WGS = 1024;
static int s2_dgrad_plan(int kpv, int C, int N, int Hi, int Wi, int* tpb_out) {
  const int cg = pick_cg(C / kpv);
  const int ntiles = N * cdiv(Hi, 8) * cdiv(Wi, 8 * (32 / cg));
  const int ncgb = cdiv(C / kpv, cg);
  int tpb = cdiv((long)ntiles * ncgb, S2_DGRAD_WGS);
  if (tpb < 1) tpb = 1;
  if (tpb_out) *tpb_out = tpb;
  return cdiv(ntiles, tpb);
}

template <typename T, int CG>
static void launch3(int mode, S2Args a, int N, const void* p0, int ld0, const float* wp, const void* p1, int ld1, void* out, int ldout,
                    float* slab, int* rows_out, hipStream_t st, DwBnStats bs) {
  constexpr int TW = 8 * (32 / CG);
  constexpr int LDS_F = lds_bytes(9, 2 * TW + 1, CG);          // forward / wgrad halo
  constexpr int LDS_D0 = lds_bytes(5, TW / 2 + 1, CG);         // dgrad dy tile
  constexpr int FOLD2 = (128 / CG) * 2 * CG * Elem<T>::kPerVec * (int)sizeof(float);   // fused BN statistics fold
  constexpr int LDS_D = LDS_D0 > FOLD2 ? LDS_D0 : FOLD2;
  constexpr int FOLD = (128 / CG) * 9 * CG * Elem<T>::kPerVec * (int)sizeof(float);
  a.ntx = cdiv(mode == 1 ? a.Wi : a.Wo, TW);
  a.nty = cdiv(mode == 1 ? a.Hi : a.Ho, mode == 1 ? 8 : 4);
  a.ncgb = cdiv(a.C / Elem<T>::kPerVec, CG);
  const int ntiles = N * a.ntx * a.nty;
  if (mode == 0) {
    DC_ONCE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dws2_fwd_kernel<T, CG>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_F));
    hipLaunchKernelGGL((dws2_fwd_kernel<T, CG>), dim3(ntiles * a.ncgb), dim3(256), LDS_F, st, (const T*)p0, ld0, wp, (T*)out, ldout, a);
  } else if (mode == 1) {
    int tpb = 1;
    const int rows = s2_dgrad_plan(Elem<T>::kPerVec, a.C, N, a.Hi, a.Wi, &tpb);
    bs.rows = rows;
    if (rows_out) *rows_out = rows;
    if (bs.wslab != nullptr) {
      constexpr int LDS_DW = LDS_D > FOLD ? LDS_D : FOLD;
      DC_ONCE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dws2_dgrad_kernel<T, CG, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DW));
      hipLaunchKernelGGL((dws2_dgrad_kernel<T, CG, true>), dim3(rows * a.ncgb), dim3(256), LDS_DW, st, (const T*)p0, ld0, wp, (const T*)p1, ld1, (T*)out, ldout, a, bs, ntiles, tpb);
    } else {
      DC_ONCE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dws2_dgrad_kernel<T, CG, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_D));
      hipLaunchKernelGGL((dws2_dgrad_kernel<T, CG, false>), dim3(rows * a.ncgb), dim3(256), LDS_D, st, (const T*)p0, ld0, wp, (const T*)p1, ld1, (T*)out, ldout, a, bs, ntiles, tpb);
    }
  } else {
    constexpr int LDS_W = LDS_F > FOLD ? LDS_F : FOLD;
    DC_ONCE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dws2_wgrad_kernel<T, CG>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_W));
    long tpb = ((long)ntiles * a.ncgb + 256) / 512;      // ~512 workgroups, as the stride-1 planner
    if (tpb < 1) tpb = 1;
    while (cdiv(ntiles, tpb) > DWT_MAX_ROWS) ++tpb;
    const int rows = cdiv(ntiles, tpb);
    *rows_out = rows;
    hipLaunchKernelGGL((dws2_wgrad_kernel<T, CG>), dim3(rows * a.ncgb), dim3(256), LDS_W, st, (const T*)p0, ld0, (const T*)p1, ld1, slab, a, ntiles, (int)tpb);
  }
}

template <typename T>
static void launch2(int mode, const S2Args& a, int N, const void* p0, int ld0, const float* wp, const void* p1, int ld1, void* out,
                    int ldout, float* slab, int* rows_out, hipStream_t st, const DwBnStats& bs) {
  const int cg = pick_cg(a.C / Elem<T>::kPerVec);
  if (cg == 32) launch3<T, 32>(mode, a, N, p0, ld0, wp, p1, ld1, out, ldout, slab, rows_out, st, bs);
  else if (cg == 16) launch3<T, 16>(mode, a, N, p0, ld0, wp, p1, ld1, out, ldout, slab, rows_out, st, bs);
  else launch3<T, 8>(mode, a, N, p0, ld0, wp, p1, ld1, out, ldout, slab, rows_out, st, bs);
}

}  // namespace

// mode 0: forward (p0 = x -> out = y); mode 1: data gradient (p0 = dy, p1 = addend or null -> out = dx);
// mode 2: weight gradient partial rows (p0 = x, p1 = dy -> slab, *rows_out rows; reduce with dw_tile_reduce)
int launch_dw_tile_s2(int dtype, int mode, int N, int Hi, int Wi, int C, const void* p0, int ld0, const float* wp, const void* p1, int ld1,
                      void* out, int ldout, float* slab, int* rows_out, hipStream_t st, const float* pscale, const float* pshift, int prelu,
                      const DwBnStats* bnstats) {
  DwBnStats bs;
  if (bnstats != nullptr && mode == 1) bs = *bnstats; else { bs.slab = nullptr; bs.y = nullptr; bs.ldy = 0; bs.mean = bs.invstd = bs.mscale = bs.mshift = nullptr; bs.relu = 0; bs.rows = 0; bs.wslab = nullptr; }
  S2Args a;
  a.pscale = mode == 1 ? nullptr : pscale; a.pshift = pshift; a.prelu = prelu;
  a.Hi = Hi; a.Wi = Wi; a.Ho = (Hi - 1) / 2 + 1; a.Wo = (Wi - 1) / 2 + 1; a.C = C;
  a.ncgb = a.ntx = a.nty = 0;
  DC_REQUIRE((long)N * Hi * Wi < (1L << 31), "dc_dwconv: tensor too large for the tiled stride-2 path");
  int rows = 0;
  if (dtype == DC_BF16) launch2<bf16>(mode, a, N, p0, ld0, wp, p1, ld1, out, ldout, slab, &rows, st, bs);
  else launch2<float>(mode, a, N, p0, ld0, wp, p1, ld1, out, ldout, slab, &rows, st, bs);
  if (rows_out) *rows_out = rows;
  DC_CHECK_LAUNCH();
  return 0;
}

// workgroups per channel block of the stride-2 data-gradient kernel = slab rows of the sums that ride along (statistics, weight-gradient rows)
int dw_tile_s2_dgrad_rows(int dtype, int C, int N, int Hi, int Wi) { return s2_dgrad_plan(dtype == DC_BF16 ? 8 : 4, C, N, Hi, Wi, nullptr); }

}  // namespace dc
