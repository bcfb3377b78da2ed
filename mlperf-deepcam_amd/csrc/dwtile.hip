// LDS-tiled stride-1 depthwise 3x3 (forward, data gradient, weight gradient): 60 of the 63 depthwise layers.
//
// The register-window kernels in dwconv.hip pull every input element through the vector L1 4.5 times (plus as many weight
// loads) and measured 2.4 TB/s where a plain copy of the same tensors reaches 4.9 TB/s.  Here a 256-thread workgroup owns
// CG channel groups (16 bytes each) x an 8 x TW pixel tile:
//   * the (8+2d) x (TW+2d) halo tile is brought in ONCE by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction,
//     no VGPRs, no ds_write); out-of-image and out-of-range lanes read a zero page, so the padding costs no branches;
//   * the nine taps of a thread's channels sit in registers for its whole life (they were re-loaded per strip before);
//   * a thread computes two 4-pixel strips, reading the 3 x (4+2d) window of each from LDS (ds_read_b128, lanes =
//     consecutive channel groups = consecutive 16-byte slots: conflict-free);
//   * workgroups that share an XCD (equal id mod 8) take consecutive tiles, so the halo a tile shares with its neighbours is
//     served by that XCD's L2 instead of travelling over the fabric again.
// Tile width follows the channel count so that thin layers still fill the workgroup: CG = 32 / 16 / 8 groups with
// TW = 8 / 16 / 32 pixels.
#include "common.h"
#include "dwtile.h"
#include "dwtile_common.h"
#include "bn_fin.h"

namespace dc {

#ifndef DT_TH_VALUE
#define DT_TH_VALUE 8
#endif
constexpr int DT_TH = DT_TH_VALUE;   // tile rows
#ifndef DT_PROBE_KY
#define DT_PROBE_KY 3                // diagnostic builds only: stencil rows the forward / data-gradient loop executes (arithmetic share)
#endif

template <int DIL, int CG>
struct TileCfg {
  static constexpr int TH = DT_TH, TW = 8 * (32 / CG);
  static constexpr int HH = TH + 2 * DIL, HW = TW + 2 * DIL, HP = HH * HW;
  static constexpr int ITER = (HP * CG + 255) / 256;     // LDS-DMA instructions per wave
  static constexpr int LDS_BYTES = ITER * 256 * 16;
  static constexpr int NSL = 128 / CG;                    // strip lanes (a thread owns half a channel group)
  static constexpr int SPR = TW / DT_PX;                  // strips per tile row
  static constexpr int SPT = TH * SPR / NSL;              // strips per thread (= 4)
  static_assert(SPT * NSL == TH * SPR, "tile does not divide into strips");
};

static __device__ __attribute__((aligned(256))) unsigned char dwt_zero_page[256];
typedef __attribute__((address_space(1))) const void* gas_ptr;
typedef __attribute__((address_space(3))) void* lds_ptr;

template <typename T, int DIL, int CG>
__device__ inline void stage_halo(char* smem, const T* __restrict__ in, int ldin, int n, int y0, int x0, int cg0, int ngroups, int H,
                                  int W) {
  typedef TileCfg<DIL, CG> K;
  constexpr int KPV = Elem<T>::kPerVec;
  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = tid % CG;
  const bool gok = cg0 + g < ngroups;
  const T* base = in + (size_t)n * H * W * ldin + (size_t)(cg0 + g) * KPV;
#pragma unroll
  for (int it = 0; it < K::ITER; ++it) {
    const int hp = (it * 256 + tid) / CG;
    const int hy = hp / K::HW, hx = hp - hy * K::HW;
    const int iy = y0 - DIL + hy, ix = x0 - DIL + hx;
    const bool ok = gok && hp < K::HP && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    const void* src = ok ? (const void*)(base + ((size_t)iy * W + ix) * ldin) : (const void*)dwt_zero_page;
    __builtin_amdgcn_global_load_lds((gas_ptr)src, (lds_ptr)(smem + (it * 256 + wv * 64) * 16), 16, 0, 0);
  }
}

// forward (FLIP = false) and data gradient (FLIP = true: the same stencil with the taps reversed, plus an optional addend)
// FIN (forward only): instantiations of their own -- FIN = 1 (a row slab): 2 KiB of coefficients and 16 KiB of partial sums beside the 52 KiB halo
// tile take the third workgroup off a CU; FIN = 2 (a sum row, bn_fin.h): one channel per thread, no partial sums, and the coefficients go
// into the bytes the halo's last LDS-DMA instruction leaves unused where there are enough of them (three workgroups per CU stay)
template <typename T, int DIL, bool FLIP, int CG, bool WG = false, int FIN = 0>
__global__ __launch_bounds__(256) void dwt_kernel(const T* __restrict__ in, int ldin, const float* __restrict__ wp,
                                                  const T* __restrict__ addend, int ldadd, T* __restrict__ out, int ldout, int H, int W,
                                                  int C, int ncgb, int ntx, int nty, const float* __restrict__ pscale,
                                                  const float* __restrict__ pshift, int prelu, const DwBnStats st, const BnFinArgs fin) {
  typedef TileCfg<DIL, CG> K;
  constexpr int KPV = Elem<T>::kPerVec, KH = KPV / 2;
  constexpr int WC = DT_PX + 2 * DIL;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = xcd_remap(blockIdx.x, gridDim.x);
  const int cgb = t % ncgb;
  const int tile_id = t / ncgb;
  int r = t / ncgb;
  const int tx = r % ntx;
  r /= ntx;
  const int ty = r % nty, n = r / nty;
  const int ngroups = C / KPV;
  const int cg0 = cgb * CG, y0 = ty * K::TH, x0 = tx * K::TW;
  stage_halo<T, DIL, CG>(smem, in, ldin, n, y0, x0, cg0, ngroups, H, W);

  const int h = threadIdx.x % (2 * CG), sl = threadIdx.x / (2 * CG);   // half-group, strip lane
  const bool cok = cg0 + (h >> 1) < ngroups;
  const int ch0 = cok ? cg0 * KPV + h * KH : 0;                          // first of this thread's channels
  float wk[9][KH];
  load_taps<KH>(wp, ch0, C, FLIP, wk);
  const float* ps = pscale;
  const float* psh = pshift;
  int cbase = 0;
  float fsc = 0.f, fsh = 0.f;
  if constexpr (FIN == 2) {
    // dc_dwconv_fwd_fin over a sum row: thread i takes channel i of the workgroup's block while the halo travels
    constexpr int CW = CG * KPV;
    if (threadIdx.x < CW) {
      const int c = cg0 * KPV + threadIdx.x;
      if (c < C) {
        double s, q;
        sum_row_load(fin.slab, C, c, s, q);
        bn_fin_coefs(fin, c, s, q, tile_id == 0, fsc, fsh);
      }
    }
    if (t == 0 && threadIdx.x == 0 && fin.nbt != nullptr) *fin.nbt += 1;
  }
  if constexpr (FIN == 1) {
    // dc_dwconv_fwd_fin: the producer's BatchNorm finalize over a short slab, by every workgroup for its own channels while its halo
    // travels (bn_fin.h: slab_quad_sum2); the workgroup of pixel tile 0 stores the vectors the backward pass reads
    constexpr int CW = CG * KPV;
    __shared__ float fincoef[2][CW];
    __shared__ double finred[2][4][CW];
    {
      double s, q;
      slab_quad_sum2<CW>(fin.slab, fin.rows, C, cg0 * KPV, finred, s, q);
      if (threadIdx.x < CW) {
        const int c = cg0 * KPV + threadIdx.x;
        float sc = 0.f, sh = 0.f;
        if (c < C) bn_fin_coefs(fin, c, s, q, tile_id == 0, sc, sh);
        fincoef[0][threadIdx.x] = sc;
        fincoef[1][threadIdx.x] = sh;
      }
      if (t == 0 && threadIdx.x == 0 && fin.nbt != nullptr) *fin.nbt += 1;
      ps = fincoef[0];
      psh = fincoef[1];
      cbase = cg0 * KPV;
    }
  }
  __syncthreads();   // vmcnt(0) + barrier: the whole halo tile has landed
  if constexpr (FIN == 2) {
    constexpr int CW = CG * KPV;
    constexpr bool SLACK = K::LDS_BYTES - K::HP * CG * 16 >= 2 * CW * (int)sizeof(float);      // slots past the halo: zero-filled by the DMA, read by nobody
    float* coef;
    if constexpr (SLACK) {
      coef = reinterpret_cast<float*>(smem + K::LDS_BYTES) - 2 * CW;
    } else {
      __shared__ float fincoef2[2 * CW];
      coef = fincoef2;
    }
    if (threadIdx.x < CW) {
      coef[threadIdx.x] = fsc;
      coef[CW + threadIdx.x] = fsh;
    }
    __syncthreads();
    ps = coef;
    psh = coef + CW;
    cbase = cg0 * KPV;
  }
  if (!FLIP && ps != nullptr) {   // fused BatchNorm(+ReLU) of the producer, applied once per staged element
    bn_transform_tile<T, K::HH, K::HW, CG>(smem, y0 - DIL, x0 - DIL, H, W, ps, psh, prelu, cg0, ngroups, cbase);
    __syncthreads();
  }
  const bool stats = FLIP && st.slab != nullptr;
  if (!cok && !stats && !WG) return;
  BnAcc<KH> bn;
  if (stats) bn.init(st, ch0);
  else if (WG) bn.init_affine(st, ch0);     // weight gradient without statistics: the layer's input is a stored tensor (or its act(y*ms+mh))
  // WG: the weight gradient of THIS depthwise layer, taken from the same dy window (see DwBnStats::wslab); dwa[t] for tap t
  float dwa[WG ? 9 : 1][KH];
  if constexpr (WG) {
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
      for (int e = 0; e < KH; ++e) dwa[t9][e] = 0.f;
  }

  const char* tile = smem + h * 8;
#pragma unroll 1
  for (int k = 0; k < (cok ? K::SPT : 0); ++k) {
    const int q = sl + K::NSL * k;
    const int row = q / K::SPR, xs = (q % K::SPR) * DT_PX;
    const int oy = y0 + row;
    // the addend (gradient already accumulated in dx by another consumer) is requested before the stencil so that its latency
    // hides behind the arithmetic; so is the BatchNorm input when the statistics ride along
    vec8 yv[DT_PX];
    if (stats || WG) {
#pragma unroll
      for (int j = 0; j < DT_PX; ++j) {
        const int ox = x0 + xs + j;
        vec8 z;
        z.w[0] = z.w[1] = 0u;
        yv[j] = (oy < H && ox < W) ? *reinterpret_cast<const vec8*>(reinterpret_cast<const T*>(st.y) + (((size_t)n * H + oy) * W + ox) * st.ldy + ch0) : z;
      }
    }
    vec8 av[DT_PX];
    if (FLIP && addend != nullptr) {
#pragma unroll
      for (int j = 0; j < DT_PX; ++j) {
        const int ox = x0 + xs + j;
        vec8 z;
        z.w[0] = z.w[1] = 0u;
        av[j] = (oy < H && ox < W) ? *reinterpret_cast<const vec8*>(addend + (((size_t)n * H + oy) * W + ox) * ldadd + ch0) : z;
      }
    }
    float acc[DT_PX][KH];
#pragma unroll
    for (int j = 0; j < DT_PX; ++j)
#pragma unroll
      for (int e = 0; e < KH; ++e) acc[j][e] = 0.f;
    // WG: the forward input of this layer at the strip's own pixels, xhat = act(y*ms + mh) rounded to T exactly as the forward
    // kernel's prologue did (bn_transform_tile), zero outside the image
    float xh[WG ? DT_PX : 1][KH];
    if constexpr (WG) {
#pragma unroll
      for (int j = 0; j < DT_PX; ++j) {
        const bool in = oy < H && x0 + xs + j < W;
        float yf[KH];
        unpack8(yv[j], yf, T());
#pragma unroll
        for (int e = 0; e < KH; ++e) {
          const float v = fmaf(yf[e], bn.ms[e], bn.mh[e]);
          yf[e] = st.relu ? fmaxf(v, 0.f) : v;
        }
        vec8 r;
        pack8(r, yf, T());
        unpack8(r, xh[j], T());
        if (!in) {
#pragma unroll
          for (int e = 0; e < KH; ++e) xh[j][e] = 0.f;
        }
      }
    }
#pragma unroll
    for (int ky = 0; ky < DT_PROBE_KY; ++ky) {
#pragma unroll
      for (int c = 0; c < WC; ++c) {
        float f[KH];
        unpack8(*reinterpret_cast<const vec8*>(tile + ((row + ky * DIL) * K::HW + xs + c) * (CG * 16)), f, T());
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int j = c - kx * DIL;   // output pixel that sees window column c through tap kx
          if (j >= 0 && j < DT_PX) {
#pragma unroll
            for (int e = 0; e < KH; ++e) acc[j][e] = fmaf(f[e], wk[ky * 3 + kx][e], acc[j][e]);
            if constexpr (WG) {
              // window element = dy[q + (ky-1, kx-1)*DIL] = dy[q - (t - 1)*DIL] for the forward tap t = 8 - (ky*3 + kx)
#pragma unroll
              for (int e = 0; e < KH; ++e) dwa[8 - (ky * 3 + kx)][e] = fmaf(f[e], xh[j][e], dwa[8 - (ky * 3 + kx)][e]);
            }
          }
        }
      }
    }
    if (oy < H) {
#pragma unroll
      for (int j = 0; j < DT_PX; ++j) {
        const int ox = x0 + xs + j;
        if (ox < W) {
          const size_t opix = ((size_t)n * H + oy) * W + ox;
          if (FLIP && addend != nullptr) {
            float a[KH];
            unpack8(av[j], a, T());
#pragma unroll
            for (int e = 0; e < KH; ++e) acc[j][e] += a[e];
          }
          vec8 v;
          pack8(v, acc[j], T());
          *reinterpret_cast<vec8*>(out + opix * ldout + ch0) = v;
          if (stats) {
            float gs[KH], yf[KH];
            unpack8(v, gs, T());
            unpack8(yv[j], yf, T());
            bn.add(gs, yf, st.relu);
          }
        }
      }
    }
  }
  if (stats) {
    __syncthreads();   // every thread of the workgroup gets here: the staged tile is dead
    bn_acc_store<KH, K::NSL, CG * KPV>(reinterpret_cast<float*>(smem), bn.a, bn.b, cok, h, sl, st, tile_id, cg0 * KPV, C);
  }
  if constexpr (WG) {
    // fold the strip lanes (as dwt_wgrad_kernel): red[sl][9][CW], one slab row per pixel tile
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
      if (c < C) {
        float sacc = 0.f;
#pragma unroll
        for (int qq = 0; qq < K::NSL; ++qq) sacc += red[(qq * 9 + tp) * CW + cl];
        st.wslab[((size_t)tile_id * 9 + tp) * C + c] = sacc;
      }
    }
  }
}

// Weight gradient.  A workgroup walks `tpb` consecutive tiles of one channel block, accumulating 9 x KH products per thread in
// registers; the strip lanes are folded through LDS and the block leaves ONE row in slab[row][9][C] with fully coalesced
// stores.  A second kernel adds the rows in a fixed order in fp64.
template <typename T, int DIL, int CG>
__global__ __launch_bounds__(256) void dwt_wgrad_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ dy, int lddy,
                                                        float* __restrict__ slab, int H, int W, int C, int ncgb, int ntx, int nty,
                                                        int ntiles, int tpb, const float* __restrict__ pscale,
                                                        const float* __restrict__ pshift, int prelu) {
  typedef TileCfg<DIL, CG> K;
  constexpr int KPV = Elem<T>::kPerVec, KH = KPV / 2;
  constexpr int WC = DT_PX + 2 * DIL;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = xcd_remap(blockIdx.x, gridDim.x);
  const int cgb = t % ncgb, srow = t / ncgb;
  const int ngroups = C / KPV;
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
    const int tx = ti % ntx;
    int r = ti / ntx;
    const int ty = r % nty, n = r / nty;
    const int y0 = ty * K::TH, x0 = tx * K::TW;
    if (ti != tbeg) __syncthreads();   // everybody is done reading the previous tile
    stage_halo<T, DIL, CG>(smem, x, ldx, n, y0, x0, cg0, ngroups, H, W);
    // this thread's dy strips come straight from global memory (each element has exactly one consumer).  The first strip is
    // requested before the wait so that it travels together with the halo; strip k+1 is requested while strip k is computed.
    auto load_dy = [&](int k, vec8 (&gv)[DT_PX]) {
      const int q = sl + K::NSL * k;
      const int row = q / K::SPR, xs = (q % K::SPR) * DT_PX;
      const int oy = y0 + row;
#pragma unroll
      for (int j = 0; j < DT_PX; ++j) {
        const int ox = x0 + xs + j;
        const bool ok = cok && oy < H && ox < W;
        vec8 z;
        z.w[0] = z.w[1] = 0u;
        gv[j] = ok ? *reinterpret_cast<const vec8*>(dy + (((size_t)n * H + oy) * W + ox) * lddy + ch0) : z;
      }
    };
    vec8 gcur[DT_PX], gnext[DT_PX];
    load_dy(0, gcur);
    __syncthreads();   // vmcnt(0) + barrier
    if (pscale != nullptr) {
      bn_transform_tile<T, K::HH, K::HW, CG>(smem, y0 - DIL, x0 - DIL, H, W, pscale, pshift, prelu, cg0, ngroups);
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
            unpack8(*reinterpret_cast<const vec8*>(tile + ((row + ky * DIL) * K::HW + xs + c) * (CG * 16)), f, T());
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
              const int j = c - kx * DIL;
              if (j >= 0 && j < DT_PX) {
#pragma unroll
                for (int e = 0; e < KH; ++e) acc[ky * 3 + kx][e] = fmaf(gf[j][e], f[e], acc[ky * 3 + kx][e]);
              }
            }
          }
#pragma unroll
        for (int j = 0; j < DT_PX; ++j) gcur[j] = gnext[j];
      }
    }
  }
  // fold the strip lanes: red[sl][9][CW], CW = channels per block
  __syncthreads();   // tile reads finished; reuse the LDS
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
    if (c < C) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < K::NSL; ++q) s += red[(q * 9 + tp) * CW + cl];
      slab[((size_t)srow * 9 + tp) * C + c] = s;
    }
  }
}

// grad[c][t] = sum over rows of slab[row][t][c]: 16 columns x 16 row-lanes per block (64-byte segments), fp64, fixed order
__global__ __launch_bounds__(256) void dwt_reduce_kernel(const float* __restrict__ slab, float* __restrict__ grad, int rows, int C) {
  __shared__ double red[16][16];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int n = 9 * C;
  const int i = blockIdx.x * 16 + cl;
  double a0 = 0.0, a1 = 0.0;
  if (i < n) {
    int r = rl;
    for (; r + 16 < rows; r += 32) {
      a0 += (double)slab[(size_t)r * n + i];
      a1 += (double)slab[(size_t)(r + 16) * n + i];
    }
    if (r < rows) a0 += (double)slab[(size_t)r * n + i];
  }
  red[rl][cl] = a0 + a1;
  __syncthreads();
  if (threadIdx.x < 16 && blockIdx.x * 16 + threadIdx.x < n) {
    const int j = blockIdx.x * 16 + threadIdx.x;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k][threadIdx.x];
    const int tp = j / C, c = j % C;
    grad[(size_t)c * 9 + tp] = (float)s;
  }
}

int dw_tile_reduce(const float* slab, float* grad_w, int rows, int C, hipStream_t st) {
  hipLaunchKernelGGL(dwt_reduce_kernel, dim3(cdiv(9 * C, 16)), dim3(256), 0, st, slab, grad_w, rows, C);
  DC_CHECK_LAUNCH();
  return 0;
}

static int g_dw_tpb = 0;   // tiles per workgroup of the weight gradient; 0 = planner
void dw_tile_set_tpb(int v) { g_dw_tpb = v < 0 ? 0 : v; }

static int g_dw_cg = 0;   // 0: by channel count; 8 / 16 / 32: force the channel-group width of a workgroup (tuning switch)
void dw_tile_set_cg(int v) { g_dw_cg = (v == 8 || v == 16 || v == 32) ? v : 0; }
static int pick_cg(int ngroups) { return g_dw_cg ? g_dw_cg : ngroups <= 8 ? 8 : ngroups <= 16 ? 16 : 32; }

struct TileGrid {
  int cg, ncgb, ntx, nty, ntiles;
};
static TileGrid tile_grid(int ngroups, int N, int H, int W) {
  TileGrid t;
  t.cg = pick_cg(ngroups);
  const int tw = 8 * (32 / t.cg);
  t.ncgb = cdiv(ngroups, t.cg);
  t.ntx = cdiv(W, tw);
  t.nty = cdiv(H, DT_TH);
  t.ntiles = N * t.ntx * t.nty;
  return t;
}

static int wgrad_tpb(const TileGrid& t) {
  if (g_dw_tpb > 0) return g_dw_tpb;
  // about 512 workgroups (2 per CU) of about 3 tiles each measured best on the 728-, 256- and 128-channel layers: fewer,
  // longer workgroups amortise the fold and keep the slab (written once, read once by the reduction) small
  long tpb = ((long)t.ntiles * t.ncgb + 256) / 512;
  if (tpb < 1) tpb = 1;
  while (cdiv(t.ntiles, tpb) > DWT_MAX_ROWS) ++tpb;
  return (int)tpb;
}

int dw_tile_rows(int dtype, int C, int N, int H, int W) { return tile_grid(C / (dtype == DC_BF16 ? 8 : 4), N, H, W).ntiles; }

size_t dw_tile_wgrad_workspace(int C, int N, int H, int W) {
  size_t best = 0;
  for (int kpv = 4; kpv <= 8; kpv += 4) {
    if (C % kpv) continue;
    const TileGrid t = tile_grid(C / kpv, N, H, W);
    const int rows = t.ntiles < DWT_MAX_ROWS ? t.ntiles : DWT_MAX_ROWS;
    const size_t b = (size_t)rows * 9 * C * sizeof(float);
    if (b > best) best = b;
  }
  return best;
}

template <typename T, int DIL, bool FLIP, int CG>
static void launch_fwd1(const TileGrid& t, const void* in, int ldin, const float* wp, const void* addend, int ldadd, void* out,
                        int ldout, int H, int W, int C, hipStream_t st, const float* pscale, const float* pshift, int prelu, const DwBnStats& bs, const BnFinArgs& fin) {
  if constexpr (FLIP) {
    if (bs.wslab != nullptr) {          // data gradient + BatchNorm sums + this layer's weight-gradient rows
      constexpr int FOLD = TileCfg<DIL, CG>::NSL * 9 * CG * Elem<T>::kPerVec * (int)sizeof(float);
      constexpr int LDSW = TileCfg<DIL, CG>::LDS_BYTES > FOLD ? TileCfg<DIL, CG>::LDS_BYTES : FOLD;
      DC_ONCE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dwt_kernel<T, DIL, true, CG, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSW));
      hipLaunchKernelGGL((dwt_kernel<T, DIL, true, CG, true>), dim3(t.ntiles * t.ncgb), dim3(256), LDSW, st, (const T*)in, ldin, wp,
                         (const T*)addend, ldadd, (T*)out, ldout, H, W, C, t.ncgb, t.ntx, t.nty, pscale, pshift, prelu, bs, fin);
      return;
    }
  }
  constexpr int LDS = TileCfg<DIL, CG>::LDS_BYTES;
  if constexpr (!FLIP) {
    if (fin.slab != nullptr && fin.rows == SUM_ROW) {
      DC_ONCE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dwt_kernel<T, DIL, false, CG, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
      hipLaunchKernelGGL((dwt_kernel<T, DIL, false, CG, false, 2>), dim3(t.ntiles * t.ncgb), dim3(256), LDS, st, (const T*)in, ldin, wp,
                         (const T*)addend, ldadd, (T*)out, ldout, H, W, C, t.ncgb, t.ntx, t.nty, pscale, pshift, prelu, bs, fin);
      return;
    }
    if (fin.slab != nullptr) {
      DC_ONCE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dwt_kernel<T, DIL, false, CG, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
      hipLaunchKernelGGL((dwt_kernel<T, DIL, false, CG, false, 1>), dim3(t.ntiles * t.ncgb), dim3(256), LDS, st, (const T*)in, ldin, wp,
                         (const T*)addend, ldadd, (T*)out, ldout, H, W, C, t.ncgb, t.ntx, t.nty, pscale, pshift, prelu, bs, fin);
      return;
    }
  }
  DC_ONCE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dwt_kernel<T, DIL, FLIP, CG>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  hipLaunchKernelGGL((dwt_kernel<T, DIL, FLIP, CG>), dim3(t.ntiles * t.ncgb), dim3(256), LDS, st, (const T*)in, ldin, wp,
                     (const T*)addend, ldadd, (T*)out, ldout, H, W, C, t.ncgb, t.ntx, t.nty, pscale, pshift, prelu, bs, fin);
}

template <typename T, int DIL, bool FLIP>
static void launch_fwd2(const TileGrid& t, const void* in, int ldin, const float* wp, const void* addend, int ldadd, void* out,
                        int ldout, int H, int W, int C, hipStream_t st, const float* pscale, const float* pshift, int prelu, const DwBnStats& bs, const BnFinArgs& fin) {
  if (t.cg == 32) launch_fwd1<T, DIL, FLIP, 32>(t, in, ldin, wp, addend, ldadd, out, ldout, H, W, C, st, pscale, pshift, prelu, bs, fin);
  else if (t.cg == 16) launch_fwd1<T, DIL, FLIP, 16>(t, in, ldin, wp, addend, ldadd, out, ldout, H, W, C, st, pscale, pshift, prelu, bs, fin);
  else launch_fwd1<T, DIL, FLIP, 8>(t, in, ldin, wp, addend, ldadd, out, ldout, H, W, C, st, pscale, pshift, prelu, bs, fin);
}

int launch_dw_tile(int dtype, int dil, bool flip, const void* in, int ldin, const float* wp, const void* addend, int ldadd,
                   void* out, int ldout, int N, int H, int W, int C, hipStream_t st, const float* pscale, const float* pshift, int prelu,
                   const DwBnStats* bnstats, const BnFinArgs* finp) {
  const int kpv = dtype == DC_BF16 ? 8 : 4;
  BnFinArgs fin;
  if (finp != nullptr) fin = *finp; else fin.slab = nullptr;
  DC_REQUIRE(finp == nullptr || (!flip && !dw_pipe_forward() && finp->rows <= FIN_RL && finp->parts == 0),
             "dc_dwconv_fwd_fin: a forward pass on the tiled kernel over a slab of at most dc_bn_bwd_apply_fin_max_rows() rows");
  // the persistent pipelined kernel serves the data gradients (52.0 -> 41.5 us with BatchNorm sums and weight gradient on the 728-channel
  // layers at local batch 8); the forward pass stays here unless option "dw_pipe" = 2 (23.4 vs 24.1 us plain, 34.5 vs 31.7 us with the
  // BatchNorm applied on load: its in-place transform pass costs eight waves more than it costs three co-resident workgroups)
  if ((flip || dw_pipe_forward()) && (long)N * H * W < (1L << 31) && dw_pipe_rows(dtype, C, dil, N, H, W) > 0)
    return launch_dw_pipe(dil, flip, in, ldin, wp, addend, ldadd, out, ldout, N, H, W, C, st, pscale, pshift, prelu, bnstats);
  DC_REQUIRE(bnstats == nullptr || bnstats->rows != SUM_ROW, "dc_dwconv_dgrad_*_sum: a sum row is served by the pipelined kernel only (dc_dwconv_dgrad_sum_row_ok)");
  const TileGrid t = tile_grid(C / kpv, N, H, W);
  DC_REQUIRE((long)t.ntiles * t.ncgb < (1L << 31) && (long)N * H * W < (1L << 31), "dc_dwconv: tensor too large for the tiled path");
  DwBnStats bs;
  if (bnstats != nullptr) bs = *bnstats; else { bs.slab = nullptr; bs.y = nullptr; bs.ldy = 0; bs.mean = bs.invstd = bs.mscale = bs.mshift = nullptr; bs.relu = 0; bs.rows = 0; bs.wslab = nullptr; }
  if (bs.slab != nullptr) bs.rows = t.ntiles;
#define DWT(TT, D, F) launch_fwd2<TT, D, F>(t, in, ldin, wp, addend, ldadd, out, ldout, H, W, C, st, pscale, pshift, prelu, bs, fin)
  if (dtype == DC_BF16) {
    if (dil == 1) { if (flip) DWT(bf16, 1, true); else DWT(bf16, 1, false); }
    else          { if (flip) DWT(bf16, 2, true); else DWT(bf16, 2, false); }
  } else {
    if (dil == 1) { if (flip) DWT(float, 1, true); else DWT(float, 1, false); }
    else          { if (flip) DWT(float, 2, true); else DWT(float, 2, false); }
  }
#undef DWT
  DC_CHECK_LAUNCH();
  return 0;
}

template <typename T, int DIL, int CG>
static void launch_wg1(const TileGrid& t, int tpb, int rows, const void* x, int ldx, const void* dy, int lddy, float* slab, int H,
                       int W, int C, hipStream_t st, const float* pscale, const float* pshift, int prelu) {
  constexpr int FOLD = TileCfg<DIL, CG>::NSL * 9 * CG * Elem<T>::kPerVec * (int)sizeof(float);   // red[strip lane][9][channels]
  constexpr int LDS = TileCfg<DIL, CG>::LDS_BYTES > FOLD ? TileCfg<DIL, CG>::LDS_BYTES : FOLD;
  DC_ONCE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dwt_wgrad_kernel<T, DIL, CG>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  hipLaunchKernelGGL((dwt_wgrad_kernel<T, DIL, CG>), dim3(rows * t.ncgb), dim3(256), LDS, st, (const T*)x, ldx, (const T*)dy, lddy,
                     slab, H, W, C, t.ncgb, t.ntx, t.nty, t.ntiles, tpb, pscale, pshift, prelu);
}

template <typename T, int DIL>
static void launch_wg2(const TileGrid& t, int tpb, int rows, const void* x, int ldx, const void* dy, int lddy, float* slab, int H,
                       int W, int C, hipStream_t st, const float* pscale, const float* pshift, int prelu) {
  if (t.cg == 32) launch_wg1<T, DIL, 32>(t, tpb, rows, x, ldx, dy, lddy, slab, H, W, C, st, pscale, pshift, prelu);
  else if (t.cg == 16) launch_wg1<T, DIL, 16>(t, tpb, rows, x, ldx, dy, lddy, slab, H, W, C, st, pscale, pshift, prelu);
  else launch_wg1<T, DIL, 8>(t, tpb, rows, x, ldx, dy, lddy, slab, H, W, C, st, pscale, pshift, prelu);
}

int launch_dw_tile_wgrad(int dtype, int dil, const void* x, int ldx, const void* dy, int lddy, float* slab, float* grad_w, int N,
                         int H, int W, int C, hipStream_t st, const float* pscale, const float* pshift, int prelu) {
  const int kpv = dtype == DC_BF16 ? 8 : 4;
  const TileGrid t = tile_grid(C / kpv, N, H, W);
  const int tpb = wgrad_tpb(t);
  const int rows = cdiv(t.ntiles, tpb);
  DC_REQUIRE(rows <= DWT_MAX_ROWS, "dc_dwconv_wgrad: tiles-per-workgroup override needs more slab rows than the workspace holds");
  DC_REQUIRE((long)N * H * W < (1L << 31), "dc_dwconv_wgrad: tensor too large for the tiled path");
  if (dtype == DC_BF16) {
    if (dil == 1) launch_wg2<bf16, 1>(t, tpb, rows, x, ldx, dy, lddy, slab, H, W, C, st, pscale, pshift, prelu);
    else launch_wg2<bf16, 2>(t, tpb, rows, x, ldx, dy, lddy, slab, H, W, C, st, pscale, pshift, prelu);
  } else {
    if (dil == 1) launch_wg2<float, 1>(t, tpb, rows, x, ldx, dy, lddy, slab, H, W, C, st, pscale, pshift, prelu);
    else launch_wg2<float, 2>(t, tpb, rows, x, ldx, dy, lddy, slab, H, W, C, st, pscale, pshift, prelu);
  }
  DC_CHECK_LAUNCH();
  hipLaunchKernelGGL(dwt_reduce_kernel, dim3(cdiv(9 * C, 16)), dim3(256), 0, st, (const float*)slab, grad_w, rows, C);
  DC_CHECK_LAUNCH();
  return 0;
}

}  // namespace dc
