// Train/eval BatchNorm2d fused with ReLU and the residual add, NHWC, all HBM-bound streaming kernels.
//
// Column reductions (statistics forward; dgamma/dbeta backward) use one lane per 16-byte channel group, 32 groups
// side by side (512 contiguous bytes per row), 8 row-lanes per workgroup, 4 rows in flight per thread, fp32 partials per
// 64-row block in a slab, and a tiny finalize kernel that combines the slab rows in fp64 in a fixed order (deterministic,
// and it avoids the E[x^2]-E[x]^2 cancellation in fp32).  Per-channel vectors reach the threads through LDS.
#include <string.h>

#include "common.h"
#include "bn_fin.h"

namespace dc {

constexpr int RED_ROWS_MIN = 64;   // rows per reduction block (= rows per slab row) on small tensors
constexpr int RED_SLAB_MAX = 2048;  // most slab rows on large ones (the finalize kernels read the slab serially per channel)
inline int red_rows(long M) {
  long r = (M + RED_SLAB_MAX - 1) / RED_SLAB_MAX;
  r = (r + 31) / 32 * 32;
  return (int)(r < RED_ROWS_MIN ? RED_ROWS_MIN : r);
}
// ... when the blocks add to a sum row: fp64 atomics onto ONE address go at about one per 17 ns (scripts/sum_row_contention.py: 1 728 blocks per
// channel block took 12 -> 42 us), so the pass is cut into at most 256 row blocks -- 4 us of adds, hidden behind the loads
constexpr int RED_SUM_MAX = 256;
inline int red_rows_sum(long M) {
  long r = (M + RED_SUM_MAX - 1) / RED_SUM_MAX;
  r = (r + 31) / 32 * 32;
  return (int)(r < RED_ROWS_MIN ? RED_ROWS_MIN : r);
}
// channel groups per block: 32 = 512 contiguous bytes per row (256-byte runs measured 25 % slower); tensors with fewer channel groups
// (the 32 / 64 / 128-channel layers) get a block as narrow as the tensor, so that no lane idles on channels that do not exist
constexpr int RED_CG_MAX = 32;
inline int narrow_cg(int ngroups, int widest) {
  int cg = 4;
  while (cg < widest && cg < ngroups) cg *= 2;
  return cg;
}

// Generic two-value column reduction.  F(row vectors...) -> (a[e], b[e]) accumulated per channel element.
template <typename T, int MODE, int RED_CG>
__global__ __launch_bounds__(256) void colred_kernel(long M, int C, int RED_ROWS, const T* __restrict__ p0, int ld0,
                                                     const T* __restrict__ p1, int ld1, const T* __restrict__ p2,
                                                     int ld2, int relu, const float* __restrict__ mean,
                                                     const float* __restrict__ invstd, float* __restrict__ slab,
                                                     const float* __restrict__ mscale, const float* __restrict__ mshift, int sum_row) {
  // MODE 0: stats of p0            -> (sum x, sum x^2)
  // MODE 1: BN backward, p0 = dout, p1 = y (pre-BN), p2 = out (post activation; only read when relu)
  //                                -> (sum g, sum g*xhat),  g = dout * (out > 0)
  //         relu == 2: the activation was never stored (it was fused into the consuming depthwise conv); the mask is
  //         recomputed as y*mscale + mshift > 0, the very expression the consumer evaluated
  // MODE 2: column sum of p0       -> (sum x, 0)
  constexpr int KPV = Elem<T>::kPerVec;
  constexpr int RED_RL = 256 / RED_CG;   // row lanes
  constexpr int CW = RED_CG * KPV;   // channels per block
  __shared__ __attribute__((aligned(16))) float red[2][RED_RL][CW];   // first the per-channel vectors, then the partial sums
  const int cgl = threadIdx.x % RED_CG, rl = threadIdx.x / RED_CG;
  const int cg = blockIdx.x * RED_CG + cgl;
  const int c0 = cg * KPV;
  const bool cok = c0 < C;
  float a[KPV], b[KPV], mu[KPV], is[KPV], ms[KPV], mh[KPV];
#pragma unroll
  for (int e = 0; e < KPV; ++e) a[e] = b[e] = mu[e] = is[e] = ms[e] = mh[e] = 0.f;
  if (MODE == 1) {
    // per-channel vectors: one thread per channel fetches them, everybody picks its 8 (4) up from LDS
    float* vecs = &red[0][0][0];   // [4][CW] <= 2*RED_RL*CW floats
    for (int i = threadIdx.x; i < CW; i += 256) {
      const int c = blockIdx.x * CW + i;
      const bool ok = c < C;
      vecs[0 * CW + i] = ok ? mean[c] : 0.f;
      vecs[1 * CW + i] = ok ? invstd[c] : 0.f;
      vecs[2 * CW + i] = (ok && relu == 2) ? mscale[c] : 0.f;
      vecs[3 * CW + i] = (ok && relu == 2) ? mshift[c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < KPV; ++e) {
      const int i = cgl * KPV + e;
      mu[e] = vecs[0 * CW + i]; is[e] = vecs[1 * CW + i]; ms[e] = vecs[2 * CW + i]; mh[e] = vecs[3 * CW + i];
    }
    __syncthreads();   // the same LDS holds the partial sums below
  }
  const long rbeg = (long)blockIdx.y * RED_ROWS;
  const long rend = rbeg + RED_ROWS < M ? rbeg + RED_ROWS : M;
  constexpr int UR = 4;   // rows in flight per thread: 4 x (1..3) independent 16-byte loads
  if (cok) {
    for (long r0 = rbeg + rl; r0 < rend; r0 += RED_RL * UR) {
      vec16 v0[UR], v1[UR], v2[UR];
#pragma unroll
      for (int u = 0; u < UR; ++u) {
        const long r = r0 + RED_RL * u;
        const bool ok = r < rend;
        const long rr = ok ? r : rbeg;   // in range; its contribution is zeroed below
        v0[u] = ldg16(p0 + (size_t)rr * ld0 + c0);
        if (!ok) v0[u] = zero16();
        if (MODE == 1) {
          v1[u] = ldg16(p1 + (size_t)rr * ld1 + c0);
          if (relu == 1) v2[u] = ldg16(p2 + (size_t)rr * ld2 + c0);
        }
      }
#pragma unroll
      for (int u = 0; u < UR; ++u) {
        float x[KPV];
        unpack(v0[u], x, T());
        if (MODE == 0) {
#pragma unroll
          for (int e = 0; e < KPV; ++e) {
            a[e] += x[e];
            b[e] = fmaf(x[e], x[e], b[e]);
          }
        } else if (MODE == 2) {
#pragma unroll
          for (int e = 0; e < KPV; ++e) a[e] += x[e];
        } else {
          float y[KPV];
          unpack(v1[u], y, T());
          if (relu == 2) {
#pragma unroll
            for (int e = 0; e < KPV; ++e) x[e] = fmaf(y[e], ms[e], mh[e]) > 0.f ? x[e] : 0.f;
          } else if (relu) {
            float o[KPV];
            unpack(v2[u], o, T());
#pragma unroll
            for (int e = 0; e < KPV; ++e) x[e] = o[e] > 0.f ? x[e] : 0.f;
          }
#pragma unroll
          for (int e = 0; e < KPV; ++e) {
            a[e] += x[e];
            b[e] = fmaf(x[e], (y[e] - mu[e]) * is[e], b[e]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < KPV; ++e) {
    red[0][rl][cgl * KPV + e] = a[e];
    red[1][rl][cgl * KPV + e] = b[e];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * CW; i += 256) {
    const int which = i / CW, cl = i % CW;
    const int c = blockIdx.x * CW + cl;
    if (c < C) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < RED_RL; ++r) s += red[which][r][cl];
      if (sum_row) unsafeAtomicAdd(reinterpret_cast<double*>(slab) + (size_t)which * C + c, (double)s);      // bn_fin.h: SUM_ROW
      else slab[((size_t)which * gridDim.y + blockIdx.y) * C + c] = s;
    }
  }
}

__global__ __launch_bounds__(256) void slab_fold_kernel(float* slab, int rows, int C) {
  __shared__ double red[2][FIN_RL][FIN_CH];
  slab_fold_block(slab, rows, C, red);
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const BnFinArgs a) {
  __shared__ double red[2][FIN_RL][FIN_CH];
  bn_finalize_block(a, blockIdx.x, red);
}

__global__ void bn_eval_coeffs_kernel(int C, const float* gamma, const float* beta, const float* rm, const float* rv,
                                      float eps, float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = gamma[c] / sqrtf(rv[c] + eps);
  scale[c] = sc;
  shift[c] = beta[c] - rm[c] * sc;
}

template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(long M, int C, const T* __restrict__ y, int ldy,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const T* __restrict__ res, int ldr, int relu,
                                                       T* __restrict__ out, int ldo) {
  constexpr int KPV = Elem<T>::kPerVec;
  const int ngroups = C / KPV;
  const long total = M * ngroups;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(i % ngroups);
    const long r = i / ngroups;
    const int c0 = cg * KPV;
    float x[KPV];
    unpack(ldg16(y + (size_t)r * ldy + c0), x, T());
#pragma unroll
    for (int e = 0; e < KPV; ++e) x[e] = fmaf(x[e], scale[c0 + e], shift[c0 + e]);
    if (res != nullptr) {
      float q[KPV];
      unpack(ldg16(res + (size_t)r * ldr + c0), q, T());
#pragma unroll
      for (int e = 0; e < KPV; ++e) x[e] += q[e];
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < KPV; ++e) x[e] = fmaxf(x[e], 0.f);
    }
    vec16 v;
    pack(v, x, T());
    stg16(out + (size_t)r * ldo + c0, v);
  }
}

// The same pass in bn_bwd_apply_kernel's thread layout (round 5): CGW channel groups x (256 / CGW) row lanes per block, APPLY_ROWS rows per block, a
// thread's channels never change, so scale / shift live in registers and four rows are in flight per thread.  The grid-stride kernel above
// divides a 64-bit element index by the channel-group count and re-loads sixteen coefficients for every 16 bytes it moves (5.0 TB/s on the
// 728-channel block outputs where this layout's backward twin reaches 5.9); same arithmetic, same bits.
template <typename T, int CGW, bool FIN = false>      // (FIN: an instantiation of its own -- the finalize's 64 loads in flight cost 36 registers)
__global__ __launch_bounds__(256) void bn_apply_rows_kernel(long M, int C, int APPLY_ROWS, const T* __restrict__ y, int ldy,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const T* __restrict__ res, int ldr, int relu, T* __restrict__ out, int ldo,
                                                            const BnFinArgs fin) {
  constexpr int KPV = Elem<T>::kPerVec;
  constexpr int RL = 256 / CGW;
  const int cgl = threadIdx.x % CGW, rl = threadIdx.x / CGW;
  const int c0 = (blockIdx.x * CGW + cgl) * KPV;
  float sc[KPV], sh[KPV];
  if constexpr (FIN) {
    // dc_bn_apply_fin: the finalize of a short slab by every block for its own channels (bn_fin.h: slab_quad_sum2); block row 0 stores
    constexpr int CW = CGW * KPV;
    __shared__ float fincoef[2][CW];
    if constexpr (CW <= 256) {
      __shared__ double finred[2][4][CW];
      double s, q;
      slab_quad_sum2<CW>(fin.slab, fin.rows, C, blockIdx.x * CW, finred, s, q);
      if (threadIdx.x < CW) {
        const int c = blockIdx.x * CW + threadIdx.x;
        float a = 0.f, b = 0.f;
        if (c < C) bn_fin_coefs(fin, c, s, q, blockIdx.y == 0, a, b);
        fincoef[0][threadIdx.x] = a;
        fincoef[1][threadIdx.x] = b;
      }
    } else {
      for (int i = threadIdx.x; i < CW; i += 256) {
        const int c = blockIdx.x * CW + i;
        float a = 0.f, b = 0.f;
        if (c < C) {
          double s, q;
          slab_quad_sum2_thread(fin.slab, fin.rows, C, c, s, q);
          bn_fin_coefs(fin, c, s, q, blockIdx.y == 0, a, b);
        }
        fincoef[0][i] = a;
        fincoef[1][i] = b;
      }
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && fin.nbt != nullptr) *fin.nbt += 1;
    __syncthreads();
    if (c0 >= C) return;
#pragma unroll
    for (int e = 0; e < KPV; ++e) {
      sc[e] = fincoef[0][cgl * KPV + e];
      sh[e] = fincoef[1][cgl * KPV + e];
    }
  } else {
    if (c0 >= C) return;
#pragma unroll
    for (int e = 0; e < KPV; ++e) {
      sc[e] = scale[c0 + e];
      sh[e] = shift[c0 + e];
    }
  }
  const long rbeg = (long)blockIdx.y * APPLY_ROWS;
  const long rend = rbeg + APPLY_ROWS < M ? rbeg + APPLY_ROWS : M;
  constexpr int UR = 4;
  for (long r0 = rbeg + rl; r0 < rend; r0 += RL * UR) {
    vec16 vy[UR], vr[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const long r = r0 + RL * u;
      const long rr = r < rend ? r : rbeg;
      vy[u] = ldg16(y + (size_t)rr * ldy + c0);
      if (res != nullptr) vr[u] = ldg16(res + (size_t)rr * ldr + c0);
    }
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const long r = r0 + RL * u;
      if (r >= rend) break;
      float x[KPV];
      unpack(vy[u], x, T());
#pragma unroll
      for (int e = 0; e < KPV; ++e) x[e] = fmaf(x[e], sc[e], sh[e]);
      if (res != nullptr) {
        float q[KPV];
        unpack(vr[u], q, T());
#pragma unroll
        for (int e = 0; e < KPV; ++e) x[e] += q[e];
      }
      if (relu) {
#pragma unroll
        for (int e = 0; e < KPV; ++e) x[e] = fmaxf(x[e], 0.f);
      }
      vec16 v;
      pack(v, x, T());
      stg16(out + (size_t)r * ldo + c0, v);
    }
  }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(int C, const float* __restrict__ slab, int rows, int parts, float* dgamma,
                                                              float* dbeta) {
  __shared__ double red[2][FIN_RL][FIN_CH];
  const int c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1));
  double s = 0.0, q = 0.0;
  if (rows == SUM_ROW) {
    if (c < C) sum_row_load(slab, C, c, s, q);
  } else {
    slab_colsum2(SlabLoad{slab, slab + (size_t)rows * C, C, c, parts}, parts ? parts : rows, c < C, red, s, q, slab_is_short(rows, parts));
  }
  if (threadIdx.x >= FIN_CH || c >= C) return;
  dbeta[c] = (float)s;
  dgamma[c] = (float)q;
}

// dy = gamma*invstd * (g - dbeta/N - xhat*dgamma/N), g = dout masked by the ReLU.  Same thread layout as the column
// reductions (16 channel groups x 16 row lanes, APPLY_ROWS rows per block): a thread's channels never change, so its per-channel
// coefficients are computed once and live in registers -- the earlier grid-stride version re-loaded five vectors per element
// and ran at 3.2 TB/s where a copy reaches 4.9.
static int g_bn_cgw = 32, g_bn_rows = 32;
static int g_bn_fin_mul_f = 2, g_bn_fin_mul_b = 2;   // rows per block of the kernels that run a finalize themselves, in units of bn_rows ("bn_fin_mul_fwd" / "_bwd")
static int g_bn_apply_rows = 1;     // dc_bn_apply on bn_apply_rows_kernel (0: the grid-stride kernel; "bn_apply_rows")
template <typename T, int CGW>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(long M, int C, int APPLY_ROWS, float inv_count, const T* __restrict__ dout,
                                                           int lddo, const T* __restrict__ y, int ldy,
                                                           const T* __restrict__ out, int ldout, int relu,
                                                           const float* __restrict__ gamma, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                           T* __restrict__ dy, int lddy, T* __restrict__ gout, int ldg,
                                                           const float* __restrict__ mscale, const float* __restrict__ mshift,
                                                           const float* __restrict__ fin_slab, int fin_rows, float* dgamma_out, float* dbeta_out) {
  constexpr int KPV = Elem<T>::kPerVec;
  constexpr int RL = 256 / CGW;   // row lanes
  constexpr int CW = CGW * KPV;   // channels per block: 128..512
  const int cgl = threadIdx.x % CGW, rl = threadIdx.x / CGW;
  const int c0 = (blockIdx.x * CGW + cgl) * KPV;
  // dy = ca*g + cb*(x - mean) + cd.  The block's coefficients are computed once, by one thread per channel, and handed out
  // through LDS: fetching the seven per-channel vectors in every thread cost as many load instructions as 8 rows of data.
  __shared__ __attribute__((aligned(16))) float coef[6][CW];
  // dc_bn_bwd_apply_fin: the finalize of a SHORT slab (at most FIN_RL rows) done here, by every block for its own channels, in
  // bn_bwd_finalize_kernel's order (bn_fin.h: the quad order, fp64) and rounded to fp32 as that kernel stores them: same bits, one launch
  // and one dependent kernel boundary less per BatchNorm.  The row blocks of a channel block all compute the same numbers; block
  // row 0 stores the parameter gradients.
  double fsb = 0.0, fsg = 0.0;      // thread i < CW: the two sums of channel blockIdx.x * CW + i
  if constexpr (CW <= 256) {
    __shared__ double finred[2][4][CW];
    if (fin_slab != nullptr) slab_quad_sum2<CW>(fin_slab, fin_rows, C, blockIdx.x * CW, finred, fsb, fsg);
  }
  for (int i = threadIdx.x; i < CW; i += 256) {
    const int c = blockIdx.x * CW + i;
    float a = 0.f, b = 0.f, d = 0.f, m = 0.f, s1 = 0.f, s2 = 0.f;
    if (c < C) {
      float dg, db;
      if (fin_slab != nullptr) {
        double sb = fsb, sg = fsg;
        if constexpr (CW > 256) slab_quad_sum2_thread(fin_slab, fin_rows, C, c, sb, sg);
        db = (float)sb;
        dg = (float)sg;
        if (blockIdx.y == 0) {
          dbeta_out[c] = db;
          dgamma_out[c] = dg;
        }
      } else {
        dg = dgamma[c];
        db = dbeta[c];
      }
      const float is = invstd[c];
      a = gamma[c] * is;
      b = -a * is * dg * inv_count;
      d = -a * db * inv_count;
      m = mean[c];
      if (relu == 2) {
        s1 = mscale[c];
        s2 = mshift[c];
      }
    }
    coef[0][i] = a; coef[1][i] = b; coef[2][i] = d; coef[3][i] = m; coef[4][i] = s1; coef[5][i] = s2;
  }
  __syncthreads();
  if (c0 >= C) return;
  float ca[KPV], cb[KPV], cd[KPV], mu[KPV], ms[KPV], mh[KPV];
#pragma unroll
  for (int e = 0; e < KPV; ++e) {
    const int i = cgl * KPV + e;
    ca[e] = coef[0][i]; cb[e] = coef[1][i]; cd[e] = coef[2][i]; mu[e] = coef[3][i]; ms[e] = coef[4][i]; mh[e] = coef[5][i];
  }
  const long rbeg = (long)blockIdx.y * APPLY_ROWS;
  const long rend = rbeg + APPLY_ROWS < M ? rbeg + APPLY_ROWS : M;
  constexpr int UR = 4;
  for (long r0 = rbeg + rl; r0 < rend; r0 += RL * UR) {
    vec16 vg[UR], vx[UR], vo[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const long r = r0 + RL * u;
      const long rr = r < rend ? r : rbeg;
      vg[u] = ldg16(dout + (size_t)rr * lddo + c0);
      vx[u] = ldg16(y + (size_t)rr * ldy + c0);
      if (relu == 1) vo[u] = ldg16(out + (size_t)rr * ldout + c0);
    }
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const long r = r0 + RL * u;
      if (r >= rend) break;
      float g[KPV], x[KPV];
      unpack(vg[u], g, T());
      unpack(vx[u], x, T());
      if (relu == 2) {
#pragma unroll
        for (int e = 0; e < KPV; ++e) g[e] = fmaf(x[e], ms[e], mh[e]) > 0.f ? g[e] : 0.f;
      } else if (relu) {
        float o[KPV];
        unpack(vo[u], o, T());
#pragma unroll
        for (int e = 0; e < KPV; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f;
      }
      if (gout != nullptr) {
        vec16 v;
        pack(v, g, T());
        stg16(gout + (size_t)r * ldg + c0, v);
      }
#pragma unroll
      for (int e = 0; e < KPV; ++e) x[e] = fmaf(ca[e], g[e], fmaf(cb[e], x[e] - mu[e], cd[e]));
      vec16 v;
      pack(v, x, T());
      stg16(dy + (size_t)r * lddy + c0, v);
    }
  }
}

static int ew_blocks(long total) {
  long b = (total + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (int)b;
}

template <typename T, int MODE>
static int launch_colred(long M, int C, const void* p0, int ld0, const void* p1, int ld1, const void* p2, int ld2,
                         int relu, const float* mean, const float* invstd, float* slab, hipStream_t st,
                         const float* mscale = nullptr, const float* mshift = nullptr, int sum_row = 0) {
  const int RED_ROWS = sum_row ? red_rows_sum(M) : red_rows(M);
  const int ngroups = C / Elem<T>::kPerVec, cg = narrow_cg(ngroups, RED_CG_MAX);
  dim3 grid(cdiv(ngroups, cg), cdiv(M, RED_ROWS));
#define DC_COLRED(W) hipLaunchKernelGGL((colred_kernel<T, MODE, W>), grid, dim3(256), 0, st, M, C, RED_ROWS, (const T*)p0, ld0, (const T*)p1, ld1, (const T*)p2, ld2, relu, mean, invstd, slab, mscale, mshift, sum_row)
  if (cg == 32) DC_COLRED(32); else if (cg == 16) DC_COLRED(16); else if (cg == 8) DC_COLRED(8); else DC_COLRED(4);
#undef DC_COLRED
  DC_CHECK_LAUNCH();
  return 0;
}

// stage one of the two-stage fold of a large slab (bn_fin.h); returns the number of parts stage two reads (0: the slab is read as it is)
static int fold_large_slab(float* slab, int rows, int C, hipStream_t st) {
  const int parts = fin_parts(rows);
  if (parts > 0) hipLaunchKernelGGL(slab_fold_kernel, dim3(cdiv(C, FIN_CH), parts), dim3(256), 0, st, slab, rows, C);
  return parts;
}

}  // namespace dc

using namespace dc;

extern "C" int dc_bn_set_option(const char* name, int value) {
  if (strcmp(name, "bn_cgw") == 0 && (value == 16 || value == 32 || value == 64)) { g_bn_cgw = value; return 0; }
  if (strcmp(name, "bn_rows") == 0 && value >= 16 && value % 16 == 0) { g_bn_rows = value; return 0; }
  if (strcmp(name, "bn_apply_rows") == 0) { g_bn_apply_rows = value != 0; return 0; }
  if (strcmp(name, "bn_fin_mul_fwd") == 0 && value >= 1 && value <= 16) { g_bn_fin_mul_f = value; return 0; }
  if (strcmp(name, "bn_fin_mul_bwd") == 0 && value >= 1 && value <= 16) { g_bn_fin_mul_b = value; return 0; }
  return -1;
}

extern "C" int dc_bn_stat_rows(long M) { return cdiv(M, red_rows(M)); }

extern "C" int dc_bn_stats(int dtype, long M, int C, const void* x, int ldx, float* slab, void* stream) {
  if (int e = dc_check_view(x, ldx, C, dtype, "dc_bn_stats x")) return e;
  DC_REQUIRE(slab != nullptr && M > 0, "dc_bn_stats: bad argument");
  hipStream_t st = (hipStream_t)stream;
  return dtype == DC_BF16 ? launch_colred<bf16, 0>(M, C, x, ldx, nullptr, 0, nullptr, 0, 0, nullptr, nullptr, slab, st)
                          : launch_colred<float, 0>(M, C, x, ldx, nullptr, 0, nullptr, 0, 0, nullptr, nullptr, slab, st);
}

extern "C" size_t dc_colsum_workspace(long M, int C) { return (size_t)2 * cdiv(M, red_rows(M)) * C * sizeof(float); }

extern "C" int dc_colsum(int dtype, long M, int C, const void* dy, int lddy, float* out, void* workspace, void* stream) {
  if (int e = dc_check_view(dy, lddy, C, dtype, "dc_colsum dy")) return e;
  DC_REQUIRE(out != nullptr && workspace != nullptr && M > 0, "dc_colsum: bad argument");
  hipStream_t st = (hipStream_t)stream;
  float* slab = (float*)workspace;
  int e = dtype == DC_BF16 ? launch_colred<bf16, 2>(M, C, dy, lddy, nullptr, 0, nullptr, 0, 0, nullptr, nullptr, slab, st)
                           : launch_colred<float, 2>(M, C, dy, lddy, nullptr, 0, nullptr, 0, 0, nullptr, nullptr, slab, st);
  if (e) return e;
  // reuse the BN backward finalize: "dbeta" = column sum; the second output goes to the slab's own tail
  const int rows = cdiv(M, red_rows(M));
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, FIN_CH)), dim3(256), 0, st, C, (const float*)slab, rows, 0,
                     slab + (size_t)rows * C, out);
  DC_CHECK_LAUNCH();
  return 0;
}

extern "C" int dc_bn_finalize(int C, long count, float* slab, int rows, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                              float eps, float* scale, float* shift, float* save_mean, float* save_invstd, void* stream) {
  DC_REQUIRE(C > 0 && (rows > 0 || rows == SUM_ROW) && slab && gamma && beta && scale && shift, "dc_bn_finalize: bad argument");
  DC_REQUIRE(rows != SUM_ROW || ((uintptr_t)slab & 7) == 0, "dc_bn_finalize: a sum row is double[2][C]");
  if (count <= 1) return dc_fail("Expected more than 1 value per channel when training", __FILE__, __LINE__);
  const int parts = rows == SUM_ROW ? 0 : fold_large_slab(slab, rows, C, (hipStream_t)stream);
  const BnFinArgs a = bn_fin_args(C, count, slab, rows, parts, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, scale,
                                  shift, save_mean, save_invstd);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, FIN_CH)), dim3(256), 0, (hipStream_t)stream, a);
  DC_CHECK_LAUNCH();
  return 0;
}

extern "C" int dc_bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* running_mean,
                                 const float* running_var, float eps, float* scale, float* shift, void* stream) {
  DC_REQUIRE(C > 0 && gamma && beta && running_mean && running_var && scale && shift, "dc_bn_eval_coeffs: bad argument");
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, C, gamma, beta,
                     running_mean, running_var, eps, scale, shift);
  DC_CHECK_LAUNCH();
  return 0;
}

static int bn_apply_impl(int dtype, long M, int C, const void* y, int ldy, const float* scale, const float* shift,
                         const void* residual, int ldr, int relu, void* out, int ldo, void* stream, const BnFinArgs* finp, int rows_per_block) {
  BnFinArgs fin;
  if (finp != nullptr) fin = *finp; else fin.slab = nullptr;
  if (int e = dc_check_view(y, ldy, C, dtype, "dc_bn_apply y")) return e;
  if (int e = dc_check_view(out, ldo, C, dtype, "dc_bn_apply out")) return e;
  if (residual)
    if (int e = dc_check_view(residual, ldr, C, dtype, "dc_bn_apply residual")) return e;
  DC_REQUIRE(scale && shift && M > 0, "dc_bn_apply: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int kpv = dtype == DC_BF16 ? 8 : 4;
  if (g_bn_apply_rows) {
    // bn_bwd_apply's block shape (a block as narrow as the tensor, rows per block growing with the row lanes)
    const int cgw = narrow_cg(C / kpv, g_bn_cgw);
    int APPLY_ROWS = (rows_per_block > 0 ? rows_per_block : g_bn_rows) * (g_bn_cgw > cgw ? g_bn_cgw / cgw : 1);
    while (cdiv(M, APPLY_ROWS) > 65535) APPLY_ROWS *= 2;      // (a block walks its rows in a loop: any row count per block is served)
    {
      const dim3 blocks2(cdiv(C / kpv, cgw), cdiv(M, APPLY_ROWS));
#define BN_AR(TT, W) do { if (finp != nullptr) hipLaunchKernelGGL((bn_apply_rows_kernel<TT, W, true>), blocks2, dim3(256), 0, st, M, C, APPLY_ROWS, (const TT*)y, ldy, scale, shift, (const TT*)residual, ldr, relu, (TT*)out, ldo, fin); \
                          else hipLaunchKernelGGL((bn_apply_rows_kernel<TT, W, false>), blocks2, dim3(256), 0, st, M, C, APPLY_ROWS, (const TT*)y, ldy, scale, shift, (const TT*)residual, ldr, relu, (TT*)out, ldo, fin); } while (0)
      if (dtype == DC_BF16) { if (cgw == 64) BN_AR(bf16, 64); else if (cgw == 32) BN_AR(bf16, 32); else if (cgw == 16) BN_AR(bf16, 16); else if (cgw == 8) BN_AR(bf16, 8); else BN_AR(bf16, 4); }
      else                  { if (cgw == 64) BN_AR(float, 64); else if (cgw == 32) BN_AR(float, 32); else if (cgw == 16) BN_AR(float, 16); else if (cgw == 8) BN_AR(float, 8); else BN_AR(float, 4); }
#undef BN_AR
      DC_CHECK_LAUNCH();
      return 0;
    }
  }
  if (finp != nullptr) {
    // dc_bn_apply_fin where the row-block kernel does not run (option bn_apply_rows = 0, or more row blocks than one launch holds): the
    // finalize as a launch of its own, then the grid-stride apply -- the two-call sequence, same results
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, FIN_CH)), dim3(256), 0, st, fin);
    DC_CHECK_LAUNCH();
  }
  const int blocks = ew_blocks(M * (C / kpv));
  if (dtype == DC_BF16)
    hipLaunchKernelGGL(bn_apply_kernel<bf16>, dim3(blocks), dim3(256), 0, st, M, C, (const bf16*)y, ldy, scale, shift, (const bf16*)residual, ldr, relu, (bf16*)out, ldo);
  else
    hipLaunchKernelGGL(bn_apply_kernel<float>, dim3(blocks), dim3(256), 0, st, M, C, (const float*)y, ldy, scale, shift, (const float*)residual, ldr, relu, (float*)out, ldo);
  DC_CHECK_LAUNCH();
  return 0;
}

extern "C" int dc_bn_apply(int dtype, long M, int C, const void* y, int ldy, const float* scale, const float* shift,
                           const void* residual, int ldr, int relu, void* out, int ldo, void* stream) {
  return bn_apply_impl(dtype, M, C, y, ldy, scale, shift, residual, ldr, relu, out, ldo, stream, nullptr, 0);
}

// dc_bn_finalize + dc_bn_apply in one launch for a SHORT slab (rows <= dc_bn_bwd_apply_fin_max_rows()): every block sums the slab for its own
// channels (same order, same bits as dc_bn_finalize), block row 0 stores scale / shift / save_mean / save_invstd and the running statistics.
// Each block repeats 2 * rows * 256 loads, so it takes 64 rows of the tensor instead of 32.
extern "C" int dc_bn_apply_fin(int dtype, long M, int C, long count, const void* y, int ldy, const float* slab, int rows, const float* gamma,
                               const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                               float eps, float* scale, float* shift, float* save_mean, float* save_invstd, const void* residual, int ldr,
                               int relu, void* out, int ldo, void* stream) {
  DC_REQUIRE(C > 0 && slab && gamma && beta && scale && shift && ((rows > 0 && rows <= FIN_RL) || rows == SUM_ROW),
             "dc_bn_apply_fin: needs a slab of at most dc_bn_bwd_apply_fin_max_rows() rows, or a sum row (rows = -1)");
  DC_REQUIRE(((uintptr_t)slab & 15) == 0 && C % 4 == 0, "dc_bn_apply_fin: the slab is read with 16-byte loads (16-byte aligned, C a multiple of 4)");
  if (count <= 1) return dc_fail("Expected more than 1 value per channel when training", __FILE__, __LINE__);
  const BnFinArgs a = bn_fin_args(C, count, slab, rows, 0, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, scale,
                                  shift, save_mean, save_invstd);
  return bn_apply_impl(dtype, M, C, y, ldy, scale, shift, residual, ldr, relu, out, ldo, stream, &a, (rows == SUM_ROW ? 1 : g_bn_fin_mul_f) * g_bn_rows);
}

static int bn_bwd_reduce_impl(int dtype, long M, int C, const void* dout, int lddo, const void* y, int ldy,
                              const void* out, int ldout, int relu, const float* save_mean, const float* save_invstd,
                              float* slab, const float* mscale, const float* mshift, void* stream, int sum_row) {
  if (int e = dc_check_view(dout, lddo, C, dtype, "dc_bn_bwd_reduce dout")) return e;
  if (int e = dc_check_view(y, ldy, C, dtype, "dc_bn_bwd_reduce y")) return e;
  if (relu == 1)
    if (int e = dc_check_view(out, ldout, C, dtype, "dc_bn_bwd_reduce out")) return e;
  DC_REQUIRE(relu != 2 || (mscale && mshift), "dc_bn_bwd_reduce: relu == 2 needs the forward scale / shift vectors");
  DC_REQUIRE(save_mean && save_invstd && slab && M > 0, "dc_bn_bwd_reduce: bad argument");
  hipStream_t st = (hipStream_t)stream;
  DC_REQUIRE(!sum_row || ((uintptr_t)slab & 7) == 0, "dc_bn_bwd_reduce_sum: a sum row is double[2][C]");
  return dtype == DC_BF16 ? launch_colred<bf16, 1>(M, C, dout, lddo, y, ldy, out, ldout, relu, save_mean, save_invstd, slab, st, mscale, mshift, sum_row)
                          : launch_colred<float, 1>(M, C, dout, lddo, y, ldy, out, ldout, relu, save_mean, save_invstd, slab, st, mscale, mshift, sum_row);
}
extern "C" int dc_bn_bwd_reduce(int dtype, long M, int C, const void* dout, int lddo, const void* y, int ldy,
                                const void* out, int ldout, int relu, const float* save_mean, const float* save_invstd,
                                float* slab, const float* mscale, const float* mshift, void* stream) {
  return bn_bwd_reduce_impl(dtype, M, C, dout, lddo, y, ldy, out, ldout, relu, save_mean, save_invstd, slab, mscale, mshift, stream, 0);
}
// ... with the two sums added to a SUM ROW (bn_fin.h): slab is double[2][C], zeroed by the caller; every block adds its share (fp64 atomics, at
// most 2 048 per channel).  dc_bn_bwd_finalize / dc_bn_bwd_apply_fin take it with rows = -1: no slab fold, no finalize launch.
extern "C" int dc_bn_bwd_reduce_sum(int dtype, long M, int C, const void* dout, int lddo, const void* y, int ldy,
                                    const void* out, int ldout, int relu, const float* save_mean, const float* save_invstd,
                                    float* slab, const float* mscale, const float* mshift, void* stream) {
  return bn_bwd_reduce_impl(dtype, M, C, dout, lddo, y, ldy, out, ldout, relu, save_mean, save_invstd, slab, mscale, mshift, stream, 1);
}

extern "C" int dc_bn_bwd_finalize(int C, float* slab, int rows, float* dgamma, float* dbeta, void* stream) {
  DC_REQUIRE(C > 0 && (rows > 0 || rows == SUM_ROW) && slab && dgamma && dbeta, "dc_bn_bwd_finalize: bad argument");
  DC_REQUIRE(rows != SUM_ROW || ((uintptr_t)slab & 7) == 0, "dc_bn_bwd_finalize: a sum row is double[2][C]");
  const int parts = rows == SUM_ROW ? 0 : fold_large_slab(slab, rows, C, (hipStream_t)stream);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, FIN_CH)), dim3(256), 0, (hipStream_t)stream, C, (const float*)slab, rows, parts, dgamma, dbeta);
  DC_CHECK_LAUNCH();
  return 0;
}

static int bn_bwd_apply_impl(int dtype, long M, int C, long count, const void* dout, int lddo, const void* y, int ldy,
                             const void* out, int ldout, int relu, const float* gamma, const float* save_mean,
                             const float* save_invstd, const float* dgamma, const float* dbeta, void* dy, int lddy,
                             void* g_out, int ldg, const float* mscale, const float* mshift, void* stream,
                             const float* fin_slab, int fin_rows, float* dgamma_out, float* dbeta_out) {
  if (int e = dc_check_view(dout, lddo, C, dtype, "dc_bn_bwd_apply dout")) return e;
  if (int e = dc_check_view(y, ldy, C, dtype, "dc_bn_bwd_apply y")) return e;
  if (int e = dc_check_view(dy, lddy, C, dtype, "dc_bn_bwd_apply dy")) return e;
  if (relu == 1)
    if (int e = dc_check_view(out, ldout, C, dtype, "dc_bn_bwd_apply out")) return e;
  DC_REQUIRE(relu != 2 || (mscale && mshift), "dc_bn_bwd_apply: relu == 2 needs the forward scale / shift vectors");
  if (g_out)
    if (int e = dc_check_view(g_out, ldg, C, dtype, "dc_bn_bwd_apply g_out")) return e;
  DC_REQUIRE(gamma && save_mean && save_invstd && ((dgamma && dbeta) || fin_slab) && M > 0 && count > 0, "dc_bn_bwd_apply: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int kpv = dtype == DC_BF16 ? 8 : 4;
  // a block as narrow as the tensor (see narrow_cg); the rows per block grow with the row lanes so that every thread has its four rows
  const int cgw = narrow_cg(C / kpv, g_bn_cgw);
  int APPLY_ROWS = g_bn_rows * (fin_slab != nullptr && fin_rows != SUM_ROW ? g_bn_fin_mul_b : 1) * (g_bn_cgw > cgw ? g_bn_cgw / cgw : 1);
  while (cdiv(M, APPLY_ROWS) > 65535) APPLY_ROWS *= 2;      // (a block walks its rows in a loop: any row count per block is served)
  const dim3 blocks(cdiv(C / kpv, cgw), cdiv(M, APPLY_ROWS));
  const float inv = 1.0f / (float)count;
#define BN_BA(TT, W) hipLaunchKernelGGL((bn_bwd_apply_kernel<TT, W>), blocks, dim3(256), 0, st, M, C, APPLY_ROWS, inv, (const TT*)dout, lddo, (const TT*)y, ldy, (const TT*)out, ldout, relu, gamma, save_mean, save_invstd, dgamma, dbeta, (TT*)dy, lddy, (TT*)g_out, ldg, mscale, mshift, fin_slab, fin_rows, dgamma_out, dbeta_out)
  if (dtype == DC_BF16) { if (cgw == 64) BN_BA(bf16, 64); else if (cgw == 32) BN_BA(bf16, 32); else if (cgw == 16) BN_BA(bf16, 16); else if (cgw == 8) BN_BA(bf16, 8); else BN_BA(bf16, 4); }
  else                  { if (cgw == 64) BN_BA(float, 64); else if (cgw == 32) BN_BA(float, 32); else if (cgw == 16) BN_BA(float, 16); else if (cgw == 8) BN_BA(float, 8); else BN_BA(float, 4); }
#undef BN_BA
  DC_CHECK_LAUNCH();
  return 0;
}

extern "C" int dc_bn_bwd_apply(int dtype, long M, int C, long count, const void* dout, int lddo, const void* y, int ldy,
                               const void* out, int ldout, int relu, const float* gamma, const float* save_mean,
                               const float* save_invstd, const float* dgamma, const float* dbeta, void* dy, int lddy,
                               void* g_out, int ldg, const float* mscale, const float* mshift, void* stream) {
  return bn_bwd_apply_impl(dtype, M, C, count, dout, lddo, y, ldy, out, ldout, relu, gamma, save_mean, save_invstd, dgamma, dbeta, dy, lddy, g_out, ldg,
                           mscale, mshift, stream, nullptr, 0, nullptr, nullptr);
}

extern "C" int dc_bn_bwd_apply_fin_max_rows(void) { return FIN_RL; }

extern "C" int dc_bn_bwd_apply_fin(int dtype, long M, int C, long count, const void* dout, int lddo, const void* y, int ldy,
                                   const void* out, int ldout, int relu, const float* gamma, const float* save_mean,
                                   const float* save_invstd, const float* slab, int rows, float* dgamma, float* dbeta, void* dy, int lddy,
                                   void* g_out, int ldg, const float* mscale, const float* mshift, void* stream) {
  DC_REQUIRE(slab != nullptr && ((rows > 0 && rows <= FIN_RL) || rows == SUM_ROW) && dgamma && dbeta,
             "dc_bn_bwd_apply_fin: needs a slab of at most dc_bn_bwd_apply_fin_max_rows() rows, or a sum row (rows = -1)");
  DC_REQUIRE(((uintptr_t)slab & 15) == 0 && C % 4 == 0, "dc_bn_bwd_apply_fin: the slab is read with 16-byte loads (16-byte aligned, C a multiple of 4)");
  return bn_bwd_apply_impl(dtype, M, C, count, dout, lddo, y, ldy, out, ldout, relu, gamma, save_mean, save_invstd, nullptr, nullptr, dy, lddy, g_out, ldg,
                           mscale, mshift, stream, slab, rows, dgamma, dbeta);
}
