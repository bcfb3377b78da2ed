// Train/eval BatchNorm2d fused with ReLU and the residual add, NHWC, all HBM-bound streaming kernels.
//
// Column reductions (statistics forward; dgamma/dbeta backward) use one lane per 16-byte channel group, 16 groups
// side by side (256 contiguous bytes per row), 16 row-lanes per workgroup, fp32 partials per 256-row block in a
// slab, and a tiny finalize kernel that combines the slab rows in fp64 in a fixed order (deterministic, and it
// avoids the E[x^2]-E[x]^2 cancellation in fp32).
#include "common.h"

namespace dc {

constexpr int RED_ROWS = 256;  // rows per reduction block
constexpr int RED_CG = 16;     // channel groups per block

// Generic two-value column reduction.  F(row vectors...) -> (a[e], b[e]) accumulated per channel element.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void colred_kernel(long M, int C, const T* __restrict__ p0, int ld0,
                                                     const T* __restrict__ p1, int ld1, const T* __restrict__ p2,
                                                     int ld2, int relu, const float* __restrict__ mean,
                                                     const float* __restrict__ invstd, float* __restrict__ slab,
                                                     const float* __restrict__ mscale, const float* __restrict__ mshift) {
  // MODE 0: stats of p0            -> (sum x, sum x^2)
  // MODE 1: BN backward, p0 = dout, p1 = y (pre-BN), p2 = out (post activation; only read when relu)
  //                                -> (sum g, sum g*xhat),  g = dout * (out > 0)
  //         relu == 2: the activation was never stored (it was fused into the consuming depthwise conv); the mask is
  //         recomputed as y*mscale + mshift > 0, the very expression the consumer evaluated
  // MODE 2: column sum of p0       -> (sum x, 0)
  constexpr int KPV = Elem<T>::kPerVec;
  __shared__ float red[2][16][RED_CG * KPV];
  const int cgl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int cg = blockIdx.x * RED_CG + cgl;
  const int c0 = cg * KPV;
  const bool cok = c0 < C;
  float a[KPV], b[KPV], mu[KPV], is[KPV];
#pragma unroll
  for (int e = 0; e < KPV; ++e) {
    a[e] = b[e] = 0.f;
    mu[e] = (MODE == 1 && cok) ? mean[c0 + e] : 0.f;
    is[e] = (MODE == 1 && cok) ? invstd[c0 + e] : 0.f;
  }
  const long rbeg = (long)blockIdx.y * RED_ROWS;
  const long rend = rbeg + RED_ROWS < M ? rbeg + RED_ROWS : M;
  if (cok) {
    for (long r = rbeg + rl; r < rend; r += 16) {
      float x[KPV];
      unpack(ldg16(p0 + (size_t)r * ld0 + c0), x, T());
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
        unpack(ldg16(p1 + (size_t)r * ld1 + c0), y, T());
        if (relu == 2) {
#pragma unroll
          for (int e = 0; e < KPV; ++e) x[e] = fmaf(y[e], mscale[c0 + e], mshift[c0 + e]) > 0.f ? x[e] : 0.f;
        } else if (relu) {
          float o[KPV];
          unpack(ldg16(p2 + (size_t)r * ld2 + c0), o, T());
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
#pragma unroll
  for (int e = 0; e < KPV; ++e) {
    red[0][rl][cgl * KPV + e] = a[e];
    red[1][rl][cgl * KPV + e] = b[e];
  }
  __syncthreads();
  const int which = threadIdx.x / (RED_CG * KPV), cl = threadIdx.x % (RED_CG * KPV);
  if (which < 2) {
    const int c = blockIdx.x * RED_CG * KPV + cl;
    if (c < C) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) s += red[which][r][cl];
      slab[((size_t)which * gridDim.y + blockIdx.y) * C + c] = s;
    }
  }
}

// Column sums of a [rows][C] fp32 slab in fp64: 4 channels x 64 row-lanes per 256-thread block, so that even the
// 3456-row slabs of the 384x576 layers cost ~50 dependent loads per thread instead of thousands.
constexpr int FIN_CH = 4, FIN_RL = 64;
__device__ inline void slab_colsum2(const float* __restrict__ s0, const float* __restrict__ s1, int rows, int C, int c,
                                    bool ok, double (&red)[2][FIN_RL][FIN_CH], double& a, double& b) {
  const int cl = threadIdx.x & (FIN_CH - 1), rl = threadIdx.x / FIN_CH;
  double x0 = 0.0, x1 = 0.0, y0 = 0.0, y1 = 0.0;
  if (ok) {
    int r = rl;
    for (; r + FIN_RL < rows; r += 2 * FIN_RL) {
      x0 += (double)s0[(size_t)r * C + c];
      x1 += (double)s0[(size_t)(r + FIN_RL) * C + c];
      y0 += (double)s1[(size_t)r * C + c];
      y1 += (double)s1[(size_t)(r + FIN_RL) * C + c];
    }
    if (r < rows) {
      x0 += (double)s0[(size_t)r * C + c];
      y0 += (double)s1[(size_t)r * C + c];
    }
  }
  red[0][rl][cl] = x0 + x1;
  red[1][rl][cl] = y0 + y1;
  __syncthreads();
  a = b = 0.0;
  if (threadIdx.x < FIN_CH) {
#pragma unroll 8
    for (int i = 0; i < FIN_RL; ++i) {
      a += red[0][i][threadIdx.x];
      b += red[1][i][threadIdx.x];
    }
  }
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(int C, double inv_count, double unbias, const float* __restrict__ slab,
                                                          int rows, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* running_mean,
                                                          float* running_var, int64_t* nbt, float momentum, float eps,
                                                          float* scale, float* shift, float* save_mean, float* save_invstd) {
  __shared__ double red[2][FIN_RL][FIN_CH];
  const int c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1));
  double s, q;
  slab_colsum2(slab, slab + (size_t)rows * C, rows, C, c, c < C, red, s, q);
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt != nullptr) *nbt += 1;
  if (threadIdx.x >= FIN_CH || c >= C) return;
  const double mean = s * inv_count;
  double var = q * inv_count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma[c], b = beta[c];
  const float sc = g * invstd;
  scale[c] = sc;
  shift[c] = b - (float)mean * sc;
  if (save_mean) save_mean[c] = (float)mean;
  if (save_invstd) save_invstd[c] = invstd;
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
  if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * unbias);
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

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(int C, const float* __restrict__ slab, int rows, float* dgamma,
                                                              float* dbeta) {
  __shared__ double red[2][FIN_RL][FIN_CH];
  const int c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1));
  double s, q;
  slab_colsum2(slab, slab + (size_t)rows * C, rows, C, c, c < C, red, s, q);
  if (threadIdx.x >= FIN_CH || c >= C) return;
  dbeta[c] = (float)s;
  dgamma[c] = (float)q;
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(long M, int C, float inv_count, const T* __restrict__ dout,
                                                           int lddo, const T* __restrict__ y, int ldy,
                                                           const T* __restrict__ out, int ldout, int relu,
                                                           const float* __restrict__ gamma, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                           T* __restrict__ dy, int lddy, T* __restrict__ gout, int ldg,
                                                           const float* __restrict__ mscale, const float* __restrict__ mshift) {
  constexpr int KPV = Elem<T>::kPerVec;
  const int ngroups = C / KPV;
  const long total = M * ngroups;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(i % ngroups);
    const long r = i / ngroups;
    const int c0 = cg * KPV;
    float g[KPV], x[KPV];
    unpack(ldg16(dout + (size_t)r * lddo + c0), g, T());
    unpack(ldg16(y + (size_t)r * ldy + c0), x, T());
    if (relu == 2) {
#pragma unroll
      for (int e = 0; e < KPV; ++e) g[e] = fmaf(x[e], mscale[c0 + e], mshift[c0 + e]) > 0.f ? g[e] : 0.f;
    } else if (relu) {
      float o[KPV];
      unpack(ldg16(out + (size_t)r * ldout + c0), o, T());
#pragma unroll
      for (int e = 0; e < KPV; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f;
    }
    if (gout != nullptr) {
      vec16 v;
      pack(v, g, T());
      stg16(gout + (size_t)r * ldg + c0, v);
    }
#pragma unroll
    for (int e = 0; e < KPV; ++e) {
      const int c = c0 + e;
      const float is = invstd[c];
      const float xhat = (x[e] - mean[c]) * is;
      x[e] = gamma[c] * is * (g[e] - dbeta[c] * inv_count - xhat * dgamma[c] * inv_count);
    }
    vec16 v;
    pack(v, x, T());
    stg16(dy + (size_t)r * lddy + c0, v);
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
                         const float* mscale = nullptr, const float* mshift = nullptr) {
  dim3 grid(cdiv(C / Elem<T>::kPerVec, RED_CG), cdiv(M, RED_ROWS));
  hipLaunchKernelGGL((colred_kernel<T, MODE>), grid, dim3(256), 0, st, M, C, (const T*)p0, ld0, (const T*)p1, ld1,
                     (const T*)p2, ld2, relu, mean, invstd, slab, mscale, mshift);
  DC_CHECK_LAUNCH();
  return 0;
}

}  // namespace dc

using namespace dc;

extern "C" int dc_bn_stat_rows(long M) { return cdiv(M, RED_ROWS); }

extern "C" int dc_bn_stats(int dtype, long M, int C, const void* x, int ldx, float* slab, void* stream) {
  if (int e = dc_check_view(x, ldx, C, dtype, "dc_bn_stats x")) return e;
  DC_REQUIRE(slab != nullptr && M > 0, "dc_bn_stats: bad argument");
  hipStream_t st = (hipStream_t)stream;
  return dtype == DC_BF16 ? launch_colred<bf16, 0>(M, C, x, ldx, nullptr, 0, nullptr, 0, 0, nullptr, nullptr, slab, st)
                          : launch_colred<float, 0>(M, C, x, ldx, nullptr, 0, nullptr, 0, 0, nullptr, nullptr, slab, st);
}

extern "C" size_t dc_colsum_workspace(long M, int C) { return (size_t)2 * cdiv(M, RED_ROWS) * C * sizeof(float); }

extern "C" int dc_colsum(int dtype, long M, int C, const void* dy, int lddy, float* out, void* workspace, void* stream) {
  if (int e = dc_check_view(dy, lddy, C, dtype, "dc_colsum dy")) return e;
  DC_REQUIRE(out != nullptr && workspace != nullptr && M > 0, "dc_colsum: bad argument");
  hipStream_t st = (hipStream_t)stream;
  float* slab = (float*)workspace;
  int e = dtype == DC_BF16 ? launch_colred<bf16, 2>(M, C, dy, lddy, nullptr, 0, nullptr, 0, 0, nullptr, nullptr, slab, st)
                           : launch_colred<float, 2>(M, C, dy, lddy, nullptr, 0, nullptr, 0, 0, nullptr, nullptr, slab, st);
  if (e) return e;
  // reuse the BN backward finalize: "dbeta" = column sum; the second output goes to the slab's own tail
  const int rows = cdiv(M, RED_ROWS);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, FIN_CH)), dim3(256), 0, st, C, (const float*)slab, rows,
                     slab + (size_t)rows * C, out);
  DC_CHECK_LAUNCH();
  return 0;
}

extern "C" int dc_bn_finalize(int C, long count, const float* slab, int rows, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                              float eps, float* scale, float* shift, float* save_mean, float* save_invstd, void* stream) {
  DC_REQUIRE(C > 0 && rows > 0 && slab && gamma && beta && scale && shift, "dc_bn_finalize: bad argument");
  if (count <= 1) return dc_fail("Expected more than 1 value per channel when training", __FILE__, __LINE__);
  const double unbias = (double)count / (double)(count - 1);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, FIN_CH)), dim3(256), 0, (hipStream_t)stream, C, 1.0 / (double)count,
                     unbias, slab, rows, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps,
                     scale, shift, save_mean, save_invstd);
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

extern "C" int dc_bn_apply(int dtype, long M, int C, const void* y, int ldy, const float* scale, const float* shift,
                           const void* residual, int ldr, int relu, void* out, int ldo, void* stream) {
  if (int e = dc_check_view(y, ldy, C, dtype, "dc_bn_apply y")) return e;
  if (int e = dc_check_view(out, ldo, C, dtype, "dc_bn_apply out")) return e;
  if (residual)
    if (int e = dc_check_view(residual, ldr, C, dtype, "dc_bn_apply residual")) return e;
  DC_REQUIRE(scale && shift && M > 0, "dc_bn_apply: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int kpv = dtype == DC_BF16 ? 8 : 4;
  const int blocks = ew_blocks(M * (C / kpv));
  if (dtype == DC_BF16)
    hipLaunchKernelGGL(bn_apply_kernel<bf16>, dim3(blocks), dim3(256), 0, st, M, C, (const bf16*)y, ldy, scale, shift, (const bf16*)residual, ldr, relu, (bf16*)out, ldo);
  else
    hipLaunchKernelGGL(bn_apply_kernel<float>, dim3(blocks), dim3(256), 0, st, M, C, (const float*)y, ldy, scale, shift, (const float*)residual, ldr, relu, (float*)out, ldo);
  DC_CHECK_LAUNCH();
  return 0;
}

extern "C" int dc_bn_bwd_reduce(int dtype, long M, int C, const void* dout, int lddo, const void* y, int ldy,
                                const void* out, int ldout, int relu, const float* save_mean, const float* save_invstd,
                                float* slab, const float* mscale, const float* mshift, void* stream) {
  if (int e = dc_check_view(dout, lddo, C, dtype, "dc_bn_bwd_reduce dout")) return e;
  if (int e = dc_check_view(y, ldy, C, dtype, "dc_bn_bwd_reduce y")) return e;
  if (relu == 1)
    if (int e = dc_check_view(out, ldout, C, dtype, "dc_bn_bwd_reduce out")) return e;
  DC_REQUIRE(relu != 2 || (mscale && mshift), "dc_bn_bwd_reduce: relu == 2 needs the forward scale / shift vectors");
  DC_REQUIRE(save_mean && save_invstd && slab && M > 0, "dc_bn_bwd_reduce: bad argument");
  hipStream_t st = (hipStream_t)stream;
  return dtype == DC_BF16 ? launch_colred<bf16, 1>(M, C, dout, lddo, y, ldy, out, ldout, relu, save_mean, save_invstd, slab, st, mscale, mshift)
                          : launch_colred<float, 1>(M, C, dout, lddo, y, ldy, out, ldout, relu, save_mean, save_invstd, slab, st, mscale, mshift);
}

extern "C" int dc_bn_bwd_finalize(int C, const float* slab, int rows, float* dgamma, float* dbeta, void* stream) {
  DC_REQUIRE(C > 0 && rows > 0 && slab && dgamma && dbeta, "dc_bn_bwd_finalize: bad argument");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, FIN_CH)), dim3(256), 0, (hipStream_t)stream, C, slab, rows, dgamma, dbeta);
  DC_CHECK_LAUNCH();
  return 0;
}

extern "C" int dc_bn_bwd_apply(int dtype, long M, int C, long count, const void* dout, int lddo, const void* y, int ldy,
                               const void* out, int ldout, int relu, const float* gamma, const float* save_mean,
                               const float* save_invstd, const float* dgamma, const float* dbeta, void* dy, int lddy,
                               void* g_out, int ldg, const float* mscale, const float* mshift, void* stream) {
  if (int e = dc_check_view(dout, lddo, C, dtype, "dc_bn_bwd_apply dout")) return e;
  if (int e = dc_check_view(y, ldy, C, dtype, "dc_bn_bwd_apply y")) return e;
  if (int e = dc_check_view(dy, lddy, C, dtype, "dc_bn_bwd_apply dy")) return e;
  if (relu == 1)
    if (int e = dc_check_view(out, ldout, C, dtype, "dc_bn_bwd_apply out")) return e;
  DC_REQUIRE(relu != 2 || (mscale && mshift), "dc_bn_bwd_apply: relu == 2 needs the forward scale / shift vectors");
  if (g_out)
    if (int e = dc_check_view(g_out, ldg, C, dtype, "dc_bn_bwd_apply g_out")) return e;
  DC_REQUIRE(gamma && save_mean && save_invstd && dgamma && dbeta && M > 0 && count > 0, "dc_bn_bwd_apply: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int kpv = dtype == DC_BF16 ? 8 : 4;
  const int blocks = ew_blocks(M * (C / kpv));
  const float inv = 1.0f / (float)count;
  if (dtype == DC_BF16)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16>, dim3(blocks), dim3(256), 0, st, M, C, inv, (const bf16*)dout, lddo, (const bf16*)y, ldy, (const bf16*)out, ldout, relu, gamma, save_mean, save_invstd, dgamma, dbeta, (bf16*)dy, lddy, (bf16*)g_out, ldg, mscale, mshift);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(blocks), dim3(256), 0, st, M, C, inv, (const float*)dout, lddo, (const float*)y, ldy, (const float*)out, ldout, relu, gamma, save_mean, save_invstd, dgamma, dbeta, (float*)dy, lddy, (float*)g_out, ldg, mscale, mshift);
  DC_CHECK_LAUNCH();
  return 0;
}
