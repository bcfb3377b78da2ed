// Depthwise 3x3 convolution of SeparableConv2d_same (fixed_padding + groups=C conv), NHWC, HBM-bound.
// One lane owns one 16-byte channel group (8 bf16 / 4 f32 channels) so every global access is a full vector and
// a wave covers up to 1 KiB of contiguous channels; each thread walks PX output pixels along W with the nine
// per-channel weights held in registers.  Weights are the fp32 master tensor [C][1][3][3] itself.
#include "common.h"

namespace dc {

constexpr int DW_PX = 4;  // output pixels per thread (along W)

template <typename T>
__device__ inline void load_w9(const float* __restrict__ w, int c0, int C, float (&wr)[9][Elem<T>::kPerVec]) {
  constexpr int KPV = Elem<T>::kPerVec;
#pragma unroll
  for (int e = 0; e < KPV; ++e)
#pragma unroll
    for (int t = 0; t < 9; ++t) wr[t][e] = (c0 + e < C) ? w[(size_t)(c0 + e) * 9 + t] : 0.f;
}

// mode 0: forward   out[n,oy,ox] = sum_t in[n, oy*s - d + ky*d, ox*s - d + kx*d] * w[t]
// mode 1: dgrad     out[n,iy,ix] = sum_t in[n, (iy + d - ky*d)/s, (ix + d - kx*d)/s] * w[t]   (when divisible) [+ addend]
template <typename T, int MODE>
__global__ __launch_bounds__(256) void dw_kernel(const T* __restrict__ in, int ldin, const float* __restrict__ w,
                                                 const T* __restrict__ addend, int ldadd, T* __restrict__ out,
                                                 int ldout, int N, int Hin, int Win, int Hout, int Wout, int C,
                                                 int stride, int dil) {
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

// Weight gradient: dw[c][t] = sum over output pixels of dy[n,oy,ox,c] * x[n, oy*s - d + ky*d, ox*s - d + kx*d, c].
// grid (channel-group chunks of 64, pixel blocks); lane <-> channel group, the block's 4 waves take interleaved
// pixels, are combined through LDS and leave one partial row per block in slab[pixel block][9][C].
template <typename T>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ dy,
                                                       int lddy, float* __restrict__ slab, int N, int Hi, int Wi,
                                                       int Ho, int Wo, int C, int stride, int dil, int pix_per_block) {
  constexpr int KPV = Elem<T>::kPerVec;
  __shared__ float red[4][9][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cg = blockIdx.x * 64 + lane;
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
    for (long pix = pbeg + wave; pix < pend; pix += 4) {
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
#pragma unroll
          for (int e = 0; e < KPV; ++e) acc[ky * 3 + kx][e] = fmaf(g[e], f[e], acc[ky * 3 + kx][e]);
        }
      }
    }
  }
  // combine the 4 waves, one channel element at a time (keeps LDS at 9 KiB)
#pragma unroll
  for (int e = 0; e < KPV; ++e) {
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 9; ++t) red[wave][t][lane] = acc[t][e];
    __syncthreads();
    for (int i = threadIdx.x; i < 9 * 64; i += 256) {
      const int t = i / 64, l = i % 64;
      const int c = (blockIdx.x * 64 + l) * KPV + e;
      if (c < C) slab[((size_t)blockIdx.y * 9 + t) * C + c] = red[0][t][l] + red[1][t][l] + red[2][t][l] + red[3][t][l];
    }
  }
}

// grad[c][t] = sum over rows of slab[row][t][c]
__global__ void dw_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ grad, int rows, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 9 * C) return;
  const int t = i / C, c = i % C;
  double a = 0.0;
  for (int r = 0; r < rows; ++r) a += (double)slab[((size_t)r * 9 + t) * C + c];
  grad[(size_t)c * 9 + t] = (float)a;
}

static int dw_out(int Hi, int stride, int dil) { return (Hi + 2 * dil - 2 * dil - 1) / stride + 1; }
static int dw_pix_per_block(long P) {
  long ppb = (P + 255) / 256;  // at most 256 pixel blocks
  if (ppb < 64) ppb = 64;
  return (int)ppb;
}

}  // namespace dc

using namespace dc;

template <typename T, int MODE>
static int launch_dw(const void* in, int ldin, const float* w, const void* addend, int ldadd, void* out, int ldout,
                     int N, int Hin, int Win, int Hout, int Wout, int C, int stride, int dil, hipStream_t st) {
  const long total = (long)N * Hout * ((Wout + DW_PX - 1) / DW_PX) * (C / Elem<T>::kPerVec);
  long blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL((dw_kernel<T, MODE>), dim3((int)blocks), dim3(256), 0, st, (const T*)in, ldin, w,
                     (const T*)addend, ldadd, (T*)out, ldout, N, Hin, Win, Hout, Wout, C, stride, dil);
  DC_CHECK_LAUNCH();
  return 0;
}

static int dw_check(int dtype, int C, int stride, int dil, int N, int Hi, int Wi) {
  DC_REQUIRE(dtype == DC_F32 || dtype == DC_BF16, "dc_dwconv: bad dtype");
  DC_REQUIRE(stride == 1 || stride == 2, "dc_dwconv: stride must be 1 or 2");
  DC_REQUIRE(dil >= 1 && C > 0 && N > 0 && Hi > 0 && Wi > 0, "dc_dwconv: bad shape");
  return 0;
}

extern "C" int dc_dwconv_fwd(int dtype, int C, int stride, int dil, int N, int Hi, int Wi, const void* x, int ldx,
                             const float* w, void* y, int ldy, void* stream) {
  if (int e = dw_check(dtype, C, stride, dil, N, Hi, Wi)) return e;
  if (int e = dc_check_view(x, ldx, C, dtype, "dc_dwconv_fwd x")) return e;
  if (int e = dc_check_view(y, ldy, C, dtype, "dc_dwconv_fwd y")) return e;
  DC_REQUIRE(w != nullptr, "dc_dwconv_fwd: null weights");
  const int Ho = dw_out(Hi, stride, dil), Wo = dw_out(Wi, stride, dil);
  hipStream_t st = (hipStream_t)stream;
  return dtype == DC_BF16 ? launch_dw<bf16, 0>(x, ldx, w, nullptr, 0, y, ldy, N, Hi, Wi, Ho, Wo, C, stride, dil, st)
                          : launch_dw<float, 0>(x, ldx, w, nullptr, 0, y, ldy, N, Hi, Wi, Ho, Wo, C, stride, dil, st);
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
  // gather from dy [Ho,Wo] into dx [Hi,Wi]
  return dtype == DC_BF16 ? launch_dw<bf16, 1>(dy, lddy, w, addend, ldadd, dx, lddx, N, Ho, Wo, Hi, Wi, C, stride, dil, st)
                          : launch_dw<float, 1>(dy, lddy, w, addend, ldadd, dx, lddx, N, Ho, Wo, Hi, Wi, C, stride, dil, st);
}

extern "C" size_t dc_dwconv_wgrad_workspace(int C, int N, int Hi, int Wi, int stride) {
  const int Ho = dw_out(Hi, stride, 1), Wo = dw_out(Wi, stride, 1);
  const long P = (long)N * Ho * Wo;
  const int ppb = dw_pix_per_block(P);
  const int rows = (int)((P + ppb - 1) / ppb);
  return (size_t)rows * 9 * C * sizeof(float);
}

extern "C" int dc_dwconv_wgrad(int dtype, int C, int stride, int dil, int N, int Hi, int Wi, const void* x, int ldx,
                               const void* dy, int lddy, void* workspace, float* grad_w, void* stream) {
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
  dim3 grid(cdiv(C / kpv, 64), rows);
  if (dtype == DC_BF16)
    hipLaunchKernelGGL(dw_wgrad_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)x, ldx, (const bf16*)dy, lddy, (float*)workspace, N, Hi, Wi, Ho, Wo, C, stride, dil, ppb);
  else
    hipLaunchKernelGGL(dw_wgrad_kernel<float>, grid, dim3(256), 0, st, (const float*)x, ldx, (const float*)dy, lddy, (float*)workspace, N, Hi, Wi, Ho, Wo, C, stride, dil, ppb);
  DC_CHECK_LAUNCH();
  hipLaunchKernelGGL(dw_wgrad_reduce_kernel, dim3(cdiv(9 * C, 256)), dim3(256), 0, st, (const float*)workspace, grad_w, rows, C);
  DC_CHECK_LAUNCH();
  return 0;
}
