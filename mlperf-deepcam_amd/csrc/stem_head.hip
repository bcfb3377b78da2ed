// The two ends of the network, both HBM/VALU-bound and too thin for MFMA tiles:
//   stem  Conv2d(16->32, k3, s2, p1) that reads the caller's NCHW fp32 batch in place (no layout pass) and writes NHWC
//   head  ConvTranspose2d(256->3, k3, s2, p1, op1) that reads NHWC and writes the NCHW fp32 logits of the reference API
#include "common.h"

namespace dc {

// ------------------------------------------------------------------------------------------------- stem forward
constexpr int STEM_CO = 32;

template <typename T>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       T* __restrict__ y, int ldy, float* __restrict__ slab, int N,
                                                       int Cin, int H, int W, int Ho, int Wo) {
  extern __shared__ __attribute__((aligned(16))) float sw[];  // [Cin*9][32] : weight of (ci,t) for all co
  __shared__ float red[2][4][STEM_CO];
  const int K = Cin * 9;
  for (int i = threadIdx.x; i < K * STEM_CO; i += 256) {
    const int co = i / K, kt = i % K;  // master layout [co][ci][3][3]
    sw[kt * STEM_CO + co] = w[i];
  }
  __syncthreads();
  const long P = (long)N * Ho * Wo;
  const long pix = (long)blockIdx.x * 256 + threadIdx.x;
  const bool ok = pix < P;
  float acc[STEM_CO];
#pragma unroll
  for (int c = 0; c < STEM_CO; ++c) acc[c] = 0.f;
  if (ok) {
    const int ox = (int)(pix % Wo);
    const long r = pix / Wo;
    const int oy = (int)(r % Ho);
    const int n = (int)(r / Ho);
    for (int ci = 0; ci < Cin; ++ci) {
      const float* xp = x + ((size_t)n * Cin + ci) * H * W;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * oy - 1 + ky;
        if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int ix = 2 * ox - 1 + kx;
          if ((unsigned)ix >= (unsigned)W) continue;
          const float v = xp[(size_t)iy * W + ix];
          const float4* wp = reinterpret_cast<const float4*>(sw + (ci * 9 + ky * 3 + kx) * STEM_CO);
#pragma unroll
          for (int q = 0; q < STEM_CO / 4; ++q) {
            const float4 ww = wp[q];
            acc[4 * q] = fmaf(v, ww.x, acc[4 * q]);
            acc[4 * q + 1] = fmaf(v, ww.y, acc[4 * q + 1]);
            acc[4 * q + 2] = fmaf(v, ww.z, acc[4 * q + 2]);
            acc[4 * q + 3] = fmaf(v, ww.w, acc[4 * q + 3]);
          }
        }
      }
    }
    constexpr int KPV = Elem<T>::kPerVec;
    T* dst = y + (size_t)pix * ldy;
#pragma unroll
    for (int q = 0; q < STEM_CO / KPV; ++q) {
      float f[KPV];
#pragma unroll
      for (int e = 0; e < KPV; ++e) f[e] = acc[q * KPV + e];
      vec16 v;
      pack(v, f, T());
      unpack(v, f, T());  // statistics of the stored (rounded) values
#pragma unroll
      for (int e = 0; e < KPV; ++e) acc[q * KPV + e] = f[e];
      stg16(dst + q * KPV, v);
    }
  }
  // per-channel partial statistics of this block's 256 pixels
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < STEM_CO; ++c) {
    const float s = wave_sum(ok ? acc[c] : 0.f);
    const float q = wave_sum(ok ? acc[c] * acc[c] : 0.f);
    if (lane == 0) {
      red[0][wave][c] = s;
      red[1][wave][c] = q;
    }
  }
  __syncthreads();
  if (threadIdx.x < 2 * STEM_CO) {
    const int which = threadIdx.x / STEM_CO, c = threadIdx.x % STEM_CO;
    slab[((size_t)which * gridDim.x + blockIdx.x) * STEM_CO + c] = red[which][0][c] + red[which][1][c] + red[which][2][c] + red[which][3][c];
  }
}

// ------------------------------------------------------------------------------------------------- stem wgrad
// dW[co][ci][t] = sum_pix dy[pix][co] * x[n][ci][2oy-1+ky][2ox-1+kx].  Per block: 64-pixel batches staged in LDS
// (dy tile 64x32, patch tile 64x(Cin*9)); thread (co = tid&31, j0 = tid>>5) owns outputs (co, j0 + 8*i).
template <typename T>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ x, const T* __restrict__ dy, int lddy,
                                                         float* __restrict__ slab, int N, int Cin, int H, int W, int Ho,
                                                         int Wo, int pix_per_block) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int K = Cin * 9;
  float* sdy = sm;              // [64][32]
  const int KP = K + 1;         // odd row stride: conflict-free column writes
  float* spt = sm + 64 * 32;    // [64][KP]
  constexpr int MAXI = 18;      // K <= 144 -> at most 18 outputs per thread
  float acc[MAXI];
#pragma unroll
  for (int i = 0; i < MAXI; ++i) acc[i] = 0.f;
  const int co = threadIdx.x & 31, j0 = threadIdx.x >> 5;
  const long P = (long)N * Ho * Wo;
  const long pbeg = (long)blockIdx.x * pix_per_block;
  const long pend = pbeg + pix_per_block < P ? pbeg + pix_per_block : P;
  for (long base = pbeg; base < pend; base += 64) {
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 32; i += 256) {
      const int pl = i >> 5, c = i & 31;
      const long pix = base + pl;
      sdy[i] = pix < pend ? Elem<T>::load(dy + (size_t)pix * lddy + c) : 0.f;
    }
    for (int i = threadIdx.x; i < 64 * K; i += 256) {
      const int kt = i / 64, pl = i % 64;  // consecutive threads -> consecutive pixels (coalesced along W)
      const long pix = base + pl;
      float v = 0.f;
      if (pix < pend) {
        const int ox = (int)(pix % Wo);
        const long r = pix / Wo;
        const int oy = (int)(r % Ho);
        const int n = (int)(r / Ho);
        const int ci = kt / 9, t = kt % 9;
        const int iy = 2 * oy - 1 + t / 3, ix = 2 * ox - 1 + t % 3;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = x[(((size_t)n * Cin + ci) * H + iy) * W + ix];
      }
      spt[pl * KP + kt] = v;
    }
    __syncthreads();
    for (int pl = 0; pl < 64; ++pl) {
      const float d = sdy[pl * 32 + co];
#pragma unroll
      for (int i = 0; i < MAXI; ++i) {
        const int j = j0 + 8 * i;
        if (j < K) acc[i] = fmaf(d, spt[pl * KP + j], acc[i]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    const int j = j0 + 8 * i;
    if (j < K) slab[(size_t)blockIdx.x * (32 * K) + co * K + j] = acc[i];
  }
}

// out[i] = sum_r slab[r][i]   (fp64 accumulate, fixed order): 8 columns x 32 row-lanes per block
__global__ __launch_bounds__(256) void slab_rows_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out, int rows, long n) {
  __shared__ double red[32][8];
  const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const long i = (long)blockIdx.x * 8 + cl;
  double a = 0.0, b = 0.0;
  if (i < n) {
    int r = rl;
    for (; r + 32 < rows; r += 64) {
      a += (double)slab[(size_t)r * n + i];
      b += (double)slab[(size_t)(r + 32) * n + i];
    }
    if (r < rows) a += (double)slab[(size_t)r * n + i];
  }
  red[rl][cl] = a + b;
  __syncthreads();
  if (threadIdx.x < 8 && (long)blockIdx.x * 8 + threadIdx.x < n) {
    double s = 0.0;
#pragma unroll 8
    for (int k = 0; k < 32; ++k) s += red[k][threadIdx.x];
    out[(long)blockIdx.x * 8 + threadIdx.x] = (float)s;
  }
}

// ------------------------------------------------------------------------------------------------- head forward
// Thread <-> input pixel (qy,qx): produces the 2x2 output quad of all 3 classes from x00,x01,x10,x11.
//   out(2qy  ,2qx  ) = x00.W[1][1]
//   out(2qy  ,2qx+1) = x00.W[1][2] + x01.W[1][0]
//   out(2qy+1,2qx  ) = x00.W[2][1] + x10.W[0][1]
//   out(2qy+1,2qx+1) = x00.W[2][2] + x01.W[2][0] + x10.W[0][2] + x11.W[0][0]
// Weights are wave-uniform -> scalar loads.
constexpr int HEAD_NC = 3;

template <typename T>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ w,
                                                       float* __restrict__ out, int N, int Cin, int Hi, int Wi) {
  constexpr int KPV = Elem<T>::kPerVec;
  const long P = (long)N * Hi * Wi;
  const long pix = (long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= P) return;
  const int qx = (int)(pix % Wi);
  const long r = pix / Wi;
  const int qy = (int)(r % Hi);
  const int n = (int)(r / Hi);
  const bool hx = qx + 1 < Wi, hy = qy + 1 < Hi;
  const T* p00 = x + (size_t)pix * ldx;
  const T* p01 = p00 + ldx;
  const T* p10 = p00 + (size_t)Wi * ldx;
  const T* p11 = p10 + ldx;
  float o[4][HEAD_NC];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int c = 0; c < HEAD_NC; ++c) o[q][c] = 0.f;
  for (int c0 = 0; c0 < Cin; c0 += KPV) {
    float a[KPV], b[KPV], c_[KPV], d[KPV];
    unpack(ldg16(p00 + c0), a, T());
    if (hx) unpack(ldg16(p01 + c0), b, T()); else { _Pragma("unroll") for (int e = 0; e < KPV; ++e) b[e] = 0.f; }
    if (hy) unpack(ldg16(p10 + c0), c_, T()); else { _Pragma("unroll") for (int e = 0; e < KPV; ++e) c_[e] = 0.f; }
    if (hx && hy) unpack(ldg16(p11 + c0), d, T()); else { _Pragma("unroll") for (int e = 0; e < KPV; ++e) d[e] = 0.f; }
#pragma unroll
    for (int e = 0; e < KPV; ++e) {
      const float* we = w + (size_t)(c0 + e) * (HEAD_NC * 9);  // [ci][co][ky][kx]
#pragma unroll
      for (int co = 0; co < HEAD_NC; ++co) {
        const float* k = we + co * 9;
        o[0][co] = fmaf(a[e], k[4], o[0][co]);
        o[1][co] = fmaf(a[e], k[5], fmaf(b[e], k[3], o[1][co]));
        o[2][co] = fmaf(a[e], k[7], fmaf(c_[e], k[1], o[2][co]));
        o[3][co] = fmaf(a[e], k[8], fmaf(b[e], k[6], fmaf(c_[e], k[2], fmaf(d[e], k[0], o[3][co]))));
      }
    }
  }
  const int Ho = 2 * Hi, Wo = 2 * Wi;
#pragma unroll
  for (int co = 0; co < HEAD_NC; ++co) {
    float* base = out + (((size_t)n * HEAD_NC + co) * Ho + 2 * qy) * Wo + 2 * qx;
    *reinterpret_cast<float2*>(base) = make_float2(o[0][co], o[1][co]);
    *reinterpret_cast<float2*>(base + Wo) = make_float2(o[2][co], o[3][co]);
  }
}

// ------------------------------------------------------------------------------------------------- head dgrad
// dx[n,qy,qx,ci] = sum_{co,ky,kx} dl[n,co,2qy-1+ky,2qx-1+kx] * W[ci][co][ky][kx];  thread <-> (pixel, channel group)
template <typename T>
__global__ __launch_bounds__(256) void head_dgrad_kernel(const float* __restrict__ dl, const float* __restrict__ w,
                                                         T* __restrict__ dx, int lddx, int N, int Cin, int Hi, int Wi) {
  constexpr int KPV = Elem<T>::kPerVec;
  const int ngroups = Cin / KPV;
  const long total = (long)N * Hi * Wi * ngroups;
  const int Ho = 2 * Hi, Wo = 2 * Wi;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % ngroups);
    const long pix = idx / ngroups;
    const int qx = (int)(pix % Wi);
    const long r = pix / Wi;
    const int qy = (int)(r % Hi);
    const int n = (int)(r / Hi);
    float g[HEAD_NC][9];
#pragma unroll
    for (int co = 0; co < HEAD_NC; ++co)
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int oy = 2 * qy - 1 + t / 3, ox = 2 * qx - 1 + t % 3;
        g[co][t] = ((unsigned)oy < (unsigned)Ho && (unsigned)ox < (unsigned)Wo)
                       ? dl[(((size_t)n * HEAD_NC + co) * Ho + oy) * Wo + ox] : 0.f;
      }
    float acc[KPV];
#pragma unroll
    for (int e = 0; e < KPV; ++e) {
      const float* we = w + (size_t)(cg * KPV + e) * (HEAD_NC * 9);
      float a = 0.f;
#pragma unroll
      for (int co = 0; co < HEAD_NC; ++co)
#pragma unroll
        for (int t = 0; t < 9; ++t) a = fmaf(g[co][t], we[co * 9 + t], a);
      acc[e] = a;
    }
    vec16 v;
    pack(v, acc, T());
    stg16(dx + (size_t)pix * lddx + cg * KPV, v);
  }
}

// ------------------------------------------------------------------------------------------------- head wgrad
// dW[ci][co][t] = sum_pix x[pix][ci] * dl[n,co,2qy-1+ky,2qx-1+kx].  grid (pixel blocks); block = 64 lanes x 3 classes;
// lane <-> Cin/64 consecutive channels; dl values are wave-uniform.
template <typename T, int CPL>
__global__ __launch_bounds__(192) void head_wgrad_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ dl,
                                                         float* __restrict__ slab, int N, int Cin, int Hi, int Wi,
                                                         int pix_per_block) {
  const int lane = threadIdx.x, co = threadIdx.y;
  const int c0 = lane * CPL;
  const long P = (long)N * Hi * Wi;
  const long pbeg = (long)blockIdx.x * pix_per_block;
  const long pend = pbeg + pix_per_block < P ? pbeg + pix_per_block : P;
  const int Ho = 2 * Hi, Wo = 2 * Wi;
  float acc[9][CPL];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < CPL; ++e) acc[t][e] = 0.f;
  for (long pix = pbeg; pix < pend; ++pix) {
    const int qx = (int)(pix % Wi);
    const long r = pix / Wi;
    const int qy = (int)(r % Hi);
    const int n = (int)(r / Hi);
    float xv[CPL];
#pragma unroll
    for (int e = 0; e < CPL; ++e) xv[e] = Elem<T>::load(x + (size_t)pix * ldx + c0 + e);
    const float* dlp = dl + ((size_t)n * HEAD_NC + co) * Ho * Wo;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int oy = 2 * qy - 1 + t / 3, ox = 2 * qx - 1 + t % 3;
      if ((unsigned)oy < (unsigned)Ho && (unsigned)ox < (unsigned)Wo) {
        const float g = dlp[(size_t)oy * Wo + ox];
#pragma unroll
        for (int e = 0; e < CPL; ++e) acc[t][e] = fmaf(xv[e], g, acc[t][e]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < CPL; ++e)
#pragma unroll
    for (int t = 0; t < 9; ++t)
      slab[(size_t)blockIdx.x * (Cin * HEAD_NC * 9) + (size_t)(c0 + e) * (HEAD_NC * 9) + co * 9 + t] = acc[t][e];
}

static int head_ppb(long P) {
  long ppb = (P + 2047) / 2048;
  if (ppb < 16) ppb = 16;
  return (int)ppb;
}
static int stem_ppb(long P) {
  long ppb = (P + 1023) / 1024;
  ppb = (ppb + 63) / 64 * 64;
  return (int)ppb;
}

}  // namespace dc

using namespace dc;

extern "C" int dc_stem_stat_rows(int N, int H, int W) { return cdiv((long)N * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1), 256); }

extern "C" int dc_stem_fwd(int dtype, int N, int Cin, int H, int W, const float* x_nchw, const float* w, void* y, int ldy,
                           float* stat_slab, void* stream) {
  DC_REQUIRE(x_nchw && w && stat_slab && N > 0 && Cin > 0 && Cin <= 16, "dc_stem_fwd: bad argument (Cin must be <= 16)");
  if (int e = dc_check_view(y, ldy, STEM_CO, dtype, "dc_stem_fwd y")) return e;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long P = (long)N * Ho * Wo;
  const size_t lds = (size_t)Cin * 9 * STEM_CO * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DC_BF16)
    hipLaunchKernelGGL(stem_fwd_kernel<bf16>, dim3(cdiv(P, 256)), dim3(256), lds, st, x_nchw, w, (bf16*)y, ldy, stat_slab, N, Cin, H, W, Ho, Wo);
  else
    hipLaunchKernelGGL(stem_fwd_kernel<float>, dim3(cdiv(P, 256)), dim3(256), lds, st, x_nchw, w, (float*)y, ldy, stat_slab, N, Cin, H, W, Ho, Wo);
  DC_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t dc_stem_wgrad_workspace(int N, int Cin, int H, int W) {
  const long P = (long)N * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1);
  const int ppb = stem_ppb(P);
  return (size_t)cdiv(P, ppb) * 32 * Cin * 9 * sizeof(float);
}

extern "C" int dc_stem_wgrad(int dtype, int N, int Cin, int H, int W, const float* x_nchw, const void* dy, int lddy,
                             void* workspace, float* grad_w, void* stream) {
  DC_REQUIRE(x_nchw && workspace && grad_w && N > 0 && Cin > 0 && Cin <= 16, "dc_stem_wgrad: bad argument");
  if (int e = dc_check_view(dy, lddy, STEM_CO, dtype, "dc_stem_wgrad dy")) return e;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long P = (long)N * Ho * Wo;
  const int ppb = stem_ppb(P);
  const int rows = cdiv(P, ppb);
  const int K = Cin * 9;
  const size_t lds = (size_t)(64 * 32 + 64 * (K + 1)) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DC_BF16)
    hipLaunchKernelGGL(stem_wgrad_kernel<bf16>, dim3(rows), dim3(256), lds, st, x_nchw, (const bf16*)dy, lddy, (float*)workspace, N, Cin, H, W, Ho, Wo, ppb);
  else
    hipLaunchKernelGGL(stem_wgrad_kernel<float>, dim3(rows), dim3(256), lds, st, x_nchw, (const float*)dy, lddy, (float*)workspace, N, Cin, H, W, Ho, Wo, ppb);
  DC_CHECK_LAUNCH();
  const long n = 32L * K;
  hipLaunchKernelGGL(slab_rows_reduce_kernel, dim3(cdiv(n, 8)), dim3(256), 0, st, (const float*)workspace, grad_w, rows, n);
  DC_CHECK_LAUNCH();
  return 0;
}

extern "C" int dc_head_fwd(int dtype, int N, int Cin, int Hi, int Wi, const void* x, int ldx, const float* w,
                           float* logits_nchw, void* stream) {
  if (int e = dc_check_view(x, ldx, Cin, dtype, "dc_head_fwd x")) return e;
  DC_REQUIRE(w && logits_nchw && N > 0, "dc_head_fwd: bad argument");
  DC_REQUIRE(((uintptr_t)logits_nchw & 7) == 0, "dc_head_fwd: logits not 8-byte aligned");
  const long P = (long)N * Hi * Wi;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DC_BF16)
    hipLaunchKernelGGL(head_fwd_kernel<bf16>, dim3(cdiv(P, 256)), dim3(256), 0, st, (const bf16*)x, ldx, w, logits_nchw, N, Cin, Hi, Wi);
  else
    hipLaunchKernelGGL(head_fwd_kernel<float>, dim3(cdiv(P, 256)), dim3(256), 0, st, (const float*)x, ldx, w, logits_nchw, N, Cin, Hi, Wi);
  DC_CHECK_LAUNCH();
  return 0;
}

extern "C" int dc_head_dgrad(int dtype, int N, int Cin, int Hi, int Wi, const float* dlogits_nchw, const float* w,
                             void* dx, int lddx, void* stream) {
  if (int e = dc_check_view(dx, lddx, Cin, dtype, "dc_head_dgrad dx")) return e;
  DC_REQUIRE(w && dlogits_nchw && N > 0, "dc_head_dgrad: bad argument");
  const int kpv = dtype == DC_BF16 ? 8 : 4;
  const long total = (long)N * Hi * Wi * (Cin / kpv);
  long blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DC_BF16)
    hipLaunchKernelGGL(head_dgrad_kernel<bf16>, dim3((int)blocks), dim3(256), 0, st, dlogits_nchw, w, (bf16*)dx, lddx, N, Cin, Hi, Wi);
  else
    hipLaunchKernelGGL(head_dgrad_kernel<float>, dim3((int)blocks), dim3(256), 0, st, dlogits_nchw, w, (float*)dx, lddx, N, Cin, Hi, Wi);
  DC_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t dc_head_wgrad_workspace(int N, int Cin, int Hi, int Wi) {
  const long P = (long)N * Hi * Wi;
  return (size_t)cdiv(P, head_ppb(P)) * Cin * HEAD_NC * 9 * sizeof(float);
}

extern "C" int dc_head_wgrad(int dtype, int N, int Cin, int Hi, int Wi, const void* x, int ldx,
                             const float* dlogits_nchw, void* workspace, float* grad_w, void* stream) {
  if (int e = dc_check_view(x, ldx, Cin, dtype, "dc_head_wgrad x")) return e;
  DC_REQUIRE(dlogits_nchw && workspace && grad_w && N > 0, "dc_head_wgrad: bad argument");
  DC_REQUIRE(Cin == 256, "dc_head_wgrad: Cin must be 256 (64 lanes x 4 channels)");
  const long P = (long)N * Hi * Wi;
  const int ppb = head_ppb(P);
  const int rows = cdiv(P, ppb);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DC_BF16)
    hipLaunchKernelGGL((head_wgrad_kernel<bf16, 4>), dim3(rows), dim3(64, 3), 0, st, (const bf16*)x, ldx, dlogits_nchw, (float*)workspace, N, Cin, Hi, Wi, ppb);
  else
    hipLaunchKernelGGL((head_wgrad_kernel<float, 4>), dim3(rows), dim3(64, 3), 0, st, (const float*)x, ldx, dlogits_nchw, (float*)workspace, N, Cin, Hi, Wi, ppb);
  DC_CHECK_LAUNCH();
  const long n = (long)Cin * HEAD_NC * 9;
  hipLaunchKernelGGL(slab_rows_reduce_kernel, dim3(cdiv(n, 8)), dim3(256), 0, st, (const float*)workspace, grad_w, rows, n);
  DC_CHECK_LAUNCH();
  return 0;
}
