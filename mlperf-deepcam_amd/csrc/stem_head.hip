// The two ends of the network, both HBM/VALU-bound and too thin for MFMA tiles:
//   stem  Conv2d(16->32, k3, s2, p1) that reads the caller's NCHW fp32 batch in place (no layout pass) and writes NHWC
//   head  ConvTranspose2d(256->3, k3, s2, p1, op1) that reads NHWC and writes the NCHW fp32 logits of the reference API
#include "common.h"
#include "lds_tr_image.h"
#include "wgrad.h"

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

// ------------------------------------------------------------------------------------------------- classifier head
// ConvTranspose2d(Cin -> 3, k3, s2, p1, op1) as GEMMs on the MFMA kernels instead of a VALU stencil:
//   forward   P[pixel][co*9+t] = x[pixel][:] . W[:, co, t]                (1x1 GEMM, N padded 27 -> 32, fp32 result)
//             logits(2qy+py, 2qx+px) = sum of the 1/2/2/4 taps of P at (qy,qx), (qy,qx+1), (qy+1,qx), (qy+1,qx+1)
//   backward  dP[pixel][co*9+t] = dlogits[n, co, 2qy-1+ky, 2qx-1+kx]      (gather, zero outside)
//             dx = dP . W^T (1x1 data-gradient GEMM),   dW = x^T . dP (pixel-reduction GEMM)
constexpr int HEAD_NC = 3;
constexpr int HEAD_NP = 32;   // 27 products padded to an MFMA-friendly width

template <typename T>
__global__ void head_pack_kernel(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wb, int Cin) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= HEAD_NP * Cin) return;
  const int j = i / Cin, ci = i % Cin;
  const float v = j < HEAD_NC * 9 ? w[(size_t)ci * (HEAD_NC * 9) + j] : 0.f;   // master [ci][co][ky][kx]
  Elem<T>::store(wf + (size_t)j * Cin + ci, v);
  Elem<T>::store(wb + (size_t)ci * HEAD_NP + j, v);
}

// out(2qy  ,2qx  ) = P00[4];                out(2qy  ,2qx+1) = P00[5] + P01[3]
// out(2qy+1,2qx  ) = P00[7] + P10[1];       out(2qy+1,2qx+1) = P00[8] + P01[6] + P10[2] + P11[0]      (per class)
__global__ __launch_bounds__(256) void head_combine_kernel(const float* __restrict__ P, float* __restrict__ out, int N, int Hi, int Wi) {
  const long total = (long)N * Hi * Wi;
  const long pix = (long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= total) return;
  const int qx = (int)(pix % Wi);
  const long r = pix / Wi;
  const int qy = (int)(r % Hi);
  const int n = (int)(r / Hi);
  const bool hx = qx + 1 < Wi, hy = qy + 1 < Hi;
  const float* p00 = P + (size_t)pix * HEAD_NP;
  const float* p01 = p00 + HEAD_NP;
  const float* p10 = p00 + (size_t)Wi * HEAD_NP;
  const float* p11 = p10 + HEAD_NP;
  const int Ho = 2 * Hi, Wo = 2 * Wi;
#pragma unroll
  for (int co = 0; co < HEAD_NC; ++co) {
    const float* a = p00 + co * 9;
    const float o00 = a[4];
    float o01 = a[5], o10 = a[7], o11 = a[8];
    if (hx) { o01 += p01[co * 9 + 3]; o11 += p01[co * 9 + 6]; }
    if (hy) { o10 += p10[co * 9 + 1]; o11 += p10[co * 9 + 2]; }
    if (hx && hy) o11 += p11[co * 9 + 0];
    float* base = out + (((size_t)n * HEAD_NC + co) * Ho + 2 * qy) * Wo + 2 * qx;
    *reinterpret_cast<float2*>(base) = make_float2(o00, o01);
    *reinterpret_cast<float2*>(base + Wo) = make_float2(o10, o11);
  }
}

// Fused forward (bf16): products and sub-pixel combination in one kernel, so the 27-product image (226 MB written and read
// back at B=8, on top of a GEMM that fills a quarter of its 128-wide tile) never exists.  A workgroup walks HF_TILES tiles of
// 4 x 32 input pixels; per tile the products of the 5 x 33 pixels it touches (the +1 halo supplies the (qy+1, qx+1) neighbours)
// are computed with MFMA straight from global memory -- a lane loads the 16 bytes of ITS pixel's K group, which is exactly the
// A-fragment layout, out-of-image pixels load zeros -- against the 32 x Cin weight image held in LDS, land in an LDS tile
// P[pixel][32] (fp32) and are combined into 8 x 64 x 3 logits with the same addition order as head_combine_kernel,
// written as coalesced NCHW rows.
constexpr int HF_TY = 4, HF_TX = 32;
constexpr int HF_PX = (HF_TY + 1) * (HF_TX + 1);        // 165 product pixels per tile
constexpr int HF_MB = (HF_PX + 15) / 16;                // 11 MFMA row blocks
constexpr int HF_TILES = 6;                             // tiles per workgroup (along x), amortises the weight image

// LOSS: the weighted cross-entropy pass (loss.hip: wce_kernel, same arithmetic statement for statement, so the same bits) runs on the
// logits of a tile while they are still in registers: the fp32 NCHW logits are then written only if the caller wants them (`out`
// may be null) and never read back (reference: upsample.last_deconv, deeplab_xception.py:374,382, followed by utils/losses.py:35-50).
// BNIN: x is the raw output y of the convolution in front of a BatchNorm(+ReLU) and the head's real input act(y * scale + shift) is formed
// while the fragments are loaded (fp32 fma, ReLU, rounded to bf16: the bits dc_bn_apply would have stored), so that activation is never
// written (906 MB per local-batch-8 step and the same again read by the apply pass).  scale / shift sit in LDS behind the weight image.
struct HeadBnIn {
  const float* scale;
  const float* shift;
  int relu;
};

struct HeadLoss {
  const void* labels;
  int lbytes;
  const float* cw;
  float grad_scale;
  double* loss_sum;
  float* dlogits;
  int64_t* pred;
  unsigned long long* counts;
};

__device__ inline int head_load_label(const void* labels, int bytes, size_t i) {
  long long v;
  if (bytes == 8) v = reinterpret_cast<const int64_t*>(labels)[i];
  else if (bytes == 4) v = reinterpret_cast<const int32_t*>(labels)[i];
  else v = reinterpret_cast<const uint8_t*>(labels)[i];
  return (v >= 0 && v < HEAD_NC) ? (int)v : -1;
}

template <bool LOSS, bool BNIN = false>
__global__ __launch_bounds__(256) void head_fused_fwd_kernel(const bf16* __restrict__ x, int ldx, const bf16* __restrict__ wf,
                                                             float* __restrict__ out, int N, int Hi, int Wi, int Cin, int ntx, const HeadLoss hl,
                                                             const HeadBnIn bi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* P = reinterpret_cast<float*>(smem);                       // [HF_MB*16][HEAD_NP]
  char* wl = smem + HF_MB * 16 * HEAD_NP * 4;                      // weight image [32][Cin] bf16, 64-byte K rows swizzled per 16-B slot
  [[maybe_unused]] float* bnv = reinterpret_cast<float*>(wl + HEAD_NP * Cin * 2);   // BNIN: scale[Cin], shift[Cin]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  if constexpr (BNIN) {
    for (int i = tid; i < Cin; i += 256) {
      bnv[i] = bi.scale[i];
      bnv[Cin + i] = bi.shift[i];
    }
  }
  // stage the weights: row j (product), K step k, slot s  ->  wl[(k*32 + j)*64 + ((s ^ ((j >> 1) & 3)) << 4)]
  for (int i = tid; i < HEAD_NP * Cin / 8; i += 256) {
    const int j = i / (Cin / 8), v = i % (Cin / 8);
    const int k = v >> 2, sl = v & 3;
    *reinterpret_cast<vec16*>(wl + (k * 32 + j) * 64 + ((sl ^ ((j >> 1) & 3)) << 4)) = ldg16(wf + (size_t)j * Cin + v * 8);
  }
  const int strips = (ntx + HF_TILES - 1) / HF_TILES;
  int b = blockIdx.x;
  const int strip = b % strips;
  b /= strips;
  const int nty = (Hi + HF_TY - 1) / HF_TY;
  const int ty = b % nty, n = b / nty;
  const int y0 = ty * HF_TY;
  const int Ho = 2 * Hi, Wo = 2 * Wi;
  [[maybe_unused]] double lsum = 0.0;
  [[maybe_unused]] unsigned long long cnt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int tt = 0; tt < HF_TILES; ++tt) {
    const int tx = strip * HF_TILES + tt;
    if (tx >= ntx) break;
    const int x0 = tx * HF_TX;
    __syncthreads();   // weights staged (first pass) / the previous tile's combine is done with P
    // A fragments of row block mb: this lane's pixel, its 16-byte K group of each of the 8 K steps (Cin = 256); the next block's
    // loads are in flight while the current one is multiplied
    // (BNIN: the validity of a block's rows travels with it: a padding row must stay zero, not act(shift))
    auto load_block = [&](int mb, vec16 (&fa)[8]) {
      const int idx = mb * 16 + fr;
      const int ry = idx / (HF_TX + 1), rx = idx - ry * (HF_TX + 1);
      const int qy = y0 + ry, qx = x0 + rx;
      const bool ok = idx < HF_PX && qy < Hi && qx < Wi;
      const bf16* src = x + (((size_t)n * Hi + (ok ? qy : 0)) * Wi + (ok ? qx : 0)) * ldx + fg * 8;
#pragma unroll
      for (int u = 0; u < 8; ++u) fa[u] = ok ? ldg16(src + u * 32) : zero16();
      return ok;
    };
    vec16 cur[8], nxt[8];
    [[maybe_unused]] bool cur_ok = false, nxt_ok = false;
    if (wave < HF_MB) cur_ok = load_block(wave, cur);
    for (int mb = wave; mb < HF_MB; mb += 4) {
      if (mb + 4 < HF_MB) nxt_ok = load_block(mb + 4, nxt);
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const char* wk = wl + u * 32 * 64;
        const vec16 b0 = *reinterpret_cast<const vec16*>(wk + fr * 64 + ((fg ^ ((fr >> 1) & 3)) << 4));
        const vec16 b1 = *reinterpret_cast<const vec16*>(wk + (16 + fr) * 64 + ((fg ^ (((16 + fr) >> 1) & 3)) << 4));
        vec16 av = cur[u];
        if constexpr (BNIN) {
          // this lane's channels of K step u: u * 32 + fg * 8 .. + 7
          const float* sc = bnv + u * 32 + fg * 8;
          const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc), s1 = *reinterpret_cast<const f32x4*>(sc + 4);
          const f32x4 h0 = *reinterpret_cast<const f32x4*>(sc + Cin), h1 = *reinterpret_cast<const f32x4*>(sc + Cin + 4);
          float f[8];
          unpack(av, f, bf16());
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            f[e] = fmaf(f[e], s0[e], h0[e]);
            f[4 + e] = fmaf(f[4 + e], s1[e], h1[e]);
          }
          if (bi.relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], 0.f);
          }
          pack(av, f, bf16());
          if (!cur_ok) av = zero16();
        }
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, b0), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, b1), acc1, 0, 0, 0);
      }
      // D[row = pixel fg*4 + r][col = product fr]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        P[(mb * 16 + fg * 4 + r) * HEAD_NP + fr] = acc0[r];
        P[(mb * 16 + fg * 4 + r) * HEAD_NP + 16 + fr] = acc1[r];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) cur[u] = nxt[u];
      cur_ok = nxt_ok;
    }
    __syncthreads();
    if constexpr (!LOSS) {
      // combine: 2*HF_TY x 2*HF_TX outputs per class, consecutive lanes along x
      for (int i = tid; i < HEAD_NC * 2 * HF_TY * 2 * HF_TX; i += 256) {
        const int ox_l = i % (2 * HF_TX);
        const int t2 = i / (2 * HF_TX);
        const int oy_l = t2 % (2 * HF_TY), co = t2 / (2 * HF_TY);
        const int qy_l = oy_l >> 1, py = oy_l & 1, qx_l = ox_l >> 1, px = ox_l & 1;
        const int qy = y0 + qy_l, qx = x0 + qx_l;
        if (qy < Hi && qx < Wi) {
          const float* p00 = P + (qy_l * (HF_TX + 1) + qx_l) * HEAD_NP + co * 9;
          const float* p01 = p00 + HEAD_NP;
          const float* p10 = p00 + (HF_TX + 1) * HEAD_NP;
          const float* p11 = p10 + HEAD_NP;
          float o;
          if (py == 0) o = px == 0 ? p00[4] : p00[5] + p01[3];
          else o = px == 0 ? p00[7] + p10[1] : ((p00[8] + p01[6]) + p10[2]) + p11[0];
          out[(((size_t)n * HEAD_NC + co) * Ho + 2 * qy + py) * Wo + 2 * qx + px] = o;
        }
      }
    } else {
      // combine one output PIXEL per thread and pass (its three logits with the same addition order as above), then the loss on them
      constexpr int NPIX = 2 * HF_TY * 2 * HF_TX;   // 512: two wave-uniform passes
#pragma unroll
      for (int it = 0; it < NPIX / 256; ++it) {
        const int i = tid + it * 256;
        const int ox_l = i % (2 * HF_TX), oy_l = i / (2 * HF_TX);
        const int qy_l = oy_l >> 1, py = oy_l & 1, qx_l = ox_l >> 1, px = ox_l & 1;
        const int qy = y0 + qy_l, qx = x0 + qx_l;
        const bool ok = qy < Hi && qx < Wi;
        float l[HEAD_NC] = {0.f, 0.f, 0.f};
        int y = 0, am = 0;
        if (ok) {
#pragma unroll
          for (int co = 0; co < HEAD_NC; ++co) {
            const float* p00 = P + (qy_l * (HF_TX + 1) + qx_l) * HEAD_NP + co * 9;
            const float* p01 = p00 + HEAD_NP;
            const float* p10 = p00 + (HF_TX + 1) * HEAD_NP;
            const float* p11 = p10 + HEAD_NP;
            if (py == 0) l[co] = px == 0 ? p00[4] : p00[5] + p01[3];
            else l[co] = px == 0 ? p00[7] + p10[1] : ((p00[8] + p01[6]) + p10[2]) + p11[0];
          }
          const size_t HWo = (size_t)Ho * Wo;
          const size_t pix = (size_t)(2 * qy + py) * Wo + 2 * qx + px;
          if (out != nullptr) {
#pragma unroll
            for (int co = 0; co < HEAD_NC; ++co) out[((size_t)n * HEAD_NC + co) * HWo + pix] = l[co];
          }
          const size_t gi = (size_t)n * HWo + pix;
          y = head_load_label(hl.labels, hl.lbytes, gi);
          float best = l[0];
          if (l[1] > best) { best = l[1]; am = 1; }
          if (l[2] > best) { best = l[2]; am = 2; }
          const float e0 = expf(l[0] - best), e1 = expf(l[1] - best), e2 = expf(l[2] - best);
          const float se = e0 + e1 + e2;
          const float lse = best + logf(se);
          float* gp = hl.dlogits != nullptr ? hl.dlogits + (size_t)n * HEAD_NC * HWo + pix : nullptr;
          if ((unsigned)y < (unsigned)HEAD_NC) {
            const float wy = hl.cw[y];
            const float ly = y == 0 ? l[0] : (y == 1 ? l[1] : l[2]);
            lsum += (double)(wy * (lse - ly));
            if (gp != nullptr) {
              const float inv = 1.0f / se;
              const float sc = wy * hl.grad_scale;
              gp[0] = sc * (e0 * inv - (y == 0 ? 1.f : 0.f));
              gp[HWo] = sc * (e1 * inv - (y == 1 ? 1.f : 0.f));
              gp[2 * HWo] = sc * (e2 * inv - (y == 2 ? 1.f : 0.f));
            }
          } else {   // a label outside [0, 3) poisons the loss and this pixel's gradient, as dc_wce_fused does
            lsum += (double)__builtin_nanf("");
            if (gp != nullptr) gp[0] = gp[HWo] = gp[2 * HWo] = __builtin_nanf("");
          }
          if (hl.pred != nullptr) hl.pred[gi] = am;
        }
        if (hl.counts != nullptr) {
          const bool eq = ok && (am == y), ne = ok && (am != y);
#pragma unroll
          for (int j = 0; j < HEAD_NC; ++j) {
            cnt[j] += __popcll(__ballot(eq && y == j));
            cnt[3 + j] += __popcll(__ballot(ne && am == j));
            cnt[6 + j] += __popcll(__ballot(ne && y == j));
          }
        }
      }
    }
  }
  if constexpr (LOSS) {
    __syncthreads();                                  // P is dead: reuse its first bytes for the block's partial sums
    double* s_loss = reinterpret_cast<double*>(smem);
    unsigned long long* s_cnt = reinterpret_cast<unsigned long long*>(smem + 64);
    lsum = wave_sum(lsum);
    if (lane == 0) {
      s_loss[wave] = lsum;
#pragma unroll
      for (int k = 0; k < 9; ++k) s_cnt[wave * 9 + k] = cnt[k];
    }
    __syncthreads();
    if (tid == 0 && hl.loss_sum != nullptr) atomicAdd(hl.loss_sum, s_loss[0] + s_loss[1] + s_loss[2] + s_loss[3]);
    if (tid < 9 && hl.counts != nullptr) {
      const unsigned long long c = s_cnt[tid] + s_cnt[9 + tid] + s_cnt[18 + tid] + s_cnt[27 + tid];
      if (c) atomicAdd(&hl.counts[tid], c);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void head_gather_kernel(const float* __restrict__ dl, T* __restrict__ dP, int N, int Hi, int Wi) {
  const long total = (long)N * Hi * Wi;
  const long pix = (long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= total) return;
  const int qx = (int)(pix % Wi);
  const long r = pix / Wi;
  const int qy = (int)(r % Hi);
  const int n = (int)(r / Hi);
  const int Ho = 2 * Hi, Wo = 2 * Wi;
  float v[HEAD_NP];
#pragma unroll
  for (int j = 0; j < HEAD_NP; ++j) v[j] = 0.f;
#pragma unroll
  for (int co = 0; co < HEAD_NC; ++co)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int oy = 2 * qy - 1 + t / 3, ox = 2 * qx - 1 + t % 3;
      if ((unsigned)oy < (unsigned)Ho && (unsigned)ox < (unsigned)Wo) v[co * 9 + t] = dl[(((size_t)n * HEAD_NC + co) * Ho + oy) * Wo + ox];
    }
  constexpr int KPV = Elem<T>::kPerVec;
  T* dst = dP + (size_t)pix * HEAD_NP;
#pragma unroll
  for (int q = 0; q < HEAD_NP / KPV; ++q) {
    float f[KPV];
#pragma unroll
    for (int e = 0; e < KPV; ++e) f[e] = v[q * KPV + e];
    vec16 o;
    pack(o, f, T());
    stg16(dst + q * KPV, o);
  }
}

// grad[ci][co][ky][kx] = tmp[(co*9+t)][ci]
__global__ void head_wfinish_kernel(const float* __restrict__ tmp, float* __restrict__ grad, int Cin) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Cin * HEAD_NC * 9) return;
  const int ci = i / (HEAD_NC * 9), j = i % (HEAD_NC * 9);
  grad[i] = tmp[(size_t)j * Cin + ci];
}

// ------------------------------------------------------------------------------------------------- input layout
// NCHW fp32 (the reference's batch layout) -> NHWC T.  One thread per PIXEL: its Cc per-channel reads are coalesced along W
// (and all in flight together), its 16-byte vectors are adjacent, so a wave writes one contiguous run of 64 pixels.  (One
// thread per (pixel, channel group) left every other 16 bytes of a 32-byte pixel to another pass: 2.65 TB/s on the 16-channel
// input.)
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ out, int ldo, int N, int Cc, long HW) {
  constexpr int KPV = Elem<T>::kPerVec;
  const int ngroups = Cc / KPV;
  const long total = (long)N * HW;
  for (long pix = blockIdx.x * (long)blockDim.x + threadIdx.x; pix < total; pix += (long)gridDim.x * blockDim.x) {
    const long n = pix / HW, p = pix % HW;
    const float* src = x + (size_t)n * Cc * HW + p;
    T* dst = out + (size_t)pix * ldo;
    if (ngroups == 2) {          // the 16-channel bf16 stem input: everything unrolled, 16 loads in flight
      float f[2][KPV];
#pragma unroll
      for (int cg = 0; cg < 2; ++cg)
#pragma unroll
        for (int e = 0; e < KPV; ++e) f[cg][e] = src[(size_t)(cg * KPV + e) * HW];
#pragma unroll
      for (int cg = 0; cg < 2; ++cg) {
        vec16 v;
        pack(v, f[cg], T());
        stg16(dst + cg * KPV, v);
      }
      continue;
    }
    for (int cg = 0; cg < ngroups; ++cg) {
      float f[KPV];
#pragma unroll
      for (int e = 0; e < KPV; ++e) f[e] = src[(size_t)(cg * KPV + e) * HW];
      vec16 v;
      pack(v, f, T());
      stg16(dst + cg * KPV, v);
    }
  }
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

// Input pipeline (reference data/cam_hdf5_dataset.py:122-129): the files hold HWC fp32 fields, i.e. they are ALREADY channels-last.
// out[p][j] = scale[j] * (x[p][channels[j]] - shift[j]) selects the requested channels, applies the min/max normalisation and
// converts to the activation dtype in one pass over the freshly copied batch; no NCHW transpose exists on this path.
template <typename T>
__global__ __launch_bounds__(256) void input_normalize_kernel(long npix, int Cfile, int Cc, const int* __restrict__ channels,
                                                              const float* __restrict__ x, const float* __restrict__ shift,
                                                              const float* __restrict__ scale, T* __restrict__ out, int ldo) {
  constexpr int KPV = Elem<T>::kPerVec;
  const int ngroups = Cc / KPV;
  const long total = npix * ngroups;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(i % ngroups);
    const long p = i / ngroups;
    float f[KPV];
#pragma unroll
    for (int e = 0; e < KPV; ++e) {
      const int j = cg * KPV + e;
      const int src = channels != nullptr ? channels[j] : j;
      f[e] = scale[j] * (x[(size_t)p * Cfile + src] - shift[j]);
    }
    vec16 v;
    pack(v, f, T());
    stg16(out + (size_t)p * ldo + cg * KPV, v);
  }
}

// The same normalisation written as the reference's NCHW fp32 batch (one thread per pixel: the sample's channels are read as one
// contiguous run, each channel plane is written coalesced).  For --channels subsets whose count the MFMA stem does not take: the
// direct stem kernel reads NCHW fp32.
__global__ __launch_bounds__(256) void input_normalize_nchw_kernel(int N, long HW, int Cfile, int Cc, const int* __restrict__ channels,
                                                                   const float* __restrict__ x, const float* __restrict__ shift,
                                                                   const float* __restrict__ scale, float* __restrict__ out) {
  const long total = (long)N * HW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long n = i / HW, p = i - n * HW;
    const float* px = x + (size_t)i * Cfile;
    for (int j = 0; j < Cc; ++j) {
      const int src = channels != nullptr ? channels[j] : j;
      out[((size_t)n * Cc + j) * HW + p] = scale[j] * (px[src] - shift[j]);
    }
  }
}

static int g_head_fused = 1;   // bf16 forward: products + combination in one kernel (0: GEMM + combine kernels)
extern "C" int dc_head_set_fused(int v) { g_head_fused = v ? 1 : 0; return 0; }

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct HeadWs {
  float* P; void* dP; void* wf; void* wb; float* tmp; void* slabs; size_t slab_bytes; size_t total;
};
static HeadWs head_ws(int dtype, int N, int Cin, int Hi, int Wi, void* base) {
  const size_t es = dtype == DC_BF16 ? 2 : 4;
  const size_t M = (size_t)N * Hi * Wi;
  dc_conv_desc d{dtype, 1, 1, 0, 1, 0, Cin, HEAD_NP};
  HeadWs w;
  char* p = (char*)base;
  size_t off = 0;
  w.P = (float*)(p + off); off += align256(M * HEAD_NP * 4);
  w.dP = p + off; off += align256(M * HEAD_NP * es);
  w.wf = p + off; off += align256((size_t)HEAD_NP * Cin * es);
  w.wb = p + off; off += align256((size_t)HEAD_NP * Cin * es);
  w.tmp = (float*)(p + off); off += align256((size_t)HEAD_NP * Cin * 4);
  w.slabs = p + off;
  w.slab_bytes = dc_conv_wgrad_workspace(&d, N, Hi, Wi);
  off += align256(w.slab_bytes);
  w.total = off;
  return w;
}

extern "C" size_t dc_head_workspace(int dtype, int N, int Cin, int Hi, int Wi) { return head_ws(dtype, N, Cin, Hi, Wi, nullptr).total; }

extern "C" int dc_wce_fused(int B, int H, int W, const float* logits_nchw, const void* labels, int label_dtype_bytes,
                            const float* class_weights, float grad_scale, double* loss_sum, float* dlogits, int64_t* pred,
                            int64_t* counts, void* stream);

static int head_fwd_impl(int dtype, int N, int Cin, int Hi, int Wi, const void* x, int ldx, const float* w, float* logits_nchw,
                         void* workspace, void* stream, const HeadLoss* hl, const HeadBnIn* bi = nullptr) {
  if (int e = dc_check_view(x, ldx, Cin, dtype, "dc_head_fwd x")) return e;
  DC_REQUIRE(w && workspace && N > 0, "dc_head_fwd: bad argument");
  DC_REQUIRE(((uintptr_t)logits_nchw & 7) == 0 && ((uintptr_t)workspace & 255) == 0, "dc_head_fwd: logits / workspace alignment");
  HeadWs ws = head_ws(dtype, N, Cin, Hi, Wi, workspace);
  hipStream_t st = (hipStream_t)stream;
  const int np = HEAD_NP * Cin;
  if (dtype == DC_BF16) hipLaunchKernelGGL(head_pack_kernel<bf16>, dim3(cdiv(np, 256)), dim3(256), 0, st, w, (bf16*)ws.wf, (bf16*)ws.wb, Cin);
  else hipLaunchKernelGGL(head_pack_kernel<float>, dim3(cdiv(np, 256)), dim3(256), 0, st, w, (float*)ws.wf, (float*)ws.wb, Cin);
  DC_CHECK_LAUNCH();
  if (dtype == DC_BF16 && g_head_fused && Cin == 256) {
    const int ntx = cdiv(Wi, HF_TX), nty = cdiv(Hi, HF_TY), strips = cdiv(ntx, HF_TILES);
    const size_t lds = (size_t)HF_MB * 16 * HEAD_NP * 4 + (size_t)HEAD_NP * Cin * 2 + (size_t)2 * Cin * 4;
    auto k0 = &head_fused_fwd_kernel<false>;
    auto k1 = &head_fused_fwd_kernel<true>;
    auto k2 = &head_fused_fwd_kernel<false, true>;
    auto k3 = &head_fused_fwd_kernel<true, true>;
    DC_ONCE({
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k3), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    if (bi != nullptr) {
      if (hl != nullptr)
        hipLaunchKernelGGL(k3, dim3(strips * nty * N), dim3(256), lds, st, (const bf16*)x, ldx, (const bf16*)ws.wf, logits_nchw, N, Hi, Wi, Cin, ntx, *hl, *bi);
      else
        hipLaunchKernelGGL(k2, dim3(strips * nty * N), dim3(256), lds, st, (const bf16*)x, ldx, (const bf16*)ws.wf, logits_nchw, N, Hi, Wi, Cin, ntx, HeadLoss{}, *bi);
      DC_CHECK_LAUNCH();
      return 0;
    }
    if (hl != nullptr)
      hipLaunchKernelGGL(k1, dim3(strips * nty * N), dim3(256), lds, st, (const bf16*)x, ldx, (const bf16*)ws.wf, logits_nchw, N, Hi, Wi, Cin, ntx, *hl, HeadBnIn{});
    else
      hipLaunchKernelGGL(k0, dim3(strips * nty * N), dim3(256), lds, st, (const bf16*)x, ldx, (const bf16*)ws.wf, logits_nchw, N, Hi, Wi, Cin, ntx, HeadLoss{}, HeadBnIn{});
    DC_CHECK_LAUNCH();
    return 0;
  }
  DC_REQUIRE(bi == nullptr, "dc_head_fwd_bnin: only the fused bf16 head kernel (256 input channels) applies the BatchNorm while loading");
  DC_REQUIRE(logits_nchw != nullptr, "dc_head_fwd: this configuration (fp32, or the unfused path) needs a logits buffer");
  dc_conv_desc d{dtype, 1, 1, 0, 1, 0, Cin, HEAD_NP};
  if (int e = dc_conv_fwd_f32out(&d, N, Hi, Wi, x, ldx, ws.wf, ws.P, HEAD_NP, stream)) return e;
  const long P = (long)N * Hi * Wi;
  hipLaunchKernelGGL(head_combine_kernel, dim3(cdiv(P, 256)), dim3(256), 0, st, (const float*)ws.P, logits_nchw, N, Hi, Wi);
  DC_CHECK_LAUNCH();
  if (hl != nullptr)     // not the fused kernel: the separate pass over the logits
    return dc_wce_fused(N, 2 * Hi, 2 * Wi, logits_nchw, hl->labels, hl->lbytes, hl->cw, hl->grad_scale, hl->loss_sum, hl->dlogits, hl->pred,
                        (int64_t*)hl->counts, stream);
  return 0;
}

extern "C" int dc_head_fwd(int dtype, int N, int Cin, int Hi, int Wi, const void* x, int ldx, const float* w,
                           float* logits_nchw, void* workspace, void* stream) {
  DC_REQUIRE(logits_nchw != nullptr, "dc_head_fwd: null logits");
  return head_fwd_impl(dtype, N, Cin, Hi, Wi, x, ldx, w, logits_nchw, workspace, stream, nullptr);
}

extern "C" int dc_head_fwd_loss(int dtype, int N, int Cin, int Hi, int Wi, const void* x, int ldx, const float* w, float* logits_nchw,
                                void* workspace, const void* labels, int label_dtype_bytes, const float* class_weights, float grad_scale,
                                double* loss_sum, float* dlogits, int64_t* pred, int64_t* counts, void* stream) {
  DC_REQUIRE(labels && class_weights, "dc_head_fwd_loss: bad argument");
  DC_REQUIRE(label_dtype_bytes == 1 || label_dtype_bytes == 4 || label_dtype_bytes == 8, "dc_head_fwd_loss: labels must be uint8, int32 or int64");
  HeadLoss hl{labels, label_dtype_bytes, class_weights, grad_scale, loss_sum, dlogits, pred, (unsigned long long*)counts};
  return head_fwd_impl(dtype, N, Cin, Hi, Wi, x, ldx, w, logits_nchw, workspace, stream, &hl);
}

// The head on a BatchNorm output that is never stored: y is the raw output of the convolution in front of the BatchNorm, the head's
// input is act(y * scale + shift) (reference: upsample.deconv3 = ConvTranspose2d + BatchNorm2d + ReLU in front of last_deconv,
// deeplab_xception.py:369-374).  bf16, 256 input channels (the fused head kernel); same logits bits as dc_bn_apply + dc_head_fwd.
extern "C" int dc_head_fwd_bnin(int dtype, int N, int Cin, int Hi, int Wi, const void* y, int ldy, const float* scale, const float* shift,
                                int relu, const float* w, float* logits_nchw, void* workspace, void* stream) {
  DC_REQUIRE(logits_nchw != nullptr && scale != nullptr && shift != nullptr, "dc_head_fwd_bnin: null argument");
  DC_REQUIRE(dtype == DC_BF16 && Cin == 256 && g_head_fused, "dc_head_fwd_bnin: bf16, 256 channels, fused head kernel only");
  const HeadBnIn bi{scale, shift, relu};
  return head_fwd_impl(dtype, N, Cin, Hi, Wi, y, ldy, w, logits_nchw, workspace, stream, nullptr, &bi);
}

extern "C" int dc_head_fwd_loss_bnin(int dtype, int N, int Cin, int Hi, int Wi, const void* y, int ldy, const float* scale, const float* shift,
                                     int relu, const float* w, float* logits_nchw, void* workspace, const void* labels, int label_dtype_bytes,
                                     const float* class_weights, float grad_scale, double* loss_sum, float* dlogits, int64_t* pred,
                                     int64_t* counts, void* stream) {
  DC_REQUIRE(labels && class_weights && scale && shift, "dc_head_fwd_loss_bnin: bad argument");
  DC_REQUIRE(dtype == DC_BF16 && Cin == 256 && g_head_fused, "dc_head_fwd_loss_bnin: bf16, 256 channels, fused head kernel only");
  DC_REQUIRE(label_dtype_bytes == 1 || label_dtype_bytes == 4 || label_dtype_bytes == 8, "dc_head_fwd_loss_bnin: labels must be uint8, int32 or int64");
  HeadLoss hl{labels, label_dtype_bytes, class_weights, grad_scale, loss_sum, dlogits, pred, (unsigned long long*)counts};
  const HeadBnIn bi{scale, shift, relu};
  return head_fwd_impl(dtype, N, Cin, Hi, Wi, y, ldy, w, logits_nchw, workspace, stream, &hl, &bi);
}

extern "C" int dc_conv_dgrad_bnstats(const dc_conv_desc* d, int N, int Hi, int Wi, const void* dy, int lddy, const void* wb,
                                     void* dx, int lddx, const void* y, int ldy, const float* mean, const float* invstd,
                                     const float* mscale, const float* mshift, int relu, float* slab, void* stream);

// The head's data gradient as ONE streaming kernel (bf16, 256 channels): dx[px][256] = dP[px][32] . wb[256][32]^T is a GEMM with a single
// 32-deep K step, i.e. 1.9 GB of traffic (dP in, dx out, the BatchNorm input in for the sums) around 14 GFLOP.  On the tiled GEMM kernels a
// workgroup's life is load -> one MFMA step -> epilogue, one or three workgroups per CU: 620 us at local batch 8.  Here nothing goes
// through LDS: a wave keeps the weight fragments of its 64 channels in registers for its whole life and, per 16 pixels, loads one dP
// fragment and (BST) the BatchNorm input at its own outputs straight from memory, four pixel groups in flight; the output leaves from
// the accumulators.  BST: the BatchNorm-backward sums of dc_conv_dgrad_bnstats (sum g, sum g * xhat, g masked by the ReLU recomputed
// from y), one slab row per 128 pixels as there.  Same MFMA per output element as the tiled kernels: dx bit-equal.
//
// The gradient dx is 906 MB at local batch 8 and its only reader is that BatchNorm's backward, so it need not exist: MODE HD_SUMS takes
// the sums without storing dx, and -- once dc_bn_bwd_finalize has made dgamma / dbeta of them -- MODE HD_APPLY forms the same dx again
// (one MFMA step from the 64-byte gathered gradient row) and writes dy = gamma*invstd*(g - dbeta/count - xhat*dgamma/count) directly,
// element for element the arithmetic of bn_bwd_apply_kernel on the rounded dx (bn.hip): 2.9 GB of traffic for the pair instead of 4.6.
enum { HD_STORE = 0, HD_SUMS = 1, HD_APPLY = 2, HD_SUMS_WG = 3 };
struct HeadDgradBst {
  const bf16* y;
  int ldy;
  const float* mean;
  const float* invstd;
  const float* mscale;
  const float* mshift;
  int relu;
  float* slab;     // [2][rows][256]
  int rows;
  // HD_APPLY only
  const float* gamma;
  const float* dgamma;
  const float* dbeta;
  float inv_count;
  // HD_SUMS_WG only: one partial weight gradient [32 products][256 channels] per workgroup
  float* wslab;
};

__device__ inline float head_row_sum16(float v) {      // sum over the 16 lanes of a DPP row (the 16 pixels of a group)
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));   // row_mirror
  return v;
}

__device__ inline uint32_t head_swap_rows16(uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F); }   // lane ^ 16

template <int MODE>
__global__ __launch_bounds__(256) void head_dgrad_kernel(const bf16* __restrict__ dP, const bf16* __restrict__ wb, bf16* __restrict__ dx,
                                                         int lddx, long M, const HeadDgradBst b) {
  constexpr int CIN = 256, GPB = MODE == HD_SUMS_WG ? 2 : 4;     // pixel groups (of 16) requested together (the weight-gradient form has 32 more accumulator registers to keep)
  constexpr bool BST = true, SUMS = MODE != HD_APPLY, APPLY = MODE == HD_APPLY, WG = MODE == HD_SUMS_WG;
  // HD_SUMS_WG: the head's WEIGHT gradient dW[product k][channel c] = sum over pixels of dP[px][k] * act(y)[px][c] rides on the statistics pass,
  // which already has dP and y of every pixel in registers (the separate pass re-read the 906 MB BatchNorm input at local batch 8).  Per 32
  // pixels a wave writes its dP fragment and the activation of its 64 channels (act = relu(y * mscale + mshift) rounded to bf16: the head's
  // forward input, bit for bit) into a wave-private LDS image and takes both MFMA operands out of it through transposing reads: 8 MFMAs,
  // 32 accumulator registers for the wave's whole life, one slab row per workgroup at the end.
  extern __shared__ __attribute__((aligned(16))) char hd_smem[];
  char* const dpimg = hd_smem + (threadIdx.x >> 6) * (2 * TRI_QUAD);
  char* const actimg = dpimg + TRI_QUAD;
  [[maybe_unused]] f32x4 wacc[2][4];
  if constexpr (WG) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) wacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const bool odd = fg & 1;
  // A operand: rows = channels.  Block i of this wave: channels 64 * wave + 16 * i + (0..15); this lane's row fr, K group fg
  vec16 fa[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) fa[i] = ldg16(wb + (size_t)(64 * wave + 16 * i + fr) * HEAD_NP + fg * 8);
  // D[row = channel fg * 4 + r][col = pixel fr].  The lane pairs (fg, fg ^ 1) trade halves of two neighbouring channel blocks
  // (igemm256.hip's register epilogue), after which a lane owns EIGHT consecutive channels of its pixel per block pair p:
  //   cpair(p) = 64 * wave + (2 * p + odd) * 16 + (fg >> 1) * 8      -> 16-byte stores, 16-byte BatchNorm-input loads
  auto cpair = [&](int p) { return 64 * wave + (2 * p + (odd ? 1 : 0)) * 16 + (fg >> 1) * 8; };
  [[maybe_unused]] float mu[2][8], is[2][8], ms[2][8], mh[2][8], cb[2][8], cd[2][8];    // APPLY: is[] holds ca = gamma * invstd
  if constexpr (BST) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = cpair(p) + e;
        mu[p][e] = b.mean[c];
        is[p][e] = b.invstd[c];
        ms[p][e] = (b.relu || WG) ? b.mscale[c] : 0.f;       // (the weight-gradient form needs the activation itself, not only its mask)
        mh[p][e] = (b.relu || WG) ? b.mshift[c] : 0.f;
        if constexpr (APPLY) {      // the coefficients of bn_bwd_apply_kernel, expression for expression
          const float inv = is[p][e];
          const float ca = b.gamma[c] * inv;
          cb[p][e] = -ca * inv * b.dgamma[c] * b.inv_count;
          cd[p][e] = -ca * b.dbeta[c] * b.inv_count;
          is[p][e] = ca;
        }
      }
  }
  const long chunks = (M + 127) / 128;
  for (long chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
    [[maybe_unused]] float s0[2][8], s1[2][8];
    if constexpr (SUMS) {
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int e = 0; e < 8; ++e) s0[p][e] = s1[p][e] = 0.f;
    }
#pragma unroll 1
    for (int g0 = 0; g0 < 8; g0 += GPB) {
      vec16 fb[GPB];
      [[maybe_unused]] vec16 yv[GPB][2];
      bool ok[GPB];
#pragma unroll
      for (int u = 0; u < GPB; ++u) {
        const long px = chunk * 128 + (g0 + u) * 16 + fr;
        ok[u] = px < M;
        fb[u] = ok[u] ? ldg16(dP + (size_t)px * HEAD_NP + fg * 8) : zero16();
        if constexpr (BST) {
#pragma unroll
          for (int p = 0; p < 2; ++p) yv[u][p] = ok[u] ? ldg16(b.y + (size_t)px * b.ldy + cpair(p)) : zero16();
        }
      }
      if constexpr (WG) {
        // two 32-pixel stages (groups 0, 1 and 2, 3 of this request): image rows 16 (u & 1) + fr
#pragma unroll
        for (int st2 = 0; st2 < GPB / 2; ++st2) {
#pragma unroll
          for (int uu = 0; uu < 2; ++uu) {
            const int u = 2 * st2 + uu;
            const int r = 16 * uu + fr;
            *reinterpret_cast<vec16*>(dpimg + tri_slot(r, fg)) = fb[u];             // (zeros past the last pixel)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
              float yf[8];
              unpack(yv[u][p], yf, bf16());
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float t = fmaf(yf[e], ms[p][e], mh[p][e]);
                yf[e] = b.relu ? fmaxf(t, 0.f) : t;
              }
              vec16 v;
              pack(v, yf, bf16());
              if (!ok[u]) v = zero16();
              *reinterpret_cast<vec16*>(actimg + tri_slot(r, (2 * p + (odd ? 1 : 0)) * 2 + (fg >> 1))) = v;
            }
          }
          bf16x8 ka[2];
#pragma unroll
          for (int kb = 0; kb < 2; ++kb) ka[kb] = tri_frag(dpimg + tri_base(lane) + ((kb ^ tri_key(lane)) << 5));
#pragma unroll
          for (int cbk = 0; cbk < 4; ++cbk) {
            const bf16x8 cbf = tri_frag(actimg + tri_base(lane) + ((cbk ^ tri_key(lane)) << 5));
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) wacc[kb][cbk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[kb], cbf, wacc[kb][cbk], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < GPB; ++u) {
        const long px = chunk * 128 + (g0 + u) * 16 + fr;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};
          accA = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[2 * p]), __builtin_bit_cast(bf16x8, fb[u]), accA, 0, 0, 0);
          accB = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[2 * p + 1]), __builtin_bit_cast(bf16x8, fb[u]), accB, 0, 0, 0);
          const uint32_t a0 = pack2_bf16(accA[0], accA[1]), a1 = pack2_bf16(accA[2], accA[3]);
          const uint32_t b0 = pack2_bf16(accB[0], accB[1]), b1 = pack2_bf16(accB[2], accB[3]);
          const uint32_t r0 = head_swap_rows16(odd ? a0 : b0), r1 = head_swap_rows16(odd ? a1 : b1);
          vec16 o;
          o.w[0] = odd ? r0 : a0;
          o.w[1] = odd ? r1 : a1;
          o.w[2] = odd ? b0 : r0;
          o.w[3] = odd ? b1 : r1;
          if constexpr (MODE == HD_STORE) {
            if (ok[u]) stg16(dx + (size_t)px * lddx + cpair(p), o);
          }
          if constexpr (APPLY) {
            if (ok[u]) {
              float f[8], yf[8];
              unpack(o, f, bf16());
              unpack(yv[u][p], yf, bf16());
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float gm = (!b.relu || fmaf(yf[e], ms[p][e], mh[p][e]) > 0.f) ? f[e] : 0.f;
                f[e] = fmaf(is[p][e], gm, fmaf(cb[p][e], yf[e] - mu[p][e], cd[p][e]));
              }
              vec16 v;
              pack(v, f, bf16());
              stg16(dx + (size_t)px * lddx + cpair(p), v);      // dx: the BatchNorm input's gradient here
            }
          }
          if constexpr (SUMS) {
            if (ok[u]) {
              // the stored (bf16-rounded) gradient and the BatchNorm input, element for element as the tiled kernels' epilogue
              float f[8], yf[8];
              unpack(o, f, bf16());
              unpack(yv[u][p], yf, bf16());
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float gm = (!b.relu || fmaf(yf[e], ms[p][e], mh[p][e]) > 0.f) ? f[e] : 0.f;
                s0[p][e] += gm;
                s1[p][e] = fmaf(gm, (yf[e] - mu[p][e]) * is[p][e], s1[p][e]);
              }
            }
          }
        }
      }
    }
    if constexpr (SUMS) {
      // fold the 16 pixels of a group (the lanes of a DPP row); lane fr == 0 of every row then owns its eight channels of every pair
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float t0 = head_row_sum16(s0[p][e]), t1 = head_row_sum16(s1[p][e]);
          if (fr == 0) {
            const int c = cpair(p) + e;
            b.slab[((size_t)0 * b.rows + chunk) * CIN + c] = t0;
            b.slab[((size_t)1 * b.rows + chunk) * CIN + c] = t1;
          }
        }
    }
  }
  if constexpr (WG) {
    // wacc[kb][cbk][r] = dW of product 16 kb + 4 fg + r and channel 64 wave + 16 cbk + fr
    float* out = b.wslab + (size_t)blockIdx.x * (HEAD_NP * CIN);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int cbk = 0; cbk < 4; ++cbk)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(size_t)(16 * kb + 4 * fg + r) * CIN + 64 * wave + 16 * cbk + fr] = wacc[kb][cbk][r];
  }
}

static int g_head_dgrad_fused = 1;   // "head_dgrad_fused": 0 = the head's data gradient on the tiled GEMM kernels (A/B, tests)
extern "C" int dc_head_set_dgrad_fused(int v) { g_head_dgrad_fused = v ? 1 : 0; return 0; }
static int g_head_wgrad_fused = 1;   // "head_wgrad_fused": the head's weight gradient rides on the statistics pass of the two-pass backward (parts = 3)
extern "C" int dc_head_set_wgrad_fused(int v) { g_head_wgrad_fused = v ? 1 : 0; return 0; }

// bn_y != nullptr: x is act(bn(bn_y)) and the data gradient also leaves that BatchNorm's backward sums in bn_slab
static int head_bwd_impl(int dtype, int N, int Cin, int Hi, int Wi, const void* x, int ldx, const float* dlogits_nchw,
                         const float* w, void* dx, int lddx, float* grad_w, void* workspace, void* stream, const void* bn_y, int bn_ldy,
                         const float* bn_mean, const float* bn_invstd, const float* bn_mscale, const float* bn_mshift, int bn_relu,
                         float* bn_slab, int parts = 3) {
  // parts: 1 = the gathered gradient image + the data gradient, 2 = the weight gradient (reads the gathered image a parts-1 call left in
  // `workspace`: the caller may run it on another stream behind that call), 3 = both
  // x == nullptr (dc_head_bwd_bnin): the head's input was never stored; the weight gradient forms act(bn_y * bn_mscale + bn_mshift) itself
  if (x != nullptr) {
    if (int e = dc_check_view(x, ldx, Cin, dtype, "dc_head_bwd x")) return e;
  } else {
    DC_REQUIRE(bn_y != nullptr && bn_mscale != nullptr && bn_mshift != nullptr && dtype == DC_BF16, "dc_head_bwd_bnin: needs the BatchNorm input and its forward scale / shift (bf16)");
    if (int e = dc_check_view(bn_y, bn_ldy, Cin, dtype, "dc_head_bwd_bnin y")) return e;
  }
  const bool streaming = dtype == DC_BF16 && Cin == 256 && g_head_dgrad_fused && bn_y != nullptr && bn_slab != nullptr;
  if (dx != nullptr) {
    if (int e = dc_check_view(dx, lddx, Cin, dtype, "dc_head_bwd dx")) return e;
  } else {
    // dx == NULL: only the BatchNorm's sums are wanted (dc_head_bwd_bnin_apply writes that BatchNorm's input gradient later)
    DC_REQUIRE(streaming || !(parts & 1), "dc_head_bwd: dx may be NULL only where the streaming data-gradient kernel takes the BatchNorm sums (bf16, 256 channels)");
  }
  DC_REQUIRE(w && dlogits_nchw && grad_w && workspace && N > 0 && (parts & 3) != 0 && (parts & ~3) == 0, "dc_head_bwd: bad argument");
  DC_REQUIRE(((uintptr_t)workspace & 255) == 0, "dc_head_bwd: workspace must be 256-byte aligned");
  HeadWs ws = head_ws(dtype, N, Cin, Hi, Wi, workspace);
  hipStream_t st = (hipStream_t)stream;
  const long P = (long)N * Hi * Wi;
  const int np = HEAD_NP * Cin;
  // (the packed weights are rebuilt here too: backward may follow a forward that ran in another engine / workspace)
  dc_conv_desc d{dtype, 1, 1, 0, 1, 0, Cin, HEAD_NP};
  if (parts & 1) {
    if (dtype == DC_BF16) {
      hipLaunchKernelGGL(head_pack_kernel<bf16>, dim3(cdiv(np, 256)), dim3(256), 0, st, w, (bf16*)ws.wf, (bf16*)ws.wb, Cin);
      hipLaunchKernelGGL(head_gather_kernel<bf16>, dim3(cdiv(P, 256)), dim3(256), 0, st, dlogits_nchw, (bf16*)ws.dP, N, Hi, Wi);
    } else {
      hipLaunchKernelGGL(head_pack_kernel<float>, dim3(cdiv(np, 256)), dim3(256), 0, st, w, (float*)ws.wf, (float*)ws.wb, Cin);
      hipLaunchKernelGGL(head_gather_kernel<float>, dim3(cdiv(P, 256)), dim3(256), 0, st, dlogits_nchw, (float*)ws.dP, N, Hi, Wi);
    }
    DC_CHECK_LAUNCH();
  }
  // the statistics pass without a stored dx (dc_head_bwd_bnin, two-pass form) asked for both parts at once: the weight gradient rides on it
  const bool wg_rides = streaming && g_head_wgrad_fused && x == nullptr && dx == nullptr && parts == 3 && bn_relu >= 0 && P >= 256;
  if ((parts & 2) && !wg_rides) {
    if (x != nullptr) {
      if (int e = dc_conv_wgrad(&d, N, Hi, Wi, x, ldx, ws.dP, HEAD_NP, ws.slabs, ws.slab_bytes, ws.tmp, stream)) return e;
    } else {
      if (int e = conv_wgrad_bnin(&d, N, Hi, Wi, bn_y, bn_ldy, bn_mscale, bn_mshift, bn_relu, ws.dP, HEAD_NP, ws.slabs, ws.slab_bytes, ws.tmp, stream)) return e;
    }
    hipLaunchKernelGGL(head_wfinish_kernel, dim3(cdiv(Cin * HEAD_NC * 9, 256)), dim3(256), 0, st, (const float*)ws.tmp, grad_w, Cin);
    DC_CHECK_LAUNCH();
  }
  if (!(parts & 1)) return 0;
  if (streaming) {
    // (without the BatchNorm sums the streaming kernel has too little in flight per wave and loses to the tiled kernels: 755 vs 714 us)
    const int chunks = cdiv(P, 128);
    const int grid = chunks < 2048 ? chunks : 2048;
    if (int e = dc_check_view(bn_y, bn_ldy, Cin, dtype, "dc_head_bwd bn_y")) return e;
    DC_REQUIRE(bn_mean && bn_invstd && (!bn_relu || (bn_mscale && bn_mshift)), "dc_head_bwd: missing BatchNorm vectors");
    HeadDgradBst b{(const bf16*)bn_y, bn_ldy, bn_mean, bn_invstd, bn_mscale, bn_mshift, bn_relu, bn_slab, chunks, nullptr, nullptr, nullptr, 0.f, nullptr};
    if (wg_rides) {
      // one slab row (32 x 256 floats) per workgroup in the product image's space of the workspace (unused by the fused forward): at most
      // chunks / 2 rows of 32 KiB fit its M x 128 bytes
      int g2 = chunks / 2 < 512 ? chunks / 2 : 512;        // (252 registers: two workgroups per CU, one round)
      if (g2 < 1) g2 = 1;
      b.wslab = ws.P;
      static_assert(HEAD_NP == 32, "product image");
      DC_ONCE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&head_dgrad_kernel<HD_SUMS_WG>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * TRI_QUAD));
      hipLaunchKernelGGL(head_dgrad_kernel<HD_SUMS_WG>, dim3(g2), dim3(256), 4 * 2 * TRI_QUAD, st, (const bf16*)ws.dP, (const bf16*)ws.wb, (bf16*)nullptr, 0, (long)P, b);
      DC_CHECK_LAUNCH();
      const long nw = (long)HEAD_NP * Cin;
      hipLaunchKernelGGL(slab_rows_reduce_kernel, dim3(cdiv(nw, 8)), dim3(256), 0, st, (const float*)ws.P, ws.tmp, g2, nw);
      hipLaunchKernelGGL(head_wfinish_kernel, dim3(cdiv(Cin * HEAD_NC * 9, 256)), dim3(256), 0, st, (const float*)ws.tmp, grad_w, Cin);
      DC_CHECK_LAUNCH();
      return 0;
    }
    if (dx != nullptr)
      hipLaunchKernelGGL(head_dgrad_kernel<HD_STORE>, dim3(grid), dim3(256), 0, st, (const bf16*)ws.dP, (const bf16*)ws.wb, (bf16*)dx, lddx, (long)P, b);
    else
      hipLaunchKernelGGL(head_dgrad_kernel<HD_SUMS>, dim3(grid), dim3(256), 0, st, (const bf16*)ws.dP, (const bf16*)ws.wb, (bf16*)nullptr, 0, (long)P, b);
    DC_CHECK_LAUNCH();
    return 0;
  }
  if (bn_y != nullptr && bn_slab != nullptr)
    return dc_conv_dgrad_bnstats(&d, N, Hi, Wi, ws.dP, HEAD_NP, ws.wb, dx, lddx, bn_y, bn_ldy, bn_mean, bn_invstd, bn_mscale, bn_mshift,
                                 bn_relu, bn_slab, stream);
  return dc_conv_dgrad(&d, N, Hi, Wi, ws.dP, HEAD_NP, ws.wb, dx, lddx, 0, stream);
}

extern "C" int dc_head_bwd(int dtype, int N, int Cin, int Hi, int Wi, const void* x, int ldx, const float* dlogits_nchw,
                           const float* w, void* dx, int lddx, float* grad_w, void* workspace, void* stream) {
  return head_bwd_impl(dtype, N, Cin, Hi, Wi, x, ldx, dlogits_nchw, w, dx, lddx, grad_w, workspace, stream, nullptr, 0, nullptr, nullptr,
                       nullptr, nullptr, 0, nullptr);
}

extern "C" int dc_head_bwd_bnstats(int dtype, int N, int Cin, int Hi, int Wi, const void* x, int ldx, const float* dlogits_nchw,
                                   const float* w, void* dx, int lddx, float* grad_w, void* workspace, const void* bn_y, int bn_ldy,
                                   const float* bn_mean, const float* bn_invstd, const float* bn_mscale, const float* bn_mshift,
                                   int bn_relu, float* bn_slab, void* stream) {
  DC_REQUIRE(bn_y != nullptr && bn_slab != nullptr, "dc_head_bwd_bnstats: needs the BatchNorm input and a slab");
  return head_bwd_impl(dtype, N, Cin, Hi, Wi, x, ldx, dlogits_nchw, w, dx, lddx, grad_w, workspace, stream, bn_y, bn_ldy, bn_mean, bn_invstd,
                       bn_mscale, bn_mshift, bn_relu, bn_slab);
}

// Backward of the head on a never-stored BatchNorm(+ReLU) output (dc_head_fwd_bnin): y is the BatchNorm's input, scale / shift its forward
// coefficients.  bn_slab != NULL additionally leaves that BatchNorm's backward sums there (as dc_head_bwd_bnstats; needs mean / invstd).
extern "C" int dc_head_bwd_bnin(int dtype, int N, int Cin, int Hi, int Wi, const void* y, int ldy, const float* scale, const float* shift,
                                int relu, const float* dlogits_nchw, const float* w, void* dx, int lddx, float* grad_w, void* workspace,
                                const float* bn_mean, const float* bn_invstd, float* bn_slab, int parts, void* stream) {
  DC_REQUIRE(y != nullptr && scale != nullptr && shift != nullptr, "dc_head_bwd_bnin: null argument");
  DC_REQUIRE(bn_slab == nullptr || (bn_mean != nullptr && bn_invstd != nullptr), "dc_head_bwd_bnin: the BatchNorm sums need mean and invstd");
  return head_bwd_impl(dtype, N, Cin, Hi, Wi, nullptr, 0, dlogits_nchw, w, dx, lddx, grad_w, workspace, stream, y, ldy, bn_mean, bn_invstd, scale,
                       shift, relu, bn_slab, parts);
}

// Second pass of dc_head_bwd_bnin(..., dx = NULL, bn_slab, parts & 1): the head's data gradient is formed again from the gathered gradient
// image that call left in `workspace` and leaves as the BatchNorm INPUT's gradient dy (see head_dgrad_kernel, HD_APPLY); dgamma / dbeta
// are dc_bn_bwd_finalize's outputs for the slab of the first pass.  Bit-equal to dc_head_bwd_bnin with a stored dx + dc_bn_bwd_apply(relu = 2).
extern "C" int dc_head_bwd_bnin_apply(int dtype, int N, int Cin, int Hi, int Wi, const void* y, int ldy, const float* scale, const float* shift,
                                      int relu, const float* gamma, const float* bn_mean, const float* bn_invstd, const float* dgamma,
                                      const float* dbeta, long count, void* dy, int lddy, void* workspace, void* stream) {
  DC_REQUIRE(dtype == DC_BF16 && Cin == 256 && g_head_dgrad_fused, "dc_head_bwd_bnin_apply: served by the streaming data-gradient kernel only (bf16, 256 channels)");
  DC_REQUIRE(y && gamma && bn_mean && bn_invstd && dgamma && dbeta && workspace && N > 0 && count > 0 && (!relu || (scale && shift)),
             "dc_head_bwd_bnin_apply: bad argument");
  DC_REQUIRE(((uintptr_t)workspace & 255) == 0, "dc_head_bwd_bnin_apply: workspace must be 256-byte aligned");
  if (int e = dc_check_view(y, ldy, Cin, dtype, "dc_head_bwd_bnin_apply y")) return e;
  if (int e = dc_check_view(dy, lddy, Cin, dtype, "dc_head_bwd_bnin_apply dy")) return e;
  HeadWs ws = head_ws(dtype, N, Cin, Hi, Wi, workspace);
  const long P = (long)N * Hi * Wi;
  const int chunks = cdiv(P, 128);
  const int grid = chunks < 2048 ? chunks : 2048;
  const HeadDgradBst b{(const bf16*)y, ldy, bn_mean, bn_invstd, scale, shift, relu, nullptr, chunks, gamma, dgamma, dbeta, 1.0f / (float)count};
  hipLaunchKernelGGL(head_dgrad_kernel<HD_APPLY>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)ws.dP, (const bf16*)ws.wb, (bf16*)dy,
                     lddy, (long)P, b);
  DC_CHECK_LAUNCH();
  return 0;
}

extern "C" int dc_nchw_to_nhwc(int dtype, int N, int C, int H, int W, const float* x_nchw, void* out, int ldo, void* stream) {
  if (int e = dc_check_view(out, ldo, C, dtype, "dc_nchw_to_nhwc out")) return e;
  DC_REQUIRE(x_nchw && N > 0 && H > 0 && W > 0, "dc_nchw_to_nhwc: bad argument");
  const long total = (long)N * H * W;
  long blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DC_BF16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16>, dim3((int)blocks), dim3(256), 0, st, x_nchw, (bf16*)out, ldo, N, C, (long)H * W);
  else hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3((int)blocks), dim3(256), 0, st, x_nchw, (float*)out, ldo, N, C, (long)H * W);
  DC_CHECK_LAUNCH();
  return 0;
}

extern "C" int dc_input_normalize_hwc(int dtype, long npix, int Cfile, int C, const int* channels, const float* x_hwc,
                                      const float* shift, const float* scale, void* out, int ldo, void* stream) {
  if (int e = dc_check_view(out, ldo, C, dtype, "dc_input_normalize_hwc out")) return e;
  DC_REQUIRE(x_hwc && shift && scale && npix > 0 && Cfile >= C, "dc_input_normalize_hwc: bad argument");
  const int kpv = dtype == DC_BF16 ? 8 : 4;
  long blocks = (npix * (C / kpv) + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DC_BF16) hipLaunchKernelGGL(input_normalize_kernel<bf16>, dim3((int)blocks), dim3(256), 0, st, npix, Cfile, C, channels, x_hwc, shift, scale, (bf16*)out, ldo);
  else hipLaunchKernelGGL(input_normalize_kernel<float>, dim3((int)blocks), dim3(256), 0, st, npix, Cfile, C, channels, x_hwc, shift, scale, (float*)out, ldo);
  DC_CHECK_LAUNCH();
  return 0;
}

extern "C" int dc_input_normalize_hwc_to_nchw(int N, long HW, int Cfile, int C, const int* channels, const float* x_hwc,
                                              const float* shift, const float* scale, float* out_nchw, void* stream) {
  DC_REQUIRE(x_hwc && shift && scale && out_nchw && N > 0 && HW > 0 && C > 0 && Cfile >= C, "dc_input_normalize_hwc_to_nchw: bad argument");
  long blocks = ((long)N * HW + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(input_normalize_nchw_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, N, HW, Cfile, C, channels, x_hwc,
                     shift, scale, out_nchw);
  DC_CHECK_LAUNCH();
  return 0;
}
