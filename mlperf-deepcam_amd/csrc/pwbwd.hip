// BatchNorm backward (apply) + pointwise-conv data gradient + pointwise-conv weight gradient in ONE pass: the thin 1x1 layers of the entry flow.
//
//   dy[m][co]  = ca[co] * g + cb[co] * (y - mean[co]) + cd[co],   g = dout masked by the ReLU (y * mscale + mshift > 0)      (bn.hip: bn_bwd_apply_kernel)
//   dx[m][ci]  = sum over co of dy[m][co] * W[co][ci]                                                                        (conv data gradient)
//   dW[co][ci] = sum over m  of dy[m][co] * x[m][ci]                                                                         (conv weight gradient)
//
// Reference: SeparableConv2d_same.pointwise + the Block's BatchNorm2d behind it (deeplab_xception.py:62-66, 84-101), their backward at
// train_hdf5_ddp.py:363.
//
// Why.  Block 1 of the entry flow runs its two stride-1 pointwise convs (64 -> 128, 128 -> 128) on 384 x 576 images: at local batch 8 every 128-channel
// tensor there is 453 MB, all three kernels above are HBM-bound, and as three launches they move 3.17 GB for the 128 -> 128 layer
// (apply: dout, y -> dy; data gradient: dy -> dx; weight gradient: dy, x) where the layer's tensors -- dout, y, x in, dx out -- are 1.81 GB.
// dy is a function of (dout, y) and per-channel constants, so a kernel that has a 32-pixel slice of dout and y in registers can compute dy, leave it in
// LDS and feed BOTH products from there; dy never exists in memory.
//
// One 256-thread workgroup walks a contiguous range of 32-pixel stages:
//   * dout, y and x of the NEXT stage are requested into registers before the MFMA phase of the current one (16-byte loads, whole NHWC rows);
//   * dy (rounded to bf16 exactly as bn_bwd_apply_kernel stores it) and x are written into LDS in wgrad384.hip's image: quads of 64 channels,
//     [quad][32 pixels][128 B], the four 32-byte chunks of a row XOR-ed with (row >> 1) & 3;
//   * weight gradient: fragments of BOTH operands through ds_read_b64_tr_b16 (channel-per-lane from a [pixel][channel] image), 128 x CI fp32
//     accumulators spread over the four waves for the whole launch, one slab row [co][ci] per workgroup at the end (dc_fold_slabs sums them);
//   * data gradient as dx^T = W^T . dy^T: the W^T fragments of a wave (its 32 input channels x 128) live in registers for the whole launch, the dy^T
//     fragments are plain 16-byte reads of the image's pixel rows; the A rows are permuted so that a lane ends up with 8 consecutive input channels of
//     one pixel: 16-byte stores, 64 contiguous bytes per pixel and instruction.
#include "common.h"
#include "igemm.h"

namespace dc {

namespace {

constexpr int PB_BP = 32;                 // pixels per stage
constexpr int PB_QUAD = PB_BP * 128;      // one quad (64 channels) of a stage image: 4 KiB
constexpr int PB_CO = 128;                // output channels of the conv = channels of the BatchNorm

typedef __attribute__((address_space(3))) short4v* lds_short4;

struct PwBwdArgs {
  const bf16* dout; int lddo;       // gradient w.r.t. the BatchNorm output
  const bf16* y; int ldy;           // BatchNorm input = conv output
  const bf16* x; int ldx;           // conv input
  const bf16* wb; int ldwb;         // packed data-gradient operand [ci][ldwb], co contiguous (dc_conv_pack_weights: wb)
  bf16* dx; int lddx;
  float* wslab;                     // [gridDim.x][128][CI]
  const float* gamma; const float* mean; const float* invstd; const float* dgamma; const float* dbeta;
  const float* mscale; const float* mshift;
  int relu;                         // 0: none, 2: mask recomputed as y * mscale + mshift > 0
  float inv_count;
  long M;
  long nstages;
};

// one MFMA operand fragment out of a [pixel][channel] image: two transposing reads, pixels {4 fg ..} and {16 + 4 fg ..} (2 KiB apart)
__device__ inline bf16x8 tr_frag(const char* p) {
  const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4)(p));
  const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4)(p + 16 * 128));
  typedef __attribute__((ext_vector_type(8))) short short8v;
  short8v f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return __builtin_bit_cast(bf16x8, f);
}

__device__ inline void lds_barrier() {
  // LDS traffic only: the register prefetch of the next stage (global loads) stays in flight across the barrier
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int CI>
__global__ __launch_bounds__(256) void pw_bn_bwd_kernel(const PwBwdArgs a) {
  static_assert(CI == 64 || CI == 128, "input channels");
  constexpr int XQ = CI / 64;               // quads of the x image
  constexpr int XL = CI / 64;               // x loads per thread and stage
  constexpr int NCIB = CI == 128 ? 2 : 1;   // weight gradient: input-channel blocks of 16 per wave
  constexpr int NPX = CI == 128 ? 2 : 1;    // data gradient: pixel blocks of 16 per wave
  __shared__ __attribute__((aligned(16))) char s_dy[2 * PB_QUAD];
  __shared__ __attribute__((aligned(16))) char s_x[XQ * PB_QUAD];
  __shared__ __attribute__((aligned(16))) float s_coef[6][PB_CO];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long s0 = a.nstages * blockIdx.x / gridDim.x, s1 = a.nstages * (blockIdx.x + 1) / gridDim.x;

  // ---- per-channel coefficients of dy = ca * g + cb * (y - mean) + cd (bn_bwd_apply_kernel's arithmetic)
  if (tid < PB_CO) {
    const float is = a.invstd[tid];
    const float ca = a.gamma[tid] * is;
    s_coef[0][tid] = ca;
    s_coef[1][tid] = -ca * is * a.dgamma[tid] * a.inv_count;
    s_coef[2][tid] = -ca * a.dbeta[tid] * a.inv_count;
    s_coef[3][tid] = a.mean[tid];
    s_coef[4][tid] = a.relu == 2 ? a.mscale[tid] : 0.f;
    s_coef[5][tid] = a.relu == 2 ? a.mshift[tid] : 0.f;
  }

  // ---- data gradient: W^T fragments of this wave's 32 input channels, all four K steps (co), for the whole launch.  Row i of virtual block v is
  // input channel cib + 8 (i >> 2) + 4 v + (i & 3): lane group fg of the result then holds cib + 8 fg + 4 v + 0..3, i.e. with v = 0, 1 eight consecutive
  const int fr = lane & 15, fg = lane >> 4;
  const int cib = CI == 128 ? 32 * wave : 32 * (wave >> 1);
  bf16x8 aw[2][4];
#pragma unroll
  for (int v = 0; v < 2; ++v)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int ci = cib + 8 * (fr >> 2) + 4 * v + (fr & 3);
      aw[v][kk] = __builtin_bit_cast(bf16x8, ldg16(a.wb + (size_t)ci * a.ldwb + 32 * kk + 8 * fg));
    }

  // ---- load side: thread = (16-byte channel group, pixel rows r0 and r0 + 16) of dout and y; x alike (CI = 64: 8 groups, one row)
  const int grp = tid & 15, r0 = tid >> 4;
  const int xg = CI == 128 ? grp : (tid & 7), xr0 = CI == 128 ? r0 : (tid >> 3);
  vec16 vdo[2], vy[2], vx[XL];
  bool ok[2], xok[XL];
  auto prefetch = [&](long s) {
    const long m0 = s * PB_BP;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long m = m0 + r0 + 16 * u;
      ok[u] = m < a.M;
      const long mm = ok[u] ? m : 0;
      vdo[u] = ldg16(a.dout + (size_t)mm * a.lddo + grp * 8);
      vy[u] = ldg16(a.y + (size_t)mm * a.ldy + grp * 8);
    }
#pragma unroll
    for (int u = 0; u < XL; ++u) {
      const long m = m0 + xr0 + 16 * u;
      xok[u] = m < a.M;
      vx[u] = ldg16(a.x + (size_t)(xok[u] ? m : 0) * a.ldx + xg * 8);
    }
  };
  // byte offset of (pixel row r, 16-byte channel group g8 of a quad) in a quad image
  auto img = [](int r, int g8) { return r * 128 + ((((g8 >> 1) ^ (r >> 1)) & 3) << 5) + (g8 & 1) * 16; };

  f32x4 accw[NCIB][8];   // weight gradient: [ci block][co block]: rows ci = 4 fg + r, column co = fr
#pragma unroll
  for (int i = 0; i < NCIB; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) accw[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addresses (transposing reads): lane (fg, fr) supplies pixel rows 4 fg + (fr >> 2) (and + 16), columns 4 (fr & 3) .. of a 16-channel chunk
  const int frow = 4 * fg + (fr >> 2);
  const int fkey = (frow >> 1) & 3;
  const int fbase = frow * 128 + 8 * (fr & 3);

  if (s0 < s1) prefetch(s0);
  __syncthreads();       // coefficients visible (also waits for the first prefetch: once per launch)
  float ca[8], cb[8], cd[8], mu[8], ms[8], mh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    ca[e] = s_coef[0][grp * 8 + e]; cb[e] = s_coef[1][grp * 8 + e]; cd[e] = s_coef[2][grp * 8 + e];
    mu[e] = s_coef[3][grp * 8 + e]; ms[e] = s_coef[4][grp * 8 + e]; mh[e] = s_coef[5][grp * 8 + e];
  }

  for (long s = s0; s < s1; ++s) {
    // ---- dy of this stage from the prefetched registers, into the image; x beside it
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float g[8], yv[8];
      unpack(vdo[u], g, bf16());
      unpack(vy[u], yv, bf16());
      if (a.relu == 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = fmaf(yv[e], ms[e], mh[e]) > 0.f ? g[e] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) yv[e] = fmaf(ca[e], g[e], fmaf(cb[e], yv[e] - mu[e], cd[e]));
      vec16 v;
      pack(v, yv, bf16());
      if (!ok[u]) v = zero16();
      const int r = r0 + 16 * u;
      *reinterpret_cast<vec16*>(s_dy + (grp >> 3) * PB_QUAD + img(r, grp & 7)) = v;
    }
#pragma unroll
    for (int u = 0; u < XL; ++u) {
      const int r = xr0 + 16 * u;
      *reinterpret_cast<vec16*>(s_x + (xg >> 3) * PB_QUAD + img(r, xg & 7)) = xok[u] ? vx[u] : zero16();
    }
    lds_barrier();
    if (s + 1 < s1) prefetch(s + 1);

    // ---- weight gradient: accw[i][j] += x^T (ci block) . dy (co block) over the stage's 32 pixels
    bf16x8 fa[NCIB];
#pragma unroll
    for (int i = 0; i < NCIB; ++i) {
      const int cblk = CI == 128 ? 2 * wave + i : wave;
      fa[i] = tr_frag(s_x + (cblk >> 2) * PB_QUAD + fbase + (((cblk & 3) ^ fkey) << 5));
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bf16x8 fb = tr_frag(s_dy + (j >> 2) * PB_QUAD + fbase + (((j & 3) ^ fkey) << 5));
#pragma unroll
      for (int i = 0; i < NCIB; ++i) accw[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb, accw[i][j], 0, 0, 0);
    }

    // ---- data gradient: dx^T (this wave's 32 input channels x 16 pixels per block) = W^T . dy^T, K = 128 output channels in four steps
#pragma unroll
    for (int pbi = 0; pbi < NPX; ++pbi) {
      const int pb = CI == 128 ? pbi : (wave & 1);
      const int r = 16 * pb + fr;
      f32x4 d0 = f32x4{0.f, 0.f, 0.f, 0.f}, d1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int g8 = 4 * (kk & 1) + fg;      // 16-byte group of the quad: channels 32 kk + 8 fg ..
        const bf16x8 b = __builtin_bit_cast(bf16x8, *reinterpret_cast<const vec16*>(s_dy + (kk >> 1) * PB_QUAD + img(r, g8)));
        d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[0][kk], b, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[1][kk], b, d1, 0, 0, 0);
      }
      const long m = s * PB_BP + r;
      if (m < a.M) {
        float o[8] = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
        vec16 v;
        pack(v, o, bf16());
        stg16(a.dx + (size_t)m * a.lddx + cib + 8 * fg, v);
      }
    }
    lds_barrier();       // every wave is done with the images
  }

  // ---- one slab row per workgroup: [co][ci]
  float* out = a.wslab + (size_t)blockIdx.x * PB_CO * CI;
#pragma unroll
  for (int i = 0; i < NCIB; ++i) {
    const int cblk = CI == 128 ? 2 * wave + i : wave;
#pragma unroll
    for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(out + (size_t)(j * 16 + fr) * CI + cblk * 16 + 4 * fg) = accw[i][j];
  }
}


// 256 output channels (block 2 of the entry flow: 128 -> 256 and 256 -> 256 at 192 x 288).  The weight-gradient accumulators (256 x CI) and the
// W^T fragments no longer fit one workgroup, so TWO 512-thread workgroups share a pixel range (same id mod 8, i.e. the same XCD, eight ids
// apart: they run side by side and the second reader of dout, y, x is served by that XCD's L2 -- the kernel is HBM-bound, L2 traffic is
// not what it pays for).  Both form the whole dy tile; member h keeps the weight-gradient rows of output channels 128 h .. 128 h + 127 and
// computes HALF of the data gradient: input channels 128 h .. (CI = 256: a wave = 32 input channels x one of the stage's two pixel blocks) or
// pixel block h of the stage (CI = 128: a wave = one 16-row virtual block of 32 input channels, 8-byte stores).
template <int CI>
__global__ __launch_bounds__(512) void pw_bn_bwd256_kernel(const PwBwdArgs a) {
  static_assert(CI == 128 || CI == 256, "input channels");
  constexpr int CO = 256;
  constexpr int XQ = CI / 64;
  constexpr int XL = CI / 128;              // x loads per thread and stage
  constexpr int NCIB = CI / 128;            // weight gradient: input-channel blocks of 16 per wave (8 waves)
  constexpr int NV = CI == 256 ? 2 : 1;     // data gradient: virtual 16-row blocks per wave
  __shared__ __attribute__((aligned(16))) char s_dy[4 * PB_QUAD];
  __shared__ __attribute__((aligned(16))) char s_x[XQ * PB_QUAD];
  __shared__ __attribute__((aligned(16))) float s_coef[6][CO];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroup b: pair q = (b / 16) * 8 + b % 8, member h = (b / 8) & 1
  const int half = (blockIdx.x >> 3) & 1;
  const int pair = (blockIdx.x >> 4) * 8 + (blockIdx.x & 7), npairs = gridDim.x >> 1;
  const long s0 = a.nstages * pair / npairs, s1 = a.nstages * (pair + 1) / npairs;

  if (tid < CO) {
    const float is = a.invstd[tid];
    const float ca = a.gamma[tid] * is;
    s_coef[0][tid] = ca;
    s_coef[1][tid] = -ca * is * a.dgamma[tid] * a.inv_count;
    s_coef[2][tid] = -ca * a.dbeta[tid] * a.inv_count;
    s_coef[3][tid] = a.mean[tid];
    s_coef[4][tid] = a.relu == 2 ? a.mscale[tid] : 0.f;
    s_coef[5][tid] = a.relu == 2 ? a.mshift[tid] : 0.f;
  }

  const int fr = lane & 15, fg = lane >> 4;
  // data gradient: this wave's 32 input channels (rows permuted as in the 128-channel kernel), all eight K steps
  const int cib = CI == 256 ? 128 * half + 32 * (wave >> 1) : 32 * (wave >> 1);
  const int dpb = CI == 256 ? (wave & 1) : half;          // pixel block of the stage this wave computes
  const int dv0 = CI == 256 ? 0 : (wave & 1);              // first virtual block
  bf16x8 aw[NV][8];
#pragma unroll
  for (int v = 0; v < NV; ++v)
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int ci = cib + 8 * (fr >> 2) + 4 * (dv0 + v) + (fr & 3);
      aw[v][kk] = __builtin_bit_cast(bf16x8, ldg16(a.wb + (size_t)ci * a.ldwb + 32 * kk + 8 * fg));
    }

  const int grp = tid & 31, r0 = tid >> 5;                  // dout / y: 32 channel groups, rows r0 and r0 + 16
  const int xg = CI == 256 ? grp : (tid & 15), xr0 = CI == 256 ? r0 : (tid >> 4);
  vec16 vdo[2], vy[2], vx[XL];
  bool ok[2], xok[XL];
  auto prefetch = [&](long s) {
    const long m0 = s * PB_BP;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long m = m0 + r0 + 16 * u;
      ok[u] = m < a.M;
      const long mm = ok[u] ? m : 0;
      vdo[u] = ldg16(a.dout + (size_t)mm * a.lddo + grp * 8);
      vy[u] = ldg16(a.y + (size_t)mm * a.ldy + grp * 8);
    }
#pragma unroll
    for (int u = 0; u < XL; ++u) {
      const long m = m0 + xr0 + 16 * u;
      xok[u] = m < a.M;
      vx[u] = ldg16(a.x + (size_t)(xok[u] ? m : 0) * a.ldx + xg * 8);
    }
  };
  auto img = [](int r, int g8) { return r * 128 + ((((g8 >> 1) ^ (r >> 1)) & 3) << 5) + (g8 & 1) * 16; };

  f32x4 accw[NCIB][8];   // [ci block][co block of this member's half]
#pragma unroll
  for (int i = 0; i < NCIB; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) accw[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frow = 4 * fg + (fr >> 2);
  const int fkey = (frow >> 1) & 3;
  const int fbase = frow * 128 + 8 * (fr & 3);

  if (s0 < s1) prefetch(s0);
  __syncthreads();

  for (long s = s0; s < s1; ++s) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float g[8], yv[8];
      unpack(vdo[u], g, bf16());
      unpack(vy[u], yv, bf16());
      // (the coefficients are read per stage: 48 registers more would not fit beside the accumulators and the W^T fragments)
      const float4* c4 = reinterpret_cast<const float4*>(&s_coef[0][0]);
      float ca[8], cb[8], cd[8], mu[8], ms[8], mh[8];
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const float4 q0 = c4[(0 * CO + grp * 8) / 4 + h2], q1 = c4[(1 * CO + grp * 8) / 4 + h2], q2 = c4[(2 * CO + grp * 8) / 4 + h2];
        const float4 q3 = c4[(3 * CO + grp * 8) / 4 + h2], q4 = c4[(4 * CO + grp * 8) / 4 + h2], q5 = c4[(5 * CO + grp * 8) / 4 + h2];
        ca[4 * h2] = q0.x; ca[4 * h2 + 1] = q0.y; ca[4 * h2 + 2] = q0.z; ca[4 * h2 + 3] = q0.w;
        cb[4 * h2] = q1.x; cb[4 * h2 + 1] = q1.y; cb[4 * h2 + 2] = q1.z; cb[4 * h2 + 3] = q1.w;
        cd[4 * h2] = q2.x; cd[4 * h2 + 1] = q2.y; cd[4 * h2 + 2] = q2.z; cd[4 * h2 + 3] = q2.w;
        mu[4 * h2] = q3.x; mu[4 * h2 + 1] = q3.y; mu[4 * h2 + 2] = q3.z; mu[4 * h2 + 3] = q3.w;
        ms[4 * h2] = q4.x; ms[4 * h2 + 1] = q4.y; ms[4 * h2 + 2] = q4.z; ms[4 * h2 + 3] = q4.w;
        mh[4 * h2] = q5.x; mh[4 * h2 + 1] = q5.y; mh[4 * h2 + 2] = q5.z; mh[4 * h2 + 3] = q5.w;
      }
      if (a.relu == 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = fmaf(yv[e], ms[e], mh[e]) > 0.f ? g[e] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) yv[e] = fmaf(ca[e], g[e], fmaf(cb[e], yv[e] - mu[e], cd[e]));
      vec16 v;
      pack(v, yv, bf16());
      if (!ok[u]) v = zero16();
      const int r = r0 + 16 * u;
      *reinterpret_cast<vec16*>(s_dy + (grp >> 3) * PB_QUAD + img(r, grp & 7)) = v;
    }
#pragma unroll
    for (int u = 0; u < XL; ++u) {
      const int r = xr0 + 16 * u;
      *reinterpret_cast<vec16*>(s_x + (xg >> 3) * PB_QUAD + img(r, xg & 7)) = xok[u] ? vx[u] : zero16();
    }
    lds_barrier();
    if (s + 1 < s1) prefetch(s + 1);

    // ---- weight gradient of this member's 128 output channels
    bf16x8 fa[NCIB];
#pragma unroll
    for (int i = 0; i < NCIB; ++i) {
      const int cblk = NCIB * wave + i;
      fa[i] = tr_frag(s_x + (cblk >> 2) * PB_QUAD + fbase + (((cblk & 3) ^ fkey) << 5));
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int cob = 8 * half + j;
      const bf16x8 fb = tr_frag(s_dy + (cob >> 2) * PB_QUAD + fbase + (((cob & 3) ^ fkey) << 5));
#pragma unroll
      for (int i = 0; i < NCIB; ++i) accw[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb, accw[i][j], 0, 0, 0);
    }

    // ---- this member's half of the data gradient, K = 256 output channels in eight steps
    {
      const int r = 16 * dpb + fr;
      f32x4 d[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) d[v] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const int g8 = 4 * (kk & 1) + fg;
        const bf16x8 b = __builtin_bit_cast(bf16x8, *reinterpret_cast<const vec16*>(s_dy + (kk >> 1) * PB_QUAD + img(r, g8)));
#pragma unroll
        for (int v = 0; v < NV; ++v) d[v] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[v][kk], b, d[v], 0, 0, 0);
      }
      const long m = s * PB_BP + r;
      if (m < a.M) {
        if constexpr (NV == 2) {
          float o[8] = {d[0][0], d[0][1], d[0][2], d[0][3], d[1][0], d[1][1], d[1][2], d[1][3]};
          vec16 v;
          pack(v, o, bf16());
          stg16(a.dx + (size_t)m * a.lddx + cib + 8 * fg, v);
        } else {
          uint2 v;
          v.x = pack2_bf16(d[0][0], d[0][1]);
          v.y = pack2_bf16(d[0][2], d[0][3]);
          *reinterpret_cast<uint2*>(a.dx + (size_t)m * a.lddx + cib + 8 * fg + 4 * dv0) = v;
        }
      }
    }
    lds_barrier();
  }

  // ---- one slab row per PAIR, this member's 128 output channels of it: [co][ci]
  float* out = a.wslab + ((size_t)pair * CO + 128 * half) * CI;
#pragma unroll
  for (int i = 0; i < NCIB; ++i) {
    const int cblk = NCIB * wave + i;
#pragma unroll
    for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(out + (size_t)(j * 16 + fr) * CI + cblk * 16 + 4 * fg) = accw[i][j];
  }
}

static int g_pw_bn_bwd = 3;       // tuning switch "pw_bn_bwd": 0 = the three separate passes, bit 0 = layers with 128 output channels, bit 1 = with 256

// workgroups = slab rows: every CU full once (the kernel is persistent; a partial second round would cost a whole one)
// 256 output channels: pairs of 512-thread workgroups, one workgroup per CU; the rows are the pairs, a multiple of 8 (pairs sit 8 ids apart)
int pwbwd256_pairs(long M) {
  static int cus = 0;
  static hipError_t err = hipSuccess;
  DC_ONCE({
    int dev = 0;
    err = hipGetDevice(&dev);
    if (err == hipSuccess) err = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  });
  if (err != hipSuccess || cus < 16) return 0;
  const long ns = (M + PB_BP - 1) / PB_BP;
  long pairs = cus / 16 * 8;
  while (pairs > 8 && pairs > ns / 8) pairs -= 8;
  return (int)pairs;
}

int pwbwd_grid(int Cin, long M) {
  static int slots[2] = {0, 0};     // resident workgroups on the device, per instantiation
  static hipError_t err = hipSuccess;
  DC_ONCE({
    int dev = 0, cus = 0, n64 = 0, n128 = 0;
    err = hipGetDevice(&dev);
    if (err == hipSuccess) err = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (err == hipSuccess) err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n64, pw_bn_bwd_kernel<64>, 256, 0);
    if (err == hipSuccess) err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n128, pw_bn_bwd_kernel<128>, 256, 0);
    slots[0] = cus * n64;
    slots[1] = cus * n128;
  });
  if (err != hipSuccess) return 0;
  const long ns = (M + PB_BP - 1) / PB_BP;
  long g = slots[Cin == 128];
  if (g > ns / 8) g = ns / 8;       // at least eight stages per workgroup: below that the slab rows cost more than the fusion saves
  return (int)g;
}

}  // namespace

void pw_bn_bwd_set(int v) { g_pw_bn_bwd = v; }

}  // namespace dc

using namespace dc;

// Slab rows (= workgroups) of dc_pw_bn_bwd for this shape; 0: not served (the caller runs dc_bn_bwd_apply, dc_conv_dgrad and the weight gradient)
extern "C" int dc_pw_bn_bwd_rows(int dtype, int Cin, int Cout, long M) {
  if (!g_pw_bn_bwd || dtype != DC_BF16 || M < 65536 || M >= (1L << 31)) return 0;
  if (Cout == 256 && (Cin == 128 || Cin == 256)) return (g_pw_bn_bwd & 2) ? pwbwd256_pairs(M) : 0;
  if (Cout != PB_CO || (Cin != 64 && Cin != 128) || !(g_pw_bn_bwd & 1)) return 0;
  return pwbwd_grid(Cin, M);
}

extern "C" int dc_pw_bn_bwd(int dtype, long M, int Cin, int Cout, long count, const void* dout, int lddo, const void* y, int ldy, int relu,
                            const float* gamma, const float* save_mean, const float* save_invstd, const float* dgamma, const float* dbeta,
                            const float* mscale, const float* mshift, const void* x, int ldx, const void* wb, void* dx, int lddx, float* wslab,
                            int wslab_rows, void* stream) {
  const int rows = dc_pw_bn_bwd_rows(dtype, Cin, Cout, M);
  DC_REQUIRE(rows > 0, "dc_pw_bn_bwd: shape not served (dc_pw_bn_bwd_rows)");
  // the caller sized its slab (and the fold that follows) from an earlier dc_pw_bn_bwd_rows call: a planner that has changed its mind since
  // (another device, a switch flipped in between) must not write past it or leave rows the fold then sums as garbage
  DC_REQUIRE(wslab_rows == rows, "dc_pw_bn_bwd: wslab_rows is not what dc_pw_bn_bwd_rows returns for this shape now");
  DC_REQUIRE(relu == 0 || relu == 2, "dc_pw_bn_bwd: the ReLU mask is recomputed from y (relu 0 or 2)");
  DC_REQUIRE(count > 0 && gamma && save_mean && save_invstd && dgamma && dbeta && wslab && wb, "dc_pw_bn_bwd: null argument");
  DC_REQUIRE(relu == 0 || (mscale && mshift), "dc_pw_bn_bwd: the mask needs the forward scale / shift");
  if (int e = dc_check_view(dout, lddo, Cout, dtype, "dc_pw_bn_bwd dout")) return e;
  if (int e = dc_check_view(y, ldy, Cout, dtype, "dc_pw_bn_bwd y")) return e;
  if (int e = dc_check_view(x, ldx, Cin, dtype, "dc_pw_bn_bwd x")) return e;
  if (int e = dc_check_view(dx, lddx, Cin, dtype, "dc_pw_bn_bwd dx")) return e;
  DC_REQUIRE(((uintptr_t)wb & 15) == 0, "dc_pw_bn_bwd: weights unaligned");
  PwBwdArgs a;
  a.dout = (const bf16*)dout; a.lddo = lddo; a.y = (const bf16*)y; a.ldy = ldy; a.x = (const bf16*)x; a.ldx = ldx;
  a.wb = (const bf16*)wb; a.ldwb = weight_ld(Cout); a.dx = (bf16*)dx; a.lddx = lddx; a.wslab = wslab;
  a.gamma = gamma; a.mean = save_mean; a.invstd = save_invstd; a.dgamma = dgamma; a.dbeta = dbeta; a.mscale = mscale; a.mshift = mshift;
  a.relu = relu; a.inv_count = 1.0f / (float)count; a.M = M; a.nstages = (M + PB_BP - 1) / PB_BP;
  if (Cout == 256) {
    if (Cin == 256) hipLaunchKernelGGL(pw_bn_bwd256_kernel<256>, dim3(2 * rows), dim3(512), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(pw_bn_bwd256_kernel<128>, dim3(2 * rows), dim3(512), 0, (hipStream_t)stream, a);
  } else if (Cin == 128) hipLaunchKernelGGL(pw_bn_bwd_kernel<128>, dim3(rows), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(pw_bn_bwd_kernel<64>, dim3(rows), dim3(256), 0, (hipStream_t)stream, a);
  DC_CHECK_LAUNCH();
  return 0;
}
