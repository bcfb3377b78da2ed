// Weight gradient of the two thin 3x3 convolutions of the entry stem (16 -> 32 stride 2 on 768x1152, 32 -> 64 stride 1 on
// 384x576; deeplab_xception.py:145,149).  With so few channels the tiled GEMM kernels (wgrad.hip / wgrad256.hip) launch one
// workgroup set PER TAP, each re-reading dy, and fill 1/8 of their MFMA tile: 0.67-0.68 ms per layer against ~0.07 ms of
// compulsory traffic.  Here ONE pass produces all nine taps:
//
//     dW[t][co][ci] = sum over output pixels m of dy[m][co] * x[S*m - 1 + t][ci]        (M = co, N = (tap, ci), K = pixels)
//
//   * a workgroup walks a strip of output rows, 32 output pixels wide; per step the 32 dy pixels and the S NEW input rows of the
//     (32 S + 2)-pixel halo arrive by LDS-DMA (a ring of input rows keeps the other two rows of the 3x3 window), so both tensors
//     leave HBM once;
//   * rows stay [pixel][channel] in LDS; both MFMA operands are fetched channel-per-lane with ds_read_b64_tr_b16 -- for the x
//     operand the tap is just an address offset (row slot for ky, pixel offset for kx), the stride a multiplier;
//   * 4 waves = 2 halves of co x 2 halves of the (tap, ci) columns; a wave keeps <= 18 accumulator tiles;
//   * one partial [9][Co][Ci] per workgroup goes to the fp32 slab of wgrad.hip ([split][tap][Co][Ci]) and its fixed-order
//     reduction kernel finishes the job (deterministic).
#include "wgrad.h"

namespace dc {

namespace {

static __device__ __attribute__((aligned(256))) unsigned char thin_zero_page[256];
typedef __attribute__((address_space(1))) const void* gas_ptr;
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef __attribute__((address_space(3))) short4v lds_s4;

struct ThinArgs {
  const void* x;
  const void* dy;
  float* slab;
  int N, Hi, Wi, Ho, Wo, ldx, lddy;
  int nseg;        // 32-pixel column segments per output row
  int nchunk;      // row chunks per image
  int rows_per;    // output rows per chunk
};

template <int CIN, int COUT, int S>
struct ThinCfg {
  static constexpr int XP = 32 * S + 2;                 // halo pixels per input row segment
  static constexpr int XROWB = XP * CIN * 2;            // bytes
  static constexpr int XSLOT = (XROWB + 1023) / 1024 * 1024;   // ring slot (whole 1-KiB DMA chunks)
  static constexpr int RX = 4 * S;                      // input-row ring: S+2 live rows + S in flight
  static constexpr int DYB = 32 * COUT * 2;             // one step of dy
  static constexpr int DYSLOT = (DYB + 1023) / 1024 * 1024;
  static constexpr int LDS = RX * XSLOT + 2 * DYSLOT;
  static constexpr int MB = COUT / 16, NBT = CIN / 16, NB = 9 * NBT;
  static constexpr int MBW = MB / 2;                    // co blocks per wave
  static constexpr int NBH = (NB + 1) / 2;              // column blocks of the first wave column (the second gets NB - NBH)
};

template <int CIN, int COUT, int S>
__global__ __launch_bounds__(256) void thin_wgrad_kernel(const ThinArgs a) {
  typedef ThinCfg<CIN, COUT, S> K;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xring = smem;
  char* dring = smem + K::RX * K::XSLOT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wa = wave >> 1, wb = wave & 1;

  // workgroup -> (image, row chunk, column segment); consecutive workgroups of an XCD take neighbouring column segments
  const int nwg = gridDim.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, xslot = blockIdx.x >> 3;
  int unit = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + xslot;
  const int seg = unit % a.nseg;
  unit /= a.nseg;
  const int chunk = unit % a.nchunk, n = unit / a.nchunk;
  const int r0 = chunk * a.rows_per, r1 = min(a.Ho, r0 + a.rows_per);
  const int qx0 = seg * 32;
  const bf16* __restrict__ xg = reinterpret_cast<const bf16*>(a.x);
  const bf16* __restrict__ dg = reinterpret_cast<const bf16*>(a.dy);

  // ---- LDS-DMA of one input row segment (iy may be outside the image: zero page) and of one step of dy
  constexpr int XSPP = CIN * 2 / 16;                   // 16-byte slots per x pixel
  constexpr int XINS = (K::XP * XSPP + 63) / 64;       // DMA instructions per input row
  constexpr int DSPP = COUT * 2 / 16;
  constexpr int DINS = 32 * DSPP / 64;
  auto issue_xrow = [&](int iy) {
    char* dst = xring + ((iy + 1 + K::RX * 4) % K::RX) * K::XSLOT;      // slot by input row (iy >= -1)
    const bool yok = (unsigned)iy < (unsigned)a.Hi;
    for (int i = wave; i < XINS; i += 4) {
      const int s = i * 64 + lane;
      const int px = s / XSPP, sub = s % XSPP;
      const int ix = qx0 * S - 1 + px;
      const bool ok = yok && px < K::XP && (unsigned)ix < (unsigned)a.Wi;
      const void* src = ok ? (const void*)(xg + (((size_t)n * a.Hi + iy) * a.Wi + ix) * a.ldx + sub * 8) : (const void*)thin_zero_page;
      __builtin_amdgcn_global_load_lds((gas_ptr)src, (lds_ptr)(dst + i * 1024), 16, 0, 0);
    }
  };
  auto issue_dy = [&](int q) {
    char* dst = dring + (q & 1) * K::DYSLOT;
    for (int i = wave; i < DINS; i += 4) {
      const int s = i * 64 + lane;
      const int px = s / DSPP, sub = s % DSPP;
      const void* src = (const void*)(dg + (((size_t)n * a.Ho + q) * a.Wo + qx0 + px) * a.lddy + sub * 8);
      __builtin_amdgcn_global_load_lds((gas_ptr)src, (lds_ptr)(dst + i * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[K::MBW][K::NBH];
#pragma unroll
  for (int i = 0; i < K::MBW; ++i)
#pragma unroll
    for (int j = 0; j < K::NBH; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
  const int p0 = 8 * fg + tq;   // this lane's pixel of the first transposed read (the second is p0 + 4)

  // prologue: the three input rows of the first step and its dy
  if (r0 < r1) {
    for (int iy = S * r0 - 1; iy <= S * r0 + 1; ++iy) issue_xrow(iy);
    issue_dy(r0);
  }
  for (int q = r0; q < r1; ++q) {
    __syncthreads();              // vmcnt(0) + barrier: step q has landed, and everybody is done reading step q-1
    if (q + 1 < r1) {             // next step: S new input rows and its dy (their latency hides behind co-resident workgroups)
      for (int iy = S * (q + 1) + 2 - S; iy <= S * (q + 1) + 1; ++iy) issue_xrow(iy);
      issue_dy(q + 1);
    }
    const char* dcur = dring + (q & 1) * K::DYSLOT;
    vec16 fa[K::MBW];
#pragma unroll
    for (int i = 0; i < K::MBW; ++i) {
      const int cblk = wa * K::MBW + i;
      const short4v a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(dcur + p0 * (COUT * 2) + cblk * 32 + tp * 8));
      const short4v a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(dcur + (p0 + 4) * (COUT * 2) + cblk * 32 + tp * 8));
      const uint2 t0 = __builtin_bit_cast(uint2, a0), t1 = __builtin_bit_cast(uint2, a1);
      fa[i].w[0] = t0.x; fa[i].w[1] = t0.y; fa[i].w[2] = t1.x; fa[i].w[3] = t1.y;
    }
#pragma unroll
    for (int j = 0; j < K::NBH; ++j) {
      const int nb = wb * K::NBH + j;                 // column block = (tap, 16-channel half)
      if (nb < K::NB) {
        const int t = nb / K::NBT, hb = nb % K::NBT;
        const int ky = t / 3, kx = t % 3;
        const char* xrow = xring + ((S * q + ky + K::RX * 4) % K::RX) * K::XSLOT;   // input row S*q - 1 + ky -> slot (iy + 1) % RX
        const short4v b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(xrow + (p0 * S + kx) * (CIN * 2) + hb * 32 + tp * 8));
        const short4v b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(xrow + ((p0 + 4) * S + kx) * (CIN * 2) + hb * 32 + tp * 8));
        const uint2 t0 = __builtin_bit_cast(uint2, b0), t1 = __builtin_bit_cast(uint2, b1);
        vec16 fb;
        fb.w[0] = t0.x; fb.w[1] = t0.y; fb.w[2] = t1.x; fb.w[3] = t1.y;
#pragma unroll
        for (int i = 0; i < K::MBW; ++i)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb), acc[i][j], 0, 0, 0);
      }
    }
  }

  // ---- partial [tap][co][ci] of this workgroup
  const int fr = lane & 15;
  float* out = a.slab + (size_t)blockIdx.x * 9 * COUT * CIN;
#pragma unroll
  for (int j = 0; j < K::NBH; ++j) {
    const int nb = wb * K::NBH + j;
    if (nb < K::NB) {
      const int t = nb / K::NBT, hb = nb % K::NBT;
      const int ci = hb * 16 + fr;
#pragma unroll
      for (int i = 0; i < K::MBW; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = (wa * K::MBW + i) * 16 + fg * 4 + r;
          out[((size_t)t * COUT + co) * CIN + ci] = acc[i][j][r];
        }
    }
  }
}

struct ThinPlan {
  int nseg, nchunk, rows_per, blocks;
};
static ThinPlan thin_plan(int N, int Ho, int Wo) {
  ThinPlan p;
  p.nseg = Wo / 32;
  int want = 512 / (N * p.nseg);       // ~2 workgroups per CU; their partials cost 9*Co*Ci*4 bytes each
  if (want < 1) want = 1;
  if (want > Ho) want = Ho;
  p.rows_per = cdiv(Ho, want);
  p.nchunk = cdiv(Ho, p.rows_per);
  p.blocks = N * p.nseg * p.nchunk;
  return p;
}

}  // namespace

bool thin_wgrad_eligible(const dc_conv_desc& d, int Hi, int Wi) {
  if (d.dtype != DC_BF16 || d.transposed || d.k != 3 || d.pad != 1 || d.dil != 1) return false;
  const bool stem = d.cin == 16 && d.cout == 32 && d.stride == 2, conv2 = d.cin == 32 && d.cout == 64 && d.stride == 1;
  if (!stem && !conv2) return false;
  const int Wo = (Wi - 1) / d.stride + 1;
  return Wo % 32 == 0 && Hi >= 2 && Wi >= 2;
}

int thin_wgrad_splits(const dc_conv_desc& d, int N, int Hi, int Wi) {
  const int Ho = (Hi - 1) / d.stride + 1, Wo = (Wi - 1) / d.stride + 1;
  return thin_plan(N, Ho, Wo).blocks;
}

// writes thin_wgrad_splits(...) partials [tap][Co][Ci] into the slab; the caller reduces them (wgrad_reduce_kernel)
int launch_thin_wgrad(const dc_conv_desc& d, int N, int Hi, int Wi, const void* x, int ldx, const void* dy, int lddy, float* slab,
                      hipStream_t st) {
  ThinArgs a;
  a.x = x; a.dy = dy; a.slab = slab;
  a.N = N; a.Hi = Hi; a.Wi = Wi; a.Ho = (Hi - 1) / d.stride + 1; a.Wo = (Wi - 1) / d.stride + 1; a.ldx = ldx; a.lddy = lddy;
  const ThinPlan p = thin_plan(N, a.Ho, a.Wo);
  a.nseg = p.nseg; a.nchunk = p.nchunk; a.rows_per = p.rows_per;
  DC_REQUIRE((long)N * Hi * Wi < (1L << 31), "dc_conv_wgrad: tensor too large for the thin path");
  if (d.cin == 16) {
    typedef ThinCfg<16, 32, 2> K;
    static bool once = false;
    if (!once) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_wgrad_kernel<16, 32, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS); once = true; }
    hipLaunchKernelGGL((thin_wgrad_kernel<16, 32, 2>), dim3(p.blocks), dim3(256), K::LDS, st, a);
  } else {
    typedef ThinCfg<32, 64, 1> K;
    static bool once = false;
    if (!once) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_wgrad_kernel<32, 64, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS); once = true; }
    hipLaunchKernelGGL((thin_wgrad_kernel<32, 64, 1>), dim3(p.blocks), dim3(256), K::LDS, st, a);
  }
  DC_CHECK_LAUNCH();
  return 0;
}

}  // namespace dc
