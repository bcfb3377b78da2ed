// SeparableConv2d_same forward as ONE operator for the entry flow's thin layers: depthwise 3x3 (stride 1) -> pointwise 1x1, bf16.
//
//   d[p][ci] = sum over the nine taps t of act(x)[p + t][ci] * wdw[t][ci]          (act = the BatchNorm + ReLU of the producer, applied on load)
//   y[p][co] = sum over ci of d[p][ci] * W[co][ci]                                  (+ the BatchNorm partial sums of the stored y)
//
// Reference: SeparableConv2d_same.forward (deeplab_xception.py:62-66: fixed_padding, conv1, pointwise) -- SURVEY section 2.3 K4: "fused as the
// A-operand producer" of the pointwise GEMM.
//
// Why only here.  On the 728-channel layers the pointwise GEMM owns every register and every LDS byte of a CU (igemm384.hip) and the depthwise
// output is a small part of its traffic.  Block 1 is the opposite: 64 / 128 channels on 384 x 576 images, every tensor is 226 - 453 MB at local
// batch 8, both kernels are HBM-bound and as two launches (+ the two-stage fold of the 13 824 statistics rows the tiled GEMM leaves) they move
// the depthwise output twice more than needed: 415 us for the 128 -> 128 layer against 1.36 GB of tensors.  Here a persistent 256-thread
// workgroup walks 8 x 16 (64 channels: 8 x 32) pixel tiles:
//   * the (8+2) x (TW+2) halo tile of x arrives by LDS-DMA (zero page outside the image), gets the producer's BatchNorm + ReLU in place
//     (dwtile_common.h: bn_transform_tile) and feeds dwtile.hip's stencil, tap for tap, so d is bit-identical to dc_dwconv_fwd's;
//   * every strip goes to memory (the backward pass needs d: the pointwise weight gradient, the depthwise data gradient's partner) AND, as the
//     same rounded bf16, into an LDS image [pixel][ci] (16-byte slots XOR-ed with the pixel index: conflict-free for the 8-byte stencil
//     stores and the 16-byte fragment reads);
//   * pointwise: y^T = W . d^T.  A wave owns 32 output channels, its W fragments live in registers for the whole launch, a lane's B fragment
//     is 16 bytes of ITS pixel's image row; the A rows are permuted so that a lane ends with 8 consecutive output channels of a pixel
//     (16-byte stores); the BatchNorm sums of the stored values stay in registers across ALL tiles: one slab row per workgroup.
#include "common.h"
#include "dwtile_common.h"
#include "igemm.h"

namespace dc {

namespace {

constexpr int SF_CO = 128;

template <int CI>
struct SfCfg {
  static constexpr int CG = CI / 8;                       // 16-byte channel groups of a pixel: 16 / 8
  static constexpr int TH = 8, TW = 8 * (32 / CG);        // 8 x 16 / 8 x 32 pixels
  static constexpr int HH = TH + 2, HW = TW + 2, HP = HH * HW;
  static constexpr int NPX = TH * TW;                     // 128 / 256
  static constexpr int ITER = (HP * CG + 255) / 256;      // LDS-DMA instructions per wave
  static constexpr int HALO = ITER * 256 * 16;            // 48 / 44 KiB
  static constexpr int AROW = CI * 2;                     // image row: 256 / 128 B
  static constexpr int AIMG = NPX * AROW;                 // 32 KiB
  static constexpr int LDS = HALO + AIMG;
  static constexpr int NSL = 128 / CG, SPR = TW / DT_PX, SPT = TH * SPR / NSL;      // strip lanes, strips per row, strips per thread (= 4)
  static constexpr int KS = CI / 32;                      // K steps of the pointwise product
  static_assert(LDS <= 80 * 1024, "two workgroups per CU");
};

struct SepFwdArgs {
  const bf16* x; int ldx;             // input (or the raw conv output in front of a BatchNorm applied on load)
  const float* pscale; const float* pshift; int prelu;
  const float* wdw;                   // packed depthwise taps [9][CI]
  bf16* d; int ldd;                   // depthwise output
  const bf16* wf;  int ldw;           // pointwise weights [co][ldw], ci contiguous
  bf16* y; int ldy;
  float* slab; int slab_rows;         // [2][slab_rows][128] or null
  int N, H, W, ntx, nty, ntiles;
  const void* zero_page;
};

static __device__ __attribute__((aligned(256))) unsigned char sf_zero_page[256];
typedef __attribute__((address_space(1))) const void* gas_ptr;
typedef __attribute__((address_space(3))) void* lds_ptr;

__device__ inline float sf_row_sum16(float v) {      // sum over the 16 lanes of a DPP row
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));
  return v;
}

template <int CI>
__global__ __launch_bounds__(256) void sepconv_fwd_kernel(const SepFwdArgs a) {
  typedef SfCfg<CI> K;
  constexpr int CG = K::CG, KH = 4, WC = DT_PX + 2, SLOTS = K::AROW / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const halo = smem;
  char* const aimg = smem + K::HALO;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, W = a.W;

  // depthwise side: half-group h (4 channels), strip lane sl
  const int h = tid % (2 * CG), sl = tid / (2 * CG);
  const int ch0 = h * KH;
  float wk[9][KH];
  load_taps<KH>(a.wdw, ch0, CI, false, wk);
  // pointwise side: this wave's 32 output channels; row i of virtual block v is channel 32 w + 8 (i >> 2) + 4 v + (i & 3)
  const int fr = lane & 15, fg = lane >> 4;
  bf16x8 aw[2][K::KS];
#pragma unroll
  for (int v = 0; v < 2; ++v)
#pragma unroll
    for (int kk = 0; kk < K::KS; ++kk) {
      const int co = 32 * wave + 8 * (fr >> 2) + 4 * v + (fr & 3);
      aw[v][kk] = __builtin_bit_cast(bf16x8, ldg16(a.wf + (size_t)co * a.ldw + 32 * kk + 8 * fg));
    }
  float ssum[8], ssq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) ssum[e] = ssq[e] = 0.f;
  const uintptr_t zp = (uintptr_t)a.zero_page;
  const int g = tid % CG;                       // DMA side: this lane's 16-byte channel group

  for (int t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
    const int tx = t % a.ntx;
    const int r = t / a.ntx;
    const int ty = r % a.nty, n = r / a.nty;
    const int y0 = ty * K::TH, x0 = tx * K::TW;
    __syncthreads();                            // the previous tile's fragment reads are done (the halo region was free since its stencil)
    {
      const bf16* base = a.x + (size_t)n * H * W * a.ldx + (size_t)g * 8;
#pragma unroll
      for (int it = 0; it < K::ITER; ++it) {
        const int hp = (it * 256 + tid) / CG;
        const int hy = hp / K::HW, hx = hp - hy * K::HW;
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        const bool ok = hp < K::HP && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        const uintptr_t src = ok ? (uintptr_t)(base + ((size_t)iy * W + ix) * a.ldx) : zp;
        __builtin_amdgcn_global_load_lds((gas_ptr)src, (lds_ptr)(halo + (it * 256 + wave * 64) * 16), 16, 0, 0);
      }
    }
    __syncthreads();                            // vmcnt(0) + barrier: the halo tile has landed
    if (a.pscale != nullptr) {
      bn_transform_tile<bf16, K::HH, K::HW, CG>(halo, y0 - 1, x0 - 1, H, W, a.pscale, a.pshift, a.prelu, 0, CG);
      __syncthreads();
    }
    // ---- depthwise stencil (dwtile.hip's, tap for tap): strips of 4 pixels x 4 channels; every result goes to memory and into the image
    const char* tile = halo + h * 8;
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
      for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
        for (int c = 0; c < WC; ++c) {
          float f[KH];
          unpack8(*reinterpret_cast<const vec8*>(tile + ((row + ky) * K::HW + xs + c) * (CG * 16)), f, bf16());
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int j = c - kx;
            if (j >= 0 && j < DT_PX) {
#pragma unroll
              for (int e = 0; e < KH; ++e) acc[j][e] = fmaf(f[e], wk[ky * 3 + kx][e], acc[j][e]);
            }
          }
        }
      }
      const int oy = y0 + row;
#pragma unroll
      for (int j = 0; j < DT_PX; ++j) {
        vec8 v;
        pack8(v, acc[j], bf16());
        const int p = row * K::TW + xs + j;     // pixel of the tile
        *reinterpret_cast<vec8*>(aimg + p * K::AROW + ((((h >> 1) ^ p) & (SLOTS - 1)) << 4) + (h & 1) * 8) = v;
        *reinterpret_cast<vec8*>(a.d + (((size_t)n * H + oy) * W + x0 + xs + j) * a.ldd + ch0) = v;
      }
    }
    __syncthreads();                            // the image is complete
    // ---- pointwise: 16 pixels per block; K = CI in steps of 32
#pragma unroll 2
    for (int pb = 0; pb < K::NPX / 16; ++pb) {
      const int p = 16 * pb + fr;
      f32x4 d0 = f32x4{0.f, 0.f, 0.f, 0.f}, d1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < K::KS; ++kk) {
        const bf16x8 b = __builtin_bit_cast(bf16x8, *reinterpret_cast<const vec16*>(aimg + p * K::AROW + ((((4 * kk + fg) ^ p) & (SLOTS - 1)) << 4)));
        d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[0][kk], b, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[1][kk], b, d1, 0, 0, 0);
      }
      float o[8] = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
      vec16 v;
      pack(v, o, bf16());
      const int oy = y0 + p / K::TW, ox = x0 + p % K::TW;
      stg16(a.y + (((size_t)n * H + oy) * W + ox) * a.ldy + 32 * wave + 8 * fg, v);
      unpack(v, o, bf16());                     // the BatchNorm sums are those of the STORED values
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        ssum[e] += o[e];
        ssq[e] = fmaf(o[e], o[e], ssq[e]);
      }
    }
  }
  if (a.slab != nullptr) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float s0 = sf_row_sum16(ssum[e]), s1 = sf_row_sum16(ssq[e]);
      if (fr == 0) {
        const int co = 32 * wave + 8 * fg + e;
        a.slab[((size_t)0 * a.slab_rows + blockIdx.x) * SF_CO + co] = s0;
        a.slab[((size_t)1 * a.slab_rows + blockIdx.x) * SF_CO + co] = s1;
      }
    }
    // the slab has the caller's row count (dc_conv_stat_rows); the rows no workgroup owns are zeros
    for (int i = tid; i < 2 * SF_CO; i += 256) {
      const int which = i / SF_CO, co = i % SF_CO;
      for (int rr = blockIdx.x + gridDim.x; rr < a.slab_rows; rr += gridDim.x) a.slab[((size_t)which * a.slab_rows + rr) * SF_CO + co] = 0.f;
    }
  }
}

static int g_sep_fwd = 1;        // tuning switch "sep_fwd": 0 = depthwise and pointwise forward as two operators

int sepfwd_grid(int Cin, int N, int H, int W) {
  const long tiles = (long)N * (H / 8) * (W / (Cin == 128 ? 16 : 32));
  long g = 512;                  // two workgroups per CU
  if (g > tiles) g = tiles;
  return (int)g;
}

}  // namespace

void sep_fwd_set(int v) { g_sep_fwd = v; }

}  // namespace dc

using namespace dc;

// Workgroups (= statistics rows the kernel fills; it zeroes the caller's other rows) of dc_sepconv_fwd; 0: the pair is not served
extern "C" int dc_sepconv_fwd_rows(int dtype, int Cin, int Cout, int stride, int dil, int N, int H, int W) {
  if (!g_sep_fwd || dtype != DC_BF16 || Cout != SF_CO || (Cin != 64 && Cin != 128) || stride != 1 || dil != 1) return 0;
  if (N <= 0 || (H % 8) || (W % (Cin == 128 ? 16 : 32)) || (long)N * H * W < 65536 || (long)N * H * W >= (1L << 31)) return 0;
  return sepfwd_grid(Cin, N, H, W);
}

extern "C" int dc_sepconv_fwd(int dtype, int Cin, int Cout, int N, int H, int W, const void* x, int ldx, const float* pscale, const float* pshift,
                              int prelu, const float* wdw, void* d, int ldd, const void* wf, void* y, int ldy, float* slab, int slab_rows,
                              void* stream) {
  const int grid = dc_sepconv_fwd_rows(dtype, Cin, Cout, 1, 1, N, H, W);
  DC_REQUIRE(grid > 0, "dc_sepconv_fwd: shape not served (dc_sepconv_fwd_rows)");
  DC_REQUIRE(wdw != nullptr && wf != nullptr && ((uintptr_t)wf & 15) == 0 && ((uintptr_t)wdw & 15) == 0, "dc_sepconv_fwd: weights null or unaligned");
  DC_REQUIRE((pscale == nullptr) == (pshift == nullptr), "dc_sepconv_fwd: scale and shift come together");
  DC_REQUIRE(slab == nullptr || slab_rows >= grid, "dc_sepconv_fwd: the statistics slab needs at least dc_sepconv_fwd_rows rows");
  if (int e = dc_check_view(x, ldx, Cin, dtype, "dc_sepconv_fwd x")) return e;
  if (int e = dc_check_view(d, ldd, Cin, dtype, "dc_sepconv_fwd d")) return e;
  if (int e = dc_check_view(y, ldy, Cout, dtype, "dc_sepconv_fwd y")) return e;
  static const void* zero_dev = nullptr;
  static hipError_t init_err = hipSuccess;
  DC_ONCE({
    void* zp = nullptr;
    init_err = hipGetSymbolAddress(&zp, HIP_SYMBOL(sf_zero_page));
    zero_dev = zp;
    if (init_err == hipSuccess) init_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&sepconv_fwd_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, SfCfg<128>::LDS);
    if (init_err == hipSuccess) init_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&sepconv_fwd_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, SfCfg<64>::LDS);
  });
  if (init_err != hipSuccess) return dc_set_error(init_err, __FILE__, __LINE__);
  SepFwdArgs a;
  a.x = (const bf16*)x; a.ldx = ldx; a.pscale = pscale; a.pshift = pshift; a.prelu = prelu; a.wdw = wdw; a.d = (bf16*)d; a.ldd = ldd;
  a.wf = (const bf16*)wf; a.ldw = weight_ld(Cin); a.y = (bf16*)y; a.ldy = ldy; a.slab = slab; a.slab_rows = slab_rows;
  a.N = N; a.H = H; a.W = W; a.ntx = W / (Cin == 128 ? 16 : 32); a.nty = H / 8; a.ntiles = N * a.ntx * a.nty;
  a.zero_page = zero_dev;
  if (Cin == 128) hipLaunchKernelGGL(sepconv_fwd_kernel<128>, dim3(grid), dim3(256), SfCfg<128>::LDS, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(sepconv_fwd_kernel<64>, dim3(grid), dim3(256), SfCfg<64>::LDS, (hipStream_t)stream, a);
  DC_CHECK_LAUNCH();
  return 0;
}
