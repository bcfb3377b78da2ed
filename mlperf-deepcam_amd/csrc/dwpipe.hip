// Persistent, software-pipelined stride-1 depthwise 3x3 (forward, data gradient with or without the BatchNorm sums / this layer's
// weight gradient riding along): the bf16 layers with at least 64 channels on images whose extents are multiples of 8.
//
// dwtile.hip's kernel is a grid of short-lived workgroups: request the halo tile, wait for ALL of it, (transform,) compute, store, exit;
// 1 296 of them on 768 (or, for the 232-register fused data gradient, 512) slots.  On the 728-channel layers it moved 2.4 - 2.9 TB/s
// (serial trace of round 4: 28.5 us forward, 49.9 us fused data gradient) where the tensors alone take 14.5 / 22 us at the copy rate: a
// workgroup's life is a chain of exposed latencies, and the last partial round of the grid runs a third full.  Here ONE 512-thread
// workgroup per CU walks its share of the tiles of ONE channel block:
//   * every input of a tile -- the (8+2d) x (8+2d) halo tile AND, for the data gradient, the BatchNorm input / layer input and the
//     addend at the tile's own pixels -- arrives by LDS-DMA into a ring of 2 - 4 stages, requested NS-1 tiles ahead; the loop holds no
//     VGPR-destination load, so the in-order vmcnt can be counted exactly: "tile i has landed" = all but the youngest
//     (NS-1) x stores + (NS-2) x DMAs of this wave are done (every wave issues the same number of each per tile; lanes without work are
//     out of their buffer resource's range -- zeros land in LDS -- / store nothing, but the instruction is issued);
//   * a thread owns ONE 4-pixel strip of half a channel group per tile: 32 half-groups x 16 strips; its nine taps live in registers for
//     the whole launch;
//   * BatchNorm sums and weight-gradient products stay in registers across ALL tiles of the workgroup and are folded once at the end:
//     one slab row per workgroup (42 rows instead of 432 for a 728-channel layer at local batch 8), so the finalize / fold kernels
//     read a tenth of what they did.
// Per-pixel arithmetic is dwtile.hip's (same taps, same order): dx / y are bit-identical; the sums are added in another order.
// Reference: SeparableConv2d_same.conv1 (deeplab_xception.py:54-66), its backward at train_hdf5_ddp.py:363.
#include "common.h"
#include "dwtile.h"
#include "dwtile_common.h"
#include "bn_fin.h"

namespace dc {

namespace {

constexpr int P_TH = 8, P_TW = 8, P_CG = 16, P_THREADS = 512;
constexpr int P_ROWB = P_CG * 16;   // bytes of one pixel's channel block in LDS
constexpr int P_CH = P_CG * 8;      // channels of a block

// PM_RES (with PM_WGRAD): the layer's stored input x is the output relu(bn(y2) + residual) of a BatchNorm whose backward sums (sum g, sum g*xhat
// with g = dx masked by x > 0) this kernel takes as well -- dx is that BatchNorm's complete output gradient once this kernel, its last
// writer, has added its share to the addend (the first separable conv of an Xception block, deeplab_xception.py:69-122)
enum { PM_FLIP = 1, PM_STATS = 2, PM_WGRAD = 4, PM_ADD = 8, PM_XFORM = 16, PM_RES = 32 };

template <int DIL>
struct PCfg {
  static constexpr int HH = P_TH + 2 * DIL, HW = P_TW + 2 * DIL, HP = HH * HW;
  static constexpr int HIT = (HP * P_CG + P_THREADS - 1) / P_THREADS;   // LDS-DMA instructions per wave for the halo tile
  static constexpr int HALO = HIT * P_THREADS * 16;
  static constexpr int LASTW = (HP * P_CG - (HIT - 1) * P_THREADS + 63) / 64;   // waves with real slots in the last halo instruction (the others skip it)
  static constexpr int TILE = P_TH * P_TW * P_ROWB;                     // own-pixel tile (BatchNorm input / addend): 16 KiB, 2 instructions per wave
};

template <int DIL, int MODE>
struct PStage {
  static constexpr bool Y = (MODE & (PM_STATS | PM_WGRAD)) != 0, A = (MODE & PM_ADD) != 0, Y2 = (MODE & PM_RES) != 0;
  static constexpr int BYTES = PCfg<DIL>::HALO + ((Y ? 1 : 0) + (Y2 ? 1 : 0) + (A ? 1 : 0)) * PCfg<DIL>::TILE;
  static constexpr int DMAS = PCfg<DIL>::HIT + 2 * ((Y ? 1 : 0) + (Y2 ? 1 : 0) + (A ? 1 : 0));
  static constexpr int OFF_Y2 = PCfg<DIL>::HALO + (Y ? PCfg<DIL>::TILE : 0), OFF_A = OFF_Y2 + (Y2 ? PCfg<DIL>::TILE : 0);
  static constexpr int NS = (160 * 1024 / BYTES) > 4 ? 4 : (160 * 1024 / BYTES);
  static constexpr int STORES = DT_PX;    // per thread per tile
  static_assert(NS >= 2, "the ring needs two stages");
  static_assert((NS - 1) * STORES + (NS - 2) * DMAS < 64, "vmcnt immediate");
};

struct DwpArgs {
  const bf16* in;       // x (forward) or dy (data gradient)
  int ldin;
  const float* wp;      // packed taps [9][C]
  const bf16* addend;   // data gradient only: dx = conv + addend (may alias out)
  int ldadd;
  bf16* out;
  int ldout;
  int H, W, C;
  int ncb, ntx, nty, ptiles;
  const float* pscale;  // forward: BatchNorm(+ReLU) of the producer applied to the staged tile
  const float* pshift;
  int prelu;
  DwBnStats st;         // data gradient: BatchNorm sums and / or this layer's weight-gradient rows; st.rows = workgroups per channel block
  const bf16* y2;       // PM_RES: input of the BatchNorm whose (residual-added, ReLU'd) output is this layer's stored input st.y
  int ldy2;
  const float* mean2;
  const float* invstd2;
  int relu2;
  float* slab2;         // [2][st.rows][C]
  int sum_row, sum_row2; // the BatchNorm sums (st.slab / slab2) go to a SUM ROW, double[2][C] (bn_fin.h), by fp64 atomics instead of a row per workgroup
};

#ifdef DWP_STAMPS
static __device__ unsigned long long dwp_stamp_buf[256 * 8 * 4];
#endif
typedef __attribute__((address_space(1))) const void* gas_ptr;
typedef __attribute__((address_space(3))) void* lds_ptr;

template <int N>
__device__ inline void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int DIL, int MODE>
__global__ __launch_bounds__(P_THREADS) void dwp_kernel(const DwpArgs a) {
  typedef PCfg<DIL> K;
  typedef PStage<DIL, MODE> S;
  constexpr bool FLIP = (MODE & PM_FLIP) != 0, STATS = (MODE & PM_STATS) != 0, WGRAD = (MODE & PM_WGRAD) != 0, ADD = (MODE & PM_ADD) != 0,
                 XFORM = (MODE & PM_XFORM) != 0, RES = (MODE & PM_RES) != 0;
  static_assert(!RES || (WGRAD && !STATS), "PM_RES rides on the weight-gradient form");
  constexpr int WC = DT_PX + 2 * DIL, KH = 4, NS = S::NS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cb = blockIdx.x % a.ncb, wj = blockIdx.x / a.ncb, wc = gridDim.x / a.ncb;
  const int t0 = (int)((long)a.ptiles * wj / wc), t1 = (int)((long)a.ptiles * (wj + 1) / wc);
  const int nt = t1 - t0;
  const int ngroups = a.C >> 3, cg0 = cb * P_CG;
  const int H = a.H, W = a.W;
  // compute side: half-group h, strip sl (row sl / 2, columns 4 * (sl & 1) ..)
  const int h = tid & 31, sl = tid >> 5;
  const int row = sl >> 1, xs = (sl & 1) * DT_PX;
  const bool cok = cg0 + (h >> 1) < ngroups;
  const int ch0 = cok ? cg0 * 8 + h * KH : 0;
  // DMA side: this lane's 16-byte slots all belong to channel group g.  Every staged tensor travels as `buffer_load_dwordx4 ... offen lds`
  // through a resource that covers ONE image: a lane's offset is its constant place in the tile (relative to the tile's first pixel, negative
  // in the halo's first rows and columns) plus the tile's origin, formed in a VGPR -- the range check takes the sum as unsigned, so rows above the
  // image (negative) and below it (past the image's bytes) come back as zeros, as do dummy requests past the last tile and channel groups past
  // C (marker offsets); columns left / right of the image are the one case the check cannot see: a lane of the halo's first / last DIL columns is
  // sent out of range when the tile touches that edge.  (Round 5's form computed a 64-bit address, two range tests and a zero-page select per
  // instruction: ~17 vector instructions each, a tenth of the loop's.)
  const int g = tid & 15;
  const bool gok = cg0 + g < ngroups;
  constexpr unsigned OOB = 0x80000000u;                   // an image is smaller than 2 GiB (dw_pipe_rows)
  unsigned hoff[K::HIT];                                  // byte offset of this lane's halo slot from the tile's first pixel (two's complement)
  unsigned hedge = 0;                                     // 2 bits per halo instruction: 1 = in the halo's first DIL columns, 2 = in its last
#pragma unroll
  for (int it = 0; it < K::HIT; ++it) {
    const int hp = (it * P_THREADS + tid) >> 4;
    const int hy = hp / K::HW, hx = hp - hy * K::HW;
    hoff[it] = (gok && hp < K::HP) ? (unsigned)((((hy - DIL) * W + (hx - DIL)) * a.ldin + (cg0 + g) * 8) * 2) : OOB;
    hedge |= ((hx < DIL ? 1u : 0u) | (hx >= K::HW - DIL ? 2u : 0u)) << (2 * it);
  }
  // own-pixel tiles (BatchNorm input, addend, second BatchNorm input): pixel (it * 512 + tid) >> 4 of the 8 x 8 tile, never past an edge
  unsigned yoff[2], aoff[2], y2off[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int pp = (it * P_THREADS + tid) >> 4;
    const int py = pp >> 3, px = pp & 7;
    yoff[it] = gok && S::Y ? (unsigned)(((py * W + px) * a.st.ldy + (cg0 + g) * 8) * 2) : OOB;
    aoff[it] = gok && S::A ? (unsigned)(((py * W + px) * a.ldadd + (cg0 + g) * 8) * 2) : OOB;
    y2off[it] = gok && S::Y2 ? (unsigned)(((py * W + px) * a.ldy2 + (cg0 + g) * 8) * 2) : OOB;
  }

  auto issue = [&](int t, int stage) {
#if defined(__HIP_DEVICE_COMPILE__)     // (the buffer-resource builtins do not exist in the host pass)
    char* sb = smem + stage * S::BYTES;
    const bool live = t < t1;
    const int tt = live ? t : t0;
    const int tx = tt % a.ntx;
    const int r = tt / a.ntx;
    const int ty = r % a.nty, n = r / a.nty;
    const int y0 = ty * P_TH, x0 = tx * P_TW;
    const int live_mask = live ? -1 : 0;                                     // (scalar) a dummy request: a resource of zero bytes, every lane out of range
    const unsigned emask = (x0 == 0 ? 0x55555555u : 0u) | (x0 + P_TW == W ? 0xAAAAAAAAu : 0u);
    {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)n * H * W * a.ldin), 0, (H * W * a.ldin * 2) & live_mask, 0x00020000);
      const unsigned torg = (unsigned)((y0 * W + x0) * a.ldin * 2);
#pragma unroll
      for (int it = 0; it < K::HIT; ++it) {
        if (it == K::HIT - 1 && wv >= K::LASTW) break;      // wave-uniform: this wave's slots of the last instruction lie past the halo tile
        const unsigned vo = ((hedge & emask) >> (2 * it)) & 3u ? OOB : hoff[it] + torg;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(sb + (it * P_THREADS + wv * 64) * 16), 16, vo, 0, 0, 0);
      }
    }
    if constexpr (S::Y) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const bf16*>(a.st.y) + (size_t)n * H * W * a.st.ldy), 0,
                                                                          (H * W * a.st.ldy * 2) & live_mask, 0x00020000);
      const unsigned torg = (unsigned)((y0 * W + x0) * a.st.ldy * 2);
#pragma unroll
      for (int it = 0; it < 2; ++it)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(sb + K::HALO + (it * P_THREADS + wv * 64) * 16), 16, yoff[it] + torg, 0, 0, 0);
    }
    if constexpr (S::A) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.addend + (size_t)n * H * W * a.ldadd), 0, (H * W * a.ldadd * 2) & live_mask, 0x00020000);
      const unsigned torg = (unsigned)((y0 * W + x0) * a.ldadd * 2);
#pragma unroll
      for (int it = 0; it < 2; ++it)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(sb + S::OFF_A + (it * P_THREADS + wv * 64) * 16), 16, aoff[it] + torg, 0, 0, 0);
    }
    if constexpr (S::Y2) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.y2 + (size_t)n * H * W * a.ldy2), 0, (H * W * a.ldy2 * 2) & live_mask, 0x00020000);
      const unsigned torg = (unsigned)((y0 * W + x0) * a.ldy2 * 2);
#pragma unroll
      for (int it = 0; it < 2; ++it)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(sb + S::OFF_Y2 + (it * P_THREADS + wv * 64) * 16), 16, y2off[it] + torg, 0, 0, 0);
    }
#endif
  };

  // ---- prologue: tiles 0 .. NS-2 requested FIRST, so that the taps and coefficients below travel beside them (the compiler's wait for
  // those loads then covers the first tiles as well: one exposed round trip per launch instead of two)
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue(t0 + s, s);

  float wk[9][KH];
  load_taps<KH>(a.wp, ch0, a.C, FLIP, wk);
  BnAcc<KH> bn;
  if constexpr (STATS) bn.init(a.st, ch0);
  else if constexpr (WGRAD) bn.init_affine(a.st, ch0);
  float dwa[WGRAD ? 9 : 1][KH];
#pragma unroll
  for (int t9 = 0; t9 < (WGRAD ? 9 : 1); ++t9)
#pragma unroll
    for (int e = 0; e < KH; ++e) dwa[t9][e] = 0.f;
  float mu2[RES ? KH : 1], is2[RES ? KH : 1], ra[RES ? KH : 1], rb[RES ? KH : 1];
  if constexpr (RES) {
#pragma unroll
    for (int e = 0; e < KH; ++e) {
      mu2[e] = a.mean2[ch0 + e];
      is2[e] = a.invstd2[ch0 + e];
      ra[e] = rb[e] = 0.f;
      asm volatile("" : "+v"(mu2[e]), "+v"(is2[e]));
    }
  }
  float xsc[8], xsh[8];   // XFORM: the coefficients of this lane's channel group (the transform pass walks the lane's DMA slots)
  if constexpr (XFORM) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      xsc[e] = gok ? a.pscale[(cg0 + g) * 8 + e] : 0.f;
      xsh[e] = gok ? a.pshift[(cg0 + g) * 8 + e] : 0.f;
    }
  }
  // every value loaded so far is made "used" here, so that the compiler's own wait for these loads sits in front of the loop and none
  // of its scoreboard entries survives into it (inside the loop the only vector-memory operations are LDS-DMAs and stores, counted by hand)
#pragma unroll
  for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
    for (int e = 0; e < KH; ++e) asm volatile("" : "+v"(wk[t9][e]));
  if constexpr (STATS || WGRAD) {
#pragma unroll
    for (int e = 0; e < KH; ++e) asm volatile("" : "+v"(bn.mu[e]), "+v"(bn.is[e]), "+v"(bn.ms[e]), "+v"(bn.mh[e]));
  }
  if constexpr (XFORM) {
#pragma unroll
    for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(xsc[e]), "+v"(xsh[e]));
  }



#ifdef DWP_STAMPS
  unsigned long long tw = 0, ti = 0, tx_ = 0, tc = 0, tprev = __builtin_amdgcn_s_memtime();
#define STAMP(acc) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc += now_ - tprev; tprev = now_; } while (0)
#else
#define STAMP(acc) do {} while (0)
#endif
  auto decode = [&](int t, int& n, int& y0, int& x0) {
    const int tx = t % a.ntx;
    const int r = t / a.ntx;
    const int ty = r % a.nty;
    n = r / a.nty;
    y0 = ty * P_TH;
    x0 = tx * P_TW;
  };
  // v = act(x * scale + shift) in place on a landed halo tile, once per staged element; the zero padding (out of the image) stays zero
  auto xform_tile = [&](int t, int stg) {
    char* sb = smem + stg * S::BYTES;
    int n, y0, x0;
    decode(t, n, y0, x0);
    (void)n;
#pragma unroll
      for (int it = 0; it < K::HIT; ++it) {
        if (it == K::HIT - 1 && wv >= K::LASTW) break;
        const int hp = (it * P_THREADS + tid) >> 4;
        const int hy = hp / K::HW, hx = hp - hy * K::HW;
        const int iy = y0 - DIL + hy, ix = x0 - DIL + hx;
        if (gok && hp < K::HP && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
          vec16* q = reinterpret_cast<vec16*>(sb) + it * P_THREADS + tid;
          float f[8];
          unpack(*q, f, bf16());
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float v = fmaf(f[e], xsc[e], xsh[e]);
            f[e] = a.prelu ? fmaxf(v, 0.f) : v;
          }
          vec16 o;
          pack(o, f, bf16());
          // the store goes through inline assembly: a compiler-visible LDS store into memory that LDS-DMAs write makes hipcc wait
          // vmcnt(0) in front of it, i.e. for the prefetched tiles as well
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          u32x4 ov;
          ov.x = o.w[0]; ov.y = o.w[1]; ov.z = o.w[2]; ov.w = o.w[3];
          const unsigned qa = (unsigned)(size_t)((__attribute__((address_space(3))) char*)reinterpret_cast<char*>(q));
          asm volatile("ds_write_b128 %0, %1" ::"v"(qa), "v"(ov) : "memory");
        }
      }
  };
  auto compute_tile = [&](int t, int stg) {
    char* sb = smem + stg * S::BYTES;
    int n, y0, x0;
    decode(t, n, y0, x0);
    // ---- one strip: output row `row`, columns xs .. xs+3
    const char* tile = sb + h * 8;
    const int oy = y0 + row;
    float yf[DT_PX][KH];
    if constexpr (S::Y) {
#pragma unroll
      for (int j = 0; j < DT_PX; ++j)
        unpack8(*reinterpret_cast<const vec8*>(sb + K::HALO + (row * P_TW + xs + j) * P_ROWB + h * 8), yf[j], bf16());
    }
    float xh[WGRAD ? DT_PX : 1][KH];
    if constexpr (WGRAD) {
      // the forward input of this layer at the strip's pixels: act(y*ms + mh) rounded to bf16 exactly as the forward kernel staged it
#pragma unroll
      for (int j = 0; j < DT_PX; ++j) {
        float v[KH];
#pragma unroll
        for (int e = 0; e < KH; ++e) {
          const float u = fmaf(yf[j][e], bn.ms[e], bn.mh[e]);
          v[e] = a.st.relu ? fmaxf(u, 0.f) : u;
        }
        vec8 rr;
        pack8(rr, v, bf16());
        unpack8(rr, xh[j], bf16());
      }
    }
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
        unpack8(*reinterpret_cast<const vec8*>(tile + ((row + ky * DIL) * K::HW + xs + c) * P_ROWB), f, bf16());
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int j = c - kx * DIL;
          if (j >= 0 && j < DT_PX) {
#pragma unroll
            for (int e = 0; e < KH; ++e) acc[j][e] = fmaf(f[e], wk[ky * 3 + kx][e], acc[j][e]);
            if constexpr (WGRAD) {
#pragma unroll
              for (int e = 0; e < KH; ++e) dwa[8 - (ky * 3 + kx)][e] = fmaf(f[e], xh[j][e], dwa[8 - (ky * 3 + kx)][e]);
            }
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < DT_PX; ++j) {
      if constexpr (ADD) {
        float av[KH];
        unpack8(*reinterpret_cast<const vec8*>(sb + S::OFF_A + (row * P_TW + xs + j) * P_ROWB + h * 8), av, bf16());
#pragma unroll
        for (int e = 0; e < KH; ++e) acc[j][e] += av[e];
      }
      vec8 v;
      pack8(v, acc[j], bf16());
      // (exactly STORES store instructions per wave per tile: the `cok` mask never empties a wave, every wave holds all 32 half-groups)
      if (cok) *reinterpret_cast<vec8*>(a.out + (((size_t)n * H + oy) * W + x0 + xs + j) * a.ldout + ch0) = v;
      if constexpr (STATS) {
        float gs[KH];
        unpack8(v, gs, bf16());
        bn.add(gs, yf[j], a.st.relu);
      }
      if constexpr (RES) {
        // g = the stored dx masked by the stored block output (yf = x here); xhat from the producing BatchNorm's input
        float gs[KH], y2f[KH];
        unpack8(v, gs, bf16());
        unpack8(*reinterpret_cast<const vec8*>(sb + S::OFF_Y2 + (row * P_TW + xs + j) * P_ROWB + h * 8), y2f, bf16());
#pragma unroll
        for (int e = 0; e < KH; ++e) {
          const float gm = (!a.relu2 || yf[j][e] > 0.f) ? gs[e] : 0.f;
          ra[e] += gm;
          rb[e] = fmaf(gm, (y2f[e] - mu2[e]) * is2[e], rb[e]);
        }
      }
    }
  };
  int stage = 0;
  if constexpr (!XFORM) {
    for (int i = 0; i < nt; ++i) {
      // tile i has landed once all but the youngest [stores of the last min(i, NS-1) tiles + DMAs of tiles i+1 .. i+NS-2] are done
      if (wv < K::LASTW) {
        if (i >= NS - 1) wait_vm<(NS - 1) * S::STORES + (NS - 2) * S::DMAS>();
        else if (i == 0) wait_vm<(NS - 2) * S::DMAS>();
        else if (i == 1) wait_vm<S::STORES + (NS - 2) * S::DMAS>();
        else wait_vm<2 * S::STORES + (NS - 2) * S::DMAS>();
      } else {          // one LDS-DMA fewer per tile
        if (i >= NS - 1) wait_vm<(NS - 1) * S::STORES + (NS - 2) * (S::DMAS - 1)>();
        else if (i == 0) wait_vm<(NS - 2) * (S::DMAS - 1)>();
        else if (i == 1) wait_vm<S::STORES + (NS - 2) * (S::DMAS - 1)>();
        else wait_vm<2 * S::STORES + (NS - 2) * (S::DMAS - 1)>();
      }
      STAMP(tx_);                       // (stamp builds, non-XFORM modes: the vmcnt wait alone goes to the "transform" column)
      __builtin_amdgcn_s_barrier();     // everybody's pieces of tile i have landed; everybody is done with the stage tile i-1 used
      STAMP(tw);
      {
        int ns = stage + NS - 1;
        if (ns >= NS) ns -= NS;
        issue(t0 + i + NS - 1, ns);     // into the stage tile i-1 has just left
      }
      STAMP(ti);
      compute_tile(t0 + i, stage);
      STAMP(tc);
      if (++stage == NS) stage = 0;
    }
  } else {
    // Forward on a lazily applied BatchNorm: the in-place transform of tile i+1 runs in the same barrier interval as the stencil of tile i
    // (different stages), so a tile costs ONE workgroup barrier and the transform's LDS round trips hide among the stencil's.  Tile i+1
    // must therefore have landed at the top of iteration i: all but the youngest [stores of the last min(i, NS-2) tiles + DMAs of tiles
    // i+2 .. i+NS-2].
    static_assert(!XFORM || NS >= 3, "the transform one tile ahead needs three stages");
    constexpr int D0 = S::DMAS, D1 = S::DMAS - 1;
    if (wv < K::LASTW) wait_vm<(NS - 2) * D0>(); else wait_vm<(NS - 2) * D1>();     // tile 0
    __builtin_amdgcn_s_barrier();
    xform_tile(t0, 0);
    for (int i = 0; i < nt; ++i) {
      if (wv < K::LASTW) {
        if (i >= NS - 2) wait_vm<(NS - 2) * S::STORES + (NS - 3) * D0>();
        else if (i == 0) wait_vm<(NS - 3) * D0>();
        else wait_vm<S::STORES + (NS - 3) * D0>();
      } else {
        if (i >= NS - 2) wait_vm<(NS - 2) * S::STORES + (NS - 3) * D1>();
        else if (i == 0) wait_vm<(NS - 3) * D1>();
        else wait_vm<S::STORES + (NS - 3) * D1>();
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this thread's transform stores of tile i
      __builtin_amdgcn_s_barrier();     // tile i is transformed, tile i+1 has landed, everybody is done with the stage tile i-1 used
      STAMP(tw);
      {
        int ns = stage + NS - 1;
        if (ns >= NS) ns -= NS;
        issue(t0 + i + NS - 1, ns);
      }
      STAMP(ti);
      if (i + 1 < nt) {
        int s1 = stage + 1;
        if (s1 == NS) s1 = 0;
        xform_tile(t0 + i + 1, s1);
      }
      STAMP(tx_);
      compute_tile(t0 + i, stage);
      STAMP(tc);
      if (++stage == NS) stage = 0;
    }
  }
#ifdef DWP_STAMPS
  if ((tid & 63) == 0 && blockIdx.x < 256) {
    unsigned long long* o = dwp_stamp_buf + ((size_t)blockIdx.x * 8 + wv) * 4;
    o[0] = tw; o[1] = ti; o[2] = tx_; o[3] = tc;
  }
#endif
  // the ring drains (dummy requests past the last tile included) before its memory becomes the fold's scratch
  wait_vm<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // ---- one slab row per workgroup: fold the 16 strip lanes through LDS in a fixed order
  if constexpr (STATS) {
    float* red = reinterpret_cast<float*>(smem);   // [16][2][P_CH]
#pragma unroll
    for (int e = 0; e < KH; ++e) {
      red[(sl * 2 + 0) * P_CH + h * KH + e] = bn.a[e];
      red[(sl * 2 + 1) * P_CH + h * KH + e] = bn.b[e];
    }
    __syncthreads();
    if (tid < 2 * P_CH) {
      const int which = tid / P_CH, cl = tid % P_CH;
      const int c = cg0 * 8 + cl;
      if (c < a.C) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) s += red[(q * 2 + which) * P_CH + cl];
        if (a.sum_row) unsafeAtomicAdd(reinterpret_cast<double*>(a.st.slab) + (size_t)which * a.C + c, (double)s);
        else a.st.slab[((size_t)which * a.st.rows + wj) * a.C + c] = s;
      }
    }
    __syncthreads();
  }
  if constexpr (RES) {
    float* red = reinterpret_cast<float*>(smem);   // [16][2][P_CH]
#pragma unroll
    for (int e = 0; e < KH; ++e) {
      red[(sl * 2 + 0) * P_CH + h * KH + e] = ra[e];
      red[(sl * 2 + 1) * P_CH + h * KH + e] = rb[e];
    }
    __syncthreads();
    if (tid < 2 * P_CH) {
      const int which = tid / P_CH, cl = tid % P_CH;
      const int c = cg0 * 8 + cl;
      if (c < a.C) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) s += red[(q * 2 + which) * P_CH + cl];
        if (a.sum_row2) unsafeAtomicAdd(reinterpret_cast<double*>(a.slab2) + (size_t)which * a.C + c, (double)s);
        else a.slab2[((size_t)which * a.st.rows + wj) * a.C + c] = s;
      }
    }
    __syncthreads();
  }
  if constexpr (WGRAD) {
    float* red = reinterpret_cast<float*>(smem);   // [16][9][P_CH]
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
      for (int e = 0; e < KH; ++e) red[(sl * 9 + tp) * P_CH + h * KH + e] = dwa[tp][e];
    __syncthreads();
    for (int i = tid; i < 9 * P_CH; i += P_THREADS) {
      const int tp = i / P_CH, cl = i % P_CH;
      const int c = cg0 * 8 + cl;
      if (c < a.C) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) s += red[(q * 9 + tp) * P_CH + cl];
        a.st.wslab[((size_t)wj * 9 + tp) * a.C + c] = s;
      }
    }
  }
}

int g_dw_pipe = 1;   // tuning switch "dw_pipe": 0 = dwtile.hip's kernels everywhere, 1 = data gradients here, 2 = the forward pass too

template <int DIL, int MODE>
int launch1(const DwpArgs& a, int grid, hipStream_t st) {
  constexpr int LDS = PStage<DIL, MODE>::NS * PStage<DIL, MODE>::BYTES;
  static_assert(LDS <= 160 * 1024, "ring larger than the LDS");
  static_assert(16 * 9 * P_CH * 4 <= LDS, "the fold's scratch lives in the drained ring");
  DC_ONCE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dwp_kernel<DIL, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  hipLaunchKernelGGL((dwp_kernel<DIL, MODE>), dim3(grid), dim3(P_THREADS), LDS, st, a);
  DC_CHECK_LAUNCH();
  return 0;
}

}  // namespace

void dw_pipe_set(int v) { g_dw_pipe = v < 0 ? 0 : v > 2 ? 2 : v; }
bool dw_pipe_forward() { return g_dw_pipe == 2; }

#ifdef DWP_STAMPS
// diagnostic builds only (make dwstamps; scripts/dw_stamps.py): the phase cycle sums the last launch left, [256 workgroups][8 waves][4]
extern "C" int dc_debug_dwp_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dwp_stamp_buf), sizeof(unsigned long long) * 256 * 8 * 4);
}
#endif

// workgroups per channel block (= slab rows of the sums that ride along), 0: the shape is not served
int dw_pipe_rows(int dtype, int C, int dil, int N, int H, int W) {
  // (64 channels -- block 1's first depthwise layer, 226 MB of gradient at local batch 8 -- fill half of a 128-channel block: the idle lanes
  // cost nothing beside one pass over dy that also takes the BatchNorm sums and the weight gradient)
  if (!g_dw_pipe || dtype != DC_BF16 || C < P_CH / 2 || (C & 7) || (H % P_TH) || (W % P_TW) || (dil != 1 && dil != 2)) return 0;
  if ((long)H * W * ((C + 63) / 64 * 64 + 64) * 2 >= (1L << 30)) return 0;      // an image per buffer resource: offsets and marker below 2 GiB
  const int ncb = cdiv(C / 8, P_CG);
  const long ptiles = (long)N * (H / P_TH) * (W / P_TW);
  long wc = 256 / ncb;               // one workgroup per CU
  if (wc < 1) wc = 1;
  if (wc > ptiles) wc = ptiles;
  // the launch takes as long as its busiest workgroup: keep that number of tiles and use the fewest workgroups that reach it (432 tiles on
  // 42 workgroups are 11 rounds; so are 40 workgroups, and the other CUs stay free for the weight-gradient stream)
  const long rounds = (ptiles + wc - 1) / wc;
  wc = (ptiles + rounds - 1) / rounds;
  return (int)wc;
}

int launch_dw_pipe(int dil, bool flip, const void* in, int ldin, const float* wp, const void* addend, int ldadd, void* out, int ldout, int N,
                   int H, int W, int C, hipStream_t st, const float* pscale, const float* pshift, int prelu, const DwBnStats* bnstats,
                   const DwResStats* res) {
  const int wc = dw_pipe_rows(DC_BF16, C, dil, N, H, W);
  DC_REQUIRE(wc > 0, "launch_dw_pipe: shape not served");
  DwpArgs a;
  a.in = (const bf16*)in; a.ldin = ldin; a.wp = wp; a.addend = (const bf16*)addend; a.ldadd = ldadd; a.out = (bf16*)out; a.ldout = ldout;
  a.H = H; a.W = W; a.C = C;
  a.ncb = cdiv(C / 8, P_CG); a.ntx = W / P_TW; a.nty = H / P_TH; a.ptiles = N * a.ntx * a.nty;
  a.pscale = pscale; a.pshift = pshift; a.prelu = prelu;
  if (bnstats != nullptr) a.st = *bnstats;
  else { a.st.slab = nullptr; a.st.y = nullptr; a.st.ldy = 0; a.st.mean = a.st.invstd = a.st.mscale = a.st.mshift = nullptr; a.st.relu = 0; a.st.rows = 0; a.st.wslab = nullptr; }
  a.sum_row = bnstats != nullptr && bnstats->rows == SUM_ROW && bnstats->slab != nullptr;
  a.sum_row2 = res != nullptr && res->sum_row;
  a.st.rows = wc;
  a.y2 = nullptr; a.ldy2 = 0; a.mean2 = a.invstd2 = nullptr; a.relu2 = 0; a.slab2 = nullptr;
  if (res != nullptr) {
    a.y2 = (const bf16*)res->y; a.ldy2 = res->ldy; a.mean2 = res->mean; a.invstd2 = res->invstd; a.relu2 = res->relu; a.slab2 = res->slab;
  }
  const int grid = a.ncb * wc;
  const bool stats = flip && a.st.slab != nullptr, wg = flip && a.st.wslab != nullptr, add = flip && addend != nullptr;
  const bool xf = !flip && pscale != nullptr;
  DC_REQUIRE(!(stats || wg) || a.st.y != nullptr, "launch_dw_pipe: the sums need the BatchNorm input");
  if (res != nullptr) {
    DC_REQUIRE(wg && !stats && a.y2 && a.mean2 && a.invstd2 && a.slab2, "launch_dw_pipe: the residual BatchNorm's sums ride on the weight-gradient form");
    DC_REQUIRE(dil == 1, "launch_dw_pipe: the residual form is built for dilation 1 (with dilation 2 its ring would hold a single stage)");
    if (add) return launch1<1, PM_FLIP | PM_WGRAD | PM_ADD | PM_RES>(a, grid, st);
    return launch1<1, PM_FLIP | PM_WGRAD | PM_RES>(a, grid, st);
  }
#define DWP(D, M) return launch1<D, M>(a, grid, st)
#define DWP_MODES(D)                                                                                      \
  if (!flip) { if (xf) DWP(D, PM_XFORM); else DWP(D, 0); }                                               \
  if (stats && wg) { if (add) DWP(D, PM_FLIP | PM_STATS | PM_WGRAD | PM_ADD); else DWP(D, PM_FLIP | PM_STATS | PM_WGRAD); } \
  if (stats) { if (add) DWP(D, PM_FLIP | PM_STATS | PM_ADD); else DWP(D, PM_FLIP | PM_STATS); }         \
  if (wg) { if (add) DWP(D, PM_FLIP | PM_WGRAD | PM_ADD); else DWP(D, PM_FLIP | PM_WGRAD); }            \
  if (add) DWP(D, PM_FLIP | PM_ADD); else DWP(D, PM_FLIP);
  if (dil == 1) { DWP_MODES(1) }
  DWP_MODES(2)
#undef DWP_MODES
#undef DWP
}

}  // namespace dc
