// EXPERIMENTS ONLY (make experiments): wgrad256.hip as it stood in round 4 with its compile-time variants -- the 32x32x16 MFMA form (DC_WG256_MFMA32),
// the five-stage ring (DC_WG256_STAGES) and the probe masks (DC_WG256_PROBE) -- none of which is part of the product library.  DESIGN section 5 has the numbers.
// Dense-conv weight gradient, 256 output channels x 256 input channels of one tap per 512-thread workgroup (bf16).
//
//     dW_t[co][ci] = sum over output pixels m of dy[m][co] * x[gather(m,t)][ci]
//
// Same reasoning as igemm256.hip: the 128 x 128 kernel (wgrad.hip) is bound by the global->LDS fill, and a 256 x 256 tile
// needs half the operand bytes per flop.  Structure:
//   waves     8 = 2 co halves x 4 ci quarters; a wave owns 128 co x 64 ci (8 x 4 MFMA tiles, 128 accumulator registers)
//   stage     32 pixels: a dy tile [32 px][256 co] and an x tile [32 px][256 ci], 16 KiB each, rows stay pixel-major (coalesced
//             NHWC loads by LDS-DMA), fragments come out channel-per-lane through ds_read_b64_tr_b16 with the XOR swizzle of
//             wgrad.hip on the 32-byte chunk index (bank pattern unchanged: a 512-byte row is two 256-byte bank rows)
//   ring      4 stages = 128 KiB, three in flight, counted vmcnt, ONE workgroup barrier per stage, every LDS read and DMA issue
//             in the shadow of the stage's 32 MFMAs, the two waves of a SIMD in opposite load/MFMA order (igemm256.hip)
//   split-K   the pixel axis is cut so that about one workgroup per CU exists; partial tiles go to the fp32 slab
//             [split][tap][Co][Ci] and wgrad_reduce_kernel (wgrad.hip) adds them in a fixed order
#include <type_traits>

#include "../wgrad.h"

namespace dc {

namespace {

constexpr int WT = 256;            // tile edge in channels
constexpr int WBP = 32;            // pixels per stage
constexpr int WROW = WT * 2;       // bytes per LDS row (one pixel, 256 channels)
constexpr int WOPER = WBP * WROW;  // 16 KiB
constexpr int WSTAGE = 2 * WOPER;
#ifndef DC_WG256_STAGES
#define DC_WG256_STAGES 4
#endif
constexpr int WNST = DC_WG256_STAGES;   // ring stages of 32 KiB (5 = the CU's whole LDS)
__device__ inline int ring(int s) { return (WNST & (WNST - 1)) == 0 ? (s & (WNST - 1)) : s % WNST; }
#ifndef DC_WG256_PROBE
#define DC_WG256_PROBE 0     // diagnostic builds only (scripts/wgrad256_probe.py): 1 = no LDS-DMA, 2 = no LDS fragment reads, 4 = L2-resident operands, 8 = no slab stores
#endif

static __device__ __attribute__((aligned(256))) unsigned char wg256_zero_page[256];
typedef __attribute__((address_space(1))) const void* gas_ptr;
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef __attribute__((address_space(3))) short4v lds_s4;

// Up to WG_MAXGROUP layers of identical geometry share one launch (dc_conv_wgrad_group): the pixel axis of each is then cut
// into fewer, longer splits (the 728 -> 728 pointwise layers at B = 8: 9 tiles x 28 splits of 31 stages each alone,
// 27 tiles x 9 splits of 96 stages as a group of three), which divides both the prologue/epilogue share of a workgroup and
// the fp32 slab traffic (256 KiB per workgroup, written once and read once by the reduction) by the group size.
struct Wgrad256Params {
  WgradParams w;           // w.x / w.dy / w.slab belong to layer 0
  const void* zero_page;
  const void* xs[WG_MAXGROUP - 1];    // layers 1..: the same geometry, strides and split plan
  const void* dys[WG_MAXGROUP - 1];
  float* slabs[WG_MAXGROUP - 1];
};

#ifndef DC_WG256_MFMA32
#define DC_WG256_MFMA32 0    // 1: v_mfma_f32_32x32x16_bf16 (half the MFMA instructions per stage: an MFMA holds the SIMD's issue port for 8 cycles
#endif                       //    whatever its shape, and this loop also issues 24 transposing LDS reads and 4 LDS-DMAs per wave and stage)
constexpr bool M32 = DC_WG256_MFMA32 != 0;
// XOR key on the 32-byte chunk index of a 512-byte pixel row.  16x16x32 fragments: a half-wave reads 8 rows x 32 B (rows 8g + 0..3 of two
// k-blocks): keys 0..7.  32x32x16 fragments: a half-wave reads 4 rows x 64 B (two adjacent chunks): the key moves whole chunk PAIRS.
__device__ inline int swz_key(int row) { return M32 ? ((row & 3) << 1) : ((row & 3) | (((row >> 3) & 1) << 2)); }

// one MFMA operand fragment (16 channels of chunk `c`, the 8 pixels 8*fg .. 8*fg+7 of the stage) from a [pixel][channel] tile
__device__ inline vec16 tr_frag(const char* tile, int c, int prow, int key0, int key1, int tp) {
  if (DC_WG256_PROBE & 2) {
    vec16 f;
    f.w[0] = f.w[1] = f.w[2] = f.w[3] = 0x3f803f80u + (unsigned)c;
    asm volatile("" : "+v"(f.w[0]), "+v"(f.w[1]), "+v"(f.w[2]), "+v"(f.w[3]));
    return f;
  }
  const short4v a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(tile + prow * WROW + ((c ^ key0) << 5) + 8 * tp));
  const short4v a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(tile + (prow + 4) * WROW + ((c ^ key1) << 5) + 8 * tp));
  const uint2 t0 = __builtin_bit_cast(uint2, a0), t1 = __builtin_bit_cast(uint2, a1);
  vec16 f;
  f.w[0] = t0.x; f.w[1] = t0.y; f.w[2] = t1.x; f.w[3] = t1.y;
  return f;
}

// 32x32x16 operand fragment: 32 channels of block `cb` x the 16 pixels of k-half `kh` of the stage; lane l holds channel 32*cb + (l & 31),
// pixels 16*kh + 8*(l >> 5) + 0..7 (two transposing reads of 4 pixels x 16 channels per 16-lane group)
__device__ inline vec16 tr_frag32(const char* tile, int cb, int kh, int lane) {
  if (DC_WG256_PROBE & 2) {
    vec16 f;
    f.w[0] = f.w[1] = f.w[2] = f.w[3] = 0x3f803f80u + (unsigned)cb;
    asm volatile("" : "+v"(f.w[0]), "+v"(f.w[1]), "+v"(f.w[2]), "+v"(f.w[3]));
    return f;
  }
  const int i = lane & 15, q = i >> 2, p = i & 3;
  const int row = 16 * kh + 8 * (lane >> 5) + q;
  const int chunk = 2 * cb + ((lane >> 4) & 1);
  const short4v a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(tile + row * WROW + ((chunk ^ swz_key(row)) << 5) + 8 * p));
  const short4v a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(tile + (row + 4) * WROW + ((chunk ^ swz_key(row + 4)) << 5) + 8 * p));
  const uint2 t0 = __builtin_bit_cast(uint2, a0), t1 = __builtin_bit_cast(uint2, a1);
  vec16 f;
  f.w[0] = t0.x; f.w[1] = t0.y; f.w[2] = t1.x; f.w[3] = t1.y;
  return f;
}

typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ __launch_bounds__(512) void wgrad256_kernel(const Wgrad256Params pp_) {
  const WgradParams& p = pp_.w;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const GatherGeom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wq = wave >> 2, wp = wave & 3;   // co half (also the load/MFMA order group), ci quarter

  // XCD-aware order (see wgrad.hip): consecutive tiles cover all (ci, co) tiles and then all taps of ONE pixel split
  const int nci = (g.Cin + WT - 1) / WT, nco = (g.Cout + WT - 1) / WT;
  const int nwg = gridDim.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, xslot = blockIdx.x >> 3;
  int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + xslot;
  const int ci0 = (tile % nci) * WT;
  tile /= nci;
  const int co0 = (tile % nco) * WT;
  tile /= nco;
  const int tapi = tile % g.ntaps;
  tile /= g.ntaps;
  const int split = tile % p.splits, layer = tile / p.splits;   // layer: wave-uniform index into the group
  const Tap tap = g.taps[tapi];
  const int py = tap.phase / g.os, px = tap.phase % g.os;
  const int mbeg = split * p.chunk;
  const int mend = min(p.M, mbeg + p.chunk);
  const int steps = mend > mbeg ? (mend - mbeg + WBP - 1) / WBP : 0;
  // a chain of scalar selects, not an indexed read: indexing the by-value argument struct would move it to scratch memory
  const void* xsel = p.x;
  const void* dsel = p.dy;
  float* ssel = p.slab;
#pragma unroll
  for (int l = 1; l < WG_MAXGROUP; ++l)
    if (layer == l) {
      xsel = pp_.xs[l - 1];
      dsel = pp_.dys[l - 1];
      ssel = pp_.slabs[l - 1];
    }
  const bf16* __restrict__ xg = reinterpret_cast<const bf16*>(xsel);
  const bf16* __restrict__ dg = reinterpret_cast<const bf16*>(dsel);
  const uintptr_t zp = (uintptr_t)pp_.zero_page;

  // ---- DMA bookkeeping: instruction j = 2*wave + i of a stage fills pixel rows 2j, 2j+1 (1 KiB); lane -> (row, 16-byte slot).
  // The pixel coordinates of the lane's two rows are decomposed once and advanced by 32 pixels per stage with carries.
  const int lrow = lane >> 5, sp = lane & 31;
  int rn[2], rqy[2], rqx[2], rlslot[2];
  size_t offq[2], offp[2];
  bool cq_ok[2], cp_ok[2];
  auto recompute = [&](int i) {
    const int n = rn[i], qy = rqy[i], qx = rqx[i];
    const int oy = qy * g.os + py, ox = qx * g.os + px;
    offq[i] = ((size_t)(n * g.Hout + oy) * g.Wout + ox) * p.lddy + co0 + rlslot[i] * 8;
    const int iy = qy * g.is + tap.dy, ix = qx * g.is + tap.dx;
    // (the x offset may be "out of the image" for halo taps: it is only dereferenced when the bounds test passes)
    offp[i] = (size_t)((long)((long)(n * g.Hin + iy) * g.Win + ix) * p.ldx + ci0 + rlslot[i] * 8);
  };
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 2 * (2 * wave + i) + lrow;
    rlslot[i] = (((sp >> 1) ^ swz_key(row)) << 1) | (sp & 1);
    const int m = mbeg + row;
    const int mm = m < p.M ? m : 0;
    rn[i] = fast_div(mm, g.div_hw);
    const int rem = mm - rn[i] * (g.Qh * g.Qw);
    rqy[i] = fast_div(rem, g.div_w);
    rqx[i] = rem - rqy[i] * g.Qw;
    cq_ok[i] = co0 + rlslot[i] * 8 < g.Cout;
    cp_ok[i] = ci0 + rlslot[i] * 8 < g.Cin;
    recompute(i);
  }
  const size_t stepq = (size_t)WBP * g.os * p.lddy, stepp = (size_t)WBP * g.is * p.ldx;
  int istage = 0;   // stage whose DMA is issued next (dy half first, x half second, then the walk advances)
  auto pick = [&](bool ok, const bf16* a) { return (gas_ptr)(ok ? (uintptr_t)a : zp); };
  auto issue_q = [&](int slot) {
    char* q = smem + slot * WSTAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int j = 2 * wave + i;
      const int m = mbeg + istage * WBP + 2 * j + lrow;
      if (!(DC_WG256_PROBE & 1)) __builtin_amdgcn_global_load_lds(pick((m < mend) & cq_ok[i], dg + offq[i]), (lds_ptr)(q + j * 1024), 16, 0, 0);
    }
  };
  auto issue_p = [&](int slot) {
    char* x_ = smem + slot * WSTAGE + WOPER;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int j = 2 * wave + i;
      const int m = mbeg + istage * WBP + 2 * j + lrow;
      const int iy = rqy[i] * g.is + tap.dy, ix = rqx[i] * g.is + tap.dx;
      const bool ok = (m < mend) & cp_ok[i] & ((unsigned)iy < (unsigned)g.Hin) & ((unsigned)ix < (unsigned)g.Win);
      if (!(DC_WG256_PROBE & 1)) __builtin_amdgcn_global_load_lds(pick(ok, xg + offp[i]), (lds_ptr)(x_ + j * 1024), 16, 0, 0);
      // advance this row by one stage
      if (DC_WG256_PROBE & 4) continue;   // diagnostic: every stage re-reads the first stage's (then L2-resident) rows
      rqx[i] += WBP;
      if (rqx[i] < g.Qw) {
        offq[i] += stepq;
        offp[i] += stepp;
      } else {
        while (rqx[i] >= g.Qw) {
          rqx[i] -= g.Qw;
          if (++rqy[i] == g.Qh) {
            rqy[i] = 0;
            ++rn[i];
          }
        }
        recompute(i);
      }
    }
    ++istage;
  };

  if constexpr (M32) {
    // ---- 32x32x16 form: wave tile 128 co x 64 ci = 4 x 2 blocks of 32 x 32; a stage is two k-halves of 16 pixels, 8 MFMAs each
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int q = 0; q < WNST - 1; ++q) {
      issue_q(q);
      issue_p(q);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (WNST - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    vec16 a0[4], b0[2], a1[4], b1[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) a0[i] = tr_frag32(smem, wq * 4 + i, 0, lane);
#pragma unroll
    for (int j = 0; j < 2; ++j) b0[j] = tr_frag32(smem + WOPER, wp * 2 + j, 0, lane);
    // H0(s): a1/b1 <- k-half 1 of stage s; DMA dy tile of stage s+3; 8 MFMAs on (a0, b0).  H1(s): a0/b0 <- k-half 0 of stage s+1; DMA x tile
    // of stage s+3; 8 MFMAs on (a1, b1).  Same hazard argument as the 16x16x32 loop below (one barrier per stage, between the halves).
    auto k_loop = [&](auto loads_first_tag) {
      constexpr bool LOADS_FIRST = decltype(loads_first_tag)::value;
      for (int s = 0; s < steps; ++s) {
        const char* q = smem + ring(s) * WSTAGE;
        const char* q1 = smem + ring(s + 1) * WSTAGE;
        const int nslot = ring(s + WNST - 1);
        auto load1 = [&]() {
#pragma unroll
          for (int i = 0; i < 4; ++i) a1[i] = tr_frag32(q, wq * 4 + i, 1, lane);
#pragma unroll
          for (int j = 0; j < 2; ++j) b1[j] = tr_frag32(q + WOPER, wp * 2 + j, 1, lane);
        };
        auto load0 = [&]() {
#pragma unroll
          for (int i = 0; i < 4; ++i) a0[i] = tr_frag32(q1, wq * 4 + i, 0, lane);
#pragma unroll
          for (int j = 0; j < 2; ++j) b0[j] = tr_frag32(q1 + WOPER, wp * 2 + j, 0, lane);
        };
        if constexpr (LOADS_FIRST) {
          load1();
          issue_q(nslot);
          __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0[i]), __builtin_bit_cast(bf16x8, b0[j]), acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if constexpr (!LOADS_FIRST) {
          __builtin_amdgcn_sched_barrier(0);
          load1();
          issue_q(nslot);
        }
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 * (WNST - 3) + 2) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (LOADS_FIRST) {
          load0();
          issue_p(nslot);
          __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1[i]), __builtin_bit_cast(bf16x8, b1[j]), acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if constexpr (!LOADS_FIRST) {
          __builtin_amdgcn_sched_barrier(0);
          load0();
          issue_p(nslot);
        }
      }
    };
    if (wq == 0) k_loop(std::true_type{});
    else k_loop(std::false_type{});
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    float* out = ssel + ((size_t)split * g.ntaps + tap.widx) * g.Cout * g.Cin;
    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wq * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (co >= g.Cout) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int ci = ci0 + wp * 64 + j * 32 + l31;
          if (ci < g.Cin && (!(DC_WG256_PROBE & 8) || acc[i][j][r] == 123.f)) out[(size_t)co * g.Cin + ci] = acc[i][j][r];
        }
      }
    return;
  }
  f32x4 acc[8][4];   // [co block][ci block]
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fg = lane >> 4;
  const int tq = (lane & 15) >> 2, tp = lane & 3;
  const int prow = 8 * fg + tq;
  const int key0 = swz_key(prow), key1 = swz_key(prow + 4);

  // ---- prologue: stages 0..2 in flight (rows past the split's end come from the zero page), stage 0 landed
#pragma unroll
  for (int q = 0; q < WNST - 1; ++q) {
    issue_q(q);
    issue_p(q);
  }
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (WNST - 2)) : "memory");   // stage 0 landed, stages 1 .. WNST-2 in flight
  __builtin_amdgcn_s_barrier();
  vec16 fa_lo[4], fa_hi[4], fb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) fa_lo[i] = tr_frag(smem, wq * 8 + i, prow, key0, key1, tp);
#pragma unroll
  for (int j = 0; j < 4; ++j) fb[j] = tr_frag(smem + WOPER, wp * 4 + j, prow, key0, key1, tp);

  // Software-pipelined loop, one barrier per stage (hazard argument: igemm256.hip, with "weight half" = dy tile, "pixel half" =
  // x tile).  M0(s): fa_hi <- co blocks 4..7 of stage s; DMA dy tile of stage s+3; 16 MFMAs on (fa_lo, fb).
  // M1(s): fa_lo <- stage s+1; DMA x tile of stage s+3; 16 MFMAs on (fa_hi, fb), each fb[j] re-read from stage s+1 after its last use.
  auto k_loop = [&](auto loads_first_tag) {
    constexpr bool LOADS_FIRST = decltype(loads_first_tag)::value;
    for (int s = 0; s < steps; ++s) {
      const char* q = smem + ring(s) * WSTAGE;
      const char* q1 = smem + ring(s + 1) * WSTAGE;
      const char* x1 = q1 + WOPER;
      const int nslot = ring(s + WNST - 1);
      // ---- M0
      if constexpr (LOADS_FIRST) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fa_hi[i] = tr_frag(q, wq * 8 + 4 + i, prow, key0, key1, tp);
        issue_q(nslot);
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa_lo[i]), __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      if constexpr (!LOADS_FIRST) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa_hi[i] = tr_frag(q, wq * 8 + 4 + i, prow, key0, key1, tp);
        issue_q(nslot);
      }
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 * (WNST - 3) + 2) : "memory");   // stage s+1 landed
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---- M1
      if constexpr (LOADS_FIRST) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fa_lo[i] = tr_frag(q1, wq * 8 + i, prow, key0, key1, tp);
        issue_p(nslot);
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa_hi[i]), __builtin_bit_cast(bf16x8, fb[j]), acc[4 + i][j], 0, 0, 0);
        fb[j] = tr_frag(x1, wp * 4 + j, prow, key0, key1, tp);
      }
      __builtin_amdgcn_s_setprio(0);
      if constexpr (!LOADS_FIRST) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa_lo[i] = tr_frag(q1, wq * 8 + i, prow, key0, key1, tp);
        issue_p(nslot);
      }
    }
  };
  if (wq == 0) k_loop(std::true_type{});
  else k_loop(std::false_type{});
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the fills of the last three slots

  float* out = ssel + ((size_t)split * g.ntaps + tap.widx) * g.Cout * g.Cin;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + wq * 128 + i * 16 + fg * 4 + r;
      if (co >= g.Cout) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ci = ci0 + wp * 64 + j * 16 + fr;
        if (ci < g.Cin && (!(DC_WG256_PROBE & 8) || acc[i][j][r] == 123.f)) out[(size_t)co * g.Cin + ci] = acc[i][j][r];
      }
    }
}

}  // namespace

// About one workgroup per CU (g_wgrad256_slots of them); every split at least g_wgrad256_min_stages stages long.  `group` layers share the launch.
static int g_wgrad256_slots = 192;   // 64 CUs stay free for the HBM-bound kernels of the backward chain (step -0.3 ms vs 256)
// A workgroup pays its ring fill, its 256 KiB fp32 slab tile and the fold's read of it whatever its share of the pixels: below ~100 stages
// (3 072 pixels) that fixed cost wins.  At local batch 2 / 4 the 728-channel groups were cut into 7 splits of 31 / 62 stages; now 3 / 5 splits
// of 72 / 87 stages: 13.88 -> 13.47 ms and 21.50 -> 20.92 ms per step (profiles/r04_ab_step_2.txt); local batch 8 keeps its 7 splits of 123.
static int g_wgrad256_min_stages = 96;
void wgrad256_set_slots(int n) { g_wgrad256_slots = n; }
void wgrad256_set_min_stages(int n) { g_wgrad256_min_stages = n < 4 ? 4 : n; }
void wgrad256_plan(const GatherGeom& g, long M, int* splits, int* chunk, int group) {
  const long tiles = (long)cdiv(g.Cin, WT) * cdiv(g.Cout, WT) * g.ntaps * group;
  long want = g_wgrad256_slots / tiles;
  const long per = (long)g_wgrad256_min_stages * WBP;
  const long maxs = (M + per - 1) / per;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  long c = (M + want - 1) / want;
  c = (c + WBP - 1) / WBP * WBP;
  *chunk = (int)c;
  *splits = (int)((M + c - 1) / c);
}

int launch_wgrad256(const WgradParams& p, hipStream_t st, int group, const void* const* xs, const void* const* dys, float* const* slabs) {
  const size_t lds = (size_t)WNST * WSTAGE;
  static const void* zero_dev = nullptr;
  static hipError_t init_err = hipSuccess;
  DC_ONCE({
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad256_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    void* zp = nullptr;
    init_err = hipGetSymbolAddress(&zp, HIP_SYMBOL(wg256_zero_page));
    zero_dev = zp;
  });
  if (init_err != hipSuccess) return dc_set_error(init_err, __FILE__, __LINE__);
  Wgrad256Params pp;
  pp.w = p;
  pp.zero_page = zero_dev;
  if (group < 1 || group > WG_MAXGROUP) return dc_fail("launch_wgrad256: group size out of range", __FILE__, __LINE__);
  for (int l = 1; l < WG_MAXGROUP; ++l) {
    pp.xs[l - 1] = l < group ? xs[l] : nullptr;
    pp.dys[l - 1] = l < group ? dys[l] : nullptr;
    pp.slabs[l - 1] = l < group ? slabs[l] : nullptr;
  }
  const long blocks = (long)cdiv(p.g.Cin, WT) * cdiv(p.g.Cout, WT) * p.g.ntaps * p.splits * group;
  hipLaunchKernelGGL(wgrad256_kernel, dim3((unsigned)blocks), dim3(512), lds, st, pp);
  DC_CHECK_LAUNCH();
  return 0;
}

}  // namespace dc
