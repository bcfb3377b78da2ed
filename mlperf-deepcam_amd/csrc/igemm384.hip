// Pointwise (1x1, stride 1) conv forward / data gradient as a GEMM on a 256 (or 128) pixel x 384 channel tile.
//
// Why a third tile shape.  The 728 -> 728 pointwise layers of the middle flow are 27.9 % of the network's FLOPs.  At local batch 8
// (M = 27 648 pixels) the 256 x 256 kernel cuts such a layer into 108 x 3 = 324 tiles for 256 CUs: two rounds, the second one a
// quarter full, and both as long (scripts/fill_bench.hip: the second round of 68 workgroups takes 14.8 us against 17.7 us for a
// full one -- the K loop of a workgroup does not speed up on an emptier chip).  256 x 384 tiles make it 108 x 2 = 216: ONE round,
// and 40 KiB of operands per 32-deep step for 3.1 M MACs (26.7 B/clk at the MFMA rate against 32 for 256 x 256: the LDS-DMA fill
// of a CU tops out at 27-33 B/clk, same benchmark).  128 x 384 tiles give the same 216 tiles at local batch 4.
//
// First version: four waves, one per SIMD, 384 accumulator registers each (a wave owned 128 pixels x 192 channels).  Correct, and
// 13 % faster than two rounds of 256 x 256 (44.1 against 50.0 us), but its K step took ~3 500 cycles for 1 536 cycles of MFMA: an
// LDS-DMA instruction holds the issuing wave until the CU's address path takes it (the fill is at its limit, so the queue is always
// full), and with one wave on the SIMD nothing multiplies meanwhile.  Hence this form:
//
//   waves     8, two per SIMD (waves w and w + 4 share one): GP pixel groups of 64 x GC channel groups of 384 / GC.
//             256 x 384: 4 x 2, a wave owns 64 pixels x 192 channels (4 x 12 MFMA tiles, 192 accumulator registers);
//             128 x 384: 2 x 4, 64 pixels x 96 channels (4 x 6 tiles).
//   K step    32 (64-byte rows, the XOR swizzle of igemm256.hip); ring of 3 stages x (24 KiB weights + 16 / 8 KiB pixels)
//   schedule  per channel block i of a step: the next block's weight fragment (ds_read_b128), one LDS-DMA of the stage two steps
//             ahead, four MFMAs; the pixel fragments of the next step are re-read in place right after their last use (last block).
//             The two waves of a SIMD issue their LDS-DMAs in different halves of the step, so one multiplies while the other is
//             held at an LDS-DMA.  ONE barrier per step, in front of the last channel block: behind it stage s+1 is visible to every wave.
//             LDS reads, MFMAs and their waits are inline assembly: the compiler's s_waitcnt placement waits lgkmcnt(0) in front of
//             every block of MFMAs, i.e. also for the fragment just requested for the NEXT block (one exposed LDS latency per block).
//   hazards   RAW: a wave waits vmcnt(IPW) (its own LDS-DMAs of stage s+1 have landed; the IPW issued during this step stay in
//             flight) before the barrier; stage s+1 is first read behind it.  WAR: slot (s+2) % 3 held stage s-1, whose last reads
//             (the last weight fragment, requested one block before the barrier of step s-1) every wave retired (lgkmcnt(0)) before
//             that barrier; the first DMA into the slot is issued in step s.  Past the last stage the DMA slots are filled from the
//             zero page, so the vmcnt arithmetic is the same in every step.
//   epilogue  from the accumulator registers (the lane-pair exchange of igemm256.hip's register epilogue: 16-byte stores, 64
//             contiguous bytes per pixel and instruction), optional bias / accumulate; BatchNorm partial sums of the STORED values
//             by DPP over the 16 pixel lanes, folded through LDS into the same slab rows (one per 128 pixels) as the other kernels.
// Outputs are bit-identical to the 256 x 256 and 128 x 128 kernels (same MFMA, same K order); the BatchNorm partial sums differ
// in summation order only (scripts/pw384_bench.py).
#include <type_traits>

#include "igemm.h"

namespace dc {

namespace {

constexpr int TN = 384;                // channels per workgroup
constexpr int ROWB = 64;               // bytes of K per row and stage (32 bf16)
constexpr int BK = 32;
constexpr int NST = 3;                 // ring stages (a fourth, i.e. the CU's whole 160 KiB, measured 4 % slower on 256 x 384: 42.3 vs 40.7 us)
constexpr int NPB = 4;                 // pixel blocks of 16 per wave

static __device__ __attribute__((aligned(256))) unsigned char zero_page384[256];
typedef __attribute__((address_space(1))) const void* gas_ptr;
typedef __attribute__((address_space(3))) void* lds_ptr;

__device__ inline int swz64(int row, int slot) { return row * 64 + ((slot ^ ((row >> 1) & 3)) << 4); }
__device__ inline uint32_t swap_rows16(uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F); }
__device__ inline float row_sum16(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));   // row_mirror
  return v;
}

// MFMA as a (volatile) statement: keeps its place between the hand-placed LDS reads and waits below.
__device__ inline void mfma_v(f32x4& c, const bf16x8& av, const bf16x8& bv) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "v"(bv));
}

// LDS fragment reads and their waits are written out: the compiler's own s_waitcnt placement waits lgkmcnt(0) in front of every
// block of MFMAs, i.e. also for the fragment it has just requested for the NEXT block (one exposed LDS latency per block on a
// SIMD that has no second wave to hide it).  Reads issued here are invisible to that pass; every consumer is one of the (volatile)
// MFMA statements above, placed behind a counted wait.
template <int OFF>
__device__ inline void lds_read16(bf16x8& dst, uint32_t addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int N>
__device__ inline void lgkm_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}

template <int I, int N, typename F>
__device__ inline void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// GP: pixel groups, 4 (256 x 384 tile) or 2 (128 x 384).  K64: 128-byte K rows (whole L2 lines per LDS-DMA row piece, 8 rows per
// wave-instruction) in a ring of TWO 64-deep stages, against 64-byte rows in a ring of three 32-deep stages.
template <int GP, bool K64>
struct Cfg {
  static constexpr int GC = 8 / GP;                       // channel groups
  static constexpr int NCB = TN / 16 / GC;                // channel blocks per wave: 12 / 6
  static constexpr int TM = GP * NPB * 16;                // pixels per workgroup: 256 / 128
  static constexpr int RB = K64 ? 128 : 64;               // bytes of K per row and stage
#ifndef DC_PW384_GP2_STAGES
#define DC_PW384_GP2_STAGES 3
#endif
  static constexpr int NSTG = K64 ? 2 : (GP == 2 ? DC_PW384_GP2_STAGES : NST);   // ring stages (128 x 384: 32 KiB each, four fit)
  static constexpr int STAGE = (TN + TM) * RB;            // 40 / 32 KiB (80 KiB: K64)
  static constexpr int NI = STAGE / 1024;                 // LDS-DMA instructions per stage
  static constexpr int IPW = NI / 8;                      // per wave: 5 / 4 (10: K64)
  // first block in which waves 4..7 issue theirs (the last block waits)
  static constexpr int LATE = K64 ? NCB - IPW : NCB - 1 - IPW;
  static constexpr int RING = NSTG * STAGE;
  static constexpr int BLK = 16 * RB;                     // bytes between the fragments of consecutive 16-row blocks
  static_assert(NI % 8 == 0 && IPW <= NCB - 1 && LATE >= 0 && RING <= 160 * 1024, "instruction split");
};

template <int GP, bool K64>
__global__ __launch_bounds__(512) void pw384_kernel(const IgemmParams p) {
  typedef Cfg<GP, K64> K;
  constexpr int KS = K64 ? 64 : 32;          // K elements per stage
  constexpr int BLK = K::BLK;
  constexpr int NCB = K::NCB, GC = K::GC;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const GatherGeom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave / GC;   // pixel group (64 pixels)
  const int wc = wave % GC;    // channel group (NCB blocks of 16)
  const bool late = wave >= 4; // the second wave of its SIMD: issues its LDS-DMAs in the second half of a step
#ifdef DC_PW384_PROBE
  // diagnostic builds (make probes; scripts/pw384_probe.py), one library per COMPILE-TIME mask: bit 0 drops the LDS-DMA issues, bit 1
  // the LDS fragment reads, bit 3 the epilogue stores -- what each costs in the K loop.  Results are garbage by construction.
  constexpr int probe = DC_PW384_PROBE;
#else
  constexpr int probe = 0;
#endif

  // XCD-aware tile order: consecutive tiles of an XCD are the channel tiles of one pixel tile (they share its pixel rows in L2)
  const int ntn = (g.Cout + TN - 1) / TN;
  const int nwg = gridDim.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, xslot = blockIdx.x >> 3;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + xslot;
  const int n0 = (tile % ntn) * TN, m0 = (tile / ntn) * K::TM;
  const int kchunks = (g.Cin + KS - 1) / KS;     // stages

  // ---- LDS-DMA bookkeeping.  Instruction i of this wave fills rows (8i + wave)*16 .. +15 of the stage image: rows [0, 384) are
  // weight rows (channels n0 ..), rows [384, 384 + TM) pixel rows (m0 ..).  A lane's source is its operand row or the zero page
  // (K tail, rows past Cout / M, stages past the last one): a select on the address, never a branch.
  // (K64: 8 rows of 128 bytes per instruction, slot swizzle pslot ^ ((row >> 1) & 7); the row's parity-of-chunk term is wave & 1)
  const int lrow = K64 ? lane >> 3 : lane >> 2, pslot = K64 ? lane & 7 : lane & 3;
  const int lslot = K64 ? pslot ^ ((((wave & 1) << 2) + (lrow >> 1)) & 7)
                        : pslot ^ ((lrow >> 1) & 3);          // logical 16-byte slot this lane fetches (swizzle on the source side)
  constexpr int RPI = K64 ? 8 : 16;                           // rows per instruction
  const uintptr_t zp = (uintptr_t)p.zero_page;
  const bf16* __restrict__ xg = reinterpret_cast<const bf16*>(p.x);
  const bf16* __restrict__ wg = reinterpret_cast<const bf16*>(p.w);
  // a lane's row as a 32-bit byte offset from its operand's base (weights for rows below TN, pixels above: uniform per instruction), ~0u
  // for a row past Cout / M.  (64-bit addresses were 2 x IPW registers: in the 128-byte-row mode, IPW = 10, that pushed the kernel over 256
  // registers and the compiler reloaded them from scratch -- each reload a vmcnt(0) -- inside the K loop, which is what round 3's first
  // "128-byte rows are slower" measurement measured.)
  const uintptr_t xbase = (uintptr_t)xg, wbase = (uintptr_t)wg;
  unsigned src[K::IPW];
#pragma unroll
  for (int i = 0; i < K::IPW; ++i) {
    const int r = (8 * i + wave) * RPI + lrow;
    if (r < TN) {          // wave-uniform per instruction: row groups never straddle the operand boundary
      const int ch = n0 + r;
      src[i] = ch < g.Cout ? (unsigned)(((size_t)ch * p.ldw + lslot * 8) * 2) : ~0u;
    } else {
      const int m = m0 + r - TN;
      src[i] = m < p.M ? (unsigned)(((size_t)m * p.ldx + lslot * 8) * 2) : ~0u;
    }
  }
  auto issue = [&](int i, int stage) {
    const int kofs = stage * KS;
    const bool ok = (src[i] != ~0u) & (kofs + lslot * 8 < g.Cin);          // stage >= kchunks fails the K test: zero page
    const uintptr_t base = (8 * i + wave) * RPI < TN ? wbase : xbase;      // (scalar)
    const uintptr_t a = ok ? base + (src[i] + (unsigned)kofs * 2u) : zp;
    if constexpr (!(probe & 1)) __builtin_amdgcn_global_load_lds((gas_ptr)a, (lds_ptr)(smem + (stage % K::NSTG) * K::STAGE + (8 * i + wave) * 1024), 16, 0, 0);
  };

#ifndef DC_LATE_PRIO
#define DC_LATE_PRIO 1      // the second-dispatched wave of every SIMD loses each arbitration at equal priority (igemm224.hip: 1 965 -> 1 924 cycles per step)
#endif
  if (DC_LATE_PRIO && late) __builtin_amdgcn_s_setprio(DC_LATE_PRIO);
  f32x4 acc[NCB][NPB];   // [channel block][pixel block]
#pragma unroll
  for (int i = 0; i < NCB; ++i)
#pragma unroll
    for (int j = 0; j < NPB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fg = lane >> 4;
  // fragment of block 0 (block i: + i * BLK; the swizzle term depends on (row >> 1) & 3 or & 7, the same for row and row + 16).
  // K64: the fragment of K half h sits at logical slot 4h + fg, physical slot (4h + fg) ^ ((row >> 1) & 7): half 1 = half 0 ^ 64 bytes.
  const int a_off = K64 ? (wc * (NCB * 16) + fr) * 128 + ((fg ^ ((fr >> 1) & 7)) << 4) : swz64(wc * (NCB * 16) + fr, fg);
  const int b_off = TN * K::RB + (K64 ? (grp * (NPB * 16) + fr) * 128 + ((fg ^ ((fr >> 1) & 7)) << 4) : swz64(grp * (NPB * 16) + fr, fg));
  // LDS byte addresses (the kernel's only LDS object is the dynamic array, so its address-space-3 pointer is the offset)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr)smem;
  // Weight fragments are requested TWO channel blocks ahead (a block is only 4 MFMAs, 64 cycles of this wave: one block of distance
  // left every block waiting on its ds_read: 2 700 cycles per step); fa[i % 3] holds block i (NCB is a multiple of 3).
  static_assert(NCB % 3 == 0, "fa ring");
  bf16x8 fa[3], fb[NPB];

  // One 32-deep step: the fragments come from cur_*; the last block requests the next step's fragments from nxt_*.  stage_end: the
  // next step reads ANOTHER ring stage, so every wave first waits for its LDS-DMAs of that stage (VMW instructions may stay in flight)
  // and all meet at a barrier.  dma: which stage to issue LDS-DMAs for during this step (-1: none).
  auto step = [&](uint32_t cur_a, uint32_t nxt_a, uint32_t nxt_b, bool stage_end, int dma, auto vmw_tag) {
    constexpr int VMW = decltype(vmw_tag)::value;
    static_for<0, NCB>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if constexpr (i + 2 < NCB && !(probe & 2)) lds_read16<(i + 2) * BLK>(fa[(i + 2) % 3], cur_a);      // weight fragment two blocks ahead
      // this step's LDS-DMAs: waves 0..3 in blocks 0 .. IPW-1, waves 4..7 in blocks LATE .. LATE+IPW-1
      if constexpr (i < K::IPW) {
        if (!late && dma >= 0) issue(i, dma);
      }
      if constexpr (i >= K::LATE && i < K::LATE + K::IPW) {
        if (late && dma >= 0) issue(i - K::LATE, dma);
      }
      if constexpr (i == NCB - 1) {
        if (stage_end) {
          asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(VMW) : "memory");
          __builtin_amdgcn_s_barrier();
        } else {
          lgkm_wait<0>();
        }
        if constexpr (!(probe & 2)) {
          lds_read16<0>(fa[0], nxt_a);                                                    // next step's first two weight fragments
          lds_read16<BLK>(fa[1], nxt_a);
        }
      }
      static_for<0, NPB>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        // counted waits (LDS reads return in order).  Block 0: fa[0], fa[1], fb[0..], then fa[2] are outstanding -> MFMA j needs all but
        // the newest NPB - j.  Blocks in between: fa[i] and the (up to two) fragments requested after it.
        if constexpr (i == 0) lgkm_wait<NPB - j>();
        else if constexpr (j == 0 && i < NCB - 1) lgkm_wait<(NCB - 1 - i < 2 ? NCB - 1 - i : 2)>();
        mfma_v(acc[i][j], fa[i % 3], fb[j]);
        if constexpr (i == NCB - 1 && !(probe & 2)) lds_read16<j * BLK>(fb[j], nxt_b);    // re-read in place for the next step
      });
    });
  };

  if constexpr (!K64) {
    // ---- prologue: stages 0 and 1 in flight, stage 0 landed, first fragments requested -------------------------------------
#pragma unroll
    for (int q = 0; q < K::NSTG - 1; ++q)
#pragma unroll
      for (int i = 0; i < K::IPW; ++i) issue(i, q);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((K::NSTG - 2) * K::IPW) : "memory");
    __builtin_amdgcn_s_barrier();
    lds_read16<0>(fa[0], lds0 + a_off);
    lds_read16<BLK>(fa[1], lds0 + a_off);
    static_for<0, NPB>([&](auto jc) { lds_read16<decltype(jc)::value * BLK>(fb[decltype(jc)::value], lds0 + b_off); });
    // outstanding LDS reads at the top of every step, oldest first: fa[0], fa[1], fb[0] .. fb[NPB-1]
    for (int s = 0; s < kchunks; ++s) {
      const uint32_t cur = lds0 + (s % K::NSTG) * K::STAGE, nxt = lds0 + ((s + 1) % K::NSTG) * K::STAGE;
      // stage s + NSTG - 1 is issued during step s; stage s+1 must have landed at the end of it
      step(cur + a_off, nxt + a_off, nxt + b_off, true, s + K::NSTG - 1, std::integral_constant<int, (K::NSTG - 2) * K::IPW>{});
    }
  } else {
    // ---- 64-deep stages, ring of two: stage s+1 is issued during the FIRST 32-deep half of stage s (its ring slot was released by the
    // barrier that ended stage s-1) and lands during the second half; nothing else is in flight at the stage's end.
#pragma unroll
    for (int i = 0; i < K::IPW; ++i) issue(i, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    lds_read16<0>(fa[0], lds0 + a_off);
    lds_read16<BLK>(fa[1], lds0 + a_off);
    static_for<0, NPB>([&](auto jc) { lds_read16<decltype(jc)::value * BLK>(fb[decltype(jc)::value], lds0 + b_off); });
    for (int s = 0; s < kchunks; ++s) {
      const uint32_t cur = lds0 + (s & 1) * K::STAGE, nxt = lds0 + ((s + 1) & 1) * K::STAGE;
      step(cur + a_off, cur + (a_off ^ 64), cur + (b_off ^ 64), false, s + 1, std::integral_constant<int, 0>{});
      step(cur + (a_off ^ 64), nxt + a_off, nxt + b_off, true, -1, std::integral_constant<int, 0>{});
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the zero-page fills of the last slots; the ring is reused below
  __builtin_amdgcn_s_barrier();

  // ---- epilogue from the accumulator registers --------------------------------------------------------------------------------
  // A lane holds, per MFMA tile (i, j), channels fg*4 .. +3 of channel block i for pixel fr of pixel block j: 8 bytes.  Lanes l and
  // l ^ 16 trade halves so that the even-fg lane keeps 8 consecutive channels of block i and the odd-fg lane 8 of block i + 1.
  const bool odd = fg & 1;
  bf16* __restrict__ yg = reinterpret_cast<bf16*>(p.y);
  const bool do_stats = p.slab != nullptr;
  float* red = reinterpret_cast<float*>(smem);      // [pixel group][sum, sum of squares][384]
#pragma unroll
  for (int pr = 0; pr < NCB / 2; ++pr) {
    const int i0 = 2 * pr;
    const int chl = wc * (NCB * 16) + (i0 + (odd ? 1 : 0)) * 16 + (fg >> 1) * 8;   // first of this lane's 8 channels after the trade
    const int ch0 = n0 + chl;
    const bool chok = ch0 < g.Cout;                                               // Cout is a multiple of 8: all or nothing
    float ba[4] = {0.f, 0.f, 0.f, 0.f}, bb[4] = {0.f, 0.f, 0.f, 0.f};             // bias of the channels this lane COMPUTED
    if (p.bias != nullptr) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ca = n0 + wc * (NCB * 16) + i0 * 16 + fg * 4 + r, cb = ca + 16;
        if (ca < g.Cout) ba[r] = p.bias[ca];
        if (cb < g.Cout) bb[r] = p.bias[cb];
      }
    }
    float st[2][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) st[0][e] = st[1][e] = 0.f;
#pragma unroll
    for (int j = 0; j < NPB; ++j) {
      const int m = m0 + grp * (NPB * 16) + j * 16 + fr;
      const uint32_t a0 = pack2_bf16(acc[i0][j][0] + ba[0], acc[i0][j][1] + ba[1]);
      const uint32_t a1 = pack2_bf16(acc[i0][j][2] + ba[2], acc[i0][j][3] + ba[3]);
      const uint32_t b0 = pack2_bf16(acc[i0 + 1][j][0] + bb[0], acc[i0 + 1][j][1] + bb[1]);
      const uint32_t b1 = pack2_bf16(acc[i0 + 1][j][2] + bb[2], acc[i0 + 1][j][3] + bb[3]);
      const uint32_t r0 = swap_rows16(odd ? a0 : b0), r1 = swap_rows16(odd ? a1 : b1);
      vec16 v;
      v.w[0] = odd ? r0 : a0;
      v.w[1] = odd ? r1 : a1;
      v.w[2] = odd ? b0 : r0;
      v.w[3] = odd ? b1 : r1;
      if (m < p.M && chok && !(probe & 8)) {
        bf16* dst = yg + (size_t)m * p.ldy + ch0;
        float f[8];
        unpack(v, f, bf16());
        if (p.accumulate) {
          float o[8];
          unpack(ldg16(dst), o, bf16());
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] += o[e];
          pack(v, f, bf16());
          unpack(v, f, bf16());
        }
        stg16(dst, v);
        if (do_stats) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            st[0][e] += f[e];
            st[1][e] = fmaf(f[e], f[e], st[1][e]);
          }
        }
      }
    }
    if (do_stats) {
      // sums over the 16 pixel lanes of a DPP row; lane fr of the row keeps value fr (which = fr >> 3, channel e = fr & 7)
      float mine = 0.f;
#pragma unroll
      for (int w = 0; w < 2; ++w)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float t = row_sum16(st[w][e]);
          if (fr == w * 8 + e) mine = t;
        }
      red[(grp * 2 + (fr >> 3)) * TN + chl + (fr & 7)] = mine;
    }
  }
  if (do_stats) {
    __syncthreads();
    // one slab row per 128 pixels = two pixel groups, folded in a fixed order
    const int rows = p.mtiles;
    for (int i = tid; i < GP * TN; i += 512) {       // GP/2 rows x 2 (sum, sum of squares) x 384
      const int c = i % TN, rw = i / TN;
      const int row = rw >> 1, which = rw & 1;
      const int mt128 = (m0 >> 7) + row;
      if (n0 + c < g.Cout && mt128 < rows)
        p.slab[((size_t)which * rows + mt128) * g.Cout + n0 + c] = red[((2 * row) * 2 + which) * TN + c] + red[((2 * row + 1) * 2 + which) * TN + c];
    }
  }
}

}  // namespace

bool pw384_eligible(const IgemmParams& p) {
  const GatherGeom& g = p.g;
  return g.ntaps == 1 && g.os == 1 && g.is == 1 && g.taps[0].dy == 0 && g.taps[0].dx == 0 && p.m_beg == 0 && p.ngroup <= 1 &&
         g.Cin >= 2 * BK && g.Cin % 8 == 0 && g.Cout % 8 == 0 &&
         (size_t)p.M * p.ldx * 2 < (1ull << 32) && (size_t)g.Cout * p.ldw * 2 < (1ull << 32);      // 32-bit row offsets
}

// npb: 8 (256-pixel tiles) or 4 (128-pixel tiles): pixel blocks of 16 per 128-pixel half, the unit pw384_tiles() counts in;
// 64: 256-pixel tiles with 64-deep stages of 128-byte rows
int launch_pw384(const IgemmParams& p_in, int npb, hipStream_t st) {
  static const void* zero_dev = nullptr;
  static hipError_t init_err = hipSuccess;
  typedef Cfg<4, false> C8;
  typedef Cfg<2, false> C4;
  typedef Cfg<4, true> C64;
  auto k8 = &pw384_kernel<4, false>;
  auto k4 = &pw384_kernel<2, false>;
  auto k64 = &pw384_kernel<4, true>;
  DC_ONCE({
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k8), hipFuncAttributeMaxDynamicSharedMemorySize, C8::RING);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k4), hipFuncAttributeMaxDynamicSharedMemorySize, C4::RING);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k64), hipFuncAttributeMaxDynamicSharedMemorySize, C64::RING);
    void* zp = nullptr;
    init_err = hipGetSymbolAddress(&zp, HIP_SYMBOL(zero_page384));
    zero_dev = zp;
  });
  if (init_err != hipSuccess) return dc_set_error(init_err, __FILE__, __LINE__);
  IgemmParams p = p_in;
  p.zero_page = zero_dev;
  const long tiles = pw384_tiles(p, npb == 64 ? 8 : npb);
  if (npb == 64) hipLaunchKernelGGL(k64, dim3((unsigned)tiles), dim3(512), C64::RING, st, p);
  else if (npb == 8) hipLaunchKernelGGL(k8, dim3((unsigned)tiles), dim3(512), C8::RING, st, p);
  else hipLaunchKernelGGL(k4, dim3((unsigned)tiles), dim3(512), C4::RING, st, p);
  DC_CHECK_LAUNCH();
  return 0;
}

}  // namespace dc
