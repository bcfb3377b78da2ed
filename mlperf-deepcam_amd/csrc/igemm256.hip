// Gather-form implicit GEMM, 256 output pixels x 256 output channels per 512-thread workgroup (bf16).
//
// Why a second tile shape: in-kernel stamps on the 128 x 128 kernel (scripts/igemm_stamps.py) show its K loop advancing at
// ~1480 cycles per 32-deep step with three workgroups per CU, i.e. the CU takes in 16 KiB of operands per ~490 cycles
// (~33 B/clk) while its MFMA pipes are busy half of that time: the loop is bound by the global->LDS fill, and the bytes to
// fill per MFMA fall with the tile edge.  A 256 x 256 tile needs half the operand bytes per flop.
//
//   waves     8 = 2 pixel halves x 4 channel quarters; a wave owns 128 pixels x 64 channels
//             (8 x 4 MFMA tiles of 16 x 16, 128 accumulator registers), so 12 ds_read_b128 feed 32 MFMAs
//   K step    32 (64-byte rows, the XOR swizzle of igemm.hip); ring of 4 stages x (16 KiB weights + 16 KiB pixels) = 128 KiB
//   schedule  software-pipelined, ONE workgroup barrier per stage (see the loop): every ds_read and LDS-DMA issue sits in the
//             shadow of the 32 MFMAs of the stage, so the two waves of a SIMD stream MFMAs back to back.  (A first version
//             alternated load and MFMA segments between two staggered wave groups with four barriers per stage: stamps
//             showed ~370-470-cycle load segments against 330-cycle MFMA segments, 54 % MFMA duty; this form reads the
//             next operands while the current ones are multiplied.)
//   hazards   RAW/WAR argument at the loop.  Past the last stage the DMA slots are filled from the zero page so that the vmcnt
//             arithmetic is the same in every iteration.
//   epilogue  as igemm.hip: accumulators -> LDS C tile (256 rows x 528 B) -> coalesced 16-byte stores, optional bias /
//             accumulate, BatchNorm partial sums of the stored values, one slab row per 128-pixel half (same slab shape as the
//             128-tile kernel, so the two are interchangeable per layer).
#include <type_traits>

#include "igemm.h"

namespace dc {

namespace {

constexpr int TM = 256, TN = 256;      // pixels, channels per workgroup
constexpr int ROWB = 64;               // bytes of K per row and stage (32 bf16)
constexpr int BK = 32;
constexpr int NSTAGE = 4;
constexpr int OPER = TM * ROWB;        // 16 KiB per operand per stage
constexpr int STAGE = 2 * OPER;
constexpr int RING = NSTAGE * STAGE;   // 128 KiB
constexpr int CROW = TN * 2 + 16;      // padded C-tile row (bytes)
constexpr int CTILE = TM * CROW;       // 132 KiB
constexpr int MAIN_BYTES = RING > CTILE ? RING : CTILE;

static __device__ __attribute__((aligned(256))) unsigned char zero_page256[256];
// the address travels in the kernel arguments (an SGPR pair): reading the symbol in the loop costs a GOT load + s_waitcnt per use
__device__ inline const void* zero_page256_ptr(const IgemmParams& p) { return p.zero_page; }
typedef __attribute__((address_space(1))) const void* gas_ptr;
typedef __attribute__((address_space(3))) void* lds_ptr;

// Diagnostic build only (-DDC_STAMPS): per-segment cycle sums of the K loop for one wave of each group (scripts/igemm_stamps.py).
#ifdef DC_STAMPS
__device__ unsigned long long* dc_stamp_buf256 = nullptr;
#define SEG_T() __builtin_amdgcn_s_memtime()
#define SEG_ADD(k, t0, t1) seg[k] += (t1) - (t0)
#else
#define SEG_T() 0ull
#define SEG_ADD(k, t0, t1)
#endif

__device__ inline int swz64(int row, int slot) { return row * 64 + ((slot ^ ((row >> 1) & 3)) << 4); }

// ---- helpers of the register epilogue ---------------------------------------------------------------------------------------
// lane ^ 16 exchange (the two 16-lane rows of a 32-lane half swap): ds_swizzle in bit-mask mode, no LDS memory involved
__device__ inline uint32_t swap_rows16(uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F); }
// sum over the 16 lanes of a DPP row (xor 1, xor 2, half mirror, mirror): every lane ends with the same bits
__device__ inline float row_sum16(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));   // row_mirror
  return v;
}

__global__ __launch_bounds__(512) void igemm256_kernel(const IgemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* s_tap = reinterpret_cast<int*>(smem + MAIN_BYTES);
  const GatherGeom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  [[maybe_unused]] const unsigned long long t_start = SEG_T();
  const int grp = wave >> 2;   // pixel half; also the stagger group
  const int wc = wave & 3;     // channel quarter

  // XCD-aware tile order (see igemm.hip): consecutive tiles of an XCD walk the channel tiles of one pixel tile first
  const int ntn = (g.Cout + TN - 1) / TN;
  const int mt256 = (p.M - p.m_beg + TM - 1) / TM;
  const int nwg = gridDim.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, xslot = blockIdx.x >> 3;
  int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + xslot;
  // split-K launch: the splits of the tile grid are the slowest index (neighbours keep sharing their operand rows as without splits)
  const int nsplit = p.ksplit > 1 ? p.ksplit : 1;
  const int tiles_per_split = nwg / nsplit;
  const int split = tile / tiles_per_split;
  tile -= split * tiles_per_split;
  const int tile_lin = tile;
  const int ntile_n = tile % ntn;
  int rest = tile / ntn;
  // grouped launch: the members of one pixel tile are neighbours in the tile order (they read the same input rows through one
  // XCD's L2).  `member` is wave-uniform; the member's pointers and dilation come out of scalar selects (indexing the by-value
  // argument struct would move it to scratch memory).
  const int ngroup = p.ngroup > 1 ? p.ngroup : 1;
  const int member = rest % ngroup;
  rest /= ngroup;
  const void* wsel = p.w;
  void* ysel = p.y;
  float* slabsel = p.slab;
  int dilmul = p.ngroup > 1 ? p.gdil[0] : 1;
#pragma unroll
  for (int b = 1; b < IgemmParams::MAXGROUP; ++b)
    if (member == b) {
      wsel = p.gw[b - 1];
      ysel = p.gy[b - 1];
      slabsel = p.gslab[b - 1];
      dilmul = p.gdil[b];
    }
  // phase fastest: the sub-pixel phases of a transposed conv have K loops of different length (1/2/2/4 taps), so neighbouring
  // workgroups finish at different times (their write bursts interleave with the others' K loops), and the four phases of one
  // pixel tile read the same input rows through one XCD's L2
  const int nph = g.os * g.os;
  const int phase = p.phase_fast ? rest % nph : rest / mt256;
  const int mtile = p.phase_fast ? rest / nph : rest % mt256;
  const int n0 = ntile_n * TN, m0 = p.m_beg + mtile * TM;
  const int py = phase / g.os, px = phase % g.os;

  const int tap_beg = g.phase_beg[phase], ntap = g.phase_beg[phase + 1] - tap_beg;
  if (tid < ntap) {
    const Tap tp = g.taps[tap_beg + tid];
    s_tap[3 * tid] = tp.dy * dilmul;
    s_tap[3 * tid + 1] = tp.dx * dilmul;
    s_tap[3 * tid + 2] = tp.widx;
  }
  __syncthreads();
  const int kchunks_all = (g.Cin + BK - 1) / BK;
  // split-K: this workgroup's share of the channel-chunk groups (KG chunks each; the K loop's outermost level)
  constexpr int KG = 4;
  const int ngk = (kchunks_all + KG - 1) / KG;
  const int kfirst = nsplit > 1 ? (ngk * split / nsplit) * KG : 0;
  const int kchunks = nsplit > 1 ? min(kchunks_all, (ngk * (split + 1) / nsplit) * KG) : kchunks_all;      // END chunk of this workgroup's range
  const int steps = ntap * (kchunks - kfirst);
  if (steps == 0 && p.accumulate) return;   // a phase without taps contributes zeros

  // ---- per-thread DMA bookkeeping: 2 rows of each operand (instruction j of wave w fills rows (8j + w)*16 .. +15) -------
  const int lrow = lane >> 2, pslot = lane & 3;
  const int lslot = pslot ^ ((lrow >> 1) & 3);          // logical 16-byte slot this lane fetches (swizzle on the source side)
  const bf16* __restrict__ xg = reinterpret_cast<const bf16*>(p.x);
  const bf16* __restrict__ wg = reinterpret_cast<const bf16*>(wsel);
  int rown[2], riy[2], rix[2];
  bool rok[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = m0 + (8 * j + wave) * 16 + lrow;
    rok[j] = m < p.M;
    const int mm = rok[j] ? m : 0;
    int n, qy, qx;
    grid_pixel(g, mm, n, qy, qx);
    rown[j] = n * g.Hin;
    riy[j] = qy * g.is;
    rix[j] = qx * g.is;
  }
  const bf16* xrow[2] = {xg, xg};
  const bf16* wrow[2] = {wg, wg};
  bool xok[2] = {false, false}, wok[2] = {false, false};
  int tapA = -1, tapB = -1;
  // Every lane's source is either its operand row or the zero page (K tail, image border, rows past M / Cout, and the dummy
  // fills past the last stage); the choice is a select on the address, never a branch: the L segments must stay short.
  const uintptr_t zp = (uintptr_t)zero_page256_ptr(p);
  auto pick = [&](bool ok, const bf16* a) { return (gas_ptr)(ok ? (uintptr_t)a : zp); };
  // weight half of a stage: 2 LDS-DMA instructions per wave
  auto issue_A = [&](int tapi, int kc, int slot, bool live) {
    if (live && tapi != tapA) {
      tapA = tapi;
      const int widx = s_tap[3 * tapi + 2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ch = n0 + (8 * j + wave) * 16 + lrow;
        wok[j] = ch < g.Cout;
        wrow[j] = wg + (((size_t)widx * g.Cout + (wok[j] ? ch : 0)) * p.ldw + lslot * 8);
      }
    }
    const int kofs = kc * BK;
    const bool kok = live & (kofs + lslot * 8 < g.Cin);
    char* base = smem + slot * STAGE;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds(pick(kok & wok[j], wrow[j] + kofs), (lds_ptr)(base + (8 * j + wave) * 16 * ROWB), 16, 0, 0);
  };
  // pixel half
  auto issue_B = [&](int tapi, int kc, int slot, bool live) {
    if (live && tapi != tapB) {
      tapB = tapi;
      const int dy = s_tap[3 * tapi], dx = s_tap[3 * tapi + 1];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int iy = riy[j] + dy, ix = rix[j] + dx;
        xok[j] = rok[j] && (unsigned)iy < (unsigned)g.Hin && (unsigned)ix < (unsigned)g.Win;
        xrow[j] = xg + (((size_t)(rown[j] + (xok[j] ? iy : 0)) * g.Win + (xok[j] ? ix : 0)) * p.ldx + lslot * 8);
      }
    }
    const int kofs = kc * BK;
    const bool kok = live & (kofs + lslot * 8 < g.Cin);
    char* base = smem + slot * STAGE + OPER;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds(pick(kok & xok[j], xrow[j] + kofs), (lds_ptr)(base + (8 * j + wave) * 16 * ROWB), 16, 0, 0);
  };

  f32x4 acc[4][8];   // [channel block][pixel block]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fg = lane >> 4;

  // ---- prologue: stages 0..2 in flight, stage 0 landed ---------------------------------------------------------------
  int itap = 0, ikc = kfirst;   // (tap, K chunk) of the next stage to issue
  // K order: groups of KG channel chunks OUTER, taps in the middle, the group's chunks inner.  With the taps outermost, a workgroup
  // swept its 256 pixels x ALL channels once per tap; the CUs of an XCD together pull more than its 4 MiB L2 through per sweep, so
  // each of the 9 taps of a 3 x 3 layer fetched the (shifted) input again from the fabric: 3.5 x the tensor on the 192 x 288 decoder
  // convolutions, 7 x on the atrous ASPP ones (FETCH_SIZE per launch, scripts/fetch_by_grid.py).  Taps innermost (every step another
  // tap) brings that down to 1.0-1.3 x but recomputes the per-tap row pointers and bounds every step: 9-24 % slower.  Groups of four
  // chunks (256 bytes per pixel; ~2.6 MB per XCD and tap sweep) keep the nine sweeps of a group in L2 and change tap every 4th step.
  int kbeg = kfirst, kend = kchunks < kfirst + KG ? kchunks : kfirst + KG;
  auto advance = [&]() {
    if (++ikc == kend) {
      ikc = kbeg;
      if (++itap == ntap) {
        itap = 0;
        kbeg = kend;
        kend = kend + KG < kchunks ? kend + KG : kchunks;
        ikc = kbeg;
      }
    }
  };
#pragma unroll
  for (int q = 0; q < NSTAGE - 1; ++q) {
    const bool live = q < steps;
    issue_A(itap, ikc, q, live);
    issue_B(itap, ikc, q, live);
    advance();
  }
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // stage 0 has landed (stages 1, 2 = 8 instructions stay in flight)
  __builtin_amdgcn_s_barrier();

  // Software-pipelined loop, ONE barrier per stage.  All LDS reads and LDS-DMA issues sit in the shadow of MFMAs:
  //   M0(s)  fb2 <- pixel blocks 4..7 of stage s;  DMA weight half of stage s+3;  16 MFMAs on (fa, fb)
  //          lgkmcnt(0) (every read of stage s-1 and the fb2 reads retired), vmcnt(6) (stage s+1 landed), barrier
  //   M1(s)  fb <- pixel blocks 0..3 of stage s+1;  DMA pixel half of stage s+3;  16 MFMAs on (fa, fb2), each weight
  //          fragment fa[i] re-read from stage s+1 right after its last use
  // RAW: stage s+1 is read only after the barrier that follows every wave's vmcnt for it.  WAR: slot (s+3)%4 held stage s-1,
  // whose last reads (fb2 in M0(s-1)) every wave retired before the barrier of stage s-1; the first DMA into it is issued after
  // that barrier.  vmcnt(6): at the wait the queue holds stages s+1 (4), s+2 (4) and the weight half of s+3 (2).
  vec16 fa[4], fb[4], fb2[4];
  {
    const char* wa = smem;
    const char* xb = wa + OPER;
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const vec16*>(wa + swz64(wc * 64 + i * 16 + fr, fg));
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const vec16*>(xb + swz64(grp * 128 + j * 16 + fr, fg));
  }
#ifdef DC_STAMPS
  unsigned long long seg[6] = {0, 0, 0, 0, 0, 0};
  const unsigned long long t_loop0 = SEG_T();
#endif
  // The two waves of a SIMD (w and w+4, i.e. the two pixel halves) run the halves of every segment in opposite order: while
  // one issues its reads and LDS-DMAs (which hold a wave's issue for 60-185 cycles apiece) the other issues MFMAs.  The
  // whole loop is instantiated twice (a branch inside the loop body made the register allocator spill).
  auto k_loop = [&](auto loads_first_tag) {
    constexpr bool LOADS_FIRST = decltype(loads_first_tag)::value;
    for (int s = 0; s < steps; ++s) {
      [[maybe_unused]] const unsigned long long t0 = SEG_T();
      const char* xb = smem + (s & (NSTAGE - 1)) * STAGE + OPER;
      const char* wa1 = smem + ((s + 1) & (NSTAGE - 1)) * STAGE;
      const char* xb1 = wa1 + OPER;
      const int nslot = (s + NSTAGE - 1) & (NSTAGE - 1);
      const bool live = s + NSTAGE - 1 < steps;
      // ---- M0
      if constexpr (LOADS_FIRST) {
#pragma unroll
        for (int j = 0; j < 4; ++j) fb2[j] = *reinterpret_cast<const vec16*>(xb + swz64(grp * 128 + (4 + j) * 16 + fr, fg));
        issue_A(itap, ikc, nslot, live);
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      if constexpr (!LOADS_FIRST) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb2[j] = *reinterpret_cast<const vec16*>(xb + swz64(grp * 128 + (4 + j) * 16 + fr, fg));
        issue_A(itap, ikc, nslot, live);
      }
      [[maybe_unused]] const unsigned long long t1 = SEG_T();
      asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
      [[maybe_unused]] const unsigned long long t2 = SEG_T();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      [[maybe_unused]] const unsigned long long t3 = SEG_T();
      // ---- M1
      if constexpr (LOADS_FIRST) {
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const vec16*>(xb1 + swz64(grp * 128 + j * 16 + fr, fg));
        issue_B(itap, ikc, nslot, live);
        advance();
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb2[j]), acc[i][4 + j], 0, 0, 0);
        fa[i] = *reinterpret_cast<const vec16*>(wa1 + swz64(wc * 64 + i * 16 + fr, fg));
      }
      __builtin_amdgcn_s_setprio(0);
      if constexpr (!LOADS_FIRST) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const vec16*>(xb1 + swz64(grp * 128 + j * 16 + fr, fg));
        issue_B(itap, ikc, nslot, live);
        advance();
      }
      [[maybe_unused]] const unsigned long long t4 = SEG_T();
      SEG_ADD(0, t0, t1); SEG_ADD(1, t1, t2); SEG_ADD(2, t2, t3); SEG_ADD(3, t3, t4);
    }
  };
  if (grp == 0) k_loop(std::true_type{});
  else k_loop(std::false_type{});
#ifdef DC_STAMPS
  if (dc_stamp_buf256 != nullptr && lane == 0 && wc == 0) {
    unsigned long long* o = dc_stamp_buf256 + ((size_t)blockIdx.x * 2 + grp) * 8;
    for (int k = 0; k < 4; ++k) o[k] = seg[k];
    o[4] = t_loop0 - t_start;      // prologue: tap table, row bookkeeping, three stages issued, stage 0 landed
    o[6] = SEG_T() - t_loop0;
    o[7] = (unsigned long long)steps;
  }
#endif
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the zero-page fills of the last three slots
  __builtin_amdgcn_s_barrier();

  // ---- split-K: the fp32 partial tile goes to this split's slab as it sits in the accumulators (64 contiguous bytes per pixel and
  // instruction); igemm256_splitk_fold_kernel does what the epilogue below does for an unsplit launch
  if (nsplit > 1) {
    float* part = p.kslab + ((size_t)split * tiles_per_split + tile_lin) * (TM * TN);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int chl = wc * 64 + i * 16 + fg * 4;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int prow = grp * 128 + j * 16 + fr;
        *reinterpret_cast<f32x4*>(part + prow * TN + chl) = acc[i][j];
      }
    }
    return;
  }

  // ---- epilogue ------------------------------------------------------------------------------------
  char* ct = smem;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int chl = wc * 64 + i * 16 + fg * 4;   // first of this lane's 4 channels, tile-local
    float b4[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n0 + chl + r < g.Cout) b4[r] = p.bias[n0 + chl + r];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int prow = grp * 128 + j * 16 + fr;
      uint2 v;
      v.x = pack2_bf16(acc[i][j][0] + b4[0], acc[i][j][1] + b4[1]);
      v.y = pack2_bf16(acc[i][j][2] + b4[2], acc[i][j][3] + b4[3]);
      *reinterpret_cast<uint2*>(ct + prow * CROW + chl * 2) = v;
    }
  }
  __syncthreads();

  constexpr int GPR = TN * 2 / 16;     // 32 sixteen-byte groups per C row
  constexpr int RPP = 512 / GPR;       // 16 rows per pass
  constexpr int PASSES = TM / RPP;     // 16; passes 0..7 cover the first 128-pixel half
  const int cgrp = tid % GPR, rsub = tid / GPR;
  const int ch0 = n0 + cgrp * 8;
  const bool chok = ch0 < g.Cout;
  float ssum[2][8], ssq[2][8];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int e = 0; e < 8; ++e) ssum[h][e] = ssq[h][e] = 0.f;
  bf16* __restrict__ yg = reinterpret_cast<bf16*>(ysel);
  const bool do_stats = slabsel != nullptr;
  // BatchNorm-backward sums instead of (sum, sum of squares): this thread's channels are fixed, their vectors live in registers
  const bool bwd_stats = p.bst.y != nullptr;
  const bf16* __restrict__ by = reinterpret_cast<const bf16*>(p.bst.y);
  float bmu[8], bis[8], bms[8], bmh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const bool ok = bwd_stats && ch0 + e < g.Cout;
    bmu[e] = ok ? p.bst.mean[ch0 + e] : 0.f;
    bis[e] = ok ? p.bst.invstd[ch0 + e] : 0.f;
    bms[e] = (ok && p.bst.relu) ? p.bst.mscale[ch0 + e] : 0.f;
    bmh[e] = (ok && p.bst.relu) ? p.bst.mshift[ch0 + e] : 0.f;
  }
  auto out_pixel = [&](int m) -> size_t {
    if (g.os == 1) return (size_t)m;
    int n, qy, qx;
    grid_pixel(g, m, n, qy, qx);
    return (size_t)(n * g.Hout + qy * g.os + py) * g.Wout + qx * g.os + px;
  };
  if (do_stats && bwd_stats) {
    // BatchNorm-backward form of the store loop.  The BatchNorm inputs of a half (8 rows per thread) are requested together in front
    // of it: one load in flight per thread (512 threads on a CU that holds nothing else) made the epilogue a chain of memory
    // latencies, slower than the separate dc_bn_bwd_reduce pass it replaces.
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      vec16 yq[PASSES / 2];
#pragma unroll
      for (int ps = 0; ps < PASSES / 2; ++ps) {
        const int m = m0 + rsub + (h * (PASSES / 2) + ps) * RPP;
        yq[ps] = (m < p.M && chok) ? ldg16(by + out_pixel(m) * p.bst.ldy + ch0) : zero16();
      }
#pragma unroll
      for (int ps = 0; ps < PASSES / 2; ++ps) {
        const int row = rsub + (h * (PASSES / 2) + ps) * RPP;
        const int m = m0 + row;
        if (m < p.M && chok) {
          const vec16 v = *reinterpret_cast<const vec16*>(ct + row * CROW + cgrp * 16);
          stg16(yg + out_pixel(m) * p.ldy + ch0, v);
          float f[8], yv[8];
          unpack(v, f, bf16());
          unpack(yq[ps], yv, bf16());
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float gm = (!p.bst.relu || fmaf(yv[e], bms[e], bmh[e]) > 0.f) ? f[e] : 0.f;
            ssum[h][e] += gm;
            ssq[h][e] = fmaf(gm, (yv[e] - bmu[e]) * bis[e], ssq[h][e]);
          }
        }
      }
    }
  } else {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll 2
    for (int ps = 0; ps < PASSES / 2; ++ps) {
      const int row = rsub + (h * (PASSES / 2) + ps) * RPP;
      const int m = m0 + row;
      if (m < p.M && chok) {
        bf16* dst = yg + out_pixel(m) * p.ldy + ch0;
        vec16 v = *reinterpret_cast<const vec16*>(ct + row * CROW + cgrp * 16);
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
        if (do_stats) {   // (uniform) the data-gradient and plain forward launches skip the BatchNorm sums
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            ssum[h][e] += f[e];
            ssq[h][e] = fmaf(f[e], f[e], ssq[h][e]);
          }
        }
      }
    }
  }
  }
  if (slabsel != nullptr) {
    __syncthreads();   // everyone is done reading the C tile
    float* red = reinterpret_cast<float*>(smem);   // [half][which][RPP][TN]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[((h * 2 + 0) * RPP + rsub) * TN + cgrp * 8 + e] = ssum[h][e];
        red[((h * 2 + 1) * RPP + rsub) * TN + cgrp * 8 + e] = ssq[h][e];
      }
    __syncthreads();
    const int rows = p.mtiles * g.os * g.os;   // slab rows: one per 128-pixel tile of every phase
    for (int i = tid; i < 4 * TN; i += 512) {
      const int c = i % TN, hw = i / TN;       // hw = half*2 + which
      const int h = hw >> 1, which = hw & 1;
      const int mt128 = (m0 >> 7) + h;
      if (n0 + c < g.Cout && mt128 < p.mtiles) {
        float a = 0.f;
#pragma unroll
        for (int r = 0; r < RPP; ++r) a += red[(hw * RPP + r) * TN + c];
        slabsel[((size_t)which * rows + phase * p.mtiles + mt128) * g.Cout + n0 + c] = a;
      }
    }
  }
#ifdef DC_STAMPS
  if (dc_stamp_buf256 != nullptr && lane == 0 && wc == 0) {
    unsigned long long* o = dc_stamp_buf256 + ((size_t)blockIdx.x * 2 + grp) * 8;
    o[5] = SEG_T() - t_start;      // whole kernel
  }
#endif
}

}  // namespace

#ifdef DC_STAMPS
extern "C" int dc_debug_stamp_buf256(void* buf) {
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(dc_stamp_buf256), &buf, sizeof(buf));
  return e == hipSuccess ? 0 : dc_set_error(e, __FILE__, __LINE__);
}
#endif

static int g_phase_fast = 1;     // tile order of multi-phase (transposed / strided) launches: phase fastest (A/B switch "igemm256_phase_fast")
void igemm256_set_phase_fast(int v) { g_phase_fast = v ? 1 : 0; }
int igemm256_phase_fast_enabled() { return g_phase_fast; }


// Sum of the split-K partial tiles of one 128-pixel half of one tile, in the order of the splits: y (bf16) and the BatchNorm partial sums of
// the STORED values, one slab row per half as the kernel's own epilogue leaves them.  256 threads = 64 channel groups of 4 x 4 row lanes.
struct SplitKFoldArgs {
  const float* kslab;
  int nsplit, tiles_per_split, ntn, ngroup, mt256, mtiles, M, Cout, ldy;
  void* y[IgemmParams::MAXGROUP];
  float* slab[IgemmParams::MAXGROUP];
};

__global__ __launch_bounds__(256) void igemm256_splitk_fold_kernel(const SplitKFoldArgs a) {
  __shared__ float red[2][4][TN];
  const int half = blockIdx.x & 1;
  const int tile_lin = blockIdx.x >> 1;
  const int ntile_n = tile_lin % a.ntn;
  int rest = tile_lin / a.ntn;
  const int member = rest % a.ngroup;
  const int mtile = rest / a.ngroup;
  bf16* yg = nullptr;
  float* sl = nullptr;
#pragma unroll
  for (int b = 0; b < IgemmParams::MAXGROUP; ++b)
    if (member == b) {
      yg = reinterpret_cast<bf16*>(a.y[b]);
      sl = a.slab[b];
    }
  const int cg = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int ch = ntile_n * TN + cg * 4;
  const bool chok = ch < a.Cout;                      // Cout % 8 == 0: all four channels or none
  const size_t tstride = (size_t)a.tiles_per_split * (TM * TN);
  const float* base = a.kslab + (size_t)tile_lin * (TM * TN) + (size_t)(half * 128) * TN + cg * 4;
  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
  for (int r = rl; r < 128; r += 4) {
    const int m = mtile * TM + half * 128 + r;
    if (m >= a.M || !chok) continue;
    f32x4 v = *reinterpret_cast<const f32x4*>(base + (size_t)r * TN);
    for (int s = 1; s < a.nsplit; ++s) v += *reinterpret_cast<const f32x4*>(base + s * tstride + (size_t)r * TN);
    uint2 o;
    o.x = pack2_bf16(v[0], v[1]);
    o.y = pack2_bf16(v[2], v[3]);
    *reinterpret_cast<uint2*>(yg + (size_t)m * a.ldy + ch) = o;
    if (sl != nullptr) {
      const float f[4] = {__uint_as_float(o.x << 16), __uint_as_float(o.x & 0xffff0000u), __uint_as_float(o.y << 16), __uint_as_float(o.y & 0xffff0000u)};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        ssum[e] += f[e];
        ssq[e] = fmaf(f[e], f[e], ssq[e]);
      }
    }
  }
  if (sl == nullptr) return;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[0][rl][cg * 4 + e] = ssum[e];
    red[1][rl][cg * 4 + e] = ssq[e];
  }
  __syncthreads();
  const int mt128 = mtile * 2 + half;
  for (int i = threadIdx.x; i < 2 * TN; i += 256) {
    const int c = i % TN, which = i / TN;
    if (ntile_n * TN + c < a.Cout && mt128 < a.mtiles)
      sl[((size_t)which * a.mtiles + mt128) * a.Cout + ntile_n * TN + c] = (red[which][0][c] + red[which][1][c]) + (red[which][2][c] + red[which][3][c]);
  }
}

static int g_splitk = 1;         // tuning switch "igemm256_splitk": 0 = never split the K loop of a 256-tile launch
void igemm256_set_splitk(int v) { g_splitk = v; }

// A launch that leaves more than half of the chip idle and has a long K loop is cut along K: as many splits as fit one round, at least two
// channel-chunk groups (2 x 4 chunks x taps K steps) each, at most eight.  Plain stride-1 launches only (no sub-pixel phases, no bias, no
// accumulation, no BatchNorm-backward epilogue: the fold kernel does the plain epilogue).
int igemm256_splitk_plan(const IgemmParams& p, size_t* slab_bytes) {
  if (slab_bytes) *slab_bytes = 0;
  if (!g_splitk || p.g.os != 1 || p.m_beg != 0 || p.bias != nullptr || p.accumulate || p.bst.y != nullptr || (p.g.Cout & 7)) return 1;
  const long tiles = igemm256_tiles(p);
  if (tiles < 1) return 1;
  const int ngk = ((p.g.Cin + BK - 1) / BK + 3) / 4;
  int s = 1;
  if (tiles <= 128) {
    s = (int)(256 / tiles);
    if (s > ngk / 2) s = ngk / 2;
    if (s > 8) s = 8;
  } else if (g_splitk > 1) {
    // more than one round with a thin last one (the grouped atrous launch at local batch 8: 324 tiles of 576 K steps, the second round a
    // quarter full): the split count of 2 - 4 that fills the rounds best, if it fills them at least a tenth better than the plain launch
    auto fill = [](long n) { return (double)n / (double)(((n + 255) / 256) * 256); };
    double best = fill(tiles) + 0.10;
    for (int c = 2; c <= 4 && c <= ngk / 2; ++c)
      if (fill(tiles * c) > best) {
        best = fill(tiles * c);
        s = c;
      }
  }
  if (s < 2) return 1;
  if (slab_bytes) *slab_bytes = (size_t)s * tiles * TM * TN * sizeof(float);
  return s;
}

int launch_igemm256(const IgemmParams& p_in, hipStream_t st) {
  const size_t lds = (size_t)MAIN_BYTES + 128;
  static const void* zero_dev = nullptr;
  static hipError_t init_err = hipSuccess;
  DC_ONCE({
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm256_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    void* zp = nullptr;
    init_err = hipGetSymbolAddress(&zp, HIP_SYMBOL(zero_page256));
    zero_dev = zp;
  });
  if (init_err != hipSuccess) return dc_set_error(init_err, __FILE__, __LINE__);
  IgemmParams p = p_in;
  p.zero_page = zero_dev;
  p.phase_fast = g_phase_fast;
  hipLaunchKernelGGL(igemm256_kernel, dim3((unsigned)(igemm256_tiles(p) * (p.ksplit > 1 ? p.ksplit : 1))), dim3(512), lds, st, p);
  DC_CHECK_LAUNCH();
  return 0;
}

int launch_igemm256_splitk(const IgemmParams& p_in, int splits, void* ws, hipStream_t st) {
  IgemmParams p = p_in;
  p.ksplit = splits;
  p.kslab = reinterpret_cast<float*>(ws);
  if (int e = launch_igemm256(p, st)) return e;
  SplitKFoldArgs a;
  a.kslab = p.kslab; a.nsplit = splits; a.tiles_per_split = (int)igemm256_tiles(p); a.ntn = (p.g.Cout + TN - 1) / TN;
  a.ngroup = p.ngroup > 1 ? p.ngroup : 1; a.mt256 = (p.M + TM - 1) / TM; a.mtiles = p.mtiles; a.M = p.M; a.Cout = p.g.Cout; a.ldy = p.ldy;
  for (int b = 0; b < IgemmParams::MAXGROUP; ++b) {
    a.y[b] = b == 0 ? p.y : p.gy[b - 1];
    a.slab[b] = b == 0 ? p.slab : p.gslab[b - 1];
  }
  hipLaunchKernelGGL(igemm256_splitk_fold_kernel, dim3((unsigned)(2 * a.tiles_per_split)), dim3(256), 0, st, a);
  DC_CHECK_LAUNCH();
  return 0;
}

}  // namespace dc
