// Gather-form implicit GEMM, 256 output pixels x 256 output channels per 512-thread workgroup (bf16).
//
// Why a second tile shape: in-kernel stamps on the 128 x 128 kernel (scripts/igemm_stamps.py) show its K loop advancing at
// ~1480 cycles per 32-deep step with three workgroups per CU, i.e. the CU takes in 16 KiB of operands per ~490 cycles
// (~33 B/clk) while its MFMA pipes are busy half of that time: the loop is bound by the global->LDS fill, and the bytes to
// fill per MFMA fall with the tile edge.  A 256 x 256 tile needs half the operand bytes per flop.
//
//   waves     8 = 2 pixel halves (the two "groups") x 4 channel quarters; a wave owns 128 pixels x 64 channels
//             (8 x 4 MFMA tiles of 16 x 16, 128 accumulator registers), so 12 ds_read_b128 feed 32 MFMAs
//   K step    32 (64-byte rows, the XOR swizzle of igemm.hip); ring of 4 stages x (16 KiB weights + 16 KiB pixels) = 128 KiB
//   schedule  per stage and wave four segments separated by workgroup barriers:
//               L0  read 4 weight + 4 pixel fragments of stage s; start the LDS-DMA of the WEIGHT half of stage s+3
//               M0  16 MFMAs (s_setprio 1)
//               L1  read the other 4 pixel fragments; start the PIXEL half of stage s+3; s_waitcnt vmcnt(8): stage s+1 landed
//               M1  16 MFMAs
//             Group 1 runs one barrier behind group 0 (an extra barrier before its loop, one after group 0's), so on every SIMD
//             one wave is in an M segment while its partner wave is in an L segment: the MFMA pipe and the LDS/DMA issue
//             alternate instead of colliding.
//   hazards   a stage is read one segment after the barrier that follows every wave's vmcnt for it (RAW); the weight half of
//             ring slot (s+3)%4 is overwritten in L0(s), two barriers after its last reader retired its reads (all of
//             them in L0(s-1)); the pixel half in L1(s), two barriers after the trailing group's L1(s-1) reads were retired
//             by the lgkmcnt(0) of its M1(s-1) (WAR).  Past the last stage the DMA slots are filled from the zero page so that
//             the vmcnt arithmetic is the same in every iteration.
//   epilogue  as igemm.hip: accumulators -> LDS C tile (256 rows x 528 B) -> coalesced 16-byte stores, optional bias /
//             accumulate, BatchNorm partial sums of the stored values, one slab row per 128-pixel half (same slab shape as the
//             128-tile kernel, so the two are interchangeable per layer).
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

__device__ inline int swz64(int row, int slot) { return row * 64 + ((slot ^ ((row >> 1) & 3)) << 4); }

__global__ __launch_bounds__(512) void igemm256_kernel(const IgemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* s_tap = reinterpret_cast<int*>(smem + MAIN_BYTES);
  const GatherGeom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;   // pixel half; also the stagger group
  const int wc = wave & 3;     // channel quarter

  // XCD-aware tile order (see igemm.hip): consecutive tiles of an XCD walk the channel tiles of one pixel tile first
  const int ntn = (g.Cout + TN - 1) / TN;
  const int mt256 = (p.M + TM - 1) / TM;
  const int nwg = gridDim.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, xslot = blockIdx.x >> 3;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + xslot;
  const int ntile_n = tile % ntn;
  const int rest = tile / ntn;
  const int mtile = rest % mt256;
  const int phase = rest / mt256;
  const int n0 = ntile_n * TN, m0 = mtile * TM;
  const int py = phase / g.os, px = phase % g.os;

  const int tap_beg = g.phase_beg[phase], ntap = g.phase_beg[phase + 1] - tap_beg;
  if (tid < ntap) {
    const Tap tp = g.taps[tap_beg + tid];
    s_tap[3 * tid] = tp.dy;
    s_tap[3 * tid + 1] = tp.dx;
    s_tap[3 * tid + 2] = tp.widx;
  }
  __syncthreads();
  const int kchunks = (g.Cin + BK - 1) / BK;
  const int steps = ntap * kchunks;
  if (steps == 0 && p.accumulate) return;   // a phase without taps contributes zeros

  // ---- per-thread DMA bookkeeping: 2 rows of each operand (instruction j of wave w fills rows (8j + w)*16 .. +15) -------
  const int lrow = lane >> 2, pslot = lane & 3;
  const int lslot = pslot ^ ((lrow >> 1) & 3);          // logical 16-byte slot this lane fetches (swizzle on the source side)
  const bf16* __restrict__ xg = reinterpret_cast<const bf16*>(p.x);
  const bf16* __restrict__ wg = reinterpret_cast<const bf16*>(p.w);
  int rown[2], riy[2], rix[2];
  bool rok[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = m0 + (8 * j + wave) * 16 + lrow;
    rok[j] = m < p.M;
    const int mm = rok[j] ? m : 0;
    const int n = fast_div(mm, g.div_hw);
    const int rem = mm - n * (g.Qh * g.Qw);
    const int qy = fast_div(rem, g.div_w), qx = rem - qy * g.Qw;
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
  int itap = 0, ikc = 0;   // (tap, K chunk) of the next stage to issue
  auto advance = [&]() {
    if (++ikc == kchunks) {
      ikc = 0;
      ++itap;
    }
  };
#pragma unroll
  for (int q = 0; q < NSTAGE - 1; ++q) {
    const bool live = q < steps;
    issue_A(itap, ikc, q, live);
    issue_B(itap, ikc, q, live);
    advance();
  }
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one barrier behind

  for (int s = 0; s < steps; ++s) {
    const char* wa = smem + (s & (NSTAGE - 1)) * STAGE;
    const char* xb = wa + OPER;
    const int nslot = (s + NSTAGE - 1) & (NSTAGE - 1);
    const bool live = s + NSTAGE - 1 < steps;
    vec16 fa[4], fb[4], fb2[4];
    // ---- L0
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const vec16*>(wa + swz64(wc * 64 + i * 16 + fr, fg));
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const vec16*>(xb + swz64(grp * 128 + j * 16 + fr, fg));
    issue_A(itap, ikc, nslot, live);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- M0
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- L1
#pragma unroll
    for (int j = 0; j < 4; ++j) fb2[j] = *reinterpret_cast<const vec16*>(xb + swz64(grp * 128 + (4 + j) * 16 + fr, fg));
    issue_B(itap, ikc, nslot, live);
    advance();
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // stage s+1 has landed (stages s+2, s+3 = 8 instructions stay in flight)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- M1
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb2[j]), acc[i][4 + j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();   // re-align the two groups
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the zero-page fills of the last three slots
  __builtin_amdgcn_s_barrier();

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
  bf16* __restrict__ yg = reinterpret_cast<bf16*>(p.y);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll 2
    for (int ps = 0; ps < PASSES / 2; ++ps) {
      const int row = rsub + (h * (PASSES / 2) + ps) * RPP;
      const int m = m0 + row;
      if (m < p.M && chok) {
        size_t opix = (size_t)m;
        if (g.os != 1) {
          const int n = fast_div(m, g.div_hw);
          const int rem = m - n * (g.Qh * g.Qw);
          const int qy = fast_div(rem, g.div_w), qx = rem - qy * g.Qw;
          opix = (size_t)(n * g.Hout + qy * g.os + py) * g.Wout + qx * g.os + px;
        }
        bf16* dst = yg + opix * p.ldy + ch0;
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
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          ssum[h][e] += f[e];
          ssq[h][e] = fmaf(f[e], f[e], ssq[h][e]);
        }
      }
    }
  }
  if (p.slab != nullptr) {
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
      const int mt128 = mtile * 2 + h;
      if (n0 + c < g.Cout && mt128 < p.mtiles) {
        float a = 0.f;
#pragma unroll
        for (int r = 0; r < RPP; ++r) a += red[(hw * RPP + r) * TN + c];
        p.slab[((size_t)which * rows + phase * p.mtiles + mt128) * g.Cout + n0 + c] = a;
      }
    }
  }
}

}  // namespace

int launch_igemm256(const IgemmParams& p_in, hipStream_t st) {
  const size_t lds = (size_t)MAIN_BYTES + 128;
  static bool attr_set = false;
  static const void* zero_dev = nullptr;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm256_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    void* zp = nullptr;
    hipError_t e = hipGetSymbolAddress(&zp, HIP_SYMBOL(zero_page256));
    if (e != hipSuccess) return dc_set_error(e, __FILE__, __LINE__);
    zero_dev = zp;
    attr_set = true;
  }
  IgemmParams p = p_in;
  p.zero_page = zero_dev;
  hipLaunchKernelGGL(igemm256_kernel, dim3((unsigned)igemm256_tiles(p)), dim3(512), lds, st, p);
  DC_CHECK_LAUNCH();
  return 0;
}

}  // namespace dc
