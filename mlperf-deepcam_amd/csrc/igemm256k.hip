// The persistent 256 x 256 implicit-GEMM kernel (igemm256p.hip) with 128-BYTE K ROWS: 64-deep stages, every LDS-DMA row piece a whole L2
// line (8 rows x 128 B per wave instruction) instead of half of one.  On the pointwise 256 x 384 tile that is worth 4 - 15 % (DESIGN
// section 5: the long-K GEMMs draw L2 requests at the rate the hardware has, 64 useful bytes per request).
//
// Differences from igemm256k_kernel:
//   * ring of TWO 64 KiB stages (pw384_kernel's 128-byte-row scheme): stage t+1 is issued during the first 32-deep half of stage t (its
//     slot was released by the barrier in the second half of stage t-1) and must have landed by the barrier in the second half of stage t:
//     every stage waits vmcnt(0), so the epilogue's stores need no counting here;
//   * served geometries: input pixel index = phase-grid pixel index + tap offset (stride-1 convolutions and their data gradients, the
//     sub-pixel phases of a transposed convolution: is == 1, Hin == Qh, Win == Qw).  A lane's four pixel rows per stage are then 64 pixels
//     apart in memory and its four weight rows 64 channels apart: ONE offset each plus scalar strides, and (qy, qx) per row for the halo
//     test only -- fewer address registers than the 64-byte form, which is what lets the doubled row count fit;
//   * both waves of a SIMD run the same load / MFMA order (see the end of the kernel);
//   * K order: groups of two 64-channel chunks outside, the taps in the middle (the 64-byte kernels' four 32-channel chunks): the same
//     products in the same order, outputs bit-identical.
#include <type_traits>

#include "igemm.h"

namespace dc {

namespace {

constexpr int TM = 256, TN = 256;      // pixels, channels per tile
constexpr int ROWB = 128;              // bytes of K per row and stage (64 bf16)
constexpr int BK = 64;
constexpr int NSTAGE = 2;
constexpr int OPER = TM * ROWB;        // 32 KiB per operand per stage
constexpr int STAGE = 2 * OPER;
constexpr int RING = NSTAGE * STAGE;   // 128 KiB
constexpr int KG = 2;                  // 64-channel chunks per tap sweep (= igemm256.hip's four 32-channel chunks)

static __device__ __attribute__((aligned(256))) unsigned char zero_page256k[256];
static __device__ __attribute__((aligned(256))) unsigned char dump_page256k[256 * 64];   // inactive epilogue lanes store here (per lane 256 B)
typedef __attribute__((address_space(1))) const void* gas_ptr;
typedef __attribute__((address_space(3))) void* lds_ptr;

// byte offset of 16-byte K group `slot` (0..3) of the FIRST 32-deep half of a row; the second half is the same address ^ 64
__device__ inline int swz128(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }
__device__ inline uint32_t swap_rows16(uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F); }
__device__ inline float row_sum16(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));   // row_mirror
  return v;
}

struct TileInfo {
  int n0, m0, phase, tap_beg, ntap;
};

}  // namespace

// The kernel's own argument block: what the tile loop needs of IgemmParams, flat.  The K loops run at the register limit (128
// accumulators + 48 fragment registers per lane, and close to a hundred live scalars), so everything that is only needed BETWEEN
// tiles (tile decode, row bookkeeping, epilogue addressing) is read again from the kernel-argument segment there, through a pointer
// the compiler cannot see through, instead of being held in scalar registers across the loops.
struct Igemm256kArgs {
  const void* x;
  const void* w;
  void* y;
  float* slab;
  const void* zero_page;
  const void* dump;
  int Hin, Win, Cin, Hout, Wout, Cout, Qw, QhQw;
  unsigned hwM;   // FastDiv by Qh*Qw, by Qw (conv_geom.h)
  int hwS;
  unsigned wM;
  int wS;
  int os, is, ldx, ldy, ldw, M, mtiles, phase_fast, ntiles;
  int phase_beg[5];
  int taps[9 * 3];   // dy, dx, widx, sorted by phase
};
typedef const __attribute__((address_space(4))) Igemm256kArgs* KArgs;

template <class K>
__device__ inline void grid_pixel_k(K k, int m, int& n, int& qy, int& qx) {
  n = k->hwM == 0 ? m : (int)(__umulhi((unsigned)m, k->hwM) >> k->hwS);
  const int rem = m - n * k->QhQw;
  qy = k->wM == 0 ? rem : (int)(__umulhi((unsigned)rem, k->wM) >> k->wS);
  qx = rem - qy * k->Qw;
}

template <bool STATS>
__global__ __launch_bounds__(512) void igemm256k_kernel(const Igemm256kArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* s_tap_all = reinterpret_cast<int*>(smem + RING);      // the layer's tap table (up to 9 x 3 ints), written once
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;   // pixel half; also the stagger group
  const int wc = wave & 3;     // channel quarter
  const KArgs k0 = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
  auto args = [&]() {          // the argument block, to be read afresh
    KArgs k = k0;
    asm volatile("" : "+s"(k));
    return k;
  };

  // Tile order as igemm256_kernel (XCD-aware: XCD x owns a contiguous share of the tiles, consecutive tiles walk the channel tiles,
  // then -- phase_fast -- the sub-pixel phases of one pixel tile).  gridDim.x is a multiple of 8; workgroup i of W = gridDim.x / 8
  // on XCD x takes, in round r, slot r * W + (i + r) % W of its XCD's share (W even; r * W + i for odd W).  The rotation by r matters: with a plain stride of W a
  // workgroup would meet the same channel tile and the same phase in every round whenever W is a multiple of their counts, and the
  // phases of a transposed convolution cost 1 : 2 : 2 : 4.
  const int W = gridDim.x >> 3, wi = blockIdx.x >> 3;
  auto slot_of = [&](int r) { return r * W + ((W & 1) ? wi : (wi + r) % W); };   // consecutive slots of a workgroup: an odd distance
  auto exists = [&](int r) {
    const KArgs k = args();
    const int ntiles = k->ntiles;
    return slot_of(r) < (ntiles >> 3) + (((int)blockIdx.x & 7) < (ntiles & 7) ? 1 : 0);
  };
  auto decode = [&](int r) {
    const KArgs k = args();
    const int ntiles = k->ntiles;
    const int ntn = (k->Cout + TN - 1) / TN;
    const int mt256 = (k->M + TM - 1) / TM;
    const int nph = k->os * k->os;
    const int q8 = ntiles >> 3, r8 = ntiles & 7;
    const int xcd = blockIdx.x & 7, xslot = slot_of(r);
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + xslot;
    const int ntile_n = tile % ntn;
    const int rest = tile / ntn;
    TileInfo t;
    t.phase = k->phase_fast ? rest % nph : rest / mt256;
    const int mtile = k->phase_fast ? rest / nph : rest % mt256;
    t.n0 = ntile_n * TN;
    t.m0 = mtile * TM;
    t.tap_beg = k->phase_beg[t.phase];
    t.ntap = k->phase_beg[t.phase + 1] - t.tap_beg;
    return t;
  };
  if (tid < 27) s_tap_all[tid] = k0->taps[tid];
  const int Cin = a.Cin, Cout = a.Cout, Hin = a.Hin, Win = a.Win, ldx = a.ldx, ldw = a.ldw;
  const int kchunks = (Cin + BK - 1) / BK;

  // ---- per-thread DMA bookkeeping of the tile whose stages are being ISSUED (one stage ahead of the multiplying) ---------------------
  // Instruction q (0..3) of this wave fills rows (8q + wave) * 8 .. + 7 of an operand's stage image, 128 bytes each: lane -> row lane >> 3,
  // physical slot lane & 7; the logical slot it fetches is swizzled on the source side (the same for all four q).
  const int lrow = lane >> 3, pslot = lane & 7;
  const int lslot = pslot ^ ((((wave & 1) << 2) + (lrow >> 1)) & 7);
  const int klim = Cin - lslot * 8;                     // this lane's slot of K chunk kc is inside the row while kc * BK < klim
  const uintptr_t xg = (uintptr_t)a.x, wg = (uintptr_t)a.w, zp = (uintptr_t)a.zero_page;
  const unsigned wstride = 64u * (unsigned)ldw * 2u, xstride = 64u * (unsigned)ldx * 2u;   // bytes between a lane's rows q and q + 1
  int ryx[4];                  // (qy << 16) | qx of pixel row q, with qy = 0x7fff for a row past M (never in bounds)
  int wlim = 0;                // row q of the weights is a channel below Cout while 64 * q < wlim
  int xoff = 0, woff = 0;      // byte offsets of row 0 (K offset 0) from x / w for the tap being issued; row q: + q * stride
  const int* s_tap = s_tap_all;                         // taps of the issuing tile's phase
  int tdy = 0, tdx = 0;        // the tap being issued
  int tapA = -1, tapB = -1;
  int itap = 0, ikc = 0, intap = 1;   // (tap, K chunk) of the next stage to issue; taps of the issuing tile
  int kbeg = 0, kend = kchunks < KG ? kchunks : KG;
  int im0 = 0;                 // first pixel of the issuing tile (this lane's row 0: + wave * 8 + (lane >> 3))
  int in0_w = 0;               // first channel of the issuing tile
  bool issuing = true;                // false past the last tile: zero-page fills
  auto load_rows = [&](const TileInfo& t) {
    const KArgs k = args();
    int lrow_o = lrow;
    asm volatile("" : "+v"(lrow_o));
    im0 = t.m0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int m = t.m0 + wave * 8 + lrow_o + 64 * q;
      const bool rok = m < k->M;
      const int mm = rok ? m : 0;
      int n, qy, qx;
      grid_pixel_k(k, mm, n, qy, qx);
      ryx[q] = ((rok ? qy : 0x7fff) << 16) | qx;
    }
    wlim = k->Cout - (t.n0 + wave * 8 + lrow_o);
    intap = t.ntap;
    s_tap = s_tap_all + 3 * t.tap_beg;
    tapA = tapB = -1;
    itap = 0;
    ikc = 0;
    kbeg = 0;
    kend = kchunks < KG ? kchunks : KG;
    // (woff's tile part: set at the first tap change below from in0)
    in0_w = t.n0;
  };
  auto issue_A = [&](int slot) {
    if (issuing && itap != tapA) {
      tapA = itap;
      const int widx = s_tap[3 * itap + 2];
      int lane_t = lane;             // lane-derived values recomputed here: not worth a register each through the K loop
      asm volatile("" : "+v"(lane_t));
      const int lrow_t = lane_t >> 3, lslot_t = (lane_t & 7) ^ ((((wave & 1) << 2) + (lane_t >> 4)) & 7);
      woff = ((widx * Cout + in0_w + wave * 8 + lrow_t) * ldw + lslot_t * 8) * 2;
    }
    const int kofs = ikc * BK;
    const bool kok = issuing & (kofs < klim);
    char* base = smem + slot * STAGE;
#pragma unroll
    for (int q = 0; q < 4; ++q)
    {
      __builtin_amdgcn_global_load_lds((gas_ptr)((kok & (64 * q < wlim)) ? wg + ((unsigned)woff + q * wstride + (unsigned)kofs * 2u) : zp),
                                       (lds_ptr)(base + (8 * q + wave) * 8 * ROWB), 16, 0, 0);
      __builtin_amdgcn_sched_barrier(0);      // one address pair at a time: the loop has no registers for four
    }
  };
  auto issue_B = [&](int slot) {
    if (issuing && itap != tapB) {
      tapB = itap;
      tdy = s_tap[3 * itap];
      tdx = s_tap[3 * itap + 1];
      int lane_t = lane;
      asm volatile("" : "+v"(lane_t));
      const int lslot_t = (lane_t & 7) ^ ((((wave & 1) << 2) + (lane_t >> 4)) & 7);
      // input pixel = phase-grid pixel + the tap's offset (is == 1, Hin == Qh, Win == Qw); only dereferenced where the halo test passes
      xoff = ((im0 + wave * 8 + (lane_t >> 3) + tdy * Win + tdx) * ldx + lslot_t * 8) * 2;
    }
    const int kofs = ikc * BK;
    const bool kok = issuing & (kofs < klim);
    char* base = smem + slot * STAGE + OPER;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int iy = (ryx[q] >> 16) + tdy, ix = (ryx[q] & 0xffff) + tdx;
      const bool ok = kok & ((unsigned)iy < (unsigned)Hin) & ((unsigned)ix < (unsigned)Win);
      __builtin_amdgcn_global_load_lds((gas_ptr)(ok ? xg + ((unsigned)xoff + q * xstride + (unsigned)kofs * 2u) : zp),
                                       (lds_ptr)(base + (8 * q + wave) * 8 * ROWB), 16, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // K order of a tile: groups of KG channel chunks outside, the taps in the middle, the group's chunks inside (igemm256.hip)
  auto advance = [&]() {
    if (++ikc == kend) {
      ikc = kbeg;
      if (++itap == intap) {
        itap = 0;
        kbeg = kend;
        kend = kend + KG < kchunks ? kend + KG : kchunks;
        ikc = kbeg;
      }
    }
  };

  f32x4 acc[4][8];   // [channel block][pixel block]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fg = lane >> 4;

  // ---- first tile: rows, stage 0 landed ----------------------------------------------------------------------------------------------
  int round = 0;                   // every workgroup owns a tile of round 0 (launcher: W <= ntiles / 8)
  TileInfo cur = decode(0);
  bool has_next = exists(1);
  __syncthreads();                 // tap table written
  load_rows(cur);
  int base = 0;                    // ring slot of the current tile's stage 0
  issue_A(0);
  issue_B(0);
  advance();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  vec16 fa[4], fb[4], fb2[4];
  {
    const char* wa = smem;
    const char* xb = wa + OPER;
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const vec16*>(wa + swz128(wc * 64 + i * 16 + fr, fg));
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const vec16*>(xb + swz128(grp * 128 + j * 16 + fr, fg));
  }

  // The two waves of a SIMD (w and w+4, the two pixel halves) run the halves of every segment in opposite order (igemm256.hip); the
  // whole tile loop is instantiated twice.
  auto tiles = [&](auto loads_first_tag) {
    constexpr bool LOADS_FIRST = decltype(loads_first_tag)::value;
    for (;;) {
      const int stages = cur.ntap * kchunks;
      // One 32-deep half of a stage.  HALF 0: the fragments of this half come from (wa, xb); the next stage's LDS-DMAs are issued (weights
      // beside M0, pixels beside M1); the pre-read for the next half is the same stage's second half (address ^ 64).  HALF 1: nothing is
      // issued; between M0 and M1 the next stage must have landed (vmcnt(0), barrier), and M1 pre-reads ITS first half.
      auto half = [&](auto half_tag, int t) {
        constexpr int H = decltype(half_tag)::value;
        // (second half of a row: the swizzled offset ^ 64)
        constexpr int X0 = H * 64, X1 = H == 0 ? 64 : 0;
        const char* wa = smem + ((base + t) & 1) * STAGE;
        const char* xb = wa + OPER;
        const char* wa1 = H == 0 ? wa : smem + ((base + t + 1) & 1) * STAGE;
        const char* xb1 = wa1 + OPER;
        const int nslot = (base + t + 1) & 1;
        // ---- M0
        if constexpr (LOADS_FIRST) {
#pragma unroll
          for (int j = 0; j < 4; ++j) fb2[j] = *reinterpret_cast<const vec16*>(xb + (swz128(grp * 128 + (4 + j) * 16 + fr, fg) ^ X0));
          if constexpr (H == 0) issue_A(nslot);
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
          for (int j = 0; j < 4; ++j) fb2[j] = *reinterpret_cast<const vec16*>(xb + (swz128(grp * 128 + (4 + j) * 16 + fr, fg) ^ X0));
          if constexpr (H == 0) issue_A(nslot);
        }
        if constexpr (H == 1) {
          // the next stage (issued during this stage's first half) has landed, and with it everything else this wave had in flight
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
        // ---- M1
        if constexpr (LOADS_FIRST) {
#pragma unroll
          for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const vec16*>(xb1 + (swz128(grp * 128 + j * 16 + fr, fg) ^ X1));
          if constexpr (H == 0) {
            issue_B(nslot);
            advance();
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb2[j]), acc[i][4 + j], 0, 0, 0);
          fa[i] = *reinterpret_cast<const vec16*>(wa1 + (swz128(wc * 64 + i * 16 + fr, fg) ^ X1));
        }
        __builtin_amdgcn_s_setprio(0);
        if constexpr (!LOADS_FIRST) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const vec16*>(xb1 + (swz128(grp * 128 + j * 16 + fr, fg) ^ X1));
          if constexpr (H == 0) {
            issue_B(nslot);
            advance();
          }
        }
      };
      // the tile's own stages 1 .. stages-1 are issued by its stages 0 .. stages-2; then the per-thread bookkeeping switches to the next tile
      // (outside the loop: the switch needs the whole geometry in scalar registers) and the last stage issues the next tile's stage 0
      for (int t = 0; t < stages - 1; ++t) {
        half(std::integral_constant<int, 0>{}, t);
        half(std::integral_constant<int, 1>{}, t);
      }
      if (has_next) load_rows(decode(round + 1));
      else issuing = false;
      half(std::integral_constant<int, 0>{}, stages - 1);
      half(std::integral_constant<int, 1>{}, stages - 1);
      // fa / fb now hold the first fragments of the NEXT tile's stage 0 (the pre-read of "stage s+1" in the last step)

      // ---- epilogue of the current tile from the accumulator registers (igemm256.hip's register epilogue; every store instruction
      // is issued by every wave whatever its lanes' validity: inactive lanes write their own 256-byte line of the dump page)
      {
        // (the lane-derived values of the epilogue are recomputed behind an opaque copy of the lane id: hoisted out of the tile loop
        // they would stay live through the K loops, which have no register to spare)
        const KArgs k = args();
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int fr = lane_e & 15, fg = lane_e >> 4;
        const bool odd = fg & 1;
        const int os = k->os;
        const int py = cur.phase / os, px = cur.phase % os;
        const int m0 = cur.m0, n0 = cur.n0;
        const int Cout_e = k->Cout, ldy = k->ldy;
        bf16* __restrict__ yg = reinterpret_cast<bf16*>(k->y);
        size_t opix[8];
        bool pok[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int m = m0 + grp * 128 + j * 16 + fr;
          pok[j] = m < k->M;
          opix[j] = (size_t)m;
          if (os != 1) {
            const int mm = pok[j] ? m : 0;
            int n, qy, qx;
            grid_pixel_k(k, mm, n, qy, qx);
            opix[j] = (size_t)(n * k->Hout + qy * os + py) * k->Wout + qx * os + px;
          }
        }
        bf16* mydump = reinterpret_cast<bf16*>(const_cast<void*>(k->dump)) + (size_t)lane_e * 128;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          const int i0 = 2 * pr;
          const int chl = wc * 64 + (i0 + (odd ? 1 : 0)) * 16 + (fg >> 1) * 8;   // first of this lane's 8 channels after the trade
          const int ch0 = n0 + chl;
          const bool chok = ch0 < Cout_e;
          float st0[8], st1[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) st0[e] = st1[e] = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const uint32_t a0 = pack2_bf16(acc[i0][j][0], acc[i0][j][1]);
            const uint32_t a1 = pack2_bf16(acc[i0][j][2], acc[i0][j][3]);
            const uint32_t b0 = pack2_bf16(acc[i0 + 1][j][0], acc[i0 + 1][j][1]);
            const uint32_t b1 = pack2_bf16(acc[i0 + 1][j][2], acc[i0 + 1][j][3]);
            const uint32_t r0 = swap_rows16(odd ? a0 : b0), r1 = swap_rows16(odd ? a1 : b1);
            vec16 vv;
            vv.w[0] = odd ? r0 : a0;
            vv.w[1] = odd ? r1 : a1;
            vv.w[2] = odd ? b0 : r0;
            vv.w[3] = odd ? b1 : r1;
            const bool ok = pok[j] && chok;
            bf16* dst = ok ? yg + opix[j] * ldy + ch0 : mydump;
            stg16(dst, vv);
            if (STATS && ok) {
              float f[8];
              unpack(vv, f, bf16());
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                st0[e] += f[e];
                st1[e] = fmaf(f[e], f[e], st1[e]);
              }
            }
          }
          if constexpr (STATS) {
            const int mtiles = k->mtiles;
            const int rows = mtiles * os * os;
            const int mt128 = (m0 >> 7) + grp;
            float mine = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float t0 = row_sum16(st0[e]), t1 = row_sum16(st1[e]);
              if (fr == e) mine = t0;
              if (fr == 8 + e) mine = t1;
            }
            const int which = fr >> 3, e = fr & 7;
            const int c = n0 + wc * 64 + (2 * pr + (odd ? 1 : 0)) * 16 + (fg >> 1) * 8 + e;
            const bool sok = c < Cout_e && mt128 < mtiles;
            float* sdst = sok ? k->slab + ((size_t)which * rows + cur.phase * mtiles + mt128) * Cout_e + c
                              : reinterpret_cast<float*>(mydump) + (lane_e & 31);
            *sdst = mine;
          }
        }
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (!has_next) break;
      // ---- next tile becomes the current one
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      base = (base + stages) & (NSTAGE - 1);
      ++round;
      cur = decode(round);
      has_next = exists(round + 1);
    }
  };
  // One order for both wave groups: instantiating the tile loop twice (igemm256p's opposite load / MFMA orders for the two waves of a SIMD)
  // spills 25 registers into the K loops here, and a loop that waits vmcnt(0) every stage cannot take a scratch reload.
  tiles(std::false_type{});
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the zero-page fills of the last slots must land before the LDS is released
}

bool igemm256k_eligible(const IgemmParams& p) {
  const GatherGeom& g = p.g;
  if (p.ngroup > 1 || p.bias != nullptr || p.accumulate || p.bst.y != nullptr || p.m_beg != 0) return false;
  // input pixel index = phase-grid pixel index + tap offset
  if (g.is != 1 || g.Hin != g.Qh || g.Win != g.Qw || g.Cin < BK) return false;
  for (int ph = 0; ph < g.os * g.os; ++ph)
    if (g.phase_beg[ph + 1] - g.phase_beg[ph] < 1) return false;      // (the data gradient of a strided 1 x 1 convolution has phases without a tap: zeros, written by igemm256_kernel)
  // the per-lane row addresses are 32-bit byte offsets from the tensor bases
  if ((size_t)p.N * g.Hin * g.Win * p.ldx * 2 >= (1ull << 32) || (size_t)9 * g.Cout * p.ldw * 2 >= (1ull << 32)) return false;
  return igemm256_tiles(p) >= 8;
}

int launch_igemm256k(const IgemmParams& p_in, int workgroups, hipStream_t st) {
  const size_t lds = (size_t)RING + 256;
  static const void* zero_dev = nullptr;
  static const void* dump_dev = nullptr;
  static hipError_t init_err = hipSuccess;
  auto k0 = &igemm256k_kernel<false>;
  auto k1 = &igemm256k_kernel<true>;
  DC_ONCE({
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    void* zp = nullptr;
    init_err = hipGetSymbolAddress(&zp, HIP_SYMBOL(zero_page256k));
    zero_dev = zp;
    if (init_err == hipSuccess) {
      init_err = hipGetSymbolAddress(&zp, HIP_SYMBOL(dump_page256k));
      dump_dev = zp;
    }
  });
  if (init_err != hipSuccess) return dc_set_error(init_err, __FILE__, __LINE__);
  const GatherGeom& g = p_in.g;
  Igemm256kArgs a;
  a.x = p_in.x;
  a.w = p_in.w;
  a.y = p_in.y;
  a.slab = p_in.slab;
  a.zero_page = zero_dev;
  a.dump = dump_dev;
  a.Hin = g.Hin; a.Win = g.Win; a.Cin = g.Cin;
  a.Hout = g.Hout; a.Wout = g.Wout; a.Cout = g.Cout;
  a.Qw = g.Qw; a.QhQw = g.Qh * g.Qw;
  a.hwM = g.div_hw.M; a.hwS = g.div_hw.sh;
  a.wM = g.div_w.M; a.wS = g.div_w.sh;
  a.os = g.os; a.is = g.is;
  a.ldx = p_in.ldx; a.ldy = p_in.ldy; a.ldw = p_in.ldw;
  a.M = p_in.M; a.mtiles = p_in.mtiles;
  a.phase_fast = igemm256_phase_fast_enabled();
  const long ntiles = igemm256_tiles(p_in);
  a.ntiles = (int)ntiles;
  for (int i = 0; i < 5; ++i) a.phase_beg[i] = g.phase_beg[i];
  for (int i = 0; i < 9; ++i) {
    a.taps[3 * i] = i < g.ntaps ? g.taps[i].dy : 0;
    a.taps[3 * i + 1] = i < g.ntaps ? g.taps[i].dx : 0;
    a.taps[3 * i + 2] = i < g.ntaps ? g.taps[i].widx : 0;
  }
  int wgs = workgroups & ~7;                  // a multiple of 8: a workgroup's tiles stay on one XCD's share of the tile order
  if (workgroups <= 0) {
    // as few workgroups as the number of rounds allows: the tile time is bound by the L2 fill rate, which fewer concurrent
    // workgroups share (864 tiles: 4 rounds on 216 workgroups 176 us, on 256 workgroups 202 us)
    const long rounds = (ntiles + 255) / 256;
    wgs = (int)(((ntiles + rounds - 1) / rounds + 7) & ~7L);
    if (wgs > 256) wgs = 256;
  }
  if (wgs > ntiles) wgs = (int)ntiles & ~7;   // every workgroup owns at least one tile (igemm256k_eligible: ntiles >= 8)
  if (wgs < 8) wgs = 8;
  if (a.slab != nullptr) hipLaunchKernelGGL(igemm256k_kernel<true>, dim3((unsigned)wgs), dim3(512), lds, st, a);
  else hipLaunchKernelGGL(igemm256k_kernel<false>, dim3((unsigned)wgs), dim3(512), lds, st, a);
  DC_CHECK_LAUNCH();
  return 0;
}

}  // namespace dc
