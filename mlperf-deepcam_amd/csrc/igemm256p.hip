// Persistent form of the 256 x 256 implicit-GEMM kernel (igemm256.hip) for launches of several rounds: one workgroup per CU walks
// its tiles, and the operand ring never drains between them.
//
// In igemm256_kernel a tile starts cold (tap table, row bookkeeping, three stages issued, one full memory latency before the first
// MFMA) and ends with every wave waiting for its stores to be accepted.  Here
//   * the LDS-DMA slots that igemm256_kernel fills from the zero page past a tile's last stage carry the NEXT tile's first three
//     stages instead (the per-thread row bookkeeping is switched to the next tile at the moment the current tile has issued its
//     last stage), so the next tile's K loop starts with its operands already in LDS, and its first fragments are pre-read by the
//     current tile's last step exactly as inside a tile;
//   * the epilogue runs from the accumulator registers (no LDS: the ring stays live) and its stores drain beside the next tile's
//     first K steps: vmcnt counts in issue order, so those two steps wait vmcnt(6 + NST) (NST = the epilogue's store instructions,
//     made an exact count by sending inactive lanes to a dump line) where the steady state waits vmcnt(6).
// Restrictions (the planner falls back to igemm256_kernel): bf16, no grouped launch, no bias, no accumulate, no BatchNorm-backward
// epilogue, at least 4 K steps in every phase, more tiles than workgroups.  Same MFMA sequence per output element as igemm256_kernel
// (outputs bit-identical); BatchNorm sums as its register epilogue.
#include <type_traits>

#include "igemm.h"

namespace dc {

namespace {

constexpr int TM = 256, TN = 256;      // pixels, channels per tile
constexpr int ROWB = 64;               // bytes of K per row and stage (32 bf16)
constexpr int BK = 32;
constexpr int NSTAGE = 4;
constexpr int OPER = TM * ROWB;        // 16 KiB per operand per stage
constexpr int STAGE = 2 * OPER;
constexpr int RING = NSTAGE * STAGE;   // 128 KiB
constexpr int KG = 4;                  // channel chunks per tap sweep (igemm256.hip)

static __device__ __attribute__((aligned(256))) unsigned char zero_page256p[256];
static __device__ __attribute__((aligned(256))) unsigned char dump_page256p[256 * 64];   // inactive epilogue lanes store here (per lane 256 B)
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

struct TileInfo {
  int n0, m0, phase, tap_beg, ntap;
};

}  // namespace

// The kernel's own argument block: what the tile loop needs of IgemmParams, flat.  The K loops run at the register limit (128
// accumulators + 48 fragment registers per lane, and close to a hundred live scalars), so everything that is only needed BETWEEN
// tiles (tile decode, row bookkeeping, epilogue addressing) is read again from the kernel-argument segment there, through a pointer
// the compiler cannot see through, instead of being held in scalar registers across the loops.
struct Igemm256pArgs {
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
typedef const __attribute__((address_space(4))) Igemm256pArgs* KArgs;

template <class K>
__device__ inline void grid_pixel_k(K k, int m, int& n, int& qy, int& qx) {
  n = k->hwM == 0 ? m : (int)(__umulhi((unsigned)m, k->hwM) >> k->hwS);
  const int rem = m - n * k->QhQw;
  qy = k->wM == 0 ? rem : (int)(__umulhi((unsigned)rem, k->wM) >> k->wS);
  qx = rem - qy * k->Qw;
}

template <bool STATS>
__global__ __launch_bounds__(512) void igemm256p_kernel(const Igemm256pArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* s_tap_all = reinterpret_cast<int*>(smem + RING);      // the layer's tap table (up to 9 x 3 ints), written once
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;   // pixel half; also the stagger group
  const int wc = wave & 3;     // channel quarter
  constexpr int NST = 16 + (STATS ? 2 : 0);   // store instructions of one epilogue, per wave
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

  // ---- per-thread DMA bookkeeping of the tile whose stages are being ISSUED (it runs three stages ahead of the multiplying) -----
  const int lrow = lane >> 2, pslot = lane & 3;
  const int lslot = pslot ^ ((lrow >> 1) & 3);          // logical 16-byte slot this lane fetches (swizzle on the source side)
  const int klim = Cin - lslot * 8;                     // this lane's slot of K chunk kc is inside the row while kc * BK < klim
  const uintptr_t xg = (uintptr_t)a.x, wg = (uintptr_t)a.w, zp = (uintptr_t)a.zero_page;
  int rown[2], ryx[2];         // image row base n * Hin; (qy*is) << 16 | qx*is, with qy*is = 0x7fff for a row past M (never in bounds)
  int in0 = 0;                 // first channel of the issuing tile
  const int* s_tap = s_tap_all;                         // taps of the issuing tile's phase
  unsigned xoff[2] = {0, 0}, woff[2] = {0, 0};          // byte offsets of this lane's rows (K offset 0) from x / w: tensors are below 4 GiB
  bool xok[2] = {false, false}, wok[2] = {false, false};
  int tapA = -1, tapB = -1;
  int itap = 0, ikc = 0, intap = 1;   // (tap, K chunk) of the next stage to issue; taps of the issuing tile
  int kbeg = 0, kend = kchunks < KG ? kchunks : KG;
  bool issuing = true;                // false past the last tile: zero-page fills keep the vmcnt arithmetic uniform
  auto load_rows = [&](const TileInfo& t) {
    const KArgs k = args();
    int lrow_o = lrow;
    asm volatile("" : "+v"(lrow_o));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = t.m0 + (8 * j + wave) * 16 + lrow_o;
      const bool rok = m < k->M;
      const int mm = rok ? m : 0;
      int n, qy, qx;
      grid_pixel_k(k, mm, n, qy, qx);
      rown[j] = n * k->Hin;
      ryx[j] = ((rok ? qy * k->is : 0x7fff) << 16) | (qx * k->is);
    }
    in0 = t.n0;
    intap = t.ntap;
    s_tap = s_tap_all + 3 * t.tap_beg;
    tapA = tapB = -1;
    itap = 0;
    ikc = 0;
    kbeg = 0;
    kend = kchunks < KG ? kchunks : KG;
  };
  auto issue_A = [&](int slot) {
    if (issuing && itap != tapA) {
      tapA = itap;
      const int widx = s_tap[3 * itap + 2];
      int lane_t = lane;             // lane-derived values recomputed here: not worth a register each through the K loop
      asm volatile("" : "+v"(lane_t));
      const int lrow_t = lane_t >> 2, lslot_t = (lane_t & 3) ^ ((lane_t >> 3) & 3);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ch = in0 + (8 * j + wave) * 16 + lrow_t;
        wok[j] = ch < Cout;
        woff[j] = (unsigned)(((widx * Cout + (wok[j] ? ch : 0)) * ldw + lslot_t * 8) * 2);
      }
    }
    const int kofs = ikc * BK;
    const bool kok = issuing & (kofs < klim);
    char* base = smem + slot * STAGE;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((gas_ptr)((kok & wok[j]) ? wg + (woff[j] + kofs * 2) : zp), (lds_ptr)(base + (8 * j + wave) * 16 * ROWB), 16, 0, 0);
  };
  auto issue_B = [&](int slot) {
    if (issuing && itap != tapB) {
      tapB = itap;
      const int dy = s_tap[3 * itap], dx = s_tap[3 * itap + 1];
      int lane_t = lane;
      asm volatile("" : "+v"(lane_t));
      const int lslot_t = (lane_t & 3) ^ ((lane_t >> 3) & 3);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int iy = (ryx[j] >> 16) + dy, ix = (ryx[j] & 0xffff) + dx;
        xok[j] = (unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Win;
        xoff[j] = (unsigned)((((rown[j] + (xok[j] ? iy : 0)) * Win + (xok[j] ? ix : 0)) * ldx + lslot_t * 8) * 2);
      }
    }
    const int kofs = ikc * BK;
    const bool kok = issuing & (kofs < klim);
    char* base = smem + slot * STAGE + OPER;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((gas_ptr)((kok & xok[j]) ? xg + (xoff[j] + kofs * 2) : zp), (lds_ptr)(base + (8 * j + wave) * 16 * ROWB), 16, 0, 0);
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

  // ---- first tile: rows, stages 0..2 in flight, stage 0 landed -------------------------------------------------------------------
  // (the next tile is decoded again where it is needed instead of being carried through the K loop: the loop is at the register limit)
  int round = 0;                   // every workgroup owns a tile of round 0 (launcher: W <= ntiles / 8)
  TileInfo cur = decode(0);
  bool has_next = exists(1);
  __syncthreads();                 // tap table written
  load_rows(cur);
  int base = 0;                    // ring slot of the current tile's stage 0
#pragma unroll
  for (int q = 0; q < NSTAGE - 1; ++q) {     // every phase has at least 4 stages
    issue_A(q);
    issue_B(q);
    advance();
  }
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // stage 0 has landed (stages 1, 2 = 8 instructions stay in flight)
  __builtin_amdgcn_s_barrier();
  vec16 fa[4], fb[4], fb2[4];
  {
    const char* wa = smem;
    const char* xb = wa + OPER;
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const vec16*>(wa + swz64(wc * 64 + i * 16 + fr, fg));
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const vec16*>(xb + swz64(grp * 128 + j * 16 + fr, fg));
  }

  bool first = true;

  // The two waves of a SIMD (w and w+4, the two pixel halves) run the halves of every segment in opposite order (igemm256.hip); the
  // whole tile loop is instantiated twice.
  auto tiles = [&](auto loads_first_tag) {
    constexpr bool LOADS_FIRST = decltype(loads_first_tag)::value;
    for (;;) {
      const int steps = cur.ntap * kchunks;
      auto step = [&](int s) {
        const char* xb = smem + ((base + s) & (NSTAGE - 1)) * STAGE + OPER;
        const char* wa1 = smem + ((base + s + 1) & (NSTAGE - 1)) * STAGE;
        const char* xb1 = wa1 + OPER;
        const int nslot = (base + s + NSTAGE - 1) & (NSTAGE - 1);
        // ---- M0
        if constexpr (LOADS_FIRST) {
#pragma unroll
          for (int j = 0; j < 4; ++j) fb2[j] = *reinterpret_cast<const vec16*>(xb + swz64(grp * 128 + (4 + j) * 16 + fr, fg));
          issue_A(nslot);
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
          issue_A(nslot);
        }
        // stage s+1 landed: all but stage s+2 (4), the weight half of s+3 (2) -- and, in the first two steps behind an epilogue, its
        // NST stores, which were issued between this tile's stage 2 and stage 3
        if (!first && s < 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(6 + NST) : "memory");
        else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- M1
        if constexpr (LOADS_FIRST) {
#pragma unroll
          for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const vec16*>(xb1 + swz64(grp * 128 + j * 16 + fr, fg));
          issue_B(nslot);
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
          issue_B(nslot);
          advance();
        }
      };
      // the tile's own stages are issued by the first steps - 3 steps; then the per-thread bookkeeping switches to the next tile
      // (outside the loops: the switch needs the whole geometry in scalar registers, the loops are at the register limit) and the
      // last three steps fill the ring with the next tile's stages 0..2
      for (int s = 0; s < steps - (NSTAGE - 1); ++s) step(s);
      if (has_next) load_rows(decode(round + 1));
      else issuing = false;
      for (int s = steps - (NSTAGE - 1); s < steps; ++s) step(s);
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
      // the stores above must stay in front of the next tile's LDS-DMA issues in program order: the vmcnt(6 + NST) waits count on it
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (!has_next) break;
      // ---- next tile becomes the current one
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      base = (base + steps) & (NSTAGE - 1);
      first = false;
      ++round;
      cur = decode(round);
      has_next = exists(round + 1);
    }
  };
  if (grp == 0) tiles(std::true_type{});
  else tiles(std::false_type{});
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the zero-page fills of the last slots must land before the LDS is released
}

bool igemm256p_eligible(const IgemmParams& p) {
  const GatherGeom& g = p.g;
  if (p.ngroup > 1 || p.bias != nullptr || p.accumulate || p.bst.y != nullptr || p.m_beg != 0) return false;
  const int kchunks = (g.Cin + BK - 1) / BK;
  for (int ph = 0; ph < g.os * g.os; ++ph)
    if ((g.phase_beg[ph + 1] - g.phase_beg[ph]) * kchunks < NSTAGE) return false;
  // the per-lane row addresses are 32-bit byte offsets from the tensor bases
  if ((size_t)p.N * g.Hin * g.Win * p.ldx * 2 >= (1ull << 32) || (size_t)9 * g.Cout * p.ldw * 2 >= (1ull << 32)) return false;
  return igemm256_tiles(p) >= 8;
}

int launch_igemm256p(const IgemmParams& p_in, int workgroups, hipStream_t st) {
  const size_t lds = (size_t)RING + 256;
  static const void* zero_dev = nullptr;
  static const void* dump_dev = nullptr;
  static hipError_t init_err = hipSuccess;
  auto k0 = &igemm256p_kernel<false>;
  auto k1 = &igemm256p_kernel<true>;
  DC_ONCE({
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    void* zp = nullptr;
    init_err = hipGetSymbolAddress(&zp, HIP_SYMBOL(zero_page256p));
    zero_dev = zp;
    if (init_err == hipSuccess) {
      init_err = hipGetSymbolAddress(&zp, HIP_SYMBOL(dump_page256p));
      dump_dev = zp;
    }
  });
  if (init_err != hipSuccess) return dc_set_error(init_err, __FILE__, __LINE__);
  const GatherGeom& g = p_in.g;
  Igemm256pArgs a;
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
  if (wgs > ntiles) wgs = (int)ntiles & ~7;   // every workgroup owns at least one tile (igemm256p_eligible: ntiles >= 8)
  if (wgs < 8) wgs = 8;
  if (a.slab != nullptr) hipLaunchKernelGGL(igemm256p_kernel<true>, dim3((unsigned)wgs), dim3(512), lds, st, a);
  else hipLaunchKernelGGL(igemm256p_kernel<false>, dim3((unsigned)wgs), dim3(512), lds, st, a);
  DC_CHECK_LAUNCH();
  return 0;
}

}  // namespace dc
