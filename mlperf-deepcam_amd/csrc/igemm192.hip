// Pointwise (1x1, stride 1) conv forward / data gradient for SMALL pixel counts: a 128 pixel x 192 channel tile per 512-thread workgroup.
//
// Why a fourth tile shape.  At local batch 2 (the reference's canonical batch, run_scripts/run_training_dgx2.sh) the 728 -> 728
// pointwise layers of the middle flow are GEMMs with M = 6 912 pixels.  On 128 x 128 tiles that is 54 x 6 = 324 workgroups: 68 CUs hold
// two of them, the LDS fill of a CU is what bounds the K loop (igemm.hip), so the launch runs as long as a CU needs for 2 x 368 KiB of
// operands (24.6 us, 297 TFLOP/s); 256 x 384 and 128 x 384 tiles give 54 / 108 workgroups for 256 CUs.  128 x 192 tiles are
// 54 x 4 = 216 workgroups -- ONE per CU, 480 KiB of operands each -- with the 128-byte K rows and the hand-placed schedule of
// igemm384.hip's K64 mode; because a stage holds only 24 MFMAs per wave (a third of the 256 x 384 tile's) the ring is FOUR 64-deep
// stages deep (three in flight) instead of two, or every stage would wait out a far-memory latency.
//
//   waves     8 = 4 pixel groups of 32 x 2 channel groups of 96: a wave owns 2 x 6 MFMA tiles (48 accumulator registers)
//   stage     64 deep: 192 weight rows + 128 pixel rows of 128 bytes = 40 KiB = 40 LDS-DMA instructions of 8 rows, five per wave
//             (waves 0..3 issue theirs in the first 32-deep half of a stage, waves 4..7 in the second); XOR swizzle as igemm384.hip
//   loop      per half: per channel block the weight fragment two blocks ahead (ds_read_b128), one LDS-DMA, two MFMAs; counted
//             lgkmcnt / vmcnt, ONE barrier per stage
//   epilogue  igemm384.hip's register epilogue (lane-pair exchange, 16-byte stores, BatchNorm sums by DPP folded through LDS into the
//             one 128-pixel slab row of the workgroup)
// Outputs are bit-identical to the other tile shapes (same MFMA, same K order); BatchNorm sums differ in summation order only.
#include <type_traits>

#include "igemm.h"

namespace dc {

namespace {

constexpr int TN = 192;                // channels per workgroup
constexpr int TM = 128;                // pixels per workgroup
constexpr int RB = 128;                // bytes of K per row and stage (64 bf16)
constexpr int KS = 64;
constexpr int NPB = 2;                 // pixel blocks of 16 per wave
constexpr int GP = 4, GC = 2;          // pixel groups x channel groups
constexpr int NCB = TN / 16 / GC;      // channel blocks per wave: 6
constexpr int STAGE = (TN + TM) * RB;  // 40 KiB
constexpr int NI = STAGE / 1024;       // 40
constexpr int IPW = NI / 8;            // 5
#ifndef DC_PW192_STAGES
#define DC_PW192_STAGES 4
#endif
constexpr int NSTG = DC_PW192_STAGES;
constexpr int RING = NSTG * STAGE;
constexpr int BLK = 16 * RB;           // bytes between the fragments of consecutive 16-row blocks
static_assert(NI % 8 == 0 && IPW <= NCB && RING <= 160 * 1024 && NSTG >= 3 && NCB % 3 == 0 && NCB % 2 == 0, "configuration");

static __device__ __attribute__((aligned(256))) unsigned char zero_page192[256];
typedef __attribute__((address_space(1))) const void* gas_ptr;
typedef __attribute__((address_space(3))) void* lds_ptr;

__device__ inline uint32_t swap_rows16(uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F); }
__device__ inline float row_sum16(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));   // row_mirror
  return v;
}
__device__ inline void mfma_v(f32x4& c, const bf16x8& av, const bf16x8& bv) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "v"(bv));
}
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

__global__ __launch_bounds__(512) void pw192_kernel(const IgemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const GatherGeom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave / GC;   // pixel group (32 pixels)
  const int wc = wave % GC;    // channel group (96 channels)
  const bool late = wave >= 4; // the second wave of its SIMD: issues its LDS-DMAs in the second half of a stage

  // XCD-aware tile order: consecutive tiles of an XCD are the channel tiles of one pixel tile (they share its pixel rows in L2)
  const int ntn = (g.Cout + TN - 1) / TN;
  const int nwg = gridDim.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, xslot = blockIdx.x >> 3;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + xslot;
  const int n0 = (tile % ntn) * TN, m0 = (tile / ntn) * TM;
  const int kchunks = (g.Cin + KS - 1) / KS;     // stages

  // ---- LDS-DMA bookkeeping (igemm384.hip, K64 mode): instruction i of this wave fills rows (8 i + wave) * 8 .. + 7 of the stage image,
  // rows [0, 192) weight rows, [192, 320) pixel rows; a lane's source is its operand row or the zero page.
  const int lrow = lane >> 3, pslot = lane & 7;
  const int lslot = pslot ^ ((((wave & 1) << 2) + (lrow >> 1)) & 7);     // logical 16-byte slot this lane fetches (swizzle on the source side)
  const uintptr_t zp = (uintptr_t)p.zero_page;
  const uintptr_t xbase = (uintptr_t)p.x, wbase = (uintptr_t)p.w;
  unsigned src[IPW];
#pragma unroll
  for (int i = 0; i < IPW; ++i) {
    const int r = (8 * i + wave) * 8 + lrow;
    if (r < TN) {          // wave-uniform per instruction: i <= 2
      const int ch = n0 + r;
      src[i] = ch < g.Cout ? (unsigned)(((size_t)ch * p.ldw + lslot * 8) * 2) : ~0u;
    } else {
      const int m = m0 + r - TN;
      src[i] = m < p.M ? (unsigned)(((size_t)m * p.ldx + lslot * 8) * 2) : ~0u;
    }
  }
  auto issue = [&](int i, int stage, int slot) {
    const int kofs = stage * KS;
    const bool ok = (src[i] != ~0u) & (kofs + lslot * 8 < g.Cin);          // stage >= kchunks fails the K test: zero page
    const uintptr_t base = (8 * i + wave) * 8 < TN ? wbase : xbase;        // (scalar)
    const uintptr_t a = ok ? base + (src[i] + (unsigned)kofs * 2u) : zp;
    __builtin_amdgcn_global_load_lds((gas_ptr)a, (lds_ptr)(smem + slot * STAGE + (8 * i + wave) * 1024), 16, 0, 0);
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
  // fragment of block 0, K half 0 (block i: + i * BLK; half 1: ^ 64 bytes)
  const int a_off = (wc * (NCB * 16) + fr) * RB + ((fg ^ ((fr >> 1) & 7)) << 4);
  const int b_off = TN * RB + (grp * (NPB * 16) + fr) * RB + ((fg ^ ((fr >> 1) & 7)) << 4);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr)smem;
  bf16x8 fa[3], fb[NPB];

  // One 32-deep half of a stage.  The fragments come from cur_*; the last block requests the next half's fragments from nxt_*.
  // stage_end: the next half reads ANOTHER ring slot: wait for this wave's LDS-DMAs of that stage, then the barrier.
  // mine: this wave issues the LDS-DMAs of stage `dstage` (ring slot `dslot`) during this half.
  auto half = [&](uint32_t cur_a, uint32_t nxt_a, uint32_t nxt_b, auto stage_end_tag, bool mine, int dstage, int dslot) {
    constexpr bool stage_end = decltype(stage_end_tag)::value;
    static_for<0, NCB>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if constexpr (i + 2 < NCB) lds_read16<(i + 2) * BLK>(fa[(i + 2) % 3], cur_a);      // weight fragment two blocks ahead
      if constexpr (i < IPW) {
        if (mine) issue(i, dstage, dslot);
      }
      if constexpr (i == NCB - 1) {
        if constexpr (stage_end) {
          asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NSTG - 2) * IPW) : "memory");
          __builtin_amdgcn_s_barrier();
        } else {
          lgkm_wait<0>();
        }
        lds_read16<0>(fa[0], nxt_a);                                                    // next half's first two weight fragments
        lds_read16<BLK>(fa[1], nxt_a);
      }
      static_for<0, NPB>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        // counted waits (LDS reads return in order).  Block 0: fa[0], fa[1], fb[0..], then fa[2] are outstanding -> MFMA j needs all but
        // the newest NPB - j.  Blocks in between: fa[i] and the (up to two) fragments requested after it.
        if constexpr (i == 0) lgkm_wait<NPB - j>();
        else if constexpr (j == 0 && i < NCB - 1) lgkm_wait<(NCB - 1 - i < 2 ? NCB - 1 - i : 2)>();
        mfma_v(acc[i][j], fa[i % 3], fb[j]);
        if constexpr (i == NCB - 1) lds_read16<j * BLK>(fb[j], nxt_b);                  // re-read in place for the next half
      });
    });
  };

  // ---- prologue: stages 0 .. NSTG-2 in flight, stage 0 landed, first fragments requested
#pragma unroll
  for (int q = 0; q < NSTG - 1; ++q)
#pragma unroll
    for (int i = 0; i < IPW; ++i) issue(i, q, q);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 2) * IPW) : "memory");
  __builtin_amdgcn_s_barrier();
  lds_read16<0>(fa[0], lds0 + a_off);
  lds_read16<BLK>(fa[1], lds0 + a_off);
  static_for<0, NPB>([&](auto jc) { lds_read16<decltype(jc)::value * BLK>(fb[decltype(jc)::value], lds0 + b_off); });
  int cslot = 0;
  for (int s = 0; s < kchunks; ++s) {
    const int nslot = cslot + 1 == NSTG ? 0 : cslot + 1;
    const int dslot = cslot == 0 ? NSTG - 1 : cslot - 1;          // the slot stage s-1 has left: stage s + NSTG - 1 goes there
    const uint32_t cur = lds0 + cslot * STAGE, nxt = lds0 + nslot * STAGE;
    half(cur + a_off, cur + (a_off ^ 64), cur + (b_off ^ 64), std::false_type{}, !late, s + NSTG - 1, dslot);
    half(cur + (a_off ^ 64), nxt + a_off, nxt + b_off, std::true_type{}, late, s + NSTG - 1, dslot);
    cslot = nslot;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // the zero-page fills of the last slots; the ring is reused below
  __builtin_amdgcn_s_barrier();

  // ---- epilogue from the accumulator registers (igemm384.hip) --------------------------------------------------------------------
  const bool odd = fg & 1;
  bf16* __restrict__ yg = reinterpret_cast<bf16*>(p.y);
  const bool do_stats = p.slab != nullptr;
  float* red = reinterpret_cast<float*>(smem);      // [pixel group][sum, sum of squares][192]
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
      if (m < p.M && chok) {
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
    // the workgroup's 128 pixels are ONE slab row: the four pixel groups folded in a fixed order
    const int rows = p.mtiles, mt128 = m0 >> 7;
    for (int i = tid; i < 2 * TN; i += 512) {
      const int c = i % TN, which = i / TN;
      if (n0 + c >= g.Cout) continue;
      const float v = (red[(0 * 2 + which) * TN + c] + red[(1 * 2 + which) * TN + c]) + (red[(2 * 2 + which) * TN + c] + red[(3 * 2 + which) * TN + c]);
      if (rows < 0) unsafeAtomicAdd(reinterpret_cast<double*>(p.slab) + (size_t)which * g.Cout + n0 + c, (double)v);      // a sum row (bn_fin.h)
      else if (mt128 < rows) p.slab[((size_t)which * rows + mt128) * g.Cout + n0 + c] = v;
    }
  }
}

}  // namespace

long pw192_tiles(const IgemmParams& p) { return (long)((p.g.Cout + TN - 1) / TN) * ((p.M + TM - 1) / TM); }

int launch_pw192(const IgemmParams& p_in, hipStream_t st) {
  static const void* zero_dev = nullptr;
  static hipError_t init_err = hipSuccess;
  DC_ONCE({
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pw192_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, RING);
    void* zp = nullptr;
    init_err = hipGetSymbolAddress(&zp, HIP_SYMBOL(zero_page192));
    zero_dev = zp;
  });
  if (init_err != hipSuccess) return dc_set_error(init_err, __FILE__, __LINE__);
  IgemmParams p = p_in;
  p.zero_page = zero_dev;
  hipLaunchKernelGGL(pw192_kernel, dim3((unsigned)pw192_tiles(p)), dim3(512), RING, st, p);
  DC_CHECK_LAUNCH();
  return 0;
}

}  // namespace dc
