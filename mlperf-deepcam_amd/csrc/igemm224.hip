// Pointwise (1x1, stride 1) conv forward / data gradient as a GEMM on a 224 pixel x 384 channel tile, weights from the [k][n] packing.
//
// Why a fourth tile shape (round 6).  pw384_kernel (igemm384.hip) runs the 728 -> 728 layers of the middle flow -- 27.9 % of the network's FLOPs --
// as 216 tiles of 256 x 384 on 256 CUs in 43 us: 4 us of prologue (an 80 KiB first stage), a 25 us K loop against 16.8 us of MFMA, 8.5 us of
// output burst.  Its 64-deep stages of 128-byte K rows (whole L2 lines per LDS-DMA row piece) are 80 KiB each, so the ring is TWO stages = the
// CU's whole LDS and only one stage is ever in flight: the fill runs in bursts, ~40 KiB in flight on average, and the probes of round 3 showed
// fill and MFMA time adding up instead of overlapping.  What changes here:
//
//   weights   come from the OTHER packing of the layer, [k][n] (for the forward pass wb = [cin][cout], for the data gradient wf = [cout][cin]:
//             both exist, dc_pack_all writes them).  A row of k is 384 contiguous channels = 768 B = six whole L2 lines, so a weight stage
//             can be 32 deep and still move whole lines: 24 KiB instead of 48.  The MFMA's A fragment (channel per lane, 8 consecutive k)
//             comes out of the [k][n] image through ds_read_b64_tr_b16, as wgrad384.hip reads its operands: quads of 64 channels,
//             [quad][32 k rows][128 B], the four 32-byte chunks of a row XOR-ed with key(row) on the DMA's source address.  Lane group fg
//             reads rows 8 fg .. 8 fg + 3 and 8 fg + 4 .. 8 fg + 7 (k = 8 fg + e as in every other GEMM kernel here: same MFMA, same K order,
//             outputs bit for bit those of the other tile shapes), so a half-wave touches rows {0..3, 8..11} and the conflict-free key is
//             ((row >> 1) & 1) | (((row >> 3) & 1) << 1); both reads of a fragment share a base register (512 B apart).
//   pixels    stay [pixel][128 B] in 64-deep stages (pw384's K64 image and swizzle, ds_read_b128 fragments re-read in place).
//   ring      THREE weight stages (3 x 24 KiB) and THREE pixel stages (3 x 28 KiB) = 156 KiB: W[s+2] and half of P[S+2] are issued during
//             step s, so 52 - 92 KiB are in flight at any time, and the first MFMA waits for 52 KiB instead of 80.
//   tile      224 pixels: 27 648 pixels are 124 x 2 = 248 tiles, i.e. 248 of the 256 CUs work (216 before) on 12.5 % fewer MFMAs and 5 % fewer
//             fill bytes each.  8 waves = 2 pixel groups of 112 (7 blocks of 16) x 4 channel groups of 96 (6 blocks): 168 accumulator
//             registers per wave.  Wave wc owns chunk pair wc & 1 (32 channels) of the quads (wc >> 1) + 2 u, u = 0..2: the two 16-channel
//             blocks of a pair are adjacent, so the register epilogue's lane-pair exchange still stores 64 contiguous bytes per pixel, and a
//             wave's six A fragments are two lane-constant bases plus immediates.
//   schedule  igemm384.hip's: fragments two channel blocks ahead, counted lgkmcnt, the two waves of a SIMD issue their LDS-DMAs in different
//             blocks of a step, ONE barrier per 32-deep step in front of its last channel block.
//   hazards   RAW: in-order vmcnt.  A wave's issue order is ... W[s+1] | P-part(s), W[s+2]; at the end of step s it waits
//             vmcnt(3 + P-part(s)): W[s+1] and everything older (all of P[S+1]) have landed; then the barrier.  WAR: W[s+2] goes into the slot
//             of W[s-1], P[S+2] into the slot of P[S-1]; their last reads were waited for (lgkmcnt(0)) before the barrier of step s-1 / 2S-1.
//             The last two steps issue nothing and wait vmcnt(0); pixel stages past the K extent are filled with zeros (every lane out of range).
//   slab      BatchNorm partial sums: ONE row per tile (row = the tile's pixel-tile index) in the layer's usual slab of cdiv(M, 128) rows; the
//             rows no tile owns are written as zeros by the first tiles, so the finalize kernels need not know the tile shape.
#include <type_traits>

#include "igemm.h"

namespace dc {

namespace {

constexpr int NWS = 3, NPS = 3;              // ring depths
constexpr int NPB = 7;                       // pixel blocks of 16 per wave

typedef __attribute__((address_space(1))) const void* gas_ptr;
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__device__ inline uint32_t swap_rows16(uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F); }
__device__ inline float row_sum16(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));   // row_mirror
  return v;
}

// MFMA, LDS fragment reads and their waits as (volatile) statements in one hand-placed order (igemm384.hip explains why)
__device__ inline void mfma_v(f32x4& c, const bf16x8& av, const bf16x8& bv) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "v"(bv));
}
template <int OFF>
__device__ inline void lds_read16(bf16x8& dst, uint32_t addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
// one A fragment (16 channels x 32 k, channel per lane) = two transposing reads: k rows 8 fg .. + 3 and 8 fg + 4 .. + 7 (512 B apart)
template <int OFF>
__device__ inline void lds_read_tr(u32x4& dst, uint32_t addr) {
  static_assert(OFF >= 0 && OFF + 512 < 65536, "ds_read offset field");
  u32x2 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(OFF) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(OFF + 512) : "memory");
  dst[0] = lo[0]; dst[1] = lo[1]; dst[2] = hi[0]; dst[3] = hi[1];
}
template <int N>
__device__ inline void lgkm_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
template <int N>
__device__ inline void vm_lgkm0_wait() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}

template <int I, int N, typename F>
__device__ inline void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// diagnostic build (make stamps224; scripts/pw224_stamps.py): wave 0 of every workgroup leaves the 100 MHz real-time counter at kernel entry,
// behind the prologue's barrier, behind the K loop and behind the drained stores, and the shader clock around the loop
#ifdef DC_PW224_STAMPS
__device__ unsigned long long pw224_stamp_buf[1024][8];
#define PW224_STAMP(i) do { if (threadIdx.x == 0) { pw224_stamp_buf[blockIdx.x & 1023][i] = __builtin_amdgcn_s_memrealtime(); \
                                                   pw224_stamp_buf[blockIdx.x & 1023][4 + (i)] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define PW224_STAMP(i)
#endif

// TN: channels per workgroup, 384 (six quads of 64; a wave owns 96 = six blocks of 16) or 192 (three quads; a wave owns 48 = three blocks):
// 224 x 192 is the tile for HALF as many pixels (local batch 4: 13 824 pixels x 728 channels = 62 x 4 = 248 tiles), 26 KiB of operands per
// 32-deep step instead of the 32 KiB of 128 x 384.
template <int TN_>
struct Cfg {
  static constexpr int TN = TN_;
  static constexpr int GP = 2;                            // pixel groups of 112
  static constexpr int GC = 8 / GP;                       // channel groups: 4
  static constexpr int NCB = TN / 16 / GC;                // channel blocks per wave: 6 / 3
  static constexpr int TM = GP * NPB * 16;                // pixels per workgroup: 224
  static constexpr int WSTG = 32 * TN * 2;                // one 32-deep weight stage: TN / 64 quads of [32 k][64 n] = 24 / 12 KiB
  static constexpr int NWI = WSTG / 1024;                 // LDS-DMA instructions per weight stage: 24 / 12
  static constexpr int WSLOTS = (NWI + 7) / 8;            // per wave: 3 / 2 (the last one only on waves < NWI - 8 * (WSLOTS - 1))
  static constexpr int WDEF = (NWI % 8) ? 1 : 0;          // waves 4..7 own one weight instruction fewer
  static constexpr int PSTG = TM * 128;                   // one 64-deep pixel stage: 28 KiB
  static constexpr int NPI = PSTG / 1024;                 // LDS-DMA instructions per pixel stage: 28
  static constexpr int PSLOTS = (NPI + 7) / 8;            // per wave: 4 (the last one only on waves < NPI - 8 * (PSLOTS - 1) = 4)
  static constexpr int HP = PSLOTS / 2;                   // per wave and 32-deep step: 2
  static constexpr int POFF = NWS * WSTG;                 // LDS offset of the pixel ring
  static constexpr int RING = NWS * WSTG + NPS * PSTG;    // 156 / 120 KiB
  static_assert(PSLOTS % 2 == 0 && NCB % 3 == 0 && RING <= 160 * 1024, "plan");
  static_assert(NPI - 8 * (PSLOTS - 1) == 4 && (WDEF == 0 || NWI - 8 * (WSLOTS - 1) == 4), "waves 0..3 own every last slot");
};

template <int TNT>
__global__ __launch_bounds__(512) void pw224_kernel(const IgemmParams p) {
  typedef Cfg<TNT> K;
  constexpr int NCB = K::NCB, GC = K::GC, HP = K::HP, TN = K::TN, WSTG = K::WSTG, GP = K::GP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const GatherGeom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63;
  PW224_STAMP(0);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave / GC;   // pixel group (112 pixels)
  const int wc = wave % GC;    // channel group.  TN 384: chunk pair wc & 1 of the quads (wc >> 1) + 2 u; TN 192: the chunks 3 wc .. 3 wc + 2
  const bool late = wave >= 4; // the second wave of its SIMD: issues its LDS-DMAs in later blocks of a step

  // XCD-aware tile order: consecutive tiles of an XCD are the channel tiles of one pixel tile (they share its pixel rows in L2)
  const int ntn = (g.Cout + TN - 1) / TN;
  const int nwg = gridDim.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, xslot = blockIdx.x >> 3;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + xslot;
  const int tm = tile / ntn;
  const int n0 = (tile % ntn) * TN, m0 = tm * K::TM;
  const int nsteps = (g.Cin + 31) / 32;       // 32-deep steps (>= 4: eligibility)

  // ---- LDS-DMA bookkeeping --------------------------------------------------------------------------------------------------
  // Weight stage (24 instructions): instruction id = 4 quad + rg fills k rows 8 rg .. + 7 of quad `quad`; wave w issues ids w, w + 8, w + 16,
  // i.e. rg = w & 3 and quads (w >> 2) + 2 i.  Lane: k row 8 rg + (lane >> 3), 16-byte piece lane & 7 of the 128-byte row = half (piece & 1) of
  // physical chunk piece >> 1, which holds logical chunk (piece >> 1) ^ key(row).
  // Pixel stage (28 instructions): id q fills pixel rows 8 q .. + 7; wave w issues q = w + 8 i (i = 3 only on waves < 4).  Lane: row 8 q +
  // (lane >> 3), physical slot lane & 7 holds logical slot (lane & 7) ^ ((row >> 1) & 7) (pw384's K64 swizzle).
  // Both operands travel as `buffer_load_dwordx4 ... offen lds`: a lane's constant 32-bit byte offset in a VGPR, the stage's K offset in an SGPR,
  // bounds from the buffer resource -- no per-instruction vector arithmetic in the K loop (the 64-bit address + zero-page select of
  // global_load_lds cost ~8 VALU instructions per LDS-DMA, 80 per step and SIMD, on a SIMD whose vector issue the MFMAs already half fill).
  // A lane out of range (a pixel row past M: its offset is past the resource's extent) gets ZEROS written to its LDS bytes.  Weight columns past
  // Cout need nothing: they only reach output channels that are never stored.  The K tail (k >= Cin: weight rows past the image, pad elements of
  // a pixel row) lies in the LAST stage of either operand only; that stage's offsets are formed in VGPRs (the range check then sees the whole
  // offset whatever it does with the SGPR part) and a pixel lane whose K slot is past Cin is sent out of range.
  const int krow = 8 * (wave & 3) + (lane >> 3);
  const int wkey = ((krow >> 1) & 1) | (((krow >> 3) & 1) << 1);
  unsigned srcw[K::WSLOTS], srcp[K::PSLOTS];
#pragma unroll
  for (int i = 0; i < K::WSLOTS; ++i) {
    const int ncol = ((wave >> 2) + 2 * i) * 64 + ((((lane & 7) >> 1) ^ wkey) << 4) + (lane & 1) * 8;
    srcw[i] = (unsigned)(((size_t)krow * p.ldw_kn + n0 + ncol) * 2);
  }
  const int pslot = (lane & 7) ^ (((4 * wave) + (lane >> 4)) & 7);     // logical 16-byte K slot this lane fetches
#pragma unroll
  for (int i = 0; i < K::PSLOTS; ++i) {
    const int m = m0 + 8 * (wave + 8 * i) + (lane >> 3);
    srcp[i] = (unsigned)(((size_t)m * p.ldx + pslot * 8) * 2);          // m >= M: past the resource's extent
  }
  const bool has_last = wave < 4;      // (wave-uniform) this wave owns the last pixel instruction and, where a weight stage is 12, the last weight one
  const unsigned wstep = (unsigned)p.ldw_kn * 64u;                     // bytes between the k rows of consecutive steps (32 rows)
  const int last_S = (nsteps - 1) >> 1;
  constexpr unsigned OOB = 0x80000000u;                                // resources are smaller than 2 GiB (eligibility)
#if defined(__HIP_DEVICE_COMPILE__)     // (the buffer-resource builtins do not exist in the host pass)
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_kn, 0, (int)((size_t)g.Cin * p.ldw_kn * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)(((size_t)(p.M - 1) * p.ldx + g.Cin) * 2), 0x00020000);
  // wdst / pdst: LDS byte offset of the ring slot the stage goes to.  The K offset of a stage rides in the SGPR operand, except in an operand's
  // last stage (and past it), where it is added to the lane's offset: one scalar select per stage, one v_add per instruction, no branch.
  const unsigned ptail = 64 * last_S + pslot * 8 < g.Cin ? 0u : OOB;    // last pixel stage: this lane's K slot is past Cin
  auto issue_w = [&](int i, int s, uint32_t wdst) {
    const bool last = s >= nsteps - 1;
    const unsigned kb = (unsigned)s * wstep;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(smem + wdst + (wave + 8 * i) * 1024), 16, srcw[i] + (last ? kb : 0u), last ? 0u : kb, 0, 0);
  };
  auto issue_p = [&](int i, int S, uint32_t pdst) {
    const bool last = S >= last_S;
    const unsigned kb = (unsigned)S * 128u;
    const unsigned vk = (last ? kb : 0u) + (S > last_S ? OOB : 0u);     // (scalar) a stage past the K extent: every lane out of range
    const unsigned vt = S == last_S ? ptail : 0u;
#ifndef DC_PW224_AUXP
#define DC_PW224_AUXP 0
#endif
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(smem + pdst + (wave + 8 * i) * 1024), 16, srcp[i] + vk + vt, last ? 0u : kb, 0, DC_PW224_AUXP);
  };
#else
  auto issue_w = [&](int, int, uint32_t) {};
  auto issue_p = [&](int, int, uint32_t) {};
#endif

#ifdef DC_PW224_PROBE
  // diagnostic builds (make probes224), one library per COMPILE-TIME mask: bit 0 drops the K loop's LDS-DMA issues, bit 1 its LDS fragment reads,
  // bit 2 its MFMAs -- what each costs in the loop.  Results are garbage by construction.
  constexpr int probe = DC_PW224_PROBE;
#else
  constexpr int probe = 0;
#endif
  f32x4 acc[NCB][NPB];   // [channel block][pixel block]
#pragma unroll
  for (int i = 0; i < NCB; ++i)
#pragma unroll
    for (int j = 0; j < NPB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fg = lane >> 4;
  // A fragment of channel block i: quad (wc >> 1) + 2 (i >> 1), chunk 2 (wc & 1) + (i & 1); the lane supplies k row 8 fg + (fr >> 2) (its key is
  // (fr >> 3) | ((fg & 1) << 1), the same for the row 4 further) and the 8-byte column group fr & 3 of the chunk
  const int akey = ((fr >> 3) & 1) | ((fg & 1) << 1);
  const int arow = 8 * fg + (fr >> 2);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr)smem;
  // TN 384: two lane-constant bases (block parity) + the immediate (i >> 1) * 8192.  TN 192: block i is chunk 3 wc + i, i.e. quad (3 wc + i) >> 2,
  // position (3 wc + i) & 3: one base per block.
  constexpr int NAB = TN == 384 ? 2 : NCB;
  uint32_t a_base[NAB];
#pragma unroll
  for (int e = 0; e < NAB; ++e) {
    const int quad = TN == 384 ? (wc >> 1) : (3 * wc + e) >> 2;
    const int pos = TN == 384 ? 2 * (wc & 1) + e : (3 * wc + e) & 3;
    a_base[e] = lds0 + quad * 4096 + arow * 128 + ((pos ^ akey) << 5) + (fr & 3) * 8;
  }
  // B fragment of pixel block j, K half h: row grp * 112 + 16 j + fr, logical slot 4 h + fg
  const uint32_t b_off = K::POFF + (grp * (NPB * 16) + fr) * 128 + ((fg ^ ((fr >> 1) & 7)) << 4);      // (without lds0: K half 1 is b_off ^ 64)
  static_assert(NCB % 3 == 0, "fa ring");
  u32x4 fa[3];
  bf16x8 fb[NPB];

  // One 32-deep step s (H = s & 1 at compile time).  Fragments of this step come from wcur (weight slot) and are already requested for
  // channel blocks 0, 1 and all pixel blocks; the last block requests the next step's from wnxt / bnxt.  dma: issue W[s+2] and the H half of
  // P[(s >> 1) + 2] during this step and wait for W[s+1] at its end; otherwise (the last two steps) issue nothing and wait for everything.
  // Outstanding LDS reads at the top of a step, oldest first: fa[0] (2 instructions), fa[1] (2), fb[0] .. fb[6].
  // wdst / pdst: the ring slots (LDS byte offsets) of W[s+2] and P[(s >> 1) + 2].
  // The last step's look-ahead reads have no consumer: the wait behind the loop must be the FIRST instruction on both exit paths (no register may
  // be handed to another value while an LDS read is still on its way to it) -- tests/test_kernel_isa_cpu.py checks the compiled kernel for it.
  auto step = [&](auto h_tag, uint32_t wcur, uint32_t wnxt, uint32_t bnxt, int s, bool dma, uint32_t wdst, uint32_t pdst) {
    constexpr int H = decltype(h_tag)::value;
    const int S = s >> 1;
    // this step's LDS-DMAs in issue order: pixel slots 2 H, 2 H + 1, then the weight instructions (the last slot of either only on waves 0..3)
    auto dma_k = [&](auto kc) {
      constexpr int k = decltype(kc)::value;
      if constexpr (k < HP) {
        constexpr int slot = HP * H + k;
        if constexpr (!(probe & 16)) { if (slot < K::PSLOTS - 1 || has_last) issue_p(slot, S + 2, pdst); }
      } else {
        constexpr int wi = k - HP;
        if constexpr (!(probe & 8)) { if (wi < K::WSLOTS - K::WDEF || has_last) issue_w(wi, s + 2, wdst); }
      }
    };
    constexpr int ND = HP + K::WSLOTS;                 // 5 / 4
    // blocks in which the early waves (0..3) and the late waves (4..7) issue instruction k
#ifndef DC_PW224_PLAN
#define DC_PW224_PLAN 0
#endif
    constexpr int EB[5] = {0, 0, 1, 1, NCB == 6 ? 2 : 1}, LB[5] = {NCB == 6 ? 2 : 1, NCB == 6 ? 3 : 1, NCB == 6 ? 3 : 2, NCB == 6 ? 4 : 2, NCB == 6 ? 4 : 2};
    static_assert(ND <= 5, "issue plan");
    static_for<0, NCB>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if constexpr (i + 2 < NCB && !(probe & 2)) {      // two blocks ahead
        if constexpr (TN == 384) lds_read_tr<((i + 2) >> 1) * 8192>(fa[(i + 2) % 3], wcur + a_base[(i + 2) & 1] - lds0);
        else lds_read_tr<0>(fa[(i + 2) % 3], wcur + a_base[i + 2] - lds0);
      }
      if (dma && !(probe & 1)) {
        static_for<0, ND>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          if constexpr (EB[k] == i) { if (!late) dma_k(kc); }
          if constexpr (LB[k] == i) { if (late) dma_k(kc); }
        });
      }
      if constexpr (i == NCB - 1) {
        // stage end: W[s+1] (and with it everything older) has landed for this wave; this step's own instructions stay in flight
        if (dma && !(probe & 1)) {
          // (all but this step's own instructions: pixel slots + weight slots, one fewer of each kind where waves 4..7 own no last slot)
          constexpr int NW = (probe & 8) ? 0 : K::WSLOTS, NP = (probe & 16) ? 0 : HP;      // (probe builds that drop one operand's LDS-DMAs)
          constexpr int DW = (probe & 8) ? 0 : K::WDEF, DP = (probe & 16) || H == 0 ? 0 : 1;
          if (has_last) vm_lgkm0_wait<NW + NP>(); else vm_lgkm0_wait<NW + NP - DW - DP>();
        } else {
          vm_lgkm0_wait<0>();
        }
        __builtin_amdgcn_s_barrier();
        if constexpr (!(probe & 2)) {
          lds_read_tr<0>(fa[0], wnxt + a_base[0] - lds0);                              // next step's first two weight fragments
          lds_read_tr<0>(fa[1], wnxt + a_base[1] - lds0);
        }
      }
      static_for<0, NPB>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        // counted waits (LDS reads return in order).  Block 0: fa[0] (2), fa[1] (2), fb[0..6], then fa[2] (2) are outstanding: MFMA j needs all
        // but the newest (NPB - 1 - j) + 2.  Blocks 1 .. NCB-3: fa[i] and the two fragments (4 instructions) requested after it; block NCB-2:
        // one fragment (2) after it; the last block waited for everything in front of the barrier.
        if constexpr (i == 0) lgkm_wait<NPB - 1 - j + 2>();
        else if constexpr (j == 0 && i < NCB - 2) lgkm_wait<4>();
        else if constexpr (j == 0 && i == NCB - 2) lgkm_wait<2>();
        if constexpr (!(probe & 4)) mfma_v(acc[i][j], __builtin_bit_cast(bf16x8, fa[i % 3]), fb[j]);
        if constexpr (i == NCB - 1 && !(probe & 2)) lds_read16<j * 2048>(fb[j], bnxt);   // re-read in place for the next step
      });
    });
  };

#ifndef DC_PW224_PRIO
#define DC_PW224_PRIO 1      // the second-dispatched wave of every SIMD loses each arbitration at equal priority: 1 965 -> 1 924 cycles per step
#endif
  if (late) __builtin_amdgcn_s_setprio(DC_PW224_PRIO);
  // ---- prologue: P[0] (the operand from beyond the L2 first), W[0], P[1], W[1] in flight (W[1] last: see "hazards"), P[0] and W[0] landed,
  // first fragments requested
#pragma unroll
  for (int q = 0; q < 2; ++q) {
#pragma unroll
    for (int i = 0; i < K::PSLOTS; ++i)
      if (i < K::PSLOTS - 1 || has_last) issue_p(i, q, K::POFF + q * K::PSTG);
#pragma unroll
    for (int i = 0; i < K::WSLOTS; ++i)
      if (i < K::WSLOTS - K::WDEF || has_last) issue_w(i, q, q * WSTG);
  }
  if (has_last) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K::WSLOTS + K::PSLOTS) : "memory");
  else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K::WSLOTS - K::WDEF + K::PSLOTS - 1) : "memory");
  __builtin_amdgcn_s_barrier();
  PW224_STAMP(1);
  lds_read_tr<0>(fa[0], a_base[0]);
  lds_read_tr<0>(fa[1], a_base[1]);
  static_for<0, NPB>([&](auto jc) { lds_read16<decltype(jc)::value * 2048>(fb[decltype(jc)::value], lds0 + b_off); });
  {
    int s = 0;
    // weight slots (byte offsets) of steps s, s + 1, s + 2; pixel slots (byte offsets inside the pixel ring) of stages S, S + 1, S + 2
    uint32_t w0 = 0, w1 = WSTG, w2 = 2 * WSTG;      // (WSTG = K::WSTG)
    uint32_t p0 = 0, p1 = K::PSTG, p2 = 2 * K::PSTG;
    typedef std::integral_constant<int, 0> H0;
    typedef std::integral_constant<int, 1> H1;
    while (true) {
      step(H0{}, lds0 + w0, lds0 + w1, lds0 + ((b_off + p0) ^ 64), s, s + 2 < nsteps, w2, K::POFF + p2);
      if (++s == nsteps) break;
      step(H1{}, lds0 + w1, lds0 + w2, lds0 + b_off + p1, s, s + 2 < nsteps, w0, K::POFF + p2);
      if (++s == nsteps) break;
      { const uint32_t t = w0; w0 = w2; w2 = w1; w1 = t; }      // two steps on: (w0, w1, w2) <- (w2, w0, w1)
      { const uint32_t t = p0; p0 = p1; p1 = p2; p2 = t; }      // one stage on
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (the ring is reused below)

  __builtin_amdgcn_s_barrier();
  PW224_STAMP(2);

  // ---- epilogue from the accumulator registers (igemm384.hip's) -------------------------------------------------------------------
  // A lane holds, per MFMA tile (i, j), channels fg*4 .. +3 of channel block i for pixel fr of pixel block j.  Blocks 2 pr and 2 pr + 1 are the
  // two halves of one 32-channel chunk pair: lanes l and l ^ 16 trade halves so that the even-fg lane keeps 8 consecutive channels of block
  // 2 pr and the odd-fg lane 8 of block 2 pr + 1 (64 contiguous bytes per pixel and store instruction).
  const bool odd = fg & 1;
  bf16* __restrict__ yg = reinterpret_cast<bf16*>(p.y);
  const bool do_stats = p.slab != nullptr;
  float* red = reinterpret_cast<float*>(smem);      // [pixel group][sum, sum of squares][384]
#pragma unroll
  for (int pr = 0; pr < NCB / 2; ++pr) {
    const int i0 = 2 * pr;
    const int cpair = TN == 384 ? ((wc >> 1) + 2 * pr) * 64 + (wc & 1) * 32 : 48 * wc;   // first channel of the pair inside the tile
    const int chl = cpair + (odd ? 16 : 0) + (fg >> 1) * 8;                        // first of this lane's 8 channels after the trade
    const int ch0 = n0 + chl;
    const bool chok = ch0 < g.Cout;                                               // Cout is a multiple of 8: all or nothing
    float ba[4] = {0.f, 0.f, 0.f, 0.f}, bb[4] = {0.f, 0.f, 0.f, 0.f};             // bias of the channels this lane COMPUTED
    if (p.bias != nullptr) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ca = n0 + cpair + fg * 4 + r, cb = ca + 16;
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
  if constexpr (NCB & 1) {
    // TN 192: the wave's third block (16 channels) has no partner block.  Lanes l and l ^ 16 trade PIXEL blocks instead: the even-fg lane keeps 8
    // consecutive channels of pixel block 2 jj, the odd-fg lane the same 8 channels of pixel block 2 jj + 1 (the last pixel block: even lanes only).
    constexpr int i2 = NCB - 1;
    const int chl = 48 * wc + 32 + (fg >> 1) * 8;
    const int ch0 = n0 + chl;
    const bool chok = ch0 < g.Cout;
    float ba[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ca = n0 + 48 * wc + 32 + fg * 4 + r;
        if (ca < g.Cout) ba[r] = p.bias[ca];
      }
    }
    float st[2][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) st[0][e] = st[1][e] = 0.f;
#pragma unroll
    for (int jj = 0; jj < (NPB + 1) / 2; ++jj) {
      const int j0 = 2 * jj, j1 = 2 * jj + 1 < NPB ? 2 * jj + 1 : 2 * jj;       // (no partner pixel block: j1 repeats j0, the odd lanes store nothing)
      const bool has1 = 2 * jj + 1 < NPB;
      const uint32_t a0 = pack2_bf16(acc[i2][j0][0] + ba[0], acc[i2][j0][1] + ba[1]);
      const uint32_t a1 = pack2_bf16(acc[i2][j0][2] + ba[2], acc[i2][j0][3] + ba[3]);
      const uint32_t b0 = pack2_bf16(acc[i2][j1][0] + ba[0], acc[i2][j1][1] + ba[1]);
      const uint32_t b1 = pack2_bf16(acc[i2][j1][2] + ba[2], acc[i2][j1][3] + ba[3]);
      const uint32_t r0 = swap_rows16(odd ? a0 : b0), r1 = swap_rows16(odd ? a1 : b1);
      vec16 v;
      v.w[0] = odd ? r0 : a0;
      v.w[1] = odd ? r1 : a1;
      v.w[2] = odd ? b0 : r0;
      v.w[3] = odd ? b1 : r1;
      const int m = m0 + grp * (NPB * 16) + (odd ? j1 : j0) * 16 + fr;
      if (m < p.M && chok && (!odd || has1)) {
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
      // the two lanes of a pair hold sums of the SAME 8 channels over different pixel blocks: add them, then sum over the 16 pixel lanes
      float mine = 0.f;
#pragma unroll
      for (int w = 0; w < 2; ++w)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float both = st[w][e] + __uint_as_float(swap_rows16(__float_as_uint(st[w][e])));
          const float t = row_sum16(both);
          if (fr == w * 8 + e) mine = t;
        }
      if (!odd) red[(grp * 2 + (fr >> 3)) * TN + chl + (fr & 7)] = mine;
    }
  }
  if (do_stats) {
    __syncthreads();
    // one slab row per tile: the pixel groups folded in a fixed order; slab rows no tile owns (the slab has a row per 128 pixels) are zeros
    const int rows = p.mtiles, ntm = (p.M + K::TM - 1) / K::TM;
    for (int i = tid; i < 2 * TN; i += 512) {
      const int c = i % TN, which = i / TN;
      if (n0 + c < g.Cout) {
        float v = red[which * TN + c];
#pragma unroll
        for (int gq = 1; gq < GP; ++gq) v += red[(gq * 2 + which) * TN + c];
        if (rows < 0) {      // a sum row (bn_fin.h: SUM_ROW): double[2][Cout], zeroed by the caller, added to by every tile
          unsafeAtomicAdd(reinterpret_cast<double*>(p.slab) + (size_t)which * g.Cout + n0 + c, (double)v);
          continue;
        }
        float* col = p.slab + (size_t)which * rows * g.Cout + n0 + c;
        col[(size_t)tm * g.Cout] = v;
        for (int r = tm + ntm; r < rows; r += ntm) col[(size_t)r * g.Cout] = 0.f;
      }
    }
  }
#ifdef DC_PW224_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  PW224_STAMP(3);
#endif
}

}  // namespace

bool pw224_eligible(const IgemmParams& p) {
  const GatherGeom& g = p.g;
  return p.w_kn != nullptr && g.ntaps == 1 && g.os == 1 && g.is == 1 && g.taps[0].dy == 0 && g.taps[0].dx == 0 && p.m_beg == 0 && p.ngroup <= 1 &&
         g.Cin >= 128 && g.Cin % 8 == 0 && g.Cout % 8 == 0 && (((uintptr_t)p.w_kn) & 15) == 0 &&
         ((size_t)p.M + 224) * p.ldx * 2 < (1ull << 31) && ((size_t)g.Cin + 64) * p.ldw_kn * 2 < (1ull << 31);      // buffer resources < 2 GiB
}

// tn: channels per tile, 384 or 192
int launch_pw224(const IgemmParams& p, int tn, hipStream_t st) {
  auto k384 = &pw224_kernel<384>;
  auto k192 = &pw224_kernel<192>;
  DC_ONCE({
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k384), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<384>::RING);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k192), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<192>::RING);
  });
  if (tn == 192) hipLaunchKernelGGL(k192, dim3((unsigned)pw224_tiles(p, 192)), dim3(512), Cfg<192>::RING, st, p);
  else hipLaunchKernelGGL(k384, dim3((unsigned)pw224_tiles(p, 384)), dim3(512), Cfg<384>::RING, st, p);
  DC_CHECK_LAUNCH();
  return 0;
}

}  // namespace dc

#ifdef DC_PW224_STAMPS
extern "C" int dc_debug_pw224_stamps(void* host_out, int blocks) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(dc::pw224_stamp_buf), (size_t)blocks * 8 * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif
