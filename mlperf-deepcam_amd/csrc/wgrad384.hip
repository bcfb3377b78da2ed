// Pointwise (1x1, stride 1) conv weight gradient on a 256 (co) x 384 (ci) tile per 512-thread workgroup (bf16).
//
//     dW[co][ci] = sum over pixels m of dy[m][co] * x[m][ci]
//
// Why a second tile shape.  The 256 x 256 kernel of rounds 2 - 4 (wgrad256.hip, retired in round 6) streamed BOTH operands of its 256 x 256 tile from beyond the L2 (activations and their
// gradients are each read once per channel tile and nothing else), and a CU takes those bytes in at ~24 GB/s whatever the ring depth
// (profiles/r04_wgrad_group_probe.txt: 170 us for the three-layer 728 launch, 109 with L2-resident operands, 84 MFMA only).  What is
// left is bytes per flop: a 256 x 384 tile needs 40 KiB per 32-pixel stage for 3.1 M MACs against 32 KiB for 2.1 M (-17 %), and the
// 728-channel layers become 3 x 2 = 6 tiles instead of 9 (a dy panel is read twice instead of three times, an x panel three times).
// Register plan and instruction schedule are igemm384.hip's (8 waves = 4 co groups of 64 x 2 ci groups of 192, 192 accumulator
// registers per wave, fragments two blocks ahead, counted lgkmcnt, the two waves of a SIMD issue their LDS-DMAs in different halves
// of a step, ONE barrier per step); what differs is where the fragments come from:
//
//   LDS image  both tiles stay [pixel][channel] as they are in memory (coalesced NHWC rows by LDS-DMA); MFMA fragments need
//              channel-per-lane, so they come out through ds_read_b64_tr_b16.  A tile is cut into QUADS of 64 channels:
//              [quad][32 pixels][128 B], one LDS-DMA instruction = 8 pixel rows x 128 B (whole L2 lines per row piece).  Inside a
//              128-byte row the four 32-byte chunks (16 channels each) are XOR-ed with (row >> 1) & 3 (on the SOURCE address of the DMA):
//              a half-wave's transposing read then touches 8 consecutive rows x 32 B = all 64 banks once.
//   K order    lane group fg = lane >> 4 of an MFMA operand holds pixels {4 fg .. 4 fg + 3} and {16 + 4 fg .. 16 + 4 fg + 3} of the stage
//              (two transposing reads, 2 KiB apart): any assignment of the 32 pixels to k works as long as both operands use the same.
//   addresses  quad and second-read offsets are instruction immediates; the XOR leaves four lane-constant bases (one per chunk of a
//              quad), kept for both tiles with the wave's own quad folded in; the ring stage is added per read (one v_add).
//   output     fp32 [co][ci] straight from the accumulators (16-byte stores, 64 contiguous bytes per co row): the split's slab tile,
//              or -- splits == 1 -- the gradient tensor itself (no slab, no fold).
//   grouping   up to WG384_MAXL layers of one geometry per launch (pointer table in the kernel-argument segment, read with scalar
//              loads): the pixel axis of each is cut into fewer, longer splits or not at all.
#include <type_traits>

#include "wgrad.h"

namespace dc {

namespace {

constexpr int TCO = 256, TCI = 384;
constexpr int BP = 32;                    // pixels per stage
constexpr int QUAD = BP * 128;            // one quad of a stage: 32 pixel rows x 128 B = 4 KiB
constexpr int QT = (TCO / 64) * QUAD;     // dy tile: 16 KiB
constexpr int PT = (TCI / 64) * QUAD;     // x tile: 24 KiB
constexpr int STAGE = QT + PT;            // 40 KiB
constexpr int NI = STAGE / 1024;          // LDS-DMA instructions per stage: 40
constexpr int IPW = NI / 8;               // per wave: 5 (instructions 0, 1: dy quads; 2, 3, 4: x quads)
constexpr int NCB = 12;                   // ci blocks of 16 per wave
constexpr int NPB = 4;                    // co blocks of 16 per wave
constexpr int LATE = NCB - 1 - IPW;       // first block in which waves 4..7 issue their LDS-DMAs
#ifndef DC_WG384_STAGES
#define DC_WG384_STAGES 4                 // ring stages (4 x 40 KiB = the CU's whole LDS: three stages in flight)
#endif
constexpr int NST = DC_WG384_STAGES;
#ifndef DC_WG384_AUX
#define DC_WG384_AUX 0                    // cache policy bits of the LDS-DMA (experiments: 2 = nt)
#endif
static_assert(NI % 8 == 0 && NST * STAGE <= 160 * 1024 && NST >= 3, "ring");

static __device__ __attribute__((aligned(256))) unsigned char wg384_zero_page[256];
typedef __attribute__((address_space(1))) const void* gas_ptr;
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

struct Wg384Params {
  const void* zero_page;
  int Cin, Cout, ldx, lddy;
  int M, splits, chunk, nlayers;
  // the x axis of the GEMM is [tap][ci] in quads of 64 channels (cq per tap; a pointwise layer has one tap); TAPS kernels: a stride-1 "same"
  // convolution of H x W images, tap t gathers at (oy + tdy[t], ox + tdx[t]) and writes slab plane twidx[t]
  int ntaps, cq, H, W;
  int tdy[9], tdx[9], twidx[9];
  FastDiv div_hw, div_w;
  const void* x[WG384_MAXL];
  const void* dy[WG384_MAXL];
  float* out[WG384_MAXL];       // [split][tap][Co][Ci] per layer
};

__device__ inline void mfma_v(f32x4& c, const bf16x8& av, const bf16x8& bv) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "v"(bv));
}
// one MFMA operand fragment = two transposing reads (pixels 4 fg .. and 16 + 4 fg ..), 2 KiB apart
template <int OFF>
__device__ inline void lds_read_frag(bf16x8& dst, uint32_t addr) {
  static_assert(OFF >= 0 && OFF + 16 * 128 < 65536, "ds_read offset field");
  u32x2 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(OFF) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(OFF + 16 * 128) : "memory");
  u32x4 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = hi[0]; f[3] = hi[1];
  dst = __builtin_bit_cast(bf16x8, f);
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
// a word of the kernel-argument segment at a wave-uniform (run-time) index: a scalar load (indexing the by-value struct would move it to
// scratch memory)
template <typename T>
__device__ inline T karg(size_t field_offset, int index) {
  const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
  return *(const T*)(ka + field_offset + sizeof(T) * index);
}

// MODE 0: pointwise layers (one tap at (0, 0)).  MODE 1: stride-1 "same" convolutions with several taps (3 x 3, dilated or not): the six
// quads of a workgroup's x tile are consecutive quads of the [tap][ci] axis, so a tile may straddle two taps; every x instruction carries its
// tap's pixel offset, and a lane tracks the image position of its pixel row for the halo test.  MODE 2: ConvTranspose2d(k 3, stride 2,
// pad 1, output_padding 1): the pixel axis walks the INPUT pixels (n, iy, ix), the 256-wide linear operand is x (the kernel's "dy" role:
// p.dy / p.lddy / p.Cout describe x) and the gathered operand of the [tap][channel] axis is dy at (2 iy + ty, 2 ix + tx), ty, tx in -1..1
// (the kernel's "x" role); the slab keeps its [tap][dy channel][x channel] layout, so the tile is stored transposed.
template <int MODE>
__global__ __launch_bounds__(512) void wgrad384_kernel(const Wg384Params p) {
  constexpr bool TAPS = MODE != 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 1;     // co group (64 channels = quad `grp` of the dy tile)
  const int wc = wave & 1;       // ci group (192 channels = quads 3 wc .. 3 wc + 2 of the x tile)
  const bool late = wave >= 4;   // the second wave of its SIMD

  // ---- tile decode.  XCD-aware order: consecutive tiles of an XCD are the (x, co) tiles of ONE pixel split of one layer.
  const int nq = p.ntaps * p.cq;                    // quads on the x axis
  const int nxt = (nq + 5) / 6, nco = (p.Cout + TCO - 1) / TCO;
  const int nwg = gridDim.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, xslot = blockIdx.x >> 3;
  int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + xslot;
  const int q0 = (tile % nxt) * 6;
  tile /= nxt;
  const int co0 = (tile % nco) * TCO;
  tile /= nco;
  const int split = tile % p.splits, layer = tile / p.splits;
  const int mbeg = split * p.chunk;
  const int mend = min(p.M, mbeg + p.chunk);
  const int npix = mend > mbeg ? mend - mbeg : 0;
  const int steps = (npix + BP - 1) / BP;
  const uintptr_t xbase = karg<uintptr_t>(__builtin_offsetof(Wg384Params, x), layer);
  const uintptr_t dbase = karg<uintptr_t>(__builtin_offsetof(Wg384Params, dy), layer);
  float* const obase = karg<float*>(__builtin_offsetof(Wg384Params, out), layer);
  const uintptr_t zp = (uintptr_t)p.zero_page;

  // ---- LDS-DMA bookkeeping.  Instruction q = 8 i + wave of a stage fills 1 KiB at q * 1024: rows 8 (q & 3) .. + 7 of quad q >> 2
  // (q < 16: dy tile, else x tile).  q & 3 = wave & 3 for every i, so a lane's pixel row is the same in all five instructions; what
  // differs per instruction is wave-uniform: the quad's channel base (and, for x, its tap).
  const int drow = 8 * (wave & 3) + (lane >> 3);                     // pixel row of the stage (0 .. 31)
  const int dsub = ((lane & 7) >> 1) ^ ((drow >> 1) & 3);            // logical 16-channel chunk of the quad this lane's 16 bytes belong to
  const int dch = dsub * 16 + (lane & 1) * 8;                        // channel inside the quad
  const unsigned lsrc_d = (unsigned)((drow * p.lddy + dch) * 2);
  // MODE 2: signed and recomputed per stage (the gathered pixel is 4 m - 2 ox + tap offset)
  typename std::conditional<MODE == 2, long, unsigned>::type lsrc_x = (unsigned)((drow * p.ldx + dch) * 2);
  const int gw = MODE == 2 ? 2 * p.W : p.W, gh = MODE == 2 ? 2 * p.H : p.H;      // extents of the gathered image
  int dlim[2], xlim[3], tdy[3], tdx[3];
  long doff[2], xoff[3];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = co0 + (2 * i + (wave >> 2)) * 64;
    doff[i] = (long)c * 2;
    dlim[i] = p.Cout - c;                                            // the lane's 8 channels exist when dch < dlim (Cout % 8 == 0)
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int q = q0 + 2 * k + (wave >> 2);                          // quad of the [tap][ci] axis
    const int tap = q / p.cq, ciq = q - tap * p.cq;
    const bool qok = q < nq;
    const int tt = qok ? tap : 0;
    tdy[k] = TAPS ? karg<int>(__builtin_offsetof(Wg384Params, tdy), tt) : 0;
    tdx[k] = TAPS ? karg<int>(__builtin_offsetof(Wg384Params, tdx), tt) : 0;
    xoff[k] = ((long)(tdy[k] * gw + tdx[k]) * p.ldx + ciq * 64) * 2;
    xlim[k] = qok ? p.Cin - ciq * 64 : 0;
  }
  const size_t dstep = (size_t)BP * p.lddy * 2, xstep = (size_t)(MODE == 2 ? 4 * BP : BP) * p.ldx * 2;
  uintptr_t dcur = dbase + (size_t)mbeg * p.lddy * 2, xcur = xbase + (size_t)(MODE == 2 ? 4 : 1) * mbeg * p.ldx * 2;   // scalar: the stage the next DMAs belong to
  int rows_left = npix;                                                                         // pixels of the split from that stage on
  int oy = 0, ox = 0;                                                                           // TAPS: image position of this lane's pixel row
  if constexpr (TAPS) {
    const int m = mbeg + drow < p.M ? mbeg + drow : 0;
    const int n = fast_div(m, p.div_hw);
    const int rem = m - n * (p.H * p.W);
    oy = fast_div(rem, p.div_w);
    ox = rem - oy * p.W;
    if constexpr (MODE == 2) lsrc_x = ((long)(4 * drow - 2 * ox) * p.ldx + dch) * 2;
  }
  auto issue = [&](int i, int slot) {
    bool ok = drow < rows_left;
    uintptr_t a;
    if (i < 2) {
      ok &= dch < dlim[i];
      a = dcur + doff[i] + lsrc_d;
    } else {
      const int k = i - 2;
      ok &= dch < xlim[k];
      if constexpr (MODE == 1) ok &= ((unsigned)(oy + tdy[k]) < (unsigned)gh) & ((unsigned)(ox + tdx[k]) < (unsigned)gw);
      if constexpr (MODE == 2) ok &= ((unsigned)(2 * oy + tdy[k]) < (unsigned)gh) & ((unsigned)(2 * ox + tdx[k]) < (unsigned)gw);
      a = xcur + xoff[k] + lsrc_x;
    }
    __builtin_amdgcn_global_load_lds((gas_ptr)(ok ? a : zp), (lds_ptr)(smem + slot * STAGE + (8 * i + wave) * 1024), 16, 0, DC_WG384_AUX);
  };
  auto advance = [&]() {
    dcur += dstep;
    xcur += xstep;
    rows_left -= BP;
    if constexpr (TAPS) {
      ox += BP;                       // W >= 32: at most one row wrap per stage
      if (ox >= p.W) {
        ox -= p.W;
        oy = oy + 1 == p.H ? 0 : oy + 1;
      }
      if constexpr (MODE == 2) lsrc_x = ((long)(4 * drow - 2 * ox) * p.ldx + dch) * 2;
    }
  };

#ifndef DC_LATE_PRIO
#define DC_LATE_PRIO 1      // the second-dispatched wave of every SIMD loses each arbitration at equal priority (igemm224.hip: 1 965 -> 1 924 cycles per step)
#endif
  if (DC_LATE_PRIO && late) __builtin_amdgcn_s_setprio(DC_LATE_PRIO);
  f32x4 acc[NCB][NPB];   // [ci block][co block]
#pragma unroll
  for (int i = 0; i < NCB; ++i)
#pragma unroll
    for (int j = 0; j < NPB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- fragment addresses.  Lane (fg, t = lane & 15 -> q = t >> 2, pc = t & 3) supplies row 4 fg + q, columns 4 pc .. of the block.
  const int fr = lane & 15, fg = lane >> 4;
  const int frow = 4 * fg + (fr >> 2);
  const int fkey = (frow >> 1) & 3;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr)smem;
  uint32_t la[4], lb[4];   // per chunk of a quad: x tile (this wave's first quad folded in), dy tile (this wave's quad folded in)
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const uint32_t l = (uint32_t)(frow * 128 + ((s ^ fkey) << 5) + 8 * (fr & 3));
    la[s] = lds0 + QT + wc * 3 * QUAD + l;
    lb[s] = lds0 + grp * QUAD + l;
  }
  static_assert(NCB % 3 == 0, "fa ring");
  bf16x8 fa[3], fb[NPB];

  // One 32-pixel step (igemm384.hip's schedule): cur / nxt = ring byte offsets of this and the next stage; dslot: the ring slot this
  // step's LDS-DMAs fill.
  auto step = [&](uint32_t cur, uint32_t nxt, int dslot) {
    static_for<0, NCB>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if constexpr (i + 2 < NCB) lds_read_frag<((i + 2) >> 2) * QUAD>(fa[(i + 2) % 3], la[(i + 2) & 3] + cur);   // ci fragment two blocks ahead
      if constexpr (i < IPW) {
        if (!late) issue(i, dslot);
      }
      if constexpr (i >= LATE && i < LATE + IPW) {
        if (late) issue(i - LATE, dslot);
      }
      if constexpr (i == NCB - 1) {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NST - 2) * IPW) : "memory");   // this wave's part of the next stage has landed
        __builtin_amdgcn_s_barrier();
        lds_read_frag<0>(fa[0], la[0] + nxt);                                               // next step's first two ci fragments
        lds_read_frag<0>(fa[1], la[1] + nxt);
      }
      static_for<0, NPB>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        // counted waits (LDS reads return in order; a fragment is two reads).  Block 0: fa[0], fa[1], fb[0 ..], then fa[2] are outstanding.
        if constexpr (i == 0) lgkm_wait<2 * (NPB - j)>();
        else if constexpr (j == 0 && i < NCB - 1) lgkm_wait<2 * (NCB - 1 - i < 2 ? NCB - 1 - i : 2)>();
        mfma_v(acc[i][j], fa[i % 3], fb[j]);
        if constexpr (i == NCB - 1) lds_read_frag<0>(fb[j], lb[j] + nxt);                   // re-read in place for the next step
      });
    });
  };

  // ---- prologue: stages 0 .. NST-2 in flight, stage 0 landed, first fragments requested
#pragma unroll
  for (int q = 0; q < NST - 1; ++q) {
#pragma unroll
    for (int i = 0; i < IPW; ++i) issue(i, q);
    advance();
  }
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * IPW) : "memory");
  __builtin_amdgcn_s_barrier();
  lds_read_frag<0>(fa[0], la[0]);
  lds_read_frag<0>(fa[1], la[1]);
  static_for<0, NPB>([&](auto jc) { lds_read_frag<0>(fb[decltype(jc)::value], lb[decltype(jc)::value]); });
  int cslot = 0;                  // ring slot of the current stage
  for (int s = 0; s < steps; ++s) {
    const int nslot = cslot + 1 == NST ? 0 : cslot + 1;
    const int dslot = cslot == 0 ? NST - 1 : cslot - 1;          // (cslot + NST - 1) % NST: the slot stage s-1 has left
    step((uint32_t)(cslot * STAGE), (uint32_t)(nslot * STAGE), dslot);
    advance();
    cslot = nslot;
  }
  // (the MFMAs are inline assembly: the compiler's hazard recogniser does not see them, so the wait states between the last MFMA's
  // write and the first read of an accumulator by a store are spelled out)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");

  // ---- epilogue: acc[i][j][r] = dW of linear-operand channel co0 + grp*64 + j*16 + fr and channel (i & 3)*16 + fg*4 + r of this wave's quad i >> 2
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int q = q0 + wc * 3 + k;
    if (q >= nq) continue;                                      // (wave-uniform)
    const int tap = q / p.cq, ciq = q - tap * p.cq;
    const int plane = split * p.ntaps + (TAPS ? karg<int>(__builtin_offsetof(Wg384Params, twidx), tap) : 0);
    if constexpr (MODE != 2) {
      float* out = obase + (size_t)plane * p.Cout * p.Cin + ciq * 64;
#pragma unroll
      for (int j = 0; j < NPB; ++j) {
        const int co = co0 + grp * 64 + j * 16 + fr;
        if (co >= p.Cout) continue;
        float* orow = out + (size_t)co * p.Cin;
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          const int c = ii * 16 + fg * 4;
          if (ciq * 64 + c < p.Cin) *reinterpret_cast<f32x4*>(orow + c) = acc[4 * k + ii][j];      // Cin % 8 == 0: all four or none
        }
      }
    } else {
      // transposed convolution: the slab plane is [dy channel (gathered, p.Cin of them)][x channel (linear, p.Cout of them)]
      float* out = obase + (size_t)plane * p.Cout * p.Cin;
#pragma unroll
      for (int j = 0; j < NPB; ++j) {
        const int cx = co0 + grp * 64 + j * 16 + fr;             // 16 consecutive x channels over the lanes fr: 64-byte runs
        if (cx >= p.Cout) continue;
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          const int cg = ciq * 64 + ii * 16 + fg * 4;
          if (cg >= p.Cin) continue;
#pragma unroll
          for (int r = 0; r < 4; ++r) out[(size_t)(cg + r) * p.Cout + cx] = acc[4 * k + ii][j][r];
        }
      }
    }
  }
}

}  // namespace

// kFwd geometry of ConvTranspose2d(k 3, stride 2, pad 1, output_padding 1): four sub-pixel phases, nine taps, the produced image twice the gathered one
bool wgrad384_is_tconv(const GatherGeom& g) {
  return g.os == 2 && g.is == 1 && g.ntaps == 9 && g.Hout == 2 * g.Hin && g.Wout == 2 * g.Win;
}

bool wgrad384_eligible(const GatherGeom& g, int ldx, int lddy, long M) {
  if (g.Cin % 8 != 0 || g.Cout % 8 != 0) return false;
  if (wgrad384_is_tconv(g)) return g.Win >= BP && (size_t)4 * BP * (ldx > lddy ? ldx : lddy) * 2 + 4096 < (1ull << 31);
  if (g.os != 1 || g.is != 1 || g.ntaps < 1 || g.ntaps > 9) return false;
  if ((size_t)BP * ldx * 2 + 2 * TCI >= (1ull << 32) || (size_t)BP * lddy * 2 + 2 * TCO >= (1ull << 32)) return false;
  if (g.ntaps == 1) return g.taps[0].dy == 0 && g.taps[0].dx == 0;
  // several taps: a "same" convolution (the kernel walks output pixels linearly and adds the tap's pixel offset) of rows at least a stage long
  if (g.Hin != g.Hout || g.Win != g.Wout || g.Win < BP) return false;
  for (int t = 0; t < g.ntaps; ++t)
    if (g.taps[t].phase != 0 || g.taps[t].dy <= -g.Hin || g.taps[t].dy >= g.Hin || g.taps[t].dx <= -g.Win || g.taps[t].dx >= g.Win) return false;
  (void)M;
  return true;
}

static int g_wgrad384_slots = 192;
static int g_wgrad384_min_stages = 96;
void wgrad384_set_slots(int n) { g_wgrad384_slots = n < 1 ? 1 : n; }
void wgrad384_set_min_stages(int n) { g_wgrad384_min_stages = n < 4 ? 4 : n; }
static long wgrad384_tiles(const GatherGeom& g) {
  if (wgrad384_is_tconv(g)) return (long)cdiv((long)g.ntaps * cdiv(g.Cout, 64), 6) * cdiv(g.Cin, TCO);     // gathered axis: [tap][dy channel]
  return (long)cdiv((long)g.ntaps * cdiv(g.Cin, 64), 6) * cdiv(g.Cout, TCO);
}
void wgrad384_plan(const GatherGeom& g, long M, int* splits, int* chunk, int group) {
  const long tiles = wgrad384_tiles(g) * group;
  long want = g_wgrad384_slots / tiles;
  const long per = (long)g_wgrad384_min_stages * BP;
  const long maxs = (M + per - 1) / per;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  long c = (M + want - 1) / want;
  c = (c + BP - 1) / BP * BP;
  *chunk = (int)c;
  *splits = (int)((M + c - 1) / c);
}

int launch_wgrad384(const WgradParams& w, hipStream_t st, int group, const void* const* xs, const void* const* dys, float* const* outs) {
  const size_t lds = (size_t)NST * STAGE;
  static const void* zero_dev = nullptr;
  static hipError_t init_err = hipSuccess;
  DC_ONCE({
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad384_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad384_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad384_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    void* zp = nullptr;
    init_err = hipGetSymbolAddress(&zp, HIP_SYMBOL(wg384_zero_page));
    zero_dev = zp;
  });
  if (init_err != hipSuccess) return dc_set_error(init_err, __FILE__, __LINE__);
  if (group < 1 || group > WG384_MAXL) return dc_fail("launch_wgrad384: group size out of range", __FILE__, __LINE__);
  const GatherGeom& g = w.g;
  // (the planner decides eligibility before the row strides are known: the kernel's 32-bit lane offsets hold for these)
  if (!wgrad384_eligible(g, w.ldx, w.lddy, w.M) || w.ldx < g.Cin || w.lddy < g.Cout)
    return dc_fail("launch_wgrad384: geometry or row strides outside what the kernel serves", __FILE__, __LINE__);
  Wg384Params pp;
  pp.zero_page = zero_dev;
  pp.Cin = g.Cin; pp.Cout = g.Cout; pp.ldx = w.ldx; pp.lddy = w.lddy;
  pp.M = w.M; pp.splits = w.splits; pp.chunk = w.chunk; pp.nlayers = group;
  pp.ntaps = g.ntaps; pp.cq = cdiv(g.Cin, 64); pp.H = g.Hin; pp.W = g.Win;
  for (int t = 0; t < 9; ++t) {
    pp.tdy[t] = t < g.ntaps ? g.taps[t].dy : 0;
    pp.tdx[t] = t < g.ntaps ? g.taps[t].dx : 0;
    pp.twidx[t] = t < g.ntaps ? g.taps[t].widx : 0;
  }
  pp.div_hw = g.div_hw; pp.div_w = g.div_w;
  for (int l = 0; l < WG384_MAXL; ++l) {
    pp.x[l] = l < group ? (group == 1 && xs == nullptr ? w.x : xs[l]) : nullptr;
    pp.dy[l] = l < group ? (group == 1 && dys == nullptr ? w.dy : dys[l]) : nullptr;
    pp.out[l] = l < group ? (group == 1 && outs == nullptr ? w.slab : outs[l]) : nullptr;
  }
  const long blocks = wgrad384_tiles(g) * w.splits * group;
  if (wgrad384_is_tconv(g)) {
    // ConvTranspose2d: the linear 256-wide operand is x, the gathered [tap][channel] operand dy (kernel comment, MODE 2); tap t of the kFwd
    // geometry carries weight plane widx = 3 ky + kx and gathers dy at (2 iy - 1 + ky, 2 ix - 1 + kx)
    pp.Cin = g.Cout; pp.Cout = g.Cin; pp.ldx = w.lddy; pp.lddy = w.ldx;
    pp.cq = cdiv(g.Cout, 64); pp.H = g.Hin; pp.W = g.Win;
    for (int t = 0; t < 9; ++t) {
      const int widx = t < g.ntaps ? g.taps[t].widx : 0;
      pp.tdy[t] = widx / 3 - 1;
      pp.tdx[t] = widx % 3 - 1;
    }
    pp.div_hw = make_fastdiv(g.Hin * g.Win);
    pp.div_w = make_fastdiv(g.Win);
    for (int l = 0; l < WG384_MAXL; ++l) {
      const void* t = pp.x[l];
      pp.x[l] = pp.dy[l];
      pp.dy[l] = t;
    }
    hipLaunchKernelGGL(wgrad384_kernel<2>, dim3((unsigned)blocks), dim3(512), lds, st, pp);
  } else if (g.ntaps == 1) hipLaunchKernelGGL(wgrad384_kernel<0>, dim3((unsigned)blocks), dim3(512), lds, st, pp);
  else hipLaunchKernelGGL(wgrad384_kernel<1>, dim3((unsigned)blocks), dim3(512), lds, st, pp);
  DC_CHECK_LAUNCH();
  return 0;
}

}  // namespace dc
