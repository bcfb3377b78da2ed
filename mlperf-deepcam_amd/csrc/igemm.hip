// Gather-form implicit GEMM on MFMA for every dense convolution of DeepLabV3+/Xception (see conv_geom.h).
//
//   tile      128 output pixels x 128 output channels per 256-thread workgroup (4 waves, 2x2, 64x64 each)
//   K loop    taps (outer) x input-channel chunks of 128 bytes (64 bf16 / 32 f32), register-staged and
//             double-buffered in LDS, one barrier per step
//   LDS       rows of 128 B, 16-B slots XOR-swizzled by (row>>1)&7 -> conflict-free ds_read_b128 fragments
//   MFMA      A operand = weight rows (output channel), B operand = pixel rows, so each lane ends up with
//             4 consecutive output channels of one pixel -> 8/16-byte LDS writes in the epilogue
//   epilogue  accumulators -> LDS C tile -> fully coalesced 16-B NHWC stores; optional bias, optional
//             read-modify-write accumulate, optional per-channel sum / sum-of-squares partials of the STORED
//             values for the following train-mode BatchNorm (deterministic slab, no atomics)
#include <string.h>

#include <type_traits>

#include "conv_geom.h"
#include "igemm.h"

namespace dc {

constexpr int BM = 128, BN = 128;
// staging modes (runtime switch "igemm_mode", A/B-able in one process):
//   0  register staging, 128-byte K rows, 2 LDS stages              (first working version)
//   1  LDS-DMA,          128-byte K rows, 2 stages, vmcnt(0)+barrier per step
//   2  LDS-DMA,           64-byte K rows, 3 stages, counted vmcnt: two stages stay in flight across the barrier; 48 KiB of
//      LDS and 146 registers -> 3 workgroups per CU, which also lets one workgroup's epilogue overlap another's K loop
//   3/4  as 2 with a 4- / 5-stage ring (64 / 80 KiB, 2 workgroups per CU): deeper prefetch.  PMC on the 728-channel GEMM shows a 90 %
//      L2 hit rate, i.e. practically every 16 KiB stage contains a few lines that come from the Infinity Cache / HBM, and the
//      stage is only usable when its slowest line has landed: prefetch distance, not bandwidth, bounds the K loop.
template <int MODE> struct StageCfg { static constexpr int ROWB = 128, NSTAGE = 2; };
template <> struct StageCfg<2> { static constexpr int ROWB = 64, NSTAGE = 3; };
template <> struct StageCfg<3> { static constexpr int ROWB = 64, NSTAGE = 4; };
template <> struct StageCfg<4> { static constexpr int ROWB = 64, NSTAGE = 5; };

template <typename T>
struct Mma;
template <>
struct Mma<bf16> {
  __device__ static inline void run(const vec16& a, const vec16& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <>
struct Mma<float> {
  // a 16-byte fragment holds k = 4g..4g+3 of lane group g; MFMA j consumes element j of every lane,
  // i.e. the k-set {4g+j}: the same permutation on both operands, so the sum over k is unchanged.
  __device__ static inline void run(const vec16& a, const vec16& b, f32x4& c) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w[j]), __uint_as_float(b.w[j]), c, 0, 0, 0);
  }
};

// Diagnostic build only (-DDC_STAMPS, scripts/igemm_stamps.py): per-workgroup s_memtime stamps at the phase boundaries of the
// kernel, written to a buffer no other code reads.  The product build contains none of this.
#ifdef DC_STAMPS
__device__ unsigned long long* dc_stamp_buf = nullptr;
#define DC_STAMP(k)                                                                                   \
  do {                                                                                                \
    if (threadIdx.x == 0 && dc_stamp_buf != nullptr)                                                  \
      dc_stamp_buf[(size_t)blockIdx.x * 8 + (k)] = ((k) == 0 || (k) == 7) ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define DC_STAMP(k)
#endif

template <int ROWB>
__device__ inline int swz(int row, int slot) {
  if constexpr (ROWB == 128) return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4);   // 8 slots per row
  else return row * 64 + ((slot ^ ((row >> 1) & 3)) << 4);                          // 4 slots per row
}

// 16 zero bytes in global memory: the source of every predicated-off LDS-DMA lane (zero fill of halos, K and M tails)
static __device__ __attribute__((aligned(256))) unsigned char dc_zero_page[256];   // (each translation unit has its own copy)

typedef __attribute__((address_space(1))) const void* gas_ptr;
typedef __attribute__((address_space(3))) void* lds_ptr;

// GLDS: stage operand tiles with the gfx950 LDS-DMA (global_load_lds_dwordx4: global -> LDS without passing through
// VGPRs and without ds_write instructions, which at ~79 B/clk/CU were the bottleneck of the register-staged loop).
// The LDS image is lane-linear per wave instruction, so the XOR swizzle is applied to the per-lane SOURCE address.
template <typename T, bool OUT32, int MODE>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmParams p) {
  constexpr bool GLDS = MODE != 0;
  constexpr int ROWB = StageCfg<MODE>::ROWB, NSTAGE = StageCfg<MODE>::NSTAGE;
  constexpr int OPER_BYTES = BM * ROWB;
  constexpr int SPR = ROWB / 16;          // 16-byte slots per row
  constexpr int KSUB = ROWB / 64;         // MFMA K sub-steps per stage (64 bytes of K each)
  constexpr int LROWS = 256 / SPR;        // rows covered by one pass of the 256 threads
  constexpr int LPASS = BM / LROWS;       // passes (= LDS-DMA instructions per operand per wave): 4 or 2
  typedef typename std::conditional<OUT32, float, T>::type TO;   // stored output type
  constexpr int KPV = Elem<T>::kPerVec;      // elements per 16 B
  constexpr int KPVO = Elem<TO>::kPerVec;
  constexpr int BK = ROWB / (int)sizeof(T);  // K elements per step
  constexpr int CROW = BN * (int)sizeof(TO) + 16;  // padded C-tile row
  // all LDS lives in the one dynamic array (keeps its base 16-byte aligned); the tap list sits behind the tiles
  static_assert(!OUT32 || true, "");
  constexpr int MAIN_BYTES = (2 * NSTAGE * OPER_BYTES) > (BM * CROW) ? (2 * NSTAGE * OPER_BYTES) : (BM * CROW);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* s_tap = reinterpret_cast<int*>(smem + MAIN_BYTES);

  const GatherGeom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  DC_STAMP(0);
  DC_STAMP(1);
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with a private L2), so the ones that
  // share an XCD (equal id mod 8) are given CONSECUTIVE tiles, and consecutive tiles walk the output-channel tiles of one
  // pixel tile first: the 128-pixel A panel is then fetched into that XCD's L2 once instead of once per channel tile.
  const int ntn = (g.Cout + BN - 1) / BN;
  const int nwg = gridDim.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, xslot = blockIdx.x >> 3;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + xslot;
  const int ntile_n = tile % ntn;
  const int rest = tile / ntn;
  // Sub-pixel phases fastest: the phases of a strided / transposed layer have K loops of different length (a 1 x 1 stride-2 data gradient:
  // 1 / 0 / 0 / 0 taps), and every XCD is dealt a CONTIGUOUS run of tiles -- with the phase as the slowest index all the tiles that do
  // work sat on two of the eight XCDs (block 1's shortcut data gradient at local batch 8: 178 us for 340 MB)
  const int nph = p.zfill ? 1 : g.os * g.os;     // (zfill: the launch covers phase (0, 0) only)
  const int phase = rest % nph;
  const int mtile = rest / nph;
  const int n0 = ntile_n * BN;
  const int m0 = p.m_beg + mtile * BM;
  const int py = phase / g.os, px = phase % g.os;

  // this phase's taps -> LDS (one thread per tap; the list is already sorted by phase on the host)
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
  if (steps == 0 && p.accumulate) return;  // a phase without taps contributes zeros

  // ---- per-thread load bookkeeping: 4 rows of each operand, one 16-byte slot -------------------------
  const int slot = tid % SPR;
  const int trow = tid / SPR;
  int rowbase[LPASS];   // ((n*Hin + iy0)*Win + ix0) is not enough (taps move iy/ix): keep n, iy0, ix0
  int riy[LPASS], rix[LPASS];
  bool rok[LPASS];
#pragma unroll
  for (int i = 0; i < LPASS; ++i) {
    const int r = trow + LROWS * i;
    const int m = m0 + r;
    rok[i] = m < p.M;
    const int mm = rok[i] ? m : 0;
    const int n = fast_div(mm, g.div_hw);
    const int rem = mm - n * (g.Qh * g.Qw);
    const int qy = fast_div(rem, g.div_w), qx = rem - qy * g.Qw;
    rowbase[i] = n * g.Hin;
    riy[i] = qy * g.is;
    rix[i] = qx * g.is;
  }
  const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ wg = reinterpret_cast<const T*>(p.w);

  f32x4 acc[4][4];  // [channel rep][pixel rep]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int wn = wave & 1, wm = wave >> 1;
  const int fr = lane & 15, fg = lane >> 4;

  auto compute = [&](int buf) {
    const char* wa = smem + buf * (2 * OPER_BYTES);
    const char* xb = wa + OPER_BYTES;
#pragma unroll
    for (int kk = 0; kk < KSUB; ++kk) {
      vec16 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ra_ = wn * 64 + i * 16 + fr;
        fa[i] = *reinterpret_cast<const vec16*>(wa + swz<ROWB>(ra_, kk * 4 + fg));
        const int rb_ = wm * 64 + i * 16 + fr;
        fb[i] = *reinterpret_cast<const vec16*>(xb + swz<ROWB>(rb_, kk * 4 + fg));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Mma<T>::run(fa[i], fb[j], acc[i][j]);
    }
  };
  auto next_step = [&](int& tapi, int& kc) {
    if (++kc == kchunks) {
      kc = 0;
      ++tapi;
    }
  };

  if constexpr (GLDS) {
    // this thread fills PHYSICAL slot `slot` of rows trow + LROWS*i; that slot holds logical slot slot ^ f(row), and f(row)
    // is the same for all of the thread's rows (LROWS/2 is a multiple of the swizzle period)
    const int lslot = ROWB == 128 ? (slot ^ ((trow >> 1) & 7)) : (slot ^ ((trow >> 1) & 3));
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    // Source addresses are strength-reduced: PMC showed the first LDS-DMA version VALU-bound on 64-bit address arithmetic
    // (36 % of wave cycles issuing, MFMA pipe ~20 % busy).  Per tap each row's base pointer and validity are computed
    // once; per K step only a channel offset is added.
    const T* xrow[LPASS];
    const T* wrow[LPASS];
    bool xok[LPASS], wok[LPASS];
    int cur_tap = -1;
    auto issue = [&](int tapi, int kc, int buf) {
      if (tapi != cur_tap) {
        cur_tap = tapi;
        const int dy = s_tap[3 * tapi], dx = s_tap[3 * tapi + 1], widx = s_tap[3 * tapi + 2];
#pragma unroll
        for (int i = 0; i < LPASS; ++i) {
          const int iy = riy[i] + dy, ix = rix[i] + dx;
          xok[i] = rok[i] && (unsigned)iy < (unsigned)g.Hin && (unsigned)ix < (unsigned)g.Win;
          xrow[i] = xg + (((size_t)(rowbase[i] + (xok[i] ? iy : 0)) * g.Win + (xok[i] ? ix : 0)) * p.ldx + lslot * KPV);
          const int ch = n0 + trow + LROWS * i;
          wok[i] = ch < g.Cout;
          wrow[i] = wg + (((size_t)widx * g.Cout + (wok[i] ? ch : 0)) * p.ldw + lslot * KPV);
        }
      }
      const int kofs = kc * BK;
      const bool kok = kofs + lslot * KPV < g.Cin;
      char* wa = smem + buf * (2 * OPER_BYTES);
      char* xb = wa + OPER_BYTES;
#pragma unroll
      for (int i = 0; i < LPASS; ++i) {
        const void* srcx = (kok && xok[i]) ? (const void*)(xrow[i] + kofs) : (const void*)dc_zero_page;
        const void* srcw = (kok && wok[i]) ? (const void*)(wrow[i] + kofs) : (const void*)dc_zero_page;
        const int rowoff = (LROWS * i + (64 / SPR) * wv) * ROWB;   // wave-uniform: one instruction fills 1 KiB of rows
        __builtin_amdgcn_global_load_lds((gas_ptr)srcw, (lds_ptr)(wa + rowoff), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gas_ptr)srcx, (lds_ptr)(xb + rowoff), 16, 0, 0);
      }
    };
    int tapi = 0, kc = 0;
    DC_STAMP(2);
    if constexpr (NSTAGE == 2) {
      if (steps > 0) issue(0, 0, 0);
      __syncthreads();   // (waits vmcnt(0): stage 0 has landed for every wave)
      for (int s = 0; s < steps; ++s) {
        next_step(tapi, kc);
        if (s + 1 < steps) issue(tapi, kc, (s + 1) & 1);
        compute(s & 1);
        __syncthreads();   // drains this wave's LDS-DMA (vmcnt(0)) and orders it before the next iteration's reads
      }
    } else {
      // 3-stage ring, counted waits.  Each issue() is 2*LPASS = 4 LDS-DMA instructions per wave, retired in order.
      // Iteration s: wait until stage s has landed (leave the younger stage in flight), barrier (everybody's stage s is
      // visible AND everybody has finished reading the buffer of stage s-1, which is the one stage s+2 will overwrite),
      // issue stage s+2, compute stage s.  Raw s_barrier: __syncthreads() would drain vmcnt to 0.
      // prologue: NSTAGE-1 stages in flight
#pragma unroll
      for (int q = 0; q < NSTAGE - 1; ++q) {
        if (q < steps) issue(tapi, kc, q);
        next_step(tapi, kc);
      }
      for (int s = 0; s < steps; ++s) {
        // stage s must have landed; up to NSTAGE-2 younger stages (4 LDS-DMA instructions each) may stay in flight
        const int younger = min(NSTAGE - 2, steps - 1 - s);
        if (younger >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (younger == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#ifdef DC_STAMPS
        if (s == 0) DC_STAMP(3);
#endif
        if (s + NSTAGE - 1 < steps) issue(tapi, kc, (s + NSTAGE - 1) % NSTAGE);
        next_step(tapi, kc);
        compute(s % NSTAGE);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // all waves done with the operand ring before the epilogue reuses it as the C tile
    }
  } else {
    vec16 ra[LPASS], rb[LPASS];
    auto load_step = [&](int tapi, int kc) {
      const int dy = s_tap[3 * tapi], dx = s_tap[3 * tapi + 1], widx = s_tap[3 * tapi + 2];
      const int kofs = kc * BK + slot * KPV;
      const bool kok = kofs < g.Cin;
#pragma unroll
      for (int i = 0; i < LPASS; ++i) {
        const int iy = riy[i] + dy, ix = rix[i] + dx;
        const bool ok = kok && rok[i] && (unsigned)iy < (unsigned)g.Hin && (unsigned)ix < (unsigned)g.Win;
        rb[i] = ok ? ldg16(xg + (((size_t)(rowbase[i] + iy) * g.Win + ix) * p.ldx + kofs)) : zero16();
        const int ch = n0 + trow + LROWS * i;
        ra[i] = (kok && ch < g.Cout) ? ldg16(wg + (((size_t)widx * g.Cout + ch) * p.ldw + kofs)) : zero16();
      }
    };
    auto store_step = [&](int buf) {
      char* wa = smem + buf * (2 * OPER_BYTES);
      char* xb = wa + OPER_BYTES;
#pragma unroll
      for (int i = 0; i < LPASS; ++i) {
        const int r = trow + LROWS * i;
        const int o = swz<ROWB>(r, slot);
        *reinterpret_cast<vec16*>(wa + o) = ra[i];
        *reinterpret_cast<vec16*>(xb + o) = rb[i];
      }
    };
    if (steps > 0) {
      load_step(0, 0);
      store_step(0);
    }
    __syncthreads();
    int tapi = 0, kc = 0;
    for (int s = 0; s < steps; ++s) {
      next_step(tapi, kc);
      const bool more = (s + 1) < steps;
      if (more) load_step(tapi, kc);
      compute(s & 1);
      if (more) store_step((s + 1) & 1);
      __syncthreads();
    }
  }

  // ---- epilogue ------------------------------------------------------------------------------------
  DC_STAMP(4);
  char* ct = smem;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int chl = wn * 64 + i * 16 + fg * 4;  // first of this lane's 4 channels, tile-local
    float b4[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n0 + chl + r < g.Cout) b4[r] = p.bias[n0 + chl + r];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int prow = wm * 64 + j * 16 + fr;
      TO* dst = reinterpret_cast<TO*>(ct + prow * CROW) + chl;
      if constexpr (sizeof(TO) == 4) {
        f32x4 v = acc[i][j];
        v[0] += b4[0]; v[1] += b4[1]; v[2] += b4[2]; v[3] += b4[3];
        *reinterpret_cast<f32x4*>(dst) = v;
      } else {
        uint2 v;
        v.x = pack2_bf16(acc[i][j][0] + b4[0], acc[i][j][1] + b4[1]);
        v.y = pack2_bf16(acc[i][j][2] + b4[2], acc[i][j][3] + b4[3]);
        *reinterpret_cast<uint2*>(dst) = v;
      }
    }
  }
  __syncthreads();
  DC_STAMP(5);

  constexpr int GPR = BN * (int)sizeof(TO) / 16;  // 16-byte groups per C row: 16 (bf16) / 32 (f32)
  constexpr int RPP = 256 / GPR;                  // rows covered per pass: 16 / 8
  constexpr int PASSES = BM / RPP;                // 8 / 16
  const int grp = tid % GPR, rsub = tid / GPR;
  const int ch0 = n0 + grp * KPVO;
  const bool chok = ch0 < g.Cout;
  float ssum[KPVO], ssq[KPVO];
#pragma unroll
  for (int e = 0; e < KPVO; ++e) ssum[e] = ssq[e] = 0.f;
  TO* __restrict__ yg = reinterpret_cast<TO*>(p.y);
  // BatchNorm-backward sums instead of (sum, sum of squares): this thread's channels are fixed, their vectors live in registers
  const bool bwd_stats = p.bst.y != nullptr;
  const TO* __restrict__ by = reinterpret_cast<const TO*>(p.bst.y);
  float bmu[KPVO], bis[KPVO], bms[KPVO], bmh[KPVO];
#pragma unroll
  for (int e = 0; e < KPVO; ++e) {
    const bool ok = bwd_stats && ch0 + e < g.Cout;
    bmu[e] = ok ? p.bst.mean[ch0 + e] : 0.f;
    bis[e] = ok ? p.bst.invstd[ch0 + e] : 0.f;
    bms[e] = (ok && p.bst.relu) ? p.bst.mscale[ch0 + e] : 0.f;
    bmh[e] = (ok && p.bst.relu) ? p.bst.mshift[ch0 + e] : 0.f;
  }
  auto out_pixel = [&](int m) -> size_t {
    if (g.os == 1) return (size_t)m;          // the phase grid IS the output, pixel index = m
    const int n = fast_div(m, g.div_hw);
    const int rem = m - n * (g.Qh * g.Qw);
    const int qy = fast_div(rem, g.div_w), qx = rem - qy * g.Qw;
    return (size_t)(n * g.Hout + qy * g.os + py) * g.Wout + qx * g.os + px;
  };
  if (bwd_stats) {
    // BatchNorm-backward form of the store loop: the BatchNorm inputs of all passes are requested together in front of it (see
    // igemm256.hip)
    vec16 yq[PASSES];
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int m = m0 + rsub + ps * RPP;
      yq[ps] = (m < p.M && chok) ? ldg16(by + out_pixel(m) * p.bst.ldy + ch0) : zero16();
    }
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int row = rsub + ps * RPP;
      const int m = m0 + row;
      if (m < p.M && chok) {
        const vec16 v = *reinterpret_cast<const vec16*>(ct + row * CROW + grp * 16);
        stg16(yg + out_pixel(m) * p.ldy + ch0, v);
        float f[KPVO], yv[KPVO];
        unpack(v, f, TO());
        unpack(yq[ps], yv, TO());
#pragma unroll
        for (int e = 0; e < KPVO; ++e) {
          const float gm = (!p.bst.relu || fmaf(yv[e], bms[e], bmh[e]) > 0.f) ? f[e] : 0.f;
          ssum[e] += gm;
          ssq[e] = fmaf(gm, (yv[e] - bmu[e]) * bis[e], ssq[e]);
        }
      }
    }
  } else {
#pragma unroll 2
  for (int ps = 0; ps < PASSES; ++ps) {
    const int row = rsub + ps * RPP;
    const int m = m0 + row;
    if (m < p.M && chok) {
      TO* dst = yg + out_pixel(m) * p.ldy + ch0;
      vec16 v = *reinterpret_cast<const vec16*>(ct + row * CROW + grp * 16);
      float f[KPVO];
      unpack(v, f, TO());
      if (p.accumulate) {
        float o[KPVO];
        unpack(ldg16(dst), o, TO());
#pragma unroll
        for (int e = 0; e < KPVO; ++e) f[e] += o[e];
        pack(v, f, TO());
        unpack(v, f, TO());
      }
      stg16(dst, v);
      if (p.zfill) {     // the three tap-less phases of this pixel: its right, lower and lower-right neighbours
        stg16(dst + p.ldy, zero16());
        stg16(dst + (size_t)g.Wout * p.ldy, zero16());
        stg16(dst + (size_t)(g.Wout + 1) * p.ldy, zero16());
      }
#pragma unroll
      for (int e = 0; e < KPVO; ++e) {
        ssum[e] += f[e];
        ssq[e] += f[e] * f[e];
      }
    }
  }
  }
  if (p.slab != nullptr) {
    __syncthreads();  // everyone is done reading the C tile
    float* red = reinterpret_cast<float*>(smem);  // [2][RPP][BN]
#pragma unroll
    for (int e = 0; e < KPVO; ++e) {
      red[(0 * RPP + rsub) * BN + grp * KPVO + e] = ssum[e];
      red[(1 * RPP + rsub) * BN + grp * KPVO + e] = ssq[e];
    }
    __syncthreads();
    const int which = tid >> 7, c = tid & 127;
    if (n0 + c < g.Cout) {
      float a = 0.f;
#pragma unroll
      for (int r = 0; r < RPP; ++r) a += red[(which * RPP + r) * BN + c];
      const int rows = p.mtiles * g.os * g.os;
      const int srow = phase * p.mtiles + m0 / BM;
      p.slab[((size_t)which * rows + srow) * g.Cout + n0 + c] = a;
    }
  }
  DC_STAMP(6);
  DC_STAMP(7);
}

// Pointwise convolution on at most 16 pixels (the image-pool branch: 2048 -> 256 on the 1x1 pooled map, M = batch).  The tiled
// kernel runs one or two workgroups for ~200 us there (64 dependent K steps of latency); this is a plain GEMV bundle instead:
// one wave per output channel, lanes stride over K in 16-byte vectors, the <= 16 pixel rows are L2-resident.
constexpr int TINY_M = 16;
template <typename T>
__global__ __launch_bounds__(256) void tiny_gemm_kernel(const IgemmParams p) {
  constexpr int KPV = Elem<T>::kPerVec;
  const int lane = threadIdx.x & 63;
  const int co = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (co >= p.g.Cout) return;
  const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ w = reinterpret_cast<const T*>(p.w) + ((size_t)p.g.taps[0].widx * p.g.Cout + co) * p.ldw;
  float acc[TINY_M];
#pragma unroll
  for (int m = 0; m < TINY_M; ++m) acc[m] = 0.f;
  for (int k0 = lane * KPV; k0 < p.g.Cin; k0 += 64 * KPV) {
    float wf[KPV];
    unpack(ldg16(w + k0), wf, T());
#pragma unroll
    for (int m = 0; m < TINY_M; ++m) {
      if (m < p.M) {
        float xf[KPV];
        unpack(ldg16(x + (size_t)m * p.ldx + k0), xf, T());
#pragma unroll
        for (int e = 0; e < KPV; ++e) acc[m] = fmaf(xf[e], wf[e], acc[m]);
      }
    }
  }
  float s = 0.f, q = 0.f;
  T* __restrict__ y = reinterpret_cast<T*>(p.y);
#pragma unroll
  for (int m = 0; m < TINY_M; ++m) {
    if (m < p.M) {
      float v = wave_sum(acc[m]);
      if (p.bias != nullptr) v += p.bias[co];
      T* dst = y + (size_t)m * p.ldy + co;
      if (p.accumulate) v += Elem<T>::load(dst);
      T tv;
      Elem<T>::store(&tv, v);
      const float st = Elem<T>::load(&tv);   // the statistics are those of the STORED (rounded) value
      if (lane == 0) {
        *dst = tv;
        s += st;
        q = fmaf(st, st, q);
      }
    }
  }
  if (p.slab != nullptr && lane == 0) {
    p.slab[co] = s;                       // one slab row: [2][1][Cout]
    p.slab[p.g.Cout + co] = q;
  }
}

static int g_igemm_mode = 2;
static int g_thin_fwd = 1;    // dedicated kernel for the thin 3x3 stem convolutions (thinconv.hip)
static int g_igemm256 = 1;   // 0: never, 1: where the planner expects it to win, 2: whenever eligible (bf16, bf16 output)

template <typename T, bool OUT32, int MODE>
static int launch_igemm2(const IgemmParams& p, hipStream_t st) {
  constexpr int CROW = BN * (int)(OUT32 ? 4 : sizeof(T)) + 16;
  constexpr size_t RING = (size_t)2 * StageCfg<MODE>::NSTAGE * BM * StageCfg<MODE>::ROWB;
  const size_t lds = (RING > (size_t)BM * CROW ? RING : (size_t)BM * CROW) + 128;
  DC_ONCE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<T, OUT32, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  dim3 grid(cdiv(p.g.Cout, BN) * cdiv(p.M - p.m_beg, BM) * (p.zfill ? 1 : p.g.os * p.g.os));
  hipLaunchKernelGGL((igemm_kernel<T, OUT32, MODE>), grid, dim3(256), lds, st, p);
  DC_CHECK_LAUNCH();
  return 0;
}
template <typename T, bool OUT32>
static int launch_igemm(const IgemmParams& p, hipStream_t st) {
  if (g_igemm_mode == 4) return launch_igemm2<T, OUT32, 4>(p, st);
  if (g_igemm_mode == 3) return launch_igemm2<T, OUT32, 3>(p, st);
  if (g_igemm_mode == 2) return launch_igemm2<T, OUT32, 2>(p, st);
  if (g_igemm_mode == 1) return launch_igemm2<T, OUT32, 1>(p, st);
  return launch_igemm2<T, OUT32, 0>(p, st);
}

// Planner for the tile shape.  The 128-tile kernel keeps 3 workgroups per CU (768 slots), the 256-tile kernel one (256 slots).
// Measured per ROUND (scripts/gemm256_bench.py): a 256-tile round costs 0.83-0.93 of a 128-tile round when the K loop is long
// (>= 64 steps of 32), about the same at 20-60 steps (the 728-channel layers) and 1.1-1.2 when it is short (its prologue and
// 132 KiB epilogue are not hidden by a co-resident workgroup) -- for 4/3 of the work per round.  When both fit one round the
// small tile (finer tail, three workgroups hiding each other's latencies) keeps the layer unless the K loop is long.
static int g_rel256 = 0;   // 0: model below; otherwise cost of a 256-tile round in percent of a 128-tile round (tuning switch)
// Pointwise layers on the one-wave-per-SIMD kernel (igemm384.hip).  0: never, 1: where the round model below expects it to win,
// 2 / 3: whenever eligible with 256- / 128-pixel tiles (tests and A/B runs).  Costs are in rounds of 256 x 256 tiles: a
// 256 x 384 tile is 1.5 of them, a 128 x 384 tile 0.8 (its weight fill is not halved), a round of the 128-tile kernel about 1 at
// the K depths in question (see igemm256_wins).  The layer must also fill most of the chip with its one workgroup per CU.
static int g_igemm256p = 1;          // persistent 256-tile kernel for multi-round launches (A/B switch "igemm256p")
static int g_igemm256p_wgs = 0;      // its workgroups ("igemm256p_wgs"); 0: fewest that keep the number of rounds
static int g_igemm256p_min = 257;    // fewest tiles it is used for ("igemm256p_min")
static int g_pw384 = 1;
static int g_pw192 = 1;             // 128 x 192 tiles for one-round launches of few pixels ("pw192")
static int g_zfill = 1;             // 1 x 1 stride-2 data gradients written first-hand: one launch over phase (0, 0) stores all four phases ("igemm_zfill")
static int g_pw384_k64 = 1;         // 256 x 384 tiles with 128-byte K rows where the planner picks that tile ("pw384_k64")
static int g_pw224 = 1;             // 224 x 384 tiles with weights from the [k][n] packing where the caller passes it ("pw224"; 2: whenever eligible)
static int pw384_plan(const IgemmParams& p) {
  if (g_pw224 == 2 && pw224_eligible(p)) return 224;
  if (g_pw224 == 3 && pw224_eligible(p)) return 226;      // 224 x 192 tiles wherever eligible (tests)
  if (g_pw384 == 0 || !pw384_eligible(p)) return 0;
  if (g_pw384 == 2) return 8;
  if (g_pw384 == 3) return 4;
  if (g_pw384 == 4) return 64;      // 256 x 384 tiles, 64-deep stages of 128-byte rows
  if (g_pw384 == 5) return 192;     // 128 x 192 tiles (igemm192.hip)
  const long t128 = (long)cdiv(p.g.Cout, BN) * p.mtiles, t256 = igemm256_tiles(p);
  const double c_old = (double)(cdiv(t256, 256) < cdiv(t128, 768) ? cdiv(t256, 256) : cdiv(t128, 768));
  if (p.g.Cout * 10 < (long)cdiv(p.g.Cout, 384) * 384 * 9) return 0;      // a 384-wide tile that is more than a tenth empty loses
  const long t8 = pw384_tiles(p, 8), t4 = pw384_tiles(p, 4);
  // 256-pixel tiles run with 64-deep stages of 128-byte K rows (whole L2 lines per LDS-DMA row piece): 4 - 15 % faster than 32-deep stages
  // of 64-byte rows on every shape (profiles/r03_pw384_bench.txt, second table), so a round of them is priced at 1.3 instead of 1.5
  const bool k64 = g_pw384_k64 && p.g.Cin >= 128;
  const double c8 = t8 >= 160 ? cdiv(t8, 256) * (k64 ? 1.3 : 1.5) : 1e9, c4 = t4 >= 160 ? cdiv(t4, 256) * 0.8 : 1e9;
  const double best = c8 < c4 ? c8 : c4;
  // few pixels (local batch 2): 128 x 192 tiles when they are ONE round of one workgroup per CU and neither 384-wide tile fills the chip
  // (728 -> 728 at M = 6 912: 216 tiles; 24.6 -> 14 us against the 128 x 128 kernel's 324 tiles, scripts/pw384_bench.py)
  const long t192 = pw192_tiles(p);
  if (g_pw192 && best >= 1e9 && t192 >= 160 && t192 <= 256 && p.g.Cin >= 128 && p.g.Cout * 10 >= (long)cdiv(p.g.Cout, 192) * 192 * 9) return 192;
  if (best >= 0.9 * c_old) return 0;
  // where the 256-pixel tile with 64-deep stages would run: the 224-pixel tile when it is ONE round too (728 -> 728 at M = 27 648: 248 tiles)
  // (multi-round launches too: 728 -> 728 at M = 110 592 137 against 156 us, 256 -> 728 there 81 against 92: profiles/r06_pw224_bench.txt)
  if (c8 < c4 && k64 && g_pw224 && pw224_eligible(p) && cdiv(pw224_tiles(p), 256) <= cdiv(t8, 256)) return 224;
  // where 128-pixel tiles would run (half as many pixels: local batch 4): 224 x 192 tiles when they are one round of nearly the whole chip
  // (13 824 pixels x 728 channels: 62 x 4 = 248 tiles, 26 KiB of operands per 32-deep step instead of 32)
  if (!(c8 < c4) && g_pw224 && pw224_eligible(p) && pw224_tiles(p, 192) > 192 && pw224_tiles(p, 192) <= 256 &&
      p.g.Cout * 10 >= (long)cdiv(p.g.Cout, 192) * 192 * 9)
    return 226;
  return c8 < c4 ? (k64 ? 64 : 8) : 4;
}

static bool igemm256_wins(const IgemmParams& p) {
  const long t128 = (long)cdiv(p.g.Cout, BN) * p.mtiles * p.g.os * p.g.os;
  const long t256 = igemm256_tiles(p);
  const long r128 = cdiv(t128, 768), r256 = cdiv(t256, 256);
  const int steps = cdiv(p.g.Cin, 32) * (p.g.ntaps / (p.g.os * p.g.os) > 0 ? p.g.ntaps / (p.g.os * p.g.os) : 1);   // average over phases
  if (r128 == 1 && r256 == 1 && steps < 64) return false;
  const int rel = g_rel256 > 0 ? g_rel256 : steps >= 64 ? 90 : steps >= 20 ? 97 : 115;
  return r256 * rel < r128 * 100;
}

static int check_view(const void* ptr, int ld, int c, int dtype, const char* what) {
  return dc_check_view(ptr, ld, c, dtype, what);
}

static int run_gather(const dc_conv_desc* d, GatherMode mode, int N, int Hi, int Wi, const void* in, int ldin,
                      const void* w, const float* bias, void* out, int ldout, float* slab, int accumulate,
                      void* stream, bool out32 = false, const BnBwdEpi* bst = nullptr, const void* w_kn = nullptr, int slab_rows = 0) {
  DC_REQUIRE(d != nullptr, "dc_conv: null descriptor");
  DC_REQUIRE(d->dtype == DC_F32 || d->dtype == DC_BF16, "dc_conv: bad dtype");
  DC_REQUIRE(d->transposed || d->k == 1 || d->k == 3, "dc_conv: kernel size must be 1 or 3");
  DC_REQUIRE(d->transposed || d->stride == 1 || d->stride == 2, "dc_conv: stride must be 1 or 2");
  DC_REQUIRE(N > 0 && Hi > 0 && Wi > 0, "dc_conv: empty input");
  IgemmParams p;
  if (!build_geom(*d, Hi, Wi, mode, &p.g)) return dc_fail("dc_conv: odd extent under a stride-2 phase split", __FILE__, __LINE__);
  if (int e = check_view(in, ldin, p.g.Cin, d->dtype, "dc_conv input")) return e;
  if (int e = check_view(out, ldout, p.g.Cout, out32 ? DC_F32 : d->dtype, "dc_conv output")) return e;
  DC_REQUIRE(w != nullptr && ((uintptr_t)w & 15) == 0, "dc_conv: weights null or unaligned");
  DC_REQUIRE(!(slab != nullptr && accumulate), "dc_conv: statistics and accumulate are exclusive");
  p.x = in; p.w = w; p.y = out; p.bias = bias; p.slab = slab;
  p.N = N; p.ldx = ldin; p.ldy = ldout;
  p.ldw = weight_ld(p.g.Cin);
  p.w_kn = w_kn;
  p.ldw_kn = weight_ld(p.g.Cout);
  const long M = (long)N * p.g.Qh * p.g.Qw;
  DC_REQUIRE(M < (1L << 31) - BM, "dc_conv: too many pixels for 32-bit indexing");
  p.M = (int)M;
  p.m_beg = 0;
  p.phase_fast = 0;
  p.zero_page = nullptr;
  p.ngroup = 0;
  p.ksplit = 0; p.kslab = nullptr;
  p.zfill = 0;
  p.bst = BnBwdEpi{nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0};
  if (bst != nullptr) {
    DC_REQUIRE(slab != nullptr && !accumulate && !out32 && bias == nullptr, "dc_conv_dgrad_bnstats: needs a slab, no bias, no accumulate");
    if (int e = check_view(bst->y, bst->ldy, p.g.Cout, d->dtype, "dc_conv_dgrad_bnstats y")) return e;
    DC_REQUIRE(bst->mean && bst->invstd && (!bst->relu || (bst->mscale && bst->mshift)), "dc_conv_dgrad_bnstats: missing BatchNorm vectors");
    p.bst = *bst;
  }
  p.mtiles = cdiv(M, BM);
  p.accumulate = accumulate;
  hipStream_t st = (hipStream_t)stream;
  const bool sum_row = slab != nullptr && slab_rows == -1;      // only the 224-pixel tile kernel below serves it
  if (!sum_row && bst == nullptr && !out32 && p.M <= TINY_M && p.g.ntaps == 1 && p.g.os == 1 && p.g.is == 1 && p.g.taps[0].dy == 0 && p.g.taps[0].dx == 0) {
    if (d->dtype == DC_BF16) hipLaunchKernelGGL(tiny_gemm_kernel<bf16>, dim3(cdiv(p.g.Cout, 4)), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(tiny_gemm_kernel<float>, dim3(cdiv(p.g.Cout, 4)), dim3(256), 0, st, p);
    DC_CHECK_LAUNCH();
    return 0;
  }
  if (!sum_row && bst == nullptr && g_thin_fwd && thin_fwd_eligible(p.g, d->dtype, bias != nullptr, accumulate, out32))
    return launch_thin_fwd(p.g, N, in, ldin, w, p.ldw, out, ldout, slab, p.mtiles * p.g.os * p.g.os, st);
  if (d->dtype == DC_BF16 && !out32 && bst == nullptr) {      // (the BatchNorm-backward epilogue lives in the LDS-epilogue kernels)
    if (const int npb = pw384_plan(p)) {
      if (npb == 224 || npb == 226) {
        // the 224-pixel tiles leave ONE slab row per tile: a caller that sized its slab by dc_conv_stat_rows_kn gets exactly those rows (no
        // zero rows behind them); any larger slab is padded with zero rows
        if (slab != nullptr && slab_rows > 0) {
          DC_REQUIRE(slab_rows >= cdiv(p.M, 224), "dc_conv_fwd_kn: slab_rows is smaller than the rows this launch writes (dc_conv_stat_rows_kn)");
          p.mtiles = slab_rows;
        }
        if (slab != nullptr && slab_rows == -1) {      // a sum row: double[2][Cout], zeroed by the caller; every tile adds its sums (fp64 atomics)
          DC_REQUIRE(((uintptr_t)slab & 7) == 0, "dc_conv_fwd_kn: a sum row is double[2][Cout]");
          p.mtiles = -1;
        }
        return launch_pw224(p, npb == 224 ? 384 : 192, st);
      }
      if (npb == 192 && sum_row) {      // the 128 x 192 tiles add to a sum row as well
        DC_REQUIRE(((uintptr_t)slab & 7) == 0, "dc_conv_fwd_kn: a sum row is double[2][Cout]");
        p.mtiles = -1;
        return launch_pw192(p, st);
      }
      DC_REQUIRE(!sum_row, "dc_conv_fwd_kn: this launch does not add to a sum row (dc_conv_sum_row_kn)");
      DC_REQUIRE(slab == nullptr || slab_rows <= 0 || slab_rows == p.mtiles, "dc_conv_fwd_kn: slab_rows is not what this launch writes now (dc_conv_stat_rows_kn)");
      return npb == 192 ? launch_pw192(p, st) : launch_pw384(p, npb, st);
    }
  }
  DC_REQUIRE(slab == nullptr || slab_rows != -1, "dc_conv_fwd_kn: this launch does not add to a sum row (dc_conv_sum_row_kn)");
  DC_REQUIRE(slab == nullptr || slab_rows <= 0 || slab_rows == p.mtiles * p.g.os * p.g.os, "dc_conv_fwd_kn: slab_rows is not what this launch writes now (dc_conv_stat_rows_kn)");
  if (d->dtype == DC_BF16 && !out32 && g_igemm256 != 0) {
    if (g_igemm256 == 2 || igemm256_wins(p)) {
      // several rounds of tiles: the persistent form keeps the operand ring full across tiles (igemm256p.hip)
      if (g_igemm256p && bst == nullptr && igemm256p_eligible(p) && igemm256_tiles(p) >= (long)g_igemm256p_min) return launch_igemm256p(p, g_igemm256p_wgs, st);
      return launch_igemm256(p, st);
    }
  }
  // every tap in phase (0, 0) of a 2 x 2 phase grid, written first-hand, nothing riding along: one launch over that phase stores all four
  if (g_zfill && p.g.os == 2 && p.g.ntaps > 0 && p.g.phase_beg[1] == p.g.ntaps && !accumulate && slab == nullptr && bias == nullptr && bst == nullptr && !out32)
    p.zfill = 1;
  if (d->dtype == DC_BF16) return out32 ? launch_igemm<bf16, true>(p, st) : launch_igemm<bf16, false>(p, st);
  return launch_igemm<float, false>(p, st);
}

// ---------------------------------------------------------------------------------------------------
// weight packing: fp32 master (PyTorch layout) -> wf[tap][cout][cin], wb[tap][cin][cout] in T
template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ master, T* __restrict__ wf, T* __restrict__ wb,
                                    int cin, int cout, int taps, int transposed) {
  const long total = (long)cin * cout * taps;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    // i enumerates the master layout
    const int t = (int)(i % taps);
    const long r = i / taps;
    int co, ci;
    if (transposed) {  // [cin][cout][t]
      co = (int)(r % cout);
      ci = (int)(r / cout);
    } else {           // [cout][cin][t]
      ci = (int)(r % cin);
      co = (int)(r / cin);
    }
    const float v = master[i];
    const int ldf = weight_ld(cin), ldb = weight_ld(cout);   // padded K strides (the pad is never read)
    if (wf) Elem<T>::store(wf + ((size_t)t * cout + co) * ldf + ci, v);
    if (wb) Elem<T>::store(wb + ((size_t)t * cin + ci) * ldb + co, v);
  }
}

// One launch repacks EVERY layer's weights (dense conv fwd/dgrad operands and depthwise [9][C]): the per-layer pack kernels
// were ~145 launches of ~10 us each, i.e. pure launch-boundary time between the optimizer and the next forward.
struct PackEntry {
  const float* master;
  void* wf;
  void* wb;
  int cin, cout, taps, kind;   // kind 0: conv [cout][cin][t], 1: transposed conv [cin][cout][t], 2: depthwise [C][9] -> fp32 [9][C]
};

// Tiles of the master tensor [A][B][t] go through LDS so that both outputs are written in 64..128-byte runs:
//   P[t][a][b] (b contiguous) and Q[t][b][a] (a contiguous).  A conv ([cout][cin][t]) has wf = P, wb = Q; a transposed conv
// ([cin][cout][t]) has wb = P, wf = Q.  (The first version stored single bf16 elements with a stride of one K row:
// 0.9 ms per step for 56 M weights; a tile pass runs near copy speed.)
constexpr int PACK_LDS_FLOATS = 32 * (32 * 9 + 1);   // fp32 storage: 32 x 32 x 9 taps (+1 pad per row); also holds 64 x (64 + 1)
// bf16 storage: the 3 x 3 tiles sit in LDS already rounded (2 bytes per element, rows of 32 * 9 + 2), so the kernel's LDS is that of the
// 64 x 65-float pointwise tile and seven workgroups fit a CU instead of three (the repack is a chain of short load - transpose - store lives)
constexpr int PACK_ROW16 = 32 * 9 + 2;
constexpr int PACK_LDS_FLOATS_BF16 = (32 * PACK_ROW16 * 2 + 3) / 4 > 64 * 65 ? (32 * PACK_ROW16 * 2 + 3) / 4 : 64 * 65;

template <typename T>
__device__ inline void pack_store_run(T* dst, float v0, float v1);
template <>
__device__ inline void pack_store_run<bf16>(bf16* dst, float v0, float v1) { *reinterpret_cast<uint32_t*>(dst) = pack2_bf16(v0, v1); }

// one TS x TS x TAPS tile of a dense entry through LDS; TAPS is a template parameter so that the index arithmetic of the three
// passes divides by constants (with a run-time tap count the kernel was bound by integer division: 0.3 ms for 450 MB)
template <typename T, int TAPS>
__device__ inline void pack_dense_tile(const PackEntry& e, int tile, float* lds) {
  const int A = e.kind == 1 ? e.cin : e.cout, B = e.kind == 1 ? e.cout : e.cin;
  T* P = reinterpret_cast<T*>(e.kind == 1 ? e.wb : e.wf);   // [t][a][b], row stride weight_ld(B)
  T* Q = reinterpret_cast<T*>(e.kind == 1 ? e.wf : e.wb);   // [t][b][a], row stride weight_ld(A)
  const int ldp = weight_ld(B), ldq = weight_ld(A);
  constexpr int TS = TAPS == 1 ? 64 : 32;       // tile edge
  constexpr int ROWF = TS * TAPS + 1;           // padded LDS row (floats): odd stride -> conflict-free column reads
  const int ntb = (B + TS - 1) / TS;
  constexpr int EPU = 4 / (int)sizeof(T);       // elements per 4-byte store unit
  const int a0 = (tile / ntb) * TS, b0 = (tile % ntb) * TS;
  const int run = min(TS, B - b0) * TAPS;       // contiguous floats per master row of this tile
  __syncthreads();                              // previous tile fully written out
  if constexpr (sizeof(T) == 2 && TAPS == 9) {
    // bf16: rounded on the way into LDS (the same rounding the stores below did), two bytes per element
    unsigned short* l16 = reinterpret_cast<unsigned short*>(lds);
#pragma unroll 4
    for (int idx = threadIdx.x; idx < TS * TS * TAPS; idx += 256) {
      const int al = idx / (TS * TAPS), j = idx - al * (TS * TAPS);
      float v = 0.f;
      if (a0 + al < A && j < run) v = e.master[((size_t)(a0 + al) * B + b0) * TAPS + j];
      l16[al * PACK_ROW16 + j] = __builtin_bit_cast(unsigned short, (bf16)v);
    }
    __syncthreads();
    constexpr int UPR16 = TS / 2;
    if (P != nullptr) {
#pragma unroll 4
      for (int idx = threadIdx.x; idx < TAPS * TS * UPR16; idx += 256) {
        const int u = idx % UPR16, al = (idx / UPR16) % TS, t = idx / (UPR16 * TS);
        const int bl = u * 2;
        if (a0 + al < A && b0 + bl < B)
          *reinterpret_cast<uint32_t*>(P + ((size_t)t * A + a0 + al) * ldp + b0 + bl) =
              (uint32_t)l16[al * PACK_ROW16 + bl * TAPS + t] | ((uint32_t)l16[al * PACK_ROW16 + (bl + 1) * TAPS + t] << 16);
      }
    }
    if (Q != nullptr) {
#pragma unroll 4
      for (int idx = threadIdx.x; idx < TAPS * TS * UPR16; idx += 256) {
        const int u = idx % UPR16, bl = (idx / UPR16) % TS, t = idx / (UPR16 * TS);
        const int al = u * 2;
        if (b0 + bl < B && a0 + al < A)
          *reinterpret_cast<uint32_t*>(Q + ((size_t)t * B + b0 + bl) * ldq + a0 + al) =
              (uint32_t)l16[al * PACK_ROW16 + bl * TAPS + t] | ((uint32_t)l16[(al + 1) * PACK_ROW16 + bl * TAPS + t] << 16);
      }
    }
    return;
  }
#pragma unroll 4
  for (int idx = threadIdx.x; idx < TS * TS * TAPS; idx += 256) {
    const int al = idx / (TS * TAPS), j = idx - al * (TS * TAPS);
    float v = 0.f;
    if (a0 + al < A && j < run) v = e.master[((size_t)(a0 + al) * B + b0) * TAPS + j];
    lds[al * ROWF + j] = v;
  }
  __syncthreads();
  constexpr int UPR = TS / EPU;                 // store units per output row
  if (P != nullptr) {
#pragma unroll 4
    for (int idx = threadIdx.x; idx < TAPS * TS * UPR; idx += 256) {
      const int u = idx % UPR, al = (idx / UPR) % TS, t = idx / (UPR * TS);
      const int bl = u * EPU;
      if (a0 + al < A && b0 + bl < B) {         // B is a multiple of EPU for every layer of the network; the row pad absorbs a tail
        T* dst = P + ((size_t)t * A + a0 + al) * ldp + b0 + bl;
        if constexpr (EPU == 2) pack_store_run<T>(dst, lds[al * ROWF + bl * TAPS + t], lds[al * ROWF + (bl + 1) * TAPS + t]);
        else *reinterpret_cast<float*>(dst) = lds[al * ROWF + bl * TAPS + t];
      }
    }
  }
  if (Q != nullptr) {
#pragma unroll 4
    for (int idx = threadIdx.x; idx < TAPS * TS * UPR; idx += 256) {
      const int u = idx % UPR, bl = (idx / UPR) % TS, t = idx / (UPR * TS);
      const int al = u * EPU;
      if (b0 + bl < B && a0 + al < A) {
        T* dst = Q + ((size_t)t * B + b0 + bl) * ldq + a0 + al;
        if constexpr (EPU == 2) pack_store_run<T>(dst, lds[al * ROWF + bl * TAPS + t], lds[(al + 1) * ROWF + bl * TAPS + t]);
        else *reinterpret_cast<float*>(dst) = lds[al * ROWF + bl * TAPS + t];
      }
    }
  }
}

// The pointwise layers (two thirds of the dense weights) in bf16: the same 64 x 64 tile, moved with 16-byte accesses -- four float4
// loads per thread instead of sixteen scalar ones, and one 16-byte store per 8 packed values instead of four 4-byte stores (the
// generic pass ran at 1.7 TB/s).  Same conversions (pack2_bf16), same bits.  Needs A and B to be multiples of 8.
__device__ inline void pack_pointwise_tile_bf16(const PackEntry& e, int tile, float* lds) {
  const int A = e.kind == 1 ? e.cin : e.cout, B = e.kind == 1 ? e.cout : e.cin;
  bf16* P = reinterpret_cast<bf16*>(e.kind == 1 ? e.wb : e.wf);   // [a][b], row stride weight_ld(B)
  bf16* Q = reinterpret_cast<bf16*>(e.kind == 1 ? e.wf : e.wb);   // [b][a], row stride weight_ld(A)
  const int ldp = weight_ld(B), ldq = weight_ld(A);
  constexpr int TS = 64, ROWF = TS + 1;
  const int ntb = (B + TS - 1) / TS;
  const int a0 = (tile / ntb) * TS, b0 = (tile % ntb) * TS;
  const int run = min(TS, B - b0);              // a multiple of 8
  __syncthreads();                              // previous tile fully written out
#pragma unroll
  for (int idx = threadIdx.x; idx < TS * (TS / 4); idx += 256) {
    const int al = idx / (TS / 4), j = (idx % (TS / 4)) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a0 + al < A && j < run) v = *reinterpret_cast<const float4*>(e.master + (size_t)(a0 + al) * B + b0 + j);
    float* d = lds + al * ROWF + j;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  __syncthreads();
  if (P != nullptr) {
#pragma unroll
    for (int idx = threadIdx.x; idx < TS * (TS / 8); idx += 256) {
      const int u = idx % (TS / 8), al = idx / (TS / 8);
      if (a0 + al < A && b0 + u * 8 < B) {
        float f[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] = lds[al * ROWF + u * 8 + k];
        vec16 v;
        pack(v, f, bf16());
        stg16(P + (size_t)(a0 + al) * ldp + b0 + u * 8, v);
      }
    }
  }
  if (Q != nullptr) {
#pragma unroll
    for (int idx = threadIdx.x; idx < TS * (TS / 8); idx += 256) {
      const int u = idx % (TS / 8), bl = idx / (TS / 8);
      if (b0 + bl < B && a0 + u * 8 < A) {
        float f[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] = lds[(u * 8 + k) * ROWF + bl];
        vec16 v;
        pack(v, f, bf16());
        stg16(Q + (size_t)(b0 + bl) * ldq + a0 + u * 8, v);
      }
    }
  }
}

// tiles of one table entry (dense: TS x TS x taps tiles of the master tensor; depthwise: runs of 256 channels)
__device__ inline int pack_entry_tiles(const PackEntry& e) {
  if (e.kind == 2) return (e.cout + 255) / 256;
  const int A = e.kind == 1 ? e.cin : e.cout, B = e.kind == 1 ? e.cout : e.cin;
  const int TS = e.taps == 1 ? 64 : 32;
  return ((A + TS - 1) / TS) * ((B + TS - 1) / TS);
}

constexpr int PACK_MAX_ENTRIES = 1024;

// One flat grid over ALL tiles of all entries: every workgroup first builds the prefix sums of the per-entry tile counts in LDS
// (a few hundred entries: one scan), then walks tiles b, b + G, ... and finds the owning entry by binary search.  (The first
// version gave every entry 128 workgroups: the 728-element depthwise entries wasted theirs and the 4.7 M-element ASPP kernels
// waited for 128 workgroups to chew through 512 tiles each -- 0.28 ms for 450 MB.)
template <typename T>
__global__ __launch_bounds__(256) void pack_all_kernel(const PackEntry* __restrict__ table, int nentries) {
  __shared__ float lds[sizeof(T) == 2 ? PACK_LDS_FLOATS_BF16 : PACK_LDS_FLOATS];
  __shared__ int prefix[PACK_MAX_ENTRIES + 1];
  for (int i = threadIdx.x; i < nentries; i += 256) prefix[i + 1] = pack_entry_tiles(table[i]);
  if (threadIdx.x == 0) prefix[0] = 0;
  __syncthreads();
  if (threadIdx.x == 0)
    for (int i = 1; i <= nentries; ++i) prefix[i] += prefix[i - 1];
  __syncthreads();
  const int total = prefix[nentries];
  for (int gt = blockIdx.x; gt < total; gt += gridDim.x) {
    int lo = 0, hi = nentries;              // invariant: prefix[lo] <= gt < prefix[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (prefix[mid] <= gt) lo = mid; else hi = mid;
    }
    const PackEntry e = table[lo];
    const int tile = gt - prefix[lo];
    if (e.kind == 2) {
      float* out = reinterpret_cast<float*>(e.wf);
      const int c = tile * 256 + threadIdx.x;
      if (c < e.cout) {
#pragma unroll
        for (int t = 0; t < 9; ++t) out[(size_t)t * e.cout + c] = e.master[(size_t)c * 9 + t];
      }
      continue;
    }
    if (e.taps == 1) {
      if constexpr (sizeof(T) == 2) {
        if (((e.cin | e.cout) & 7) == 0) pack_pointwise_tile_bf16(e, tile, lds);
        else pack_dense_tile<T, 1>(e, tile, lds);
      } else {
        pack_dense_tile<T, 1>(e, tile, lds);
      }
    } else {
      pack_dense_tile<T, 9>(e, tile, lds);
    }
  }
}

}  // namespace dc

using namespace dc;

#ifdef DC_STAMPS
extern "C" int dc_debug_stamp_buf(void* buf) {
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(dc_stamp_buf), &buf, sizeof(buf));
  return e == hipSuccess ? 0 : dc_set_error(e, __FILE__, __LINE__);
}
#endif

static int g_pack_blocks = 2048;   // workgroups of the flat dc_pack_all grid (tuning switch "pack_blocks")
extern "C" int dc_pack_all(int dtype, const void* table_dev, int nentries, void* stream) {
  DC_REQUIRE(table_dev != nullptr && nentries > 0, "dc_pack_all: bad argument");
  DC_REQUIRE(sizeof(PackEntry) == 40, "dc_pack_all: entry layout changed");
  DC_REQUIRE(nentries <= PACK_MAX_ENTRIES, "dc_pack_all: too many entries for one launch");
  dim3 grid(g_pack_blocks);
  if (dtype == DC_BF16) hipLaunchKernelGGL(pack_all_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, (const PackEntry*)table_dev, nentries);
  else hipLaunchKernelGGL(pack_all_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const PackEntry*)table_dev, nentries);
  DC_CHECK_LAUNCH();
  return 0;
}

// Tuning switches for A/B measurements in one process: "igemm_mode" (0/1/2, see StageCfg), "wgrad_target_blocks".
extern "C" int dc_wgrad_set_target_blocks(int n);
extern "C" int dc_wgrad_set_mode(int m);
extern "C" int dc_wgrad_set_min_steps(int n);
extern "C" int dc_wgrad_set_thin(int m);
extern "C" int dc_wgrad_set_384(int m);
extern "C" int dc_wgrad_set_384_slots(int n);
extern "C" int dc_wgrad_set_384_fill(int pct);
extern "C" int dc_wgrad_set_384_min_stages(int n);
extern "C" int dc_head_set_fused(int v);
extern "C" int dc_head_set_dgrad_fused(int v);
extern "C" int dc_head_set_wgrad_fused(int v);
extern "C" int dc_dw_set_option(const char* name, int value);
extern "C" int dc_bn_set_option(const char* name, int value);
extern "C" int dc_set_option(const char* name, int value) {
  if (name != nullptr && strcmp(name, "igemm_mode") == 0) {
    if (value < 0 || value > 4) return dc_fail("dc_set_option: igemm_mode must be 0..4", __FILE__, __LINE__);
    g_igemm_mode = value;
    return 0;
  }
  if (name != nullptr && strcmp(name, "igemm256") == 0) { g_igemm256 = value; return 0; }
  if (name != nullptr && strcmp(name, "pw384") == 0) { g_pw384 = value; return 0; }
  if (name != nullptr && strcmp(name, "igemm256p") == 0) { g_igemm256p = value; return 0; }
  if (name != nullptr && strcmp(name, "igemm256p_wgs") == 0) { g_igemm256p_wgs = value; return 0; }
  if (name != nullptr && strcmp(name, "igemm256p_min") == 0) { g_igemm256p_min = value; return 0; }
  if (name != nullptr && strcmp(name, "pw384_k64") == 0) { g_pw384_k64 = value; return 0; }
  if (name != nullptr && strcmp(name, "pw224") == 0) { g_pw224 = value; return 0; }
  if (name != nullptr && strcmp(name, "pw192") == 0) { g_pw192 = value; return 0; }
  if (name != nullptr && strcmp(name, "igemm_zfill") == 0) { g_zfill = value; return 0; }
  if (name != nullptr && strcmp(name, "igemm256_splitk") == 0) { igemm256_set_splitk(value); return 0; }
  if (name != nullptr && strcmp(name, "thin_fwd") == 0) { g_thin_fwd = value != 0; return 0; }
  if (name != nullptr && strcmp(name, "thin_tile") == 0) { thin_set_tile(value); return 0; }
  if (name != nullptr && strcmp(name, "igemm256_rel") == 0) { g_rel256 = value; return 0; }
  if (name != nullptr && strcmp(name, "pack_blocks") == 0 && value > 0) { g_pack_blocks = value; return 0; }
  if (name != nullptr && strcmp(name, "igemm256_phase_fast") == 0) { igemm256_set_phase_fast(value); return 0; }
  if (name != nullptr && strcmp(name, "wgrad_target_blocks") == 0) return dc_wgrad_set_target_blocks(value);
  if (name != nullptr && strcmp(name, "wgrad_mode") == 0) return dc_wgrad_set_mode(value);
  if (name != nullptr && strcmp(name, "wgrad_min_steps") == 0) return dc_wgrad_set_min_steps(value);
  if (name != nullptr && strcmp(name, "thin_wgrad") == 0) return dc_wgrad_set_thin(value);
  if (name != nullptr && strcmp(name, "wgrad384") == 0) return dc_wgrad_set_384(value);
  if (name != nullptr && strcmp(name, "wgrad384_slots") == 0) return dc_wgrad_set_384_slots(value);
  if (name != nullptr && strcmp(name, "wgrad384_fill") == 0) return dc_wgrad_set_384_fill(value);
  if (name != nullptr && strcmp(name, "wgrad384_min_stages") == 0) return dc_wgrad_set_384_min_stages(value);
  if (name != nullptr && strcmp(name, "head_fused") == 0) return dc_head_set_fused(value);
  if (name != nullptr && strcmp(name, "head_dgrad_fused") == 0) return dc_head_set_dgrad_fused(value);
  if (name != nullptr && strcmp(name, "head_wgrad_fused") == 0) return dc_head_set_wgrad_fused(value);
  if (name != nullptr && dc_dw_set_option(name, value) == 0) return 0;
  if (name != nullptr && dc_bn_set_option(name, value) == 0) return 0;
  return dc_fail("dc_set_option: unknown option", __FILE__, __LINE__);
}

// Every tuning switch with its default, in ONE place: dc_reset_options() puts the whole set back (the test suite calls it after every test,
// so that a switch one test leaves behind cannot change what the next one measures -- a stale "restore" of this kind once ran the model
// tests on an experimental kernel), and the library applies the same table when it is loaded, so the table IS the default.
static const struct { const char* name; int value; } kOptionDefaults[] = {
    {"igemm_mode", 2}, {"igemm256", 1}, {"pw384", 1}, {"pw384_k64", 1}, {"pw224", 1}, {"pw192", 1}, {"igemm256p", 1}, {"igemm256p_wgs", 0}, {"igemm256p_min", 257},
    {"thin_fwd", 1}, {"thin_tile", 1}, {"igemm256_rel", 0}, {"pack_blocks", 2048},
    {"igemm256_phase_fast", 1}, {"wgrad_target_blocks", 768},
    {"wgrad_mode", 1}, {"wgrad_min_steps", 16}, {"thin_wgrad", 1}, 
    {"wgrad384", 1}, {"wgrad384_slots", 192}, {"wgrad384_fill", 66}, {"wgrad384_min_stages", 96}, {"head_fused", 1},
    {"head_dgrad_fused", 1}, {"head_wgrad_fused", 1}, {"igemm256_splitk", 1}, {"igemm_zfill", 1}, {"pw_bn_bwd", 3}, {"sep_fwd", 1}, {"dw_tile", 1}, {"dw_wgrad_tpb", 0}, {"dw_cg", 0}, {"dw_pipe", 1}, {"bn_cgw", 32}, {"bn_rows", 32}, {"bn_apply_rows", 1}, {"bn_fin_mul_fwd", 2}, {"bn_fin_mul_bwd", 2},
};

extern "C" int dc_reset_options(void) {
  for (const auto& o : kOptionDefaults)
    if (int e = dc_set_option(o.name, o.value)) return e;
  return 0;
}

namespace {
struct OptionDefaultsAtLoad {
  OptionDefaultsAtLoad() { (void)dc_reset_options(); }
} g_option_defaults_at_load;
}  // namespace

extern "C" int dc_conv_out_hw(const dc_conv_desc* d, int Hi, int Wi, int* Ho, int* Wo) {
  DC_REQUIRE(d != nullptr, "dc_conv_out_hw: null descriptor");
  GatherGeom g;
  if (!build_geom(*d, Hi, Wi, kFwd, &g)) return dc_fail("dc_conv_out_hw: unsupported geometry", __FILE__, __LINE__);
  if (Ho) *Ho = g.Hout;
  if (Wo) *Wo = g.Wout;
  return 0;
}

extern "C" int dc_conv_stat_rows(const dc_conv_desc* d, int N, int Hi, int Wi) {
  GatherGeom g;
  if (d == nullptr || !build_geom(*d, Hi, Wi, kFwd, &g)) return -1;
  return cdiv((long)N * g.Qh * g.Qw, BM) * g.os * g.os;
}

extern "C" int dc_conv_fwd(const dc_conv_desc* d, int N, int Hi, int Wi, const void* x, int ldx, const void* wf,
                           const float* bias, void* y, int ldy, float* stat_slab, int accumulate, void* stream) {
  return run_gather(d, kFwd, N, Hi, Wi, x, ldx, wf, bias, y, ldy, stat_slab, accumulate, stream);
}

// dc_conv_fwd / dc_conv_dgrad with BOTH packed images of the layer: w is the one the plain entry point takes, w_kn the other one (forward: wb,
// data gradient: wf), whose rows are [k][n].  Pointwise layers may then run on the 224 x 384 tile kernel (igemm224.hip), which streams its weight
// stages from w_kn; every other layer ignores it.  Same results as the plain entry points, bit for bit (BatchNorm sums: another summation order).
extern "C" int dc_conv_fwd_kn(const dc_conv_desc* d, int N, int Hi, int Wi, const void* x, int ldx, const void* wf, const void* wb,
                              const float* bias, void* y, int ldy, float* stat_slab, int slab_rows, int accumulate, void* stream) {
  return run_gather(d, kFwd, N, Hi, Wi, x, ldx, wf, bias, y, ldy, stat_slab, accumulate, stream, false, nullptr, wb, slab_rows);
}
// Rows of the slab dc_conv_fwd_kn writes for this layer when it is handed both weight images: one per 224-pixel tile where the planner gives
// the layer to igemm224.hip (27 648 pixels: 124 rows instead of the 216 of dc_conv_stat_rows; 13 824 pixels: 62, short enough for the kernels
// that run the BatchNorm finalize themselves), dc_conv_stat_rows everywhere else.  Pass the value as slab_rows.
// the pointwise tile a forward launch with both weight images would run on (pw384_plan's code: 224 / 226 / 192 / ...), 0: another kernel
static int kn_forward_plan(const dc_conv_desc* d, int N, int Hi, int Wi, long* Mout) {
  const int plain = dc_conv_stat_rows(d, N, Hi, Wi);
  if (plain <= 0 || d->dtype != DC_BF16) return 0;
  IgemmParams p;
  if (!build_geom(*d, Hi, Wi, kFwd, &p.g)) return 0;
  const long M = (long)N * p.g.Qh * p.g.Qw;
  if (M >= (1L << 31) - BM) return 0;
  if (M <= TINY_M || (g_thin_fwd && thin_fwd_eligible(p.g, d->dtype, false, 0, false))) return 0;      // (run_gather asks these first)
  p.x = nullptr; p.w = nullptr; p.y = nullptr; p.bias = nullptr; p.slab = nullptr;
  p.N = N; p.ldx = weight_ld(p.g.Cin); p.ldy = weight_ld(p.g.Cout); p.ldw = weight_ld(p.g.Cin);
  alignas(16) static const char some_image[16] = {0};
  p.w_kn = some_image;                       // (any 16-byte aligned non-null pointer: eligibility only looks at it)
  p.ldw_kn = weight_ld(p.g.Cout);
  p.M = (int)M; p.m_beg = 0; p.phase_fast = 0; p.zero_page = nullptr; p.ngroup = 0; p.ksplit = 0; p.kslab = nullptr; p.zfill = 0;
  p.bst = BnBwdEpi{nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0};
  p.mtiles = cdiv(M, BM); p.accumulate = 0;
  *Mout = M;
  return pw384_plan(p);
}
extern "C" int dc_conv_stat_rows_kn(const dc_conv_desc* d, int N, int Hi, int Wi) {
  long M = 0;
  const int npb = kn_forward_plan(d, N, Hi, Wi, &M);
  return (npb == 224 || npb == 226) ? cdiv(M, 224) : dc_conv_stat_rows(d, N, Hi, Wi);
}
// 1: dc_conv_fwd_kn takes slab_rows = -1 for this layer -- stat_slab is then a SUM ROW, double[2][Cout], that the caller has zeroed and every
// tile of the launch adds its channel sums to (fp64 atomics; the sum of a few hundred fp32 numbers is exact in fp64, so the row does not
// depend on the order they arrive in).  dc_bn_finalize, dc_bn_apply_fin and dc_dwconv_fwd_fin take it with rows = -1; reading two numbers
// per channel, the consumers run the finalize themselves at any tensor size.
extern "C" int dc_conv_sum_row_kn(const dc_conv_desc* d, int N, int Hi, int Wi) {
  if (d == nullptr) return 0;
  long M = 0;
  const int npb = kn_forward_plan(d, N, Hi, Wi, &M);
  return npb == 224 || npb == 226 || npb == 192 ? 1 : 0;      // igemm224.hip's tiles and igemm192.hip's
}
extern "C" int dc_conv_dgrad_kn(const dc_conv_desc* d, int N, int Hi, int Wi, const void* dy, int lddy, const void* wb, const void* wf,
                                void* dx, int lddx, int accumulate, void* stream) {
  return run_gather(d, kDgrad, N, Hi, Wi, dy, lddy, wb, nullptr, dx, lddx, nullptr, accumulate, stream, false, nullptr, wf);
}

// `count` "same" dilated 3x3 convolutions (stride 1, pad == dil) of ONE input in ONE launch of the 256-tile kernel: the three
// atrous branches of the ASPP head read the same encoder output and are 108 tiles each at local batch 8 (27 at batch 2) on 256
// CUs; together they fill the chip.  Per output element the arithmetic is that of dc_conv_fwd (same tile, same K order): the
// results are bit-identical to `count` separate calls, which is also the fallback when the kernel does not serve the layer.
// With a workspace of dc_conv_fwd_dilated_group_workspace(...) bytes (> 0 where the launch would leave more than half of the chip idle under a
// long K loop: local batch 2, 81 tiles of 576 K steps) the launch is cut along K: `splits` workgroups per tile leave fp32 partial tiles in
// the workspace and a second kernel sums them in the order of the splits, stores the outputs and takes the BatchNorm sums.  The results are
// then those of another order of the K sum (not the bits of the unsplit launch).
static int dilated_group_impl(const dc_conv_desc* d, int N, int Hi, int Wi, int count, const int* dils, const void* x, int ldx,
                              const void* const* wfs, void* const* ys, int ldy, float* const* stat_slabs, void* ws, size_t ws_bytes,
                              size_t* want_bytes, void* stream) {
  DC_REQUIRE(d != nullptr && dils != nullptr && (want_bytes != nullptr || (wfs != nullptr && ys != nullptr)), "dc_conv_fwd_dilated_group: null argument");
  DC_REQUIRE(count >= 1 && count <= IgemmParams::MAXGROUP, "dc_conv_fwd_dilated_group: count must be 1..4");
  DC_REQUIRE(!d->transposed && d->k == 3 && d->stride == 1, "dc_conv_fwd_dilated_group: 3x3, stride 1 convolutions only");
  if (want_bytes != nullptr) *want_bytes = 0;
  for (int b = 0; b < count; ++b) DC_REQUIRE(dils[b] >= 1 && (want_bytes != nullptr || (wfs[b] != nullptr && ys[b] != nullptr)), "dc_conv_fwd_dilated_group: bad member");
  const bool fused = count > 1 && d->dtype == DC_BF16 && g_igemm256 != 0;
  if (!fused) {
    if (want_bytes != nullptr) return 0;
    for (int b = 0; b < count; ++b) {
      dc_conv_desc db = *d;
      db.dil = db.pad = dils[b];
      if (int e = run_gather(&db, kFwd, N, Hi, Wi, x, ldx, wfs[b], nullptr, ys[b], ldy, stat_slabs ? stat_slabs[b] : nullptr, 0, stream)) return e;
    }
    return 0;
  }
  dc_conv_desc du = *d;
  du.dil = du.pad = 1;                      // unit-dilation tap table; the kernel scales the offsets per member
  IgemmParams p;
  if (!build_geom(du, Hi, Wi, kFwd, &p.g)) return dc_fail("dc_conv_fwd_dilated_group: unsupported geometry", __FILE__, __LINE__);
  if (want_bytes == nullptr) {
    if (int e = check_view(x, ldx, p.g.Cin, d->dtype, "dc_conv_fwd_dilated_group input")) return e;
    for (int b = 0; b < count; ++b) {
      if (int e = check_view(ys[b], ldy, p.g.Cout, d->dtype, "dc_conv_fwd_dilated_group output")) return e;
      DC_REQUIRE(((uintptr_t)wfs[b] & 15) == 0, "dc_conv_fwd_dilated_group: weights unaligned");
    }
  }
  const long M = (long)N * p.g.Qh * p.g.Qw;
  DC_REQUIRE(N > 0 && M < (1L << 31) - BM, "dc_conv_fwd_dilated_group: bad pixel count");
  p.x = x; p.w = wfs ? wfs[0] : nullptr; p.y = ys ? ys[0] : nullptr; p.bias = nullptr; p.slab = stat_slabs ? stat_slabs[0] : nullptr;
  p.ksplit = 0; p.kslab = nullptr; p.zfill = 0;
  p.bst = BnBwdEpi{nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0};
  p.N = N; p.ldx = ldx; p.ldy = ldy;
  p.ldw = weight_ld(p.g.Cin);
  p.w_kn = nullptr; p.ldw_kn = 0;
  p.M = (int)M; p.m_beg = 0; p.phase_fast = 0; p.zero_page = nullptr;
  p.mtiles = cdiv(M, BM);
  p.accumulate = 0;
  p.ngroup = count;
  size_t need = 0;
  const int splits = igemm256_splitk_plan(p, &need);
  if (want_bytes != nullptr) {
    *want_bytes = need;
    return 0;
  }
  for (int b = 0; b < IgemmParams::MAXGROUP; ++b) p.gdil[b] = b < count ? dils[b] : 1;
  for (int b = 1; b < IgemmParams::MAXGROUP; ++b) {
    p.gw[b - 1] = b < count ? wfs[b] : nullptr;
    p.gy[b - 1] = b < count ? ys[b] : nullptr;
    p.gslab[b - 1] = (b < count && stat_slabs) ? stat_slabs[b] : nullptr;
  }
  if (stat_slabs)
    for (int b = 0; b < count; ++b) DC_REQUIRE(stat_slabs[b] != nullptr, "dc_conv_fwd_dilated_group: statistics for all members or none");
  if (splits > 1 && ws != nullptr && ws_bytes >= need) {
    DC_REQUIRE(((uintptr_t)ws & 15) == 0, "dc_conv_fwd_dilated_group_ws: workspace unaligned");
    return launch_igemm256_splitk(p, splits, ws, (hipStream_t)stream);
  }
  return launch_igemm256(p, (hipStream_t)stream);
}

extern "C" int dc_conv_fwd_dilated_group(const dc_conv_desc* d, int N, int Hi, int Wi, int count, const int* dils, const void* x,
                                         int ldx, const void* const* wfs, void* const* ys, int ldy, float* const* stat_slabs,
                                         void* stream) {
  return dilated_group_impl(d, N, Hi, Wi, count, dils, x, ldx, wfs, ys, ldy, stat_slabs, nullptr, 0, nullptr, stream);
}

extern "C" size_t dc_conv_fwd_dilated_group_workspace(const dc_conv_desc* d, int N, int Hi, int Wi, int count, const int* dils) {
  size_t need = 0;
  if (dilated_group_impl(d, N, Hi, Wi, count, dils, nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, 0, &need, nullptr) != 0) return 0;
  return need;
}

extern "C" int dc_conv_fwd_dilated_group_ws(const dc_conv_desc* d, int N, int Hi, int Wi, int count, const int* dils, const void* x,
                                            int ldx, const void* const* wfs, void* const* ys, int ldy, float* const* stat_slabs,
                                            void* ws, size_t ws_bytes, void* stream) {
  return dilated_group_impl(d, N, Hi, Wi, count, dils, x, ldx, wfs, ys, ldy, stat_slabs, ws, ws_bytes, nullptr, stream);
}

extern "C" int dc_conv_fwd_f32out(const dc_conv_desc* d, int N, int Hi, int Wi, const void* x, int ldx, const void* wf,
                                  float* y, int ldy, void* stream) {
  return run_gather(d, kFwd, N, Hi, Wi, x, ldx, wf, nullptr, y, ldy, nullptr, 0, stream, true);
}

extern "C" int dc_conv_dgrad(const dc_conv_desc* d, int N, int Hi, int Wi, const void* dy, int lddy,
                             const void* wb, void* dx, int lddx, int accumulate, void* stream) {
  return run_gather(d, kDgrad, N, Hi, Wi, dy, lddy, wb, nullptr, dx, lddx, nullptr, accumulate, stream);
}

// Data gradient whose output is the gradient w.r.t. a BatchNorm(+ReLU) output y_bn = act(bn(y)): also leaves that BatchNorm's
// backward sums in `slab` ([2][rows][C], rows = dc_conv_dgrad_bnstats_rows), so that dc_bn_bwd_reduce is not needed.
extern "C" int dc_conv_dgrad_bnstats_rows(const dc_conv_desc* d, int N, int Hi, int Wi) {
  if (d == nullptr) return 0;
  GatherGeom g;
  if (!build_geom(*d, Hi, Wi, kDgrad, &g)) return 0;
  const long M = (long)N * g.Qh * g.Qw;
  return (int)(cdiv(M, BM) * g.os * g.os);
}

extern "C" int dc_conv_dgrad_bnstats(const dc_conv_desc* d, int N, int Hi, int Wi, const void* dy, int lddy, const void* wb,
                                     void* dx, int lddx, const void* y, int ldy, const float* mean, const float* invstd,
                                     const float* mscale, const float* mshift, int relu, float* slab, void* stream) {
  const BnBwdEpi bst{y, ldy, mean, invstd, mscale, mshift, relu};
  return run_gather(d, kDgrad, N, Hi, Wi, dy, lddy, wb, nullptr, dx, lddx, slab, 0, stream, false, &bst);
}

extern "C" int dc_conv_packed_elems(const dc_conv_desc* d, size_t* wf_elems, size_t* wb_elems) {
  DC_REQUIRE(d != nullptr, "dc_conv_packed_elems: null descriptor");
  const size_t k = d->transposed ? 3 : d->k;
  if (wf_elems) *wf_elems = k * k * d->cout * (size_t)weight_ld(d->cin);
  if (wb_elems) *wb_elems = k * k * d->cin * (size_t)weight_ld(d->cout);
  return 0;
}

extern "C" int dc_conv_pack_weights(const dc_conv_desc* d, const float* master, void* wf, void* wb, void* stream) {
  DC_REQUIRE(d != nullptr && master != nullptr, "dc_conv_pack_weights: null argument");
  const int k = d->transposed ? 3 : d->k;
  const long total = (long)d->cin * d->cout * k * k;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  if (d->dtype == DC_BF16)
    hipLaunchKernelGGL(pack_weights_kernel<bf16>, dim3(blocks), dim3(256), 0, st, master, (bf16*)wf, (bf16*)wb, d->cin, d->cout, k * k, d->transposed);
  else
    hipLaunchKernelGGL(pack_weights_kernel<float>, dim3(blocks), dim3(256), 0, st, master, (float*)wf, (float*)wb, d->cin, d->cout, k * k, d->transposed);
  DC_CHECK_LAUNCH();
  return 0;
}
