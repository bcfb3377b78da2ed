// Weight gradient of every dense convolution:  dW_t[co][ci] = sum over output pixels m of dy[m][co] * x[gather(m,t)][ci].
//
// A GEMM whose reduction axis is the PIXEL axis, i.e. the slow axis of both NHWC operands.  The bf16 MFMA wants
// 8 consecutive reduction elements per lane, so the LDS tiles stay [pixel][channel] (coalesced global loads) and
// fragments are fetched with the gfx950 transposed LDS read ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane
// group, delivered channel-per-lane).  The f32 MFMA (16x16x4) takes ONE element per lane, so it reads the same
// image with plain ds_read_b32.  The pixel axis is split across workgroups (split-K); partial 128x128 fp32 tiles
// go to a slab with plain stores and a second kernel reduces them in a fixed order straight into the PyTorch
// master layout of the gradient -- deterministic, no float atomics.
#include "conv_geom.h"
#include "wgrad.h"

namespace dc {

template <typename T>
struct WgTraits;
template <>
struct WgTraits<bf16> {
  static constexpr int BP = 64;            // pixels per step
  static constexpr int ROW = 128 * 2 + 32; // padded LDS row bytes (conflict-free transposed reads)
};
template <>
struct WgTraits<float> {
  static constexpr int BP = 32;
  static constexpr int ROW = 128 * 4 + 64;
};

template <typename T>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) {
  using TR = WgTraits<T>;
  constexpr int KPV = Elem<T>::kPerVec;
  constexpr int BP = TR::BP, ROW = TR::ROW;
  constexpr int VPR = 128 * (int)sizeof(T) / 16;   // 16-byte vectors per tile row: 16 / 32
  constexpr int TILE = BP * ROW;                   // bytes per operand tile
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const GatherGeom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware order (see igemm.hip): workgroups sharing an XCD take consecutive tiles, and consecutive tiles cover all
  // (ci, co) tiles and then all taps of ONE pixel split, so its x / dy pixel panels (the 9 taps read the same pixels,
  // shifted) are pulled into that XCD's L2 once.
  const int nci = (g.Cin + 127) / 128, nco = (g.Cout + 127) / 128;
  const int nwg = gridDim.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
  const int ci0 = (tile % nci) * 128;
  tile /= nci;
  const int co0 = (tile % nco) * 128;
  tile /= nco;
  const int tapi = tile % g.ntaps, split = tile / g.ntaps;
  const Tap tap = g.taps[tapi];
  const int py = tap.phase / g.os, px = tap.phase % g.os;
  const int mbeg = split * p.chunk;
  const int mend = min(p.M, mbeg + p.chunk);
  const int steps = mend > mbeg ? (mend - mbeg + BP - 1) / BP : 0;

  const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ dg = reinterpret_cast<const T*>(p.dy);

  // each thread moves 4 vectors per operand per step: vector v = tid + 256*i -> (row = v / VPR, col = v % VPR)
  vec16 rq[4], rp[4];
  // x = act(y * xscale + xshift) formed here (bf16 only): a thread's column, i.e. its KPV channels, is the same for its four vectors
  // (256 is a multiple of VPR); vectors that were not loaded (padding, halo) stay zero
  [[maybe_unused]] unsigned xvalid = 0;
  [[maybe_unused]] float xsc[KPV], xsh[KPV];
  const bool xform = sizeof(T) == 2 && p.xscale != nullptr;
  if (xform) {
    const int ci = ci0 + (tid % VPR) * KPV;
#pragma unroll
    for (int e = 0; e < KPV; ++e) {
      xsc[e] = ci + e < g.Cin ? p.xscale[ci + e] : 0.f;
      xsh[e] = ci + e < g.Cin ? p.xshift[ci + e] : 0.f;
    }
  }
  auto load_step = [&](int s) {
    xvalid = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int v = tid + 256 * i;
      const int row = v / VPR, col = v % VPR;
      const int m = mbeg + s * BP + row;
      rq[i] = zero16();
      rp[i] = zero16();
      if (m < mend) {
        const int n = m / (g.Qh * g.Qw);
        const int rem = m - n * (g.Qh * g.Qw);
        const int qy = rem / g.Qw, qx = rem - qy * g.Qw;
        const int co = co0 + col * KPV;
        if (co < g.Cout) {
          const int oy = qy * g.os + py, ox = qx * g.os + px;
          rq[i] = ldg16(dg + ((size_t)(n * g.Hout + oy) * g.Wout + ox) * p.lddy + co);
        }
        const int ci = ci0 + col * KPV;
        const int iy = qy * g.is + tap.dy, ix = qx * g.is + tap.dx;
        if (ci < g.Cin && (unsigned)iy < (unsigned)g.Hin && (unsigned)ix < (unsigned)g.Win) {
          rp[i] = ldg16(xg + ((size_t)(n * g.Hin + iy) * g.Win + ix) * p.ldx + ci);
          xvalid |= 1u << i;
        }
      }
    }
  };
  auto store_step = [&](int buf) {
    char* q = smem + buf * (2 * TILE);
    char* pp = q + TILE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int v = tid + 256 * i;
      const int row = v / VPR, col = v % VPR;
      if constexpr (sizeof(T) == 2) {
        if (xform && (xvalid >> i & 1)) {
          float f[KPV];
          unpack(rp[i], f, T());
#pragma unroll
          for (int e = 0; e < KPV; ++e) {
            f[e] = fmaf(f[e], xsc[e], xsh[e]);
            if (p.xrelu) f[e] = fmaxf(f[e], 0.f);
          }
          pack(rp[i], f, T());
        }
      }
      *reinterpret_cast<vec16*>(q + row * ROW + col * 16) = rq[i];
      *reinterpret_cast<vec16*>(pp + row * ROW + col * 16) = rp[i];
    }
  };

  f32x4 acc[4][4];  // [co rep][ci rep]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int wq = wave >> 1, wp = wave & 1;
  const int fr = lane & 15, fg = lane >> 4;

  if (steps > 0) {
    load_step(0);
    store_step(0);
  }
  __syncthreads();
  for (int s = 0; s < steps; ++s) {
    const bool more = (s + 1) < steps;
    if (more) load_step(s + 1);
    const char* q = smem + (s & 1) * (2 * TILE);
    const char* pp = q + TILE;
    if constexpr (sizeof(T) == 2) {
      // two K=32 sub-steps; lane group fg owns pixels 8*fg .. 8*fg+7 of the sub-step
      const int tq = (lane & 15) >> 2, tp = lane & 3;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int prow = ks * 32 + 8 * fg + tq;
        vec16 fa[4], fb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int cq = (wq * 64 + i * 16 + 4 * tp) * 2;
          const int cp = (wp * 64 + i * 16 + 4 * tp) * 2;
          typedef __attribute__((address_space(3))) short4v lds_s4;
          const short4v a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(q + prow * ROW + cq));
          const short4v a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(q + (prow + 4) * ROW + cq));
          const short4v b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(pp + prow * ROW + cp));
          const short4v b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(pp + (prow + 4) * ROW + cp));
          uint2 t0 = __builtin_bit_cast(uint2, a0), t1 = __builtin_bit_cast(uint2, a1);
          fa[i].w[0] = t0.x; fa[i].w[1] = t0.y; fa[i].w[2] = t1.x; fa[i].w[3] = t1.y;
          t0 = __builtin_bit_cast(uint2, b0); t1 = __builtin_bit_cast(uint2, b1);
          fb[i].w[0] = t0.x; fb[i].w[1] = t0.y; fb[i].w[2] = t1.x; fb[i].w[3] = t1.y;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
      }
    } else {
      // 8 sub-steps of 4 pixels: lane group fg owns pixel 4*ks + fg
#pragma unroll
      for (int ks = 0; ks < BP / 4; ++ks) {
        const int prow = ks * 4 + fg;
        float fa[4], fb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          fa[i] = *reinterpret_cast<const float*>(q + prow * ROW + (wq * 64 + i * 16 + fr) * 4);
          fb[i] = *reinterpret_cast<const float*>(pp + prow * ROW + (wp * 64 + i * 16 + fr) * 4);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    }
    if (more) store_step((s + 1) & 1);
    __syncthreads();
  }

  // D[row = co][col = ci]: lane holds co = (lane>>4)*4 + r, ci = lane&15 of each 16x16 tile
  float* out = p.slab + ((size_t)split * g.ntaps + tap.widx) * g.Cout * g.Cin;  // widx: the master's (ky,kx) index
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + wq * 64 + i * 16 + fg * 4 + r;
      if (co >= g.Cout) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ci = ci0 + wp * 64 + j * 16 + fr;
        if (ci < g.Cin) out[(size_t)co * g.Cin + ci] = acc[i][j][r];
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// LDS-DMA version (default): operands go global -> LDS with global_load_lds_dwordx4 into a 3-stage ring (16 KiB per stage:
// 32 pixels x 128 channels bf16, or 16 pixels fp32), counted vmcnt so that two stages stay in flight across the raw
// s_barrier, 48 KiB of LDS -> 3 workgroups per CU.  LDS-DMA writes are lane-linear, so rows cannot be padded; the
// transposed reads are kept conflict-free by an XOR swizzle of 32-byte chunks (64-byte chunks for fp32) applied to the
// per-lane SOURCE address and mirrored in the fragment reads.
static __device__ __attribute__((aligned(256))) unsigned char dc_wg_zero_page[256];
typedef __attribute__((address_space(1))) const void* wg_gas_ptr;
typedef __attribute__((address_space(3))) void* wg_lds_ptr;

template <typename T>
__global__ __launch_bounds__(256) void wgrad_dma_kernel(const WgradParams p) {
  constexpr int KPV = Elem<T>::kPerVec;
  constexpr int ROW = 128 * (int)sizeof(T);        // bytes per LDS row (one pixel, 128 channels)
  constexpr int SPR = ROW / 16;                    // 16-byte slots per row: 16 / 32
  constexpr int BP = 8192 / ROW;                   // pixels per stage: 32 / 16
  constexpr int RPI = 1024 / ROW;                  // rows per LDS-DMA instruction: 4 / 2
  constexpr int TILE = 8192;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const GatherGeom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nci = (g.Cin + 127) / 128, nco = (g.Cout + 127) / 128;
  const int nwg = gridDim.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, xslot = blockIdx.x >> 3;
  int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + xslot;
  const int ci0 = (tile % nci) * 128;
  tile /= nci;
  const int co0 = (tile % nco) * 128;
  tile /= nco;
  const int tapi = tile % g.ntaps, split = tile / g.ntaps;
  const Tap tap = g.taps[tapi];
  const int py = tap.phase / g.os, px = tap.phase % g.os;
  const int mbeg = split * p.chunk;
  const int mend = min(p.M, mbeg + p.chunk);
  const int steps = mend > mbeg ? (mend - mbeg + BP - 1) / BP : 0;
  const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ dg = reinterpret_cast<const T*>(p.dy);

  // lane -> (row within the instruction, physical slot); the swizzle key depends on the row only.
  // The pixel coordinates (n, qy, qx) of each of the lane's two rows are decomposed ONCE and then advanced by BP pixels per
  // stage with carries: integer divisions per stage made the first version of this kernel VALU-bound.
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int lrow = lane / SPR, sp = lane % SPR;
  int rn[2], rqy[2], rqx[2], rlslot[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = RPI * (2 * wv + i) + lrow;
    if constexpr (sizeof(T) == 2) {
      const int key = (row & 3) | (((row >> 3) & 1) << 2);
      rlslot[i] = (((sp >> 1) ^ key) << 1) | (sp & 1);
    } else {
      rlslot[i] = (((sp >> 2) ^ (row & 1)) << 2) | (sp & 3);
    }
    const int m = mbeg + row;
    const int mm = m < p.M ? m : 0;
    rn[i] = mm / (g.Qh * g.Qw);
    const int rem = mm - rn[i] * (g.Qh * g.Qw);
    rqy[i] = rem / g.Qw;
    rqx[i] = rem - rqy[i] * g.Qw;
  }
  // running element offsets of the lane's two rows into dy and x (recomputed only when the pixel walk wraps a row)
  size_t offq[2], offp[2];
  bool cq_ok[2], cp_ok[2];
  auto recompute = [&](int i) {
    const int n = rn[i], qy = rqy[i], qx = rqx[i];
    const int oy = qy * g.os + py, ox = qx * g.os + px;
    offq[i] = ((size_t)(n * g.Hout + oy) * g.Wout + ox) * p.lddy + co0 + rlslot[i] * KPV;
    const int iy = qy * g.is + tap.dy, ix = qx * g.is + tap.dx;
    // (the x offset may be "out of the image" for halo taps: it is only dereferenced when the bounds test passes)
    offp[i] = (size_t)((long)((long)(n * g.Hin + iy) * g.Win + ix) * p.ldx + ci0 + rlslot[i] * KPV);
  };
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    cq_ok[i] = co0 + rlslot[i] * KPV < g.Cout;
    cp_ok[i] = ci0 + rlslot[i] * KPV < g.Cin;
    recompute(i);
  }
  const size_t stepq = (size_t)BP * g.os * p.lddy, stepp = (size_t)BP * g.is * p.ldx;
  auto issue = [&](int s, int buf) {
    char* q = smem + buf * (2 * TILE);
    char* pp = q + TILE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int j = 2 * wv + i;                    // instruction index within the stage: rows RPI*j ..
      const int row = RPI * j + lrow;
      const int m = mbeg + s * BP + row;
      const bool mok = m < mend;
      const int iy = rqy[i] * g.is + tap.dy, ix = rqx[i] * g.is + tap.dx;
      const bool pok = mok && cp_ok[i] && (unsigned)iy < (unsigned)g.Hin && (unsigned)ix < (unsigned)g.Win;
      const void* srcq = (mok && cq_ok[i]) ? (const void*)(dg + offq[i]) : (const void*)dc_wg_zero_page;
      const void* srcp = pok ? (const void*)(xg + offp[i]) : (const void*)dc_wg_zero_page;
      __builtin_amdgcn_global_load_lds((wg_gas_ptr)srcq, (wg_lds_ptr)(q + j * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((wg_gas_ptr)srcp, (wg_lds_ptr)(pp + j * 1024), 16, 0, 0);
      // advance this row by BP pixels
      rqx[i] += BP;
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
  };

  f32x4 acc[4][4];  // [co rep][ci rep]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int wq = wave >> 1, wp = wave & 1;
  const int fr = lane & 15, fg = lane >> 4;

  auto compute = [&](int buf) {
    const char* q = smem + buf * (2 * TILE);
    const char* pp = q + TILE;
    if constexpr (sizeof(T) == 2) {
      const int tq = (lane & 15) >> 2, tp = lane & 3;
      const int prow = 8 * fg + tq;                               // pixels 8fg..8fg+3, then +4
      const int key0 = (prow & 3) | (((prow >> 3) & 1) << 2);
      const int key1 = ((prow + 4) & 3) | ((((prow + 4) >> 3) & 1) << 2);
      vec16 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int cq = wq * 4 + i, cp = wp * 4 + i;                // 32-byte chunk = 16 channels
        typedef __attribute__((address_space(3))) short4v lds_s4;
        const short4v a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(q + prow * ROW + ((cq ^ key0) << 5) + 8 * tp));
        const short4v a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(q + (prow + 4) * ROW + ((cq ^ key1) << 5) + 8 * tp));
        const short4v b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(pp + prow * ROW + ((cp ^ key0) << 5) + 8 * tp));
        const short4v b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(pp + (prow + 4) * ROW + ((cp ^ key1) << 5) + 8 * tp));
        uint2 t0 = __builtin_bit_cast(uint2, a0), t1 = __builtin_bit_cast(uint2, a1);
        fa[i].w[0] = t0.x; fa[i].w[1] = t0.y; fa[i].w[2] = t1.x; fa[i].w[3] = t1.y;
        t0 = __builtin_bit_cast(uint2, b0); t1 = __builtin_bit_cast(uint2, b1);
        fb[i].w[0] = t0.x; fb[i].w[1] = t0.y; fb[i].w[2] = t1.x; fb[i].w[3] = t1.y;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int ks = 0; ks < BP / 4; ++ks) {
        const int prow = ks * 4 + fg;
        float fa[4], fb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          fa[i] = *reinterpret_cast<const float*>(q + prow * ROW + (((wq * 4 + i) ^ (prow & 1)) << 6) + fr * 4);
          fb[i] = *reinterpret_cast<const float*>(pp + prow * ROW + (((wp * 4 + i) ^ (prow & 1)) << 6) + fr * 4);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    }
  };

  if (steps > 0) issue(0, 0);
  if (steps > 1) issue(1, 1);
  for (int s = 0; s < steps; ++s) {
    if (s + 1 < steps) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (s + 2 < steps) issue(s + 2, (s + 2) % 3);
    compute(s % 3);
  }

  float* out = p.slab + ((size_t)split * g.ntaps + tap.widx) * g.Cout * g.Cin;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + wq * 64 + i * 16 + fg * 4 + r;
      if (co >= g.Cout) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ci = ci0 + wp * 64 + j * 16 + fr;
        if (ci < g.Cin) out[(size_t)co * g.Cin + ci] = acc[i][j][r];
      }
    }
}

// grad[master layout] = sum over splits of slab[split][tap][co][ci].  One thread = 4 consecutive ci of one (tap, co):
// float4 loads, the split loop unrolled with independent accumulators (the first version walked the splits with one
// dependent scalar load at a time and cost 6 ms per step under load).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ grad, int splits, int taps,
                                                           int Co, int Ci, int transposed) {
  const long per = (long)Co * Ci;
  const long per4 = per >> 2;                       // Ci is a multiple of 4 for every layer (checked on the host)
  const long total = per4 * taps;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int t = (int)(idx / per4);
    const long i = (idx % per4) << 2;
    const float4* src = reinterpret_cast<const float4*>(slab + (size_t)t * per + i);
    const size_t stride4 = ((size_t)taps * per) >> 2;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
    int s = 0;
    // eight slabs in flight first (the 728-channel layers have 21 splits: 6 dependent rounds of four became 2 of eight + 1 of
    // four + 1); fixed combination order, so the result does not depend on anything but the split count
    float4 b0 = a0, b1 = a0, b2 = a0, b3 = a0;
    for (; s + 8 <= splits; s += 8) {
      const float4 v0 = src[(size_t)s * stride4], v1 = src[(size_t)(s + 1) * stride4];
      const float4 v2 = src[(size_t)(s + 2) * stride4], v3 = src[(size_t)(s + 3) * stride4];
      const float4 v4 = src[(size_t)(s + 4) * stride4], v5 = src[(size_t)(s + 5) * stride4];
      const float4 v6 = src[(size_t)(s + 6) * stride4], v7 = src[(size_t)(s + 7) * stride4];
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
      a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
      a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
      a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
      b0.x += v4.x; b0.y += v4.y; b0.z += v4.z; b0.w += v4.w;
      b1.x += v5.x; b1.y += v5.y; b1.z += v5.z; b1.w += v5.w;
      b2.x += v6.x; b2.y += v6.y; b2.z += v6.z; b2.w += v6.w;
      b3.x += v7.x; b3.y += v7.y; b3.z += v7.z; b3.w += v7.w;
    }
    a0.x += b0.x; a0.y += b0.y; a0.z += b0.z; a0.w += b0.w;
    a1.x += b1.x; a1.y += b1.y; a1.z += b1.z; a1.w += b1.w;
    a2.x += b2.x; a2.y += b2.y; a2.z += b2.z; a2.w += b2.w;
    a3.x += b3.x; a3.y += b3.y; a3.z += b3.z; a3.w += b3.w;
    for (; s + 4 <= splits; s += 4) {
      const float4 v0 = src[(size_t)s * stride4], v1 = src[(size_t)(s + 1) * stride4];
      const float4 v2 = src[(size_t)(s + 2) * stride4], v3 = src[(size_t)(s + 3) * stride4];
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
      a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
      a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
      a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
    }
    for (; s < splits; ++s) {
      const float4 v0 = src[(size_t)s * stride4];
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
    }
    const float r[4] = {(a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z), (a0.w + a1.w) + (a2.w + a3.w)};
    const int co = (int)(i / Ci), ci = (int)(i % Ci);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long base = transposed ? ((long)(ci + e) * Co + co) * taps : ((long)co * Ci + ci + e) * taps;
      grad[base + t] = r[e];
    }
  }
}

static int g_wgrad_mode = 1;               // 1 = LDS-DMA 3-stage kernel, 0 = register-staged kernel
static int g_thin_wgrad = 1;               // one-pass kernel for the two thin stem convolutions (thinconv.hip)
static int g_wgrad384_fill = 66;           // fewest percent of a 384-wide tile row a pointwise layer's input channels must fill ("wgrad384_fill")
static int g_wgrad384 = 1;                 // 256 x 384 pointwise kernel: 0 never, 1 planner, 2 wherever eligible (tests)
static int g_wgrad_target_blocks = 768;    // resident capacity: 256 CUs x 3 workgroups (48 KiB LDS, 146 registers)

// Fewest K steps (of BP pixels) a pixel split may have (tuning switch "wgrad_min_steps").  Every split costs a 64 KiB fp32 slab
// tile written and read back: with two steps as the floor, the 128 -> 128 layer at 384 x 576 and local batch 2 was cut into 768
// splits of 9 steps (50 MB of slabs for a 64 KiB result).  Whole-step A/B at 2 / 8 / 16 / 32 / 64 steps: local batch 2 14.71 /
// 14.54 / 14.43 / 14.45 / 14.86 ms, batch 4 23.26 / 23.20 / 23.08 / 23.18 / 23.32, batch 8 unchanged (40.9-41.0).
static int g_wgrad_min_steps = 16;
static void plan_splits(const GatherGeom& g, long M, int BP, int* splits, int* chunk) {
  const long tiles = (long)cdiv(g.Cin, 128) * cdiv(g.Cout, 128) * g.ntaps;
  long want = g_wgrad_target_blocks / tiles;   // FLOOR: all workgroups must be co-resident (3 per CU), a second partial wave costs more than it buys
  const long maxs = (M + (long)g_wgrad_min_steps * BP - 1) / ((long)g_wgrad_min_steps * BP);   // at least that many steps per split
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  if (want > 1024) want = 1024;
  long c = (M + want - 1) / want;
  c = (c + BP - 1) / BP * BP;
  *chunk = (int)c;
  *splits = (int)((M + c - 1) / c);
}

}  // namespace dc

using namespace dc;

extern "C" int dc_wgrad_set_min_steps(int n) {
  if (n < 1) return dc_fail("dc_set_option: wgrad_min_steps must be positive", __FILE__, __LINE__);
  g_wgrad_min_steps = n;
  return 0;
}

extern "C" int dc_wgrad_set_mode(int m) {
  g_wgrad_mode = m ? 1 : 0;
  return 0;
}


extern "C" int dc_wgrad_set_384(int m) {
  g_wgrad384 = m;
  return 0;
}
extern "C" int dc_wgrad_set_384_fill(int pct) {
  g_wgrad384_fill = pct;
  return 0;
}
extern "C" int dc_wgrad_set_384_slots(int n) {
  if (n < 1) return dc_fail("dc_set_option: wgrad384_slots must be positive", __FILE__, __LINE__);
  wgrad384_set_slots(n);
  return 0;
}
extern "C" int dc_wgrad_set_384_min_stages(int n) {
  if (n < 1) return dc_fail("dc_set_option: wgrad384_min_stages must be positive", __FILE__, __LINE__);
  wgrad384_set_min_stages(n);
  return 0;
}

extern "C" int dc_wgrad_set_thin(int m) {
  g_thin_wgrad = m ? 1 : 0;
  return 0;
}

// (Round 6: the 256 x 256 weight-gradient kernel of rounds 2 - 4, wgrad256.hip, served one launch per step after wgrad384.hip took the pointwise,
// 3 x 3 and transposed layers; with that launch on the 128-tile kernel the step is the same -- 31.14 against 31.17 ms -- and the file is gone.)
// Pointwise layers whose input-channel extent pads no more on 384-wide tiles than on 256-wide ones go to the 256 x 384 kernel (wgrad384.hip)
static bool wgrad384_wins(const GatherGeom& g) {
  if (g_wgrad384 == 0 || !wgrad384_eligible(g, 0, 0, 0)) return false;
  if (g_wgrad384 == 2) return true;
  if (wgrad384_is_tconv(g))      // ConvTranspose2d: x is the 256-wide linear operand, [tap][dy channel] the gathered axis; "wgrad384" = 4: off
    return g_wgrad384 != 3 && g_wgrad384 != 4 && g.Cout >= 64 && (long)cdiv(g.Cin, 256) * 256 * 3 <= (long)g.Cin * 4;
  if (g.ntaps > 1) {
    // 3 x 3 "same" convolutions: the x axis is [tap][ci] in quads of 64 channels, so any input width fills the 384-wide tiles to within
    // half a tile (304 -> 256: 45 quads = 7.5 tiles); the layer must fill the 256 output channels of a tile to three quarters
    return g_wgrad384 != 3 && g.Cin >= 64 && (long)cdiv(g.Cout, 256) * 256 * 3 <= (long)g.Cout * 4;
  }
  // pointwise: the x axis in quads of 64 channels must fill its 384-wide tiles to g_wgrad384_fill percent (256 channels: 4 of 6 quads = 67 %),
  // the output channels their 256-wide tiles to three quarters
  const long nq = cdiv(g.Cin, 64), nxt = cdiv(nq, 6);
  return nq * 100 >= nxt * 6 * g_wgrad384_fill && (long)cdiv(g.Cout, 256) * 256 * 3 <= (long)g.Cout * 4;
}
// the "big tile" plan and launch of a layer the 256 x 384 planner accepts
static void wgrad_big_plan(const GatherGeom& g, long M, int* splits, int* chunk, int group = 1) { wgrad384_plan(g, M, splits, chunk, group); }
static int launch_wgrad_big(const WgradParams& p, hipStream_t st, int group = 1, const void* const* xs = nullptr, const void* const* dys = nullptr,
                            float* const* slabs = nullptr) {
  return launch_wgrad384(p, st, group, xs, dys, slabs);
}

static int launch_wgrad_reduce(const float* slab, float* grad_w, int splits, const GatherGeom& g, int transposed, hipStream_t st) {
  const long per = (long)g.Cout * g.Cin;
  const long work = (per >> 2) * g.ntaps;
  const int blocks = (int)((work + 255) / 256 > 4096 ? 4096 : (work + 255) / 256);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, slab, grad_w, splits, g.ntaps, g.Cout, g.Cin, transposed);
  DC_CHECK_LAUNCH();
  return 0;
}

extern "C" int dc_wgrad_set_target_blocks(int n) {
  if (n > 0) g_wgrad_target_blocks = n;
  return 0;
}

extern "C" size_t dc_conv_wgrad_workspace(const dc_conv_desc* d, int N, int Hi, int Wi) {
  GatherGeom g;
  if (d == nullptr || !build_geom(*d, Hi, Wi, kFwd, &g)) return 0;
  int splits, chunk;
  plan_splits(g, (long)N * g.Qh * g.Qw, d->dtype == DC_BF16 ? 64 : 32, &splits, &chunk);
  if (d->dtype == DC_BF16) {   // either tile shape may serve the layer: size for the larger plan
    int s2, c2;
    wgrad_big_plan(g, (long)N * g.Qh * g.Qw, &s2, &c2);
    if (s2 > splits) splits = s2;
    if (thin_wgrad_eligible(*d, Hi, Wi)) {
      const int s3 = thin_wgrad_splits(*d, N, Hi, Wi);
      if (s3 > splits) splits = s3;
    }
  }
  return (size_t)splits * g.ntaps * g.Cout * g.Cin * sizeof(float);
}

// Which kernel serves a layer and how its pixel axis is split: shared by the reducing calls (dc_conv_wgrad, dc_conv_wgrad_group) and the
// slab-only calls (dc_conv_wgrad_partial + dc_fold_slabs).  transform: the operand is act(x * xscale + xshift) (register-staged kernel).
enum WgradKernel { WK_THIN, WK_256, WK_DMA, WK_REG };
static int plan_wgrad(const dc_conv_desc* d, int N, int Hi, int Wi, bool transform, int count, WgradParams* p, WgradKernel* kind) {
  if (!build_geom(*d, Hi, Wi, kFwd, &p->g)) return dc_fail("dc_conv_wgrad: unsupported geometry", __FILE__, __LINE__);
  const long M = (long)N * p->g.Qh * p->g.Qw;
  DC_REQUIRE(M < (1L << 31) - 256, "dc_conv_wgrad: too many pixels for 32-bit indexing");
  p->N = N; p->M = (int)M;
  if (count > 1) {
    DC_REQUIRE(!transform && d->dtype == DC_BF16 && !(g_thin_wgrad && thin_wgrad_eligible(*d, Hi, Wi)) && wgrad384_wins(p->g) &&
                   count <= WG384_MAXL,
               "dc_conv_wgrad: this layer is not served by the grouped launch");
    *kind = WK_256;
    wgrad_big_plan(p->g, M, &p->splits, &p->chunk, count);
    return 0;
  }
  const int BP = d->dtype == DC_BF16 ? 64 : 32;    // chunk granularity (a multiple of both kernels' pixels per stage)
  const bool thin = !transform && g_thin_wgrad && thin_wgrad_eligible(*d, Hi, Wi);
  const bool big = !transform && !thin && d->dtype == DC_BF16 && wgrad384_wins(p->g);
  if (thin) { p->splits = thin_wgrad_splits(*d, N, Hi, Wi); p->chunk = 0; *kind = WK_THIN; }
  else if (big) { wgrad_big_plan(p->g, M, &p->splits, &p->chunk); *kind = WK_256; }
  else { plan_splits(p->g, M, BP, &p->splits, &p->chunk); *kind = (g_wgrad_mode == 1 && !transform) ? WK_DMA : WK_REG; }
  return 0;
}

// the split-K partial sums of one layer (or of `group` layers of one geometry: 256-tile kernel only) into p.slab (slabs[]); no reduction
static int launch_wgrad_partial(const dc_conv_desc* d, const WgradParams& p, WgradKernel kind, int N, int Hi, int Wi, hipStream_t st, int group = 1,
                                const void* const* xs = nullptr, const void* const* dys = nullptr, float* const* slabs = nullptr) {
  dim3 grid(cdiv(p.g.Cin, 128) * cdiv(p.g.Cout, 128) * p.g.ntaps * p.splits);
  if (kind == WK_THIN) {
    if (int e = launch_thin_wgrad(*d, N, Hi, Wi, p.x, p.ldx, p.dy, p.lddy, p.slab, st)) return e;
  } else if (kind == WK_256) {
    if (int e = launch_wgrad_big(p, st, group, xs, dys, slabs)) return e;
  } else if (kind == WK_DMA) {
    const size_t lds = 3 * 2 * 8192;
    if (d->dtype == DC_BF16) {
      DC_ONCE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_dma_kernel<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(wgrad_dma_kernel<bf16>, grid, dim3(256), lds, st, p);
    } else {
      DC_ONCE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_dma_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(wgrad_dma_kernel<float>, grid, dim3(256), lds, st, p);
    }
  } else if (d->dtype == DC_BF16) {
    const size_t lds = 4 * (size_t)WgTraits<bf16>::BP * WgTraits<bf16>::ROW;
    DC_ONCE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(wgrad_kernel<bf16>, grid, dim3(256), lds, st, p);
  } else {
    const size_t lds = 4 * (size_t)WgTraits<float>::BP * WgTraits<float>::ROW;
    DC_ONCE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(wgrad_kernel<float>, grid, dim3(256), lds, st, p);
  }
  DC_CHECK_LAUNCH();
  return 0;
}

// xscale != nullptr: x is a raw convolution output and the operand is act(x * xscale + xshift) (WgradParams::xscale): the
// register-staged 128-tile kernel, whatever the planner would have picked
static int conv_wgrad_impl(const dc_conv_desc* d, int N, int Hi, int Wi, const void* x, int ldx, const void* dy,
                           int lddy, void* workspace, size_t workspace_bytes, float* grad_w, void* stream, const float* xscale,
                           const float* xshift, int xrelu) {
  DC_REQUIRE(d != nullptr && grad_w != nullptr && workspace != nullptr, "dc_conv_wgrad: null argument");
  DC_REQUIRE(xscale == nullptr || (d->dtype == DC_BF16 && xshift != nullptr), "dc_conv_wgrad: the operand transform is bf16 only");
  WgradParams p;
  WgradKernel kind;
  p.xscale = xscale; p.xshift = xshift; p.xrelu = xrelu;
  if (int e = plan_wgrad(d, N, Hi, Wi, xscale != nullptr, 1, &p, &kind)) return e;
  if (int e = dc_check_view(x, ldx, p.g.Cin, d->dtype, "dc_conv_wgrad x")) return e;
  if (int e = dc_check_view(dy, lddy, p.g.Cout, d->dtype, "dc_conv_wgrad dy")) return e;
  const size_t need = (size_t)p.splits * p.g.ntaps * p.g.Cout * p.g.Cin * sizeof(float);
  DC_REQUIRE(workspace_bytes >= need, "dc_conv_wgrad: workspace too small");
  p.x = x; p.dy = dy; p.slab = (float*)workspace;
  p.ldx = ldx; p.lddy = lddy;
  hipStream_t st = (hipStream_t)stream;
  if (int e = launch_wgrad_partial(d, p, kind, N, Hi, Wi, st)) return e;
  return launch_wgrad_reduce(p.slab, grad_w, p.splits, p.g, d->transposed, st);
}

extern "C" int dc_conv_wgrad(const dc_conv_desc* d, int N, int Hi, int Wi, const void* x, int ldx, const void* dy,
                             int lddy, void* workspace, size_t workspace_bytes, float* grad_w, void* stream) {
  return conv_wgrad_impl(d, N, Hi, Wi, x, ldx, dy, lddy, workspace, workspace_bytes, grad_w, stream, nullptr, nullptr, 0);
}

namespace dc {
int conv_wgrad_bnin(const dc_conv_desc* d, int N, int Hi, int Wi, const void* y, int ldy, const float* xscale, const float* xshift, int xrelu,
                    const void* dy, int lddy, void* workspace, size_t workspace_bytes, float* grad_w, void* stream) {
  return conv_wgrad_impl(d, N, Hi, Wi, y, ldy, dy, lddy, workspace, workspace_bytes, grad_w, stream, xscale, xshift, xrelu);
}
}  // namespace dc

// Grouped form: `count` layers of ONE geometry (same descriptor, extents and row strides) in one launch of the 256-tile
// kernel and one reduction per layer.  Layers the 256-tile kernel does not serve fall back to `count` plain calls.
static bool wgrad_group_eligible(const dc_conv_desc& d, const GatherGeom& g, int Hi, int Wi) {
  return d.dtype == DC_BF16 && !(g_thin_wgrad && thin_wgrad_eligible(d, Hi, Wi)) && wgrad384_wins(g);
}

extern "C" size_t dc_conv_wgrad_group_workspace(const dc_conv_desc* d, int N, int Hi, int Wi, int count) {
  GatherGeom g;
  if (d == nullptr || count < 1 || !build_geom(*d, Hi, Wi, kFwd, &g)) return 0;
  const size_t single = dc_conv_wgrad_workspace(d, N, Hi, Wi);
  if (count == 1 || !wgrad_group_eligible(*d, g, Hi, Wi) || count > WG384_MAXL) return single;
  int splits, chunk;
  wgrad_big_plan(g, (long)N * g.Qh * g.Qw, &splits, &chunk, count);
  const size_t grouped = (size_t)count * splits * g.ntaps * g.Cout * g.Cin * sizeof(float);
  return grouped > single ? grouped : single;
}

extern "C" int dc_conv_wgrad_group(const dc_conv_desc* d, int N, int Hi, int Wi, int count, const void* const* xs, int ldx,
                                   const void* const* dys, int lddy, void* workspace, size_t workspace_bytes, float* const* grad_ws,
                                   void* stream) {
  DC_REQUIRE(d != nullptr && xs != nullptr && dys != nullptr && grad_ws != nullptr && workspace != nullptr, "dc_conv_wgrad_group: null argument");
  DC_REQUIRE(count >= 1, "dc_conv_wgrad_group: empty group");
  WgradParams p;
  if (!build_geom(*d, Hi, Wi, kFwd, &p.g)) return dc_fail("dc_conv_wgrad_group: unsupported geometry", __FILE__, __LINE__);
  if (count == 1 || !wgrad_group_eligible(*d, p.g, Hi, Wi) || count > WG384_MAXL) {
    for (int l = 0; l < count; ++l)
      if (int e = dc_conv_wgrad(d, N, Hi, Wi, xs[l], ldx, dys[l], lddy, workspace, workspace_bytes, grad_ws[l], stream)) return e;
    return 0;
  }
  for (int l = 0; l < count; ++l) {
    DC_REQUIRE(grad_ws[l] != nullptr, "dc_conv_wgrad_group: null gradient pointer");
    if (int e = dc_check_view(xs[l], ldx, p.g.Cin, d->dtype, "dc_conv_wgrad_group x")) return e;
    if (int e = dc_check_view(dys[l], lddy, p.g.Cout, d->dtype, "dc_conv_wgrad_group dy")) return e;
  }
  const long M = (long)N * p.g.Qh * p.g.Qw;
  DC_REQUIRE(M < (1L << 31) - 256, "dc_conv_wgrad_group: too many pixels for 32-bit indexing");
  wgrad_big_plan(p.g, M, &p.splits, &p.chunk, count);
  const size_t per_layer = (size_t)p.splits * p.g.ntaps * p.g.Cout * p.g.Cin;   // floats
  DC_REQUIRE(workspace_bytes >= per_layer * count * sizeof(float), "dc_conv_wgrad_group: workspace too small");
  float* slabs[WG384_MAXL];
  for (int l = 0; l < count; ++l) slabs[l] = (float*)workspace + per_layer * l;
  p.x = xs[0]; p.dy = dys[0]; p.slab = slabs[0];
  p.N = N; p.ldx = ldx; p.lddy = lddy; p.M = (int)M;
  hipStream_t st = (hipStream_t)stream;
  if (int e = launch_wgrad_big(p, st, count, xs, dys, slabs)) return e;
  for (int l = 0; l < count; ++l)
    if (int e = launch_wgrad_reduce(slabs[l], grad_ws[l], p.splits, p.g, d->transposed, st)) return e;
  return 0;
}

// ---- slab-only form: the partial sums stay in per-layer slabs and dc_fold_slabs (fold.hip) adds the slabs of many layers in one launch.
// The 83 per-layer reductions of a step (17 MB of slabs each on average: launch- and latency-bound at 1.8 TB/s) become part of ~50 larger
// folds that also take the depthwise layers' rows.
extern "C" int dc_conv_wgrad_plan(const dc_conv_desc* d, int N, int Hi, int Wi, int count, int* splits, size_t* slab_bytes) {
  DC_REQUIRE(d != nullptr && splits != nullptr && slab_bytes != nullptr && count >= 1, "dc_conv_wgrad_plan: bad argument");
  WgradParams p;
  WgradKernel kind;
  if (int e = plan_wgrad(d, N, Hi, Wi, false, count, &p, &kind)) return e;
  *splits = p.splits;
  *slab_bytes = (size_t)p.splits * p.g.ntaps * p.g.Cout * p.g.Cin * sizeof(float);
  return 0;
}

extern "C" int dc_conv_wgrad_partial(const dc_conv_desc* d, int N, int Hi, int Wi, int count, const void* const* xs, int ldx,
                                     const void* const* dys, int lddy, float* const* slabs, int splits, void* stream) {
  DC_REQUIRE(d != nullptr && xs != nullptr && dys != nullptr && slabs != nullptr && count >= 1, "dc_conv_wgrad_partial: bad argument");
  WgradParams p;
  WgradKernel kind;
  if (int e = plan_wgrad(d, N, Hi, Wi, false, count, &p, &kind)) return e;
  DC_REQUIRE(p.splits == splits, "dc_conv_wgrad_partial: the split plan changed since dc_conv_wgrad_plan (tuning options were switched in between)");
  for (int l = 0; l < count; ++l) {
    DC_REQUIRE(slabs[l] != nullptr, "dc_conv_wgrad_partial: null slab");
    if (int e = dc_check_view(xs[l], ldx, p.g.Cin, d->dtype, "dc_conv_wgrad_partial x")) return e;
    if (int e = dc_check_view(dys[l], lddy, p.g.Cout, d->dtype, "dc_conv_wgrad_partial dy")) return e;
  }
  p.x = xs[0]; p.dy = dys[0]; p.slab = slabs[0];
  p.ldx = ldx; p.lddy = lddy;
  return launch_wgrad_partial(d, p, kind, N, Hi, Wi, (hipStream_t)stream, count, xs, dys, slabs);
}
