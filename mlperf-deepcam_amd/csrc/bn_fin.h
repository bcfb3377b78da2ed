// The BatchNorm finalize step (slab of per-tile partial sums -> scale / shift / saved mean / invstd / running statistics) as a device
// function, so that it can run inside the kernel that consumes the coefficients instead of as a launch of its own.
//
// Why: at local batch 8 the 77 dc_bn_finalize launches of a forward pass cost the chain 1.5 ms (DC_DEBUG_SKIP_BN_FINALIZE=async) --
// 4-11 us of kernel each plus two dispatch boundaries of 6-7 us on a chip whose eight L2s are made coherent at every boundary.
// Hand-over inside the consumer (dwtile.hip): the first workgroups of the grid ("leaders", lowest block ids, dispatched first) each run
// bn_finalize_block for a few 4-channel blocks -- the same code and the same summation order as the stand-alone kernel, hence the same
// bits -- and publish them with a release at agent scope; every workgroup waits (bounded) on the per-channel-block counter before it
// reads scale / shift.  A wait that runs out is not an error: the workgroup then computes the coefficients it needs itself, serially in
// the same order (same bits again), so a dispatch order that starved the leaders would cost time, never correctness or a hang.
#pragma once
#include "common.h"

namespace dc {

struct BnFinArgs {
  const float* slab;      // [2][rows][C]: sum, sum of squares
  int rows, C;
  double inv_count, unbias;
  const float* gamma;
  const float* beta;
  float* running_mean;    // may be null
  float* running_var;
  long long* nbt;         // num_batches_tracked, may be null
  float momentum, eps;
  float* scale;
  float* shift;
  float* save_mean;       // may be null
  float* save_invstd;
  // in-consumer form only: sync[channel block of the consumer's grid] counts the 4-channel blocks finalized so far over all launches;
  // launch number `epoch` (0, 1, ...) of this layer is complete for a channel block of nblk blocks at (epoch + 1) * nblk
  unsigned* sync;
  unsigned epoch;
};

// Column sums of a [rows][C] fp32 slab in fp64: 4 channels x 64 row-lanes per 256-thread block, so that even the
// 3456-row slabs of the 384x576 layers cost ~50 dependent loads per thread instead of thousands.
constexpr int FIN_CH = 4, FIN_RL = 64;
__device__ inline void slab_colsum2(const float* __restrict__ s0, const float* __restrict__ s1, int rows, int C, int c,
                                    bool ok, double (&red)[2][FIN_RL][FIN_CH], double& a, double& b) {
  const int cl = threadIdx.x & (FIN_CH - 1), rl = threadIdx.x / FIN_CH;
  // FIN_UR independent partial sums per thread and slab: the 13 824-row slabs of the 384 x 576 layers are 216 rows per thread, and
  // with two loads in flight the kernel was a chain of ~100 memory latencies (63-83 us); the partials are combined in a fixed order
  constexpr int FIN_UR = 8;
  double xs[FIN_UR], ys[FIN_UR];
#pragma unroll
  for (int u = 0; u < FIN_UR; ++u) xs[u] = ys[u] = 0.0;
  if (ok) {
    int r = rl;
    for (; r + (FIN_UR - 1) * FIN_RL < rows; r += FIN_UR * FIN_RL) {
      float fx[FIN_UR], fy[FIN_UR];
#pragma unroll
      for (int u = 0; u < FIN_UR; ++u) {
        fx[u] = s0[(size_t)(r + u * FIN_RL) * C + c];
        fy[u] = s1[(size_t)(r + u * FIN_RL) * C + c];
      }
#pragma unroll
      for (int u = 0; u < FIN_UR; ++u) {
        xs[u] += (double)fx[u];
        ys[u] += (double)fy[u];
      }
    }
#pragma unroll
    for (int u = 0; u < FIN_UR - 1; ++u) {          // at most FIN_UR - 1 rows left (static indices: the partials stay in registers)
      const int rr = r + u * FIN_RL;
      if (rr < rows) {
        xs[u] += (double)s0[(size_t)rr * C + c];
        ys[u] += (double)s1[(size_t)rr * C + c];
      }
    }
  }
  red[0][rl][cl] = ((xs[0] + xs[1]) + (xs[2] + xs[3])) + ((xs[4] + xs[5]) + (xs[6] + xs[7]));
  red[1][rl][cl] = ((ys[0] + ys[1]) + (ys[2] + ys[3])) + ((ys[4] + ys[5]) + (ys[6] + ys[7]));
  __syncthreads();
  a = b = 0.0;
  if (threadIdx.x < FIN_CH) {
#pragma unroll 8
    for (int i = 0; i < FIN_RL; ++i) {
      a += red[0][i][threadIdx.x];
      b += red[1][i][threadIdx.x];
    }
  }
}


// One 4-channel block of dc_bn_finalize (256 threads; red: 4 KiB of LDS).  COHERENT: scale / shift are stored write-through at agent
// scope (the in-consumer form: workgroups on other XCDs read them in the same kernel; see bn_fin_lead).
template <bool COHERENT = false>
__device__ inline void bn_finalize_block(const BnFinArgs& a, int cblock, double (&red)[2][FIN_RL][FIN_CH]) {
  const int C = a.C;
  const int c = cblock * FIN_CH + (threadIdx.x & (FIN_CH - 1));
  double s, q;
  slab_colsum2(a.slab, a.slab + (size_t)a.rows * C, a.rows, C, c, c < C, red, s, q);
  if (cblock == 0 && threadIdx.x == 0 && a.nbt != nullptr) *a.nbt += 1;
  if (threadIdx.x >= FIN_CH || c >= C) return;
  const double mean = s * a.inv_count;
  double var = q * a.inv_count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)a.eps));
  const float g = a.gamma[c], b = a.beta[c];
  const float sc = g * invstd;
  if constexpr (COHERENT) {
    __hip_atomic_store(a.scale + c, sc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(a.shift + c, b - (float)mean * sc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    a.scale[c] = sc;
    a.shift[c] = b - (float)mean * sc;
  }
  if (a.save_mean) a.save_mean[c] = (float)mean;
  if (a.save_invstd) a.save_invstd[c] = invstd;
  if (a.running_mean) a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * (float)mean;
  if (a.running_var) a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * (float)(var * a.unbias);
}

// One channel's scale / shift by ONE thread, in slab_colsum2's summation order (row lane by row lane, eight partials each, the same
// pairwise combination, then the 64 row lanes in sequence): the bits of bn_finalize_block.  The fall-back of a consumer workgroup whose
// wait ran out; it does not touch the running statistics (the leaders still will).
__device__ inline void bn_coeffs_serial(const BnFinArgs& a, int c, float& scale, float& shift) {
  constexpr int FIN_UR = 8;
  const float* __restrict__ s0 = a.slab;
  const float* __restrict__ s1 = a.slab + (size_t)a.rows * a.C;
  const int rows = a.rows, C = a.C;
  double s = 0.0, q = 0.0;
  for (int rl = 0; rl < FIN_RL; ++rl) {
    double xs[FIN_UR], ys[FIN_UR];
#pragma unroll
    for (int u = 0; u < FIN_UR; ++u) xs[u] = ys[u] = 0.0;
    int r = rl;
    for (; r + (FIN_UR - 1) * FIN_RL < rows; r += FIN_UR * FIN_RL) {
#pragma unroll
      for (int u = 0; u < FIN_UR; ++u) {
        xs[u] += (double)s0[(size_t)(r + u * FIN_RL) * C + c];
        ys[u] += (double)s1[(size_t)(r + u * FIN_RL) * C + c];
      }
    }
#pragma unroll
    for (int u = 0; u < FIN_UR - 1; ++u) {
      const int rr = r + u * FIN_RL;
      if (rr < rows) {
        xs[u] += (double)s0[(size_t)rr * C + c];
        ys[u] += (double)s1[(size_t)rr * C + c];
      }
    }
    s += ((xs[0] + xs[1]) + (xs[2] + xs[3])) + ((xs[4] + xs[5]) + (xs[6] + xs[7]));
    q += ((ys[0] + ys[1]) + (ys[2] + ys[3])) + ((ys[4] + ys[5]) + (ys[6] + ys[7]));
  }
  const double mean = s * a.inv_count;
  double var = q * a.inv_count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)a.eps));
  scale = a.gamma[c] * invstd;
  shift = a.beta[c] - (float)mean * scale;
}

// Consumer side of the hand-over.  The consumer's grid is cut into `ncb` channel blocks of `cw` channels (cw a multiple of 4); a
// workgroup needs the coefficients of its own channel block.  The finalize jobs (4-channel blocks, all channel blocks interleaved) go to
// the FIRST workgroups of the grid in dispatch order -- `rank` = blockIdx.x < nlead -- whatever tile those workgroups compute themselves:
// a leader that the dispatcher reaches late would keep everybody waiting.  `red`: 4 KiB of LDS nobody else uses meanwhile.
// Call bn_fin_lead first (rank < nlead only), start whatever loads can run meanwhile, then bn_fin_wait (everybody).
constexpr int BN_FIN_SPINS = 4096;
__device__ inline int bn_fin_leaders(int ncb, int cw, int grid) {
  const int jobs = ncb * (cw / FIN_CH);
  return jobs < grid ? jobs : grid;
}
__device__ inline void bn_fin_lead(const BnFinArgs& a, int ncb, int cw, int rank, int nlead, double (&red)[2][FIN_RL][FIN_CH]) {
  const int jobs = ncb * (cw / FIN_CH);
  for (int job = rank; job < jobs; job += nlead) {
    const int cb = job % ncb, b = job / ncb;
    const int c0 = cb * cw + b * FIN_CH;
    if (c0 >= a.C) continue;      // the last channel block is ragged
    // No fence instructions here: an agent-scope release writes back, and an acquire invalidates, a whole L2 -- issued by every workgroup
    // of the consumer's grid that cost 86 us per layer.  Instead the coefficients themselves travel write-through (agent-scope
    // stores), the count is bumped once they have been acknowledged (vmcnt), and the readers fetch them with agent-scope loads.
    bn_finalize_block<true>(a, c0 / FIN_CH, red);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();            // every thread's stores acknowledged (and red is free again)
    if (threadIdx.x == 0) __hip_atomic_fetch_add(a.sync + cb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// (sync[ncb] counts the waits that ran out: zero in a healthy run, tests look at it)
__device__ inline void bn_fin_wait(const BnFinArgs& a, int ncb, int chan_block, int c_lo, int c_hi, int force_fallback) {
  __shared__ int fin_ok;
  const int nblk = (c_hi - c_lo + FIN_CH - 1) / FIN_CH;
  const unsigned target = (a.epoch + 1u) * (unsigned)nblk;
  if (threadIdx.x == 0) {
    int it = 0;
    bool ok = false;
    if (!force_fallback) {
      // One look at once (a workgroup of a later round finds the count complete), then a pause of about the leaders' 4 - 5 us, then a
      // look per microsecond: 768 resident workgroups polling three words without a pause queue up in front of the leaders' own
      // stores and counts at that memory channel.
      for (;;) {
        const unsigned v = __hip_atomic_load(a.sync + chan_block, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = (int)(v - target) >= 0;
        if (ok || ++it >= BN_FIN_SPINS) break;
        if (it == 1) {
          __builtin_amdgcn_s_sleep(127);
          __builtin_amdgcn_s_sleep(32);
        } else {
          __builtin_amdgcn_s_sleep(32);
        }
      }
    }
    fin_ok = ok ? 1 : 0;
    if (!ok && !force_fallback) __hip_atomic_fetch_add(a.sync + ncb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (fin_ok) return;      // (the caller reads scale / shift with agent-scope loads: bn_coef_load)
  // the wait ran out (or the test switch asks for this path): the same coefficients, computed here
  for (int c = c_lo + (int)threadIdx.x; c < c_hi; c += (int)blockDim.x) {
    float sc, sh;
    bn_coeffs_serial(a, c, sc, sh);
    __hip_atomic_store(a.scale + c, sc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(a.shift + c, sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// scale / shift as the consumer reads them after bn_fin_wait: at agent scope, i.e. from the coherence point rather than from a line this
// XCD's L2 may still hold
__device__ inline float bn_coef_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

}  // namespace dc
