// The BatchNorm finalize step (slab of per-tile partial sums -> scale / shift / saved mean / invstd / running statistics) as a device
// function, used by bn_finalize_kernel (bn.hip).  (Round 3 also ran it inside the depthwise kernel that consumes the coefficients, with a
// leader / waiter hand-over at agent scope; on eight XCDs that hand-over cost ~12 us per layer against 6.5 us for the launch it replaced
// -- profiles/r03_bnfin_bench.txt, DESIGN section 5 -- and was removed in round 4.)
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


// One 4-channel block of dc_bn_finalize (256 threads; red: 4 KiB of LDS).
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
  a.scale[c] = sc;
  a.shift[c] = b - (float)mean * sc;
  if (a.save_mean) a.save_mean[c] = (float)mean;
  if (a.save_invstd) a.save_invstd[c] = invstd;
  if (a.running_mean) a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * (float)mean;
  if (a.running_var) a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * (float)(var * a.unbias);
}

}  // namespace dc
