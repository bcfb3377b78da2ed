// The BatchNorm finalize step (slab of per-tile partial sums -> scale / shift / saved mean / invstd / running statistics) as a device
// function, used by bn_finalize_kernel (bn.hip).  (Round 3 also ran it inside the depthwise kernel that consumes the coefficients, with a
// leader / waiter hand-over at agent scope; on eight XCDs that hand-over cost ~12 us per layer against 6.5 us for the launch it replaced
// -- profiles/r03_bnfin_bench.txt, DESIGN section 5 -- and was removed in round 4.)
#pragma once
#include "common.h"

namespace dc {

// A SUM ROW (rows == SUM_ROW = -1): instead of one fp32 row per producing workgroup the slab is ONE row of fp64 sums, double[2][C], that the
// caller zeroes and the producing launch adds to with global fp64 atomics (igemm224.hip; dwpipe.hip's BatchNorm-backward sums).  The
// addends are the same per-workgroup fp32 sums a row slab would hold; an fp64 sum of a few hundred fp32 numbers of one channel is exact
// (53 bits against 24 + the spread of their exponents), i.e. independent of the order the atomics arrive in, and equal to what the
// finalize kernels compute from the row slab.  The consumer then reads two numbers per channel, so it can run the finalize itself at
// any size (dc_dwconv_fwd_fin / dc_bn_apply_fin / dc_bn_bwd_apply_fin) and the finalize launch leaves the dependent chain.
constexpr int SUM_ROW = -1;
__device__ inline void sum_row_load(const float* slab, int C, int c, double& s, double& q) {
  const double* d = reinterpret_cast<const double*>(slab);
  s = d[c];
  q = d[(size_t)C + c];
}

struct BnFinArgs {
  const float* slab;      // [2][rows][C]: sum, sum of squares; rows == SUM_ROW: double[2][C]
  int rows, C, parts;     // parts > 0: the slab went through slab_fold_kernel (stage-one results in place)
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

inline BnFinArgs bn_fin_args(int C, long count, const float* slab, int rows, int parts, const float* gamma, const float* beta,
                             float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, float eps, float* scale,
                             float* shift, float* save_mean, float* save_invstd) {
  BnFinArgs a;
  a.slab = slab; a.rows = rows; a.C = C; a.parts = parts; a.inv_count = 1.0 / (double)count; a.unbias = (double)count / (double)(count - 1);
  a.gamma = gamma; a.beta = beta; a.running_mean = running_mean; a.running_var = running_var;
  a.nbt = reinterpret_cast<long long*>(num_batches_tracked); a.momentum = momentum; a.eps = eps;
  a.scale = scale; a.shift = shift; a.save_mean = save_mean; a.save_invstd = save_invstd;
  return a;
}

// Column sums of a [rows][C] fp32 slab in fp64: 4 channels x 64 row-lanes per 256-thread block, so that even the
// 3456-row slabs of the 192x288 layers cost ~50 dependent loads per thread instead of thousands.  `ld(which, row)` returns element
// (row, this thread's channel) of slab `which` as a double.
constexpr int FIN_CH = 4, FIN_RL = 64;
// `quad` (a short slab, rows <= FIN_RL: one row per row lane): the lanes are summed as FOUR interleaved sequences p_j = row j + row j+4 + ...,
// then (p0 + p1) + (p2 + p3) -- the order in which the kernels that run the finalize themselves can take the slab with 16-byte loads, four
// row lanes per channel quad (slab_quad_sum2 below): same bits either way.
template <typename Load>
__device__ inline void slab_colsum2(Load ld, int rows, bool ok, double (&red)[2][FIN_RL][FIN_CH], double& a, double& b, bool quad = false) {
  const int cl = threadIdx.x & (FIN_CH - 1), rl = threadIdx.x / FIN_CH;
  // FIN_UR independent partial sums per thread and slab: with two loads in flight the kernel was a chain of ~100 memory latencies on
  // the largest slabs; the partials are combined in a fixed order
  constexpr int FIN_UR = 8;
  double xs[FIN_UR], ys[FIN_UR];
#pragma unroll
  for (int u = 0; u < FIN_UR; ++u) xs[u] = ys[u] = 0.0;
  if (ok) {
    int r = rl;
    for (; r + (FIN_UR - 1) * FIN_RL < rows; r += FIN_UR * FIN_RL) {
      double fx[FIN_UR], fy[FIN_UR];
#pragma unroll
      for (int u = 0; u < FIN_UR; ++u) {
        fx[u] = ld(0, r + u * FIN_RL);
        fy[u] = ld(1, r + u * FIN_RL);
      }
#pragma unroll
      for (int u = 0; u < FIN_UR; ++u) {
        xs[u] += fx[u];
        ys[u] += fy[u];
      }
    }
#pragma unroll
    for (int u = 0; u < FIN_UR - 1; ++u) {          // at most FIN_UR - 1 rows left (static indices: the partials stay in registers)
      const int rr = r + u * FIN_RL;
      if (rr < rows) {
        xs[u] += ld(0, rr);
        ys[u] += ld(1, rr);
      }
    }
  }
  red[0][rl][cl] = ((xs[0] + xs[1]) + (xs[2] + xs[3])) + ((xs[4] + xs[5]) + (xs[6] + xs[7]));
  red[1][rl][cl] = ((ys[0] + ys[1]) + (ys[2] + ys[3])) + ((ys[4] + ys[5]) + (ys[6] + ys[7]));
  __syncthreads();
  a = b = 0.0;
  if (threadIdx.x < FIN_CH) {
    if (quad) {
      double pa[4] = {0.0, 0.0, 0.0, 0.0}, pb[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
      for (int i = 0; i < FIN_RL; i += 4)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          pa[j] += red[0][i + j][threadIdx.x];
          pb[j] += red[1][i + j][threadIdx.x];
        }
      a = (pa[0] + pa[1]) + (pa[2] + pa[3]);
      b = (pb[0] + pb[1]) + (pb[2] + pb[3]);
    } else {
#pragma unroll 8
      for (int i = 0; i < FIN_RL; ++i) {
        a += red[0][i][threadIdx.x];
        b += red[1][i][threadIdx.x];
      }
    }
  }
}
inline __host__ __device__ bool slab_is_short(int rows, int parts) { return parts == 0 && rows <= FIN_RL; }

// Slabs of thousands of rows (one row per 128-pixel tile: 3456 on the 192 x 288 layers, 13 824 on the 384 x 576 ones) leave the finalize
// kernels with C / 4 workgroups of 50 - 200 dependent rows per thread (8 workgroups for the 32-channel stem; 36 - 80 us per launch).
// Those slabs are folded in two stages: stage one (slab_fold_kernel, C / 4 x rows / FIN_CHUNK workgroups) sums FIN_CHUNK rows and leaves
// the fp64 result IN the slab, in the first two rows of its own chunk (low word, high word; rows and channels no other workgroup
// touches), stage two sums those.
constexpr int FIN_CHUNK = 256, FIN_TWO_STAGE_ROWS = 1024;
inline int fin_parts(int rows) {
  if (rows <= FIN_TWO_STAGE_ROWS || rows % FIN_CHUNK == 1) return 0;      // (a last chunk of one row could not hold its two words)
  return (rows + FIN_CHUNK - 1) / FIN_CHUNK;
}
// the two element loaders: a plain slab, and the stage-one results (parts > 0)
struct SlabLoad {
  const float* s0; const float* s1; int C, c, parts;
  __device__ inline double operator()(int which, int row) const {
    const float* s = which ? s1 : s0;
    if (parts == 0) return (double)s[(size_t)row * C + c];
    const size_t at = (size_t)row * FIN_CHUNK * C + c;
    const unsigned lo = __float_as_uint(s[at]), hi = __float_as_uint(s[at + C]);
    return __hiloint2double((int)hi, (int)lo);
  }
};
__device__ inline void slab_fold_block(float* slab, int rows, int C, double (&red)[2][FIN_RL][FIN_CH]) {
  const int c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1));
  const int r0 = blockIdx.y * FIN_CHUNK;
  const int n = rows - r0 < FIN_CHUNK ? rows - r0 : FIN_CHUNK;
  float* s0 = slab + (size_t)r0 * C;
  float* s1 = slab + ((size_t)rows + r0) * C;
  double a, b;
  slab_colsum2(SlabLoad{s0, s1, C, c, 0}, n, c < C, red, a, b);     // ends with a barrier behind every read of the chunk
  if (threadIdx.x < FIN_CH && c < C) {
    s0[c] = __uint_as_float((unsigned)__double2loint(a)); s0[C + c] = __uint_as_float((unsigned)__double2hiint(a));
    s1[c] = __uint_as_float((unsigned)__double2loint(b)); s1[C + c] = __uint_as_float((unsigned)__double2hiint(b));
  }
}

// From the sums of one channel to its coefficients (and, with `store`, to the stored vectors and the running statistics).
__device__ inline void bn_fin_coefs(const BnFinArgs& a, int c, double s, double q, bool store, float& sc, float& sh) {
  const double mean = s * a.inv_count;
  double var = q * a.inv_count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)a.eps));
  const float g = a.gamma[c], b = a.beta[c];
  sc = g * invstd;
  sh = b - (float)mean * sc;
  if (!store) return;
  a.scale[c] = sc;
  a.shift[c] = sh;
  if (a.save_mean) a.save_mean[c] = (float)mean;
  if (a.save_invstd) a.save_invstd[c] = invstd;
  if (a.running_mean) a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * (float)mean;
  if (a.running_var) a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * (float)(var * a.unbias);
}

// One 4-channel block of dc_bn_finalize (256 threads; red: 4 KiB of LDS).
__device__ inline void bn_finalize_block(const BnFinArgs& a, int cblock, double (&red)[2][FIN_RL][FIN_CH]) {
  const int C = a.C;
  const int c = cblock * FIN_CH + (threadIdx.x & (FIN_CH - 1));
  double s = 0.0, q = 0.0;
  if (a.rows == SUM_ROW) {
    if (c < C) sum_row_load(a.slab, C, c, s, q);
  } else {
    slab_colsum2(SlabLoad{a.slab, a.slab + (size_t)a.rows * C, C, c, a.parts}, a.parts ? a.parts : a.rows, c < C, red, s, q, slab_is_short(a.rows, a.parts));
  }
  if (cblock == 0 && threadIdx.x == 0 && a.nbt != nullptr) *a.nbt += 1;
  if (threadIdx.x >= FIN_CH || c >= C) return;
  float sc, sh;
  bn_fin_coefs(a, c, s, q, true, sc, sh);
}

// dc_dwconv_fwd_fin / dc_bn_apply_fin / dc_bn_bwd_apply_fin: the finalize of a SHORT slab (at most FIN_RL rows, not folded) done by the kernel that
// consumes the coefficients, every workgroup for the CW channels it works on, in slab_colsum2's quad order -- the same bits as dc_bn_finalize /
// dc_bn_bwd_finalize, one launch and one dependent kernel boundary less per BatchNorm.
//
// slab_quad_sum2 (all 256 threads; CW <= 256): a thread takes one of the four row sequences of a channel QUAD with 16-byte loads -- a 54-row slab
// is 28 loads per thread, requested in two batches, where one thread per channel needed 108 --, the partial sums meet in LDS (red: 64 x CW
// bytes) and thread i < CW returns the two sums of channel cbase + i.  The slab was written by the kernel in front of this one, on other XCDs:
// every load is a trip beyond this XCD's L2, and every workgroup of the launch asks for the same lines at the same moment.
template <int CW>
__device__ inline void slab_quad_sum2(const float* __restrict__ slab, int rows, int C, int cbase, double (&red)[2][4][CW], double& s, double& q) {
  static_assert(CW % 4 == 0 && CW <= 256, "one channel quad per thread and row sequence");
  if (rows == SUM_ROW) {      // (uniform) the sums are there already
    s = q = 0.0;
    if (threadIdx.x < CW && cbase + (int)threadIdx.x < C) sum_row_load(slab, C, cbase + threadIdx.x, s, q);
    return;
  }
  constexpr int NQ = CW / 4;
  const int qd = threadIdx.x % NQ, j = threadIdx.x / NQ;
  if (j < 4) {
    double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
    const int c4 = cbase + 4 * qd;
    if (c4 < C) {
      const float* s0 = slab + c4;
      const float* s1 = slab + (size_t)rows * C + c4;
      constexpr int FB = 8;
      for (int r0 = j; r0 < rows; r0 += 4 * FB) {
        float4 x[FB], y[FB];
#pragma unroll
        for (int i = 0; i < FB; ++i) {
          const int r = r0 + 4 * i < rows ? r0 + 4 * i : r0;       // (a row past the end re-reads row r0 and adds +0.0)
          x[i] = *reinterpret_cast<const float4*>(s0 + (size_t)r * C);
          y[i] = *reinterpret_cast<const float4*>(s1 + (size_t)r * C);
        }
#pragma unroll
        for (int i = 0; i < FB; ++i) {
          const bool ok = r0 + 4 * i < rows;
          a[0] += (double)(ok ? x[i].x : 0.f); a[1] += (double)(ok ? x[i].y : 0.f); a[2] += (double)(ok ? x[i].z : 0.f); a[3] += (double)(ok ? x[i].w : 0.f);
          b[0] += (double)(ok ? y[i].x : 0.f); b[1] += (double)(ok ? y[i].y : 0.f); b[2] += (double)(ok ? y[i].z : 0.f); b[3] += (double)(ok ? y[i].w : 0.f);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[0][j][4 * qd + e] = a[e];
      red[1][j][4 * qd + e] = b[e];
    }
  }
  __syncthreads();
  s = q = 0.0;
  if (threadIdx.x < CW) {
    const int i = threadIdx.x;
    s = (red[0][0][i] + red[0][1][i]) + (red[0][2][i] + red[0][3][i]);
    q = (red[1][0][i] + red[1][1][i]) + (red[1][2][i] + red[1][3][i]);
  }
}
// The same sums by ONE thread for channel c (blocks of more than 256 channels): the four row sequences in turn.
__device__ inline void slab_quad_sum2_thread(const float* __restrict__ slab, int rows, int C, int c, double& s, double& q) {
  if (rows == SUM_ROW) {
    sum_row_load(slab, C, c, s, q);
    return;
  }
  const float* s0 = slab + c;
  const float* s1 = slab + (size_t)rows * C + c;
  double pa[4] = {0.0, 0.0, 0.0, 0.0}, pb[4] = {0.0, 0.0, 0.0, 0.0};
  for (int r0 = 0; r0 < rows; r0 += 16) {
    float x[16], y[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = r0 + i < rows ? r0 + i : rows - 1;
      x[i] = s0[(size_t)r * C];
      y[i] = s1[(size_t)r * C];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      pa[i & 3] += (double)(r0 + i < rows ? x[i] : 0.f);
      pb[i & 3] += (double)(r0 + i < rows ? y[i] : 0.f);
    }
  }
  s = (pa[0] + pa[1]) + (pa[2] + pa[3]);
  q = (pb[0] + pb[1]) + (pb[2] + pb[3]);
}

}  // namespace dc
