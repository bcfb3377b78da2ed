// dc_fold_slabs: the fixed-order sum of split-K / per-tile partial slabs of MANY layers in one launch.
//
// Every weight-gradient kernel of the step leaves fp32 partial sums (dense convs: [splits][taps][Co][Ci] from wgrad*.hip / thinconv.hip;
// depthwise convs: one row of [9][C] per pixel tile from the fused data-gradient kernel, dwtile.hip).  Folded per layer they were 146
// launches of 4 - 50 MB each on the weight-gradient stream, 1.8 TB/s on average (launch- and latency-bound) and one stream fence each.
// Here a launch takes up to FOLD_MAX layers: the entry table travels in the kernel-argument segment (scalar loads, no device table to
// keep), a block finds its layer from the running block counts and then does exactly what the per-layer kernels did, in the same
// order -- results are bit-identical to wgrad_reduce_kernel / dwt_reduce_kernel and independent of how layers are grouped into launches.
// Replaces: the accumulation autograd performs inside conv_backward_weight (train_hdf5_ddp.py:363).
#include "common.h"

namespace dc {

namespace {

constexpr int FOLD_MAX = 24;

struct FoldArgs {
  dc_fold_entry e[FOLD_MAX];
  int first[FOLD_MAX + 1];   // first block of entry i; first[n] = grid size
  int n;
};

// dense: one thread = 4 consecutive ci of one (tap, co), float4 loads, eight slabs in flight (wgrad_reduce_kernel's order)
__device__ inline void fold_dense(const dc_fold_entry& en, long idx) {
  const int taps = en.taps, Co = en.co, Ci = en.ci, splits = en.splits;
  const long per = (long)Co * Ci;
  const long per4 = per >> 2;
  if (idx >= per4 * taps) return;
  const int t = (int)(idx / per4);
  const long i = (idx % per4) << 2;
  const float4* src = reinterpret_cast<const float4*>(en.slab + (size_t)t * per + i);
  const size_t stride4 = ((size_t)taps * per) >> 2;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
  float4 b0 = a0, b1 = a0, b2 = a0, b3 = a0;
  int s = 0;
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
  const bool transposed = en.kind == DC_FOLD_CONVT;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const long base = transposed ? ((long)(ci + e) * Co + co) * taps : ((long)co * Ci + ci + e) * taps;
    en.grad[base + t] = r[e];
  }
}

// depthwise: grad[c][t] = sum over rows of slab[row][t][c]; 16 columns x 16 row-lanes per block, fp64, fixed order (dwt_reduce_kernel's)
__device__ inline void fold_dw(const dc_fold_entry& en, int blk, double (*red)[16]) {
  const int C = en.co, rows = en.splits;
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int n = 9 * C;
  const int i = blk * 16 + cl;
  double a0 = 0.0, a1 = 0.0;
  if (i < n) {
    int r = rl;
    for (; r + 16 < rows; r += 32) {
      a0 += (double)en.slab[(size_t)r * n + i];
      a1 += (double)en.slab[(size_t)(r + 16) * n + i];
    }
    if (r < rows) a0 += (double)en.slab[(size_t)r * n + i];
  }
  red[rl][cl] = a0 + a1;
  __syncthreads();
  if (threadIdx.x < 16 && blk * 16 + threadIdx.x < n) {
    const int j = blk * 16 + threadIdx.x;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k][threadIdx.x];
    const int tp = j / C, c = j % C;
    en.grad[(size_t)c * 9 + tp] = (float)s;
  }
}

__global__ __launch_bounds__(256) void fold_kernel(const FoldArgs a_) {
  __shared__ double red[16][16];
  // the table is read through the kernel-argument segment with a run-time index (scalar loads); indexing the by-value struct would
  // make the compiler copy it to scratch memory
  typedef const __attribute__((address_space(4))) FoldArgs* KArgs;
  KArgs a = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(a));
  const int b = blockIdx.x;
  int e = 0;
  const int n = a->n;
  while (e + 1 < n && b >= a->first[e + 1]) ++e;
  dc_fold_entry en;
  en.slab = a->e[e].slab; en.grad = a->e[e].grad; en.kind = a->e[e].kind; en.splits = a->e[e].splits;
  en.taps = a->e[e].taps; en.co = a->e[e].co; en.ci = a->e[e].ci;
  const int lb = b - a->first[e];
  if (en.kind == DC_FOLD_DW) fold_dw(en, lb, red);
  else fold_dense(en, (long)lb * 256 + threadIdx.x);
  (void)a_;
}

int fold_blocks(const dc_fold_entry& en) {
  if (en.kind == DC_FOLD_DW) return cdiv(9L * en.co, 16);
  return cdiv((((long)en.co * en.ci) >> 2) * en.taps, 256);
}

}  // namespace

}  // namespace dc

extern "C" int dc_fold_slabs(const dc_fold_entry* entries, int n, void* stream) {
  using namespace dc;
  DC_REQUIRE(entries != nullptr && n >= 0, "dc_fold_slabs: bad argument");
  for (int i = 0; i < n; ++i) {
    const dc_fold_entry& en = entries[i];
    DC_REQUIRE(en.slab != nullptr && en.grad != nullptr && en.splits >= 1 && en.co >= 1, "dc_fold_slabs: bad entry");
    if (en.kind == DC_FOLD_DW) DC_REQUIRE(en.taps == 9 && en.ci == 1, "dc_fold_slabs: a depthwise entry has taps = 9, ci = 1");
    else {
      DC_REQUIRE(en.kind == DC_FOLD_CONV || en.kind == DC_FOLD_CONVT, "dc_fold_slabs: unknown entry kind");
      DC_REQUIRE(en.taps >= 1 && en.ci >= 4 && en.ci % 4 == 0 && ((uintptr_t)en.slab & 15) == 0, "dc_fold_slabs: a dense entry needs ci % 4 == 0 and a 16-byte aligned slab");
    }
  }
  hipStream_t st = (hipStream_t)stream;
  for (int i0 = 0; i0 < n; i0 += FOLD_MAX) {
    FoldArgs a;
    a.n = n - i0 < FOLD_MAX ? n - i0 : FOLD_MAX;
    long blocks = 0;
    for (int i = 0; i < a.n; ++i) {
      a.e[i] = entries[i0 + i];
      a.first[i] = (int)blocks;
      blocks += fold_blocks(a.e[i]);
    }
    for (int i = a.n; i < FOLD_MAX; ++i) {
      a.e[i] = a.e[0];
      a.first[i] = (int)blocks;
    }
    a.first[FOLD_MAX] = (int)blocks;
    DC_REQUIRE(blocks < (1L << 31), "dc_fold_slabs: too many blocks");
    if (blocks == 0) continue;
    hipLaunchKernelGGL(fold_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
    DC_CHECK_LAUNCH();
  }
  return 0;
}
