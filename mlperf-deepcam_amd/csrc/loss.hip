// Fused weighted cross-entropy forward + backward, first-max argmax and IoU confusion counts in ONE pass over the
// NCHW fp32 logits (HBM-bound: 12 B/pixel read, 12 B/pixel gradient write, 8 B/pixel labels).
//   reference: utils/losses.py:28-52 (plain mean of w[y]*CE; the "false positive" factors are identities),
//              torch.max(outputs,1)[1] (train_hdf5_ddp.py:406,458), utils/utils.py:32-60.
#include "common.h"

namespace dc {

constexpr int NC = 3;

// labels outside [0, NC) come back as -1 (the full 64-bit value is compared, not its low half)
__device__ inline int load_label(const void* labels, int bytes, long i) {
  long long v;
  if (bytes == 8) v = reinterpret_cast<const int64_t*>(labels)[i];
  else if (bytes == 4) v = reinterpret_cast<const int32_t*>(labels)[i];
  else v = reinterpret_cast<const uint8_t*>(labels)[i];
  return (v >= 0 && v < NC) ? (int)v : -1;
}

__device__ inline unsigned long long wave_count(bool pred) { return __popcll(__ballot(pred)); }

__global__ __launch_bounds__(256) void wce_kernel(int B, long HW, const float* __restrict__ logits,
                                                  const void* __restrict__ labels, int lbytes,
                                                  const float* __restrict__ cw, float grad_scale, double* loss_sum,
                                                  float* __restrict__ dlogits, int64_t* __restrict__ pred,
                                                  unsigned long long* counts) {
  __shared__ double s_loss[4];
  __shared__ unsigned long long s_cnt[4][9];
  const long total = (long)B * HW;
  const float w0 = cw[0], w1 = cw[1], w2 = cw[2];
  double lsum = 0.0;
  unsigned long long cnt[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) cnt[k] = 0ull;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // the trip count is made wave-uniform so that the ballots below see all 64 lanes
  const long stride = (long)gridDim.x * blockDim.x;
  const long first = blockIdx.x * (long)blockDim.x + threadIdx.x;
  const long iters = (total + stride - 1) / stride;
  for (long it = 0; it < iters; ++it) {
    const long i = first + it * stride;
    const bool ok = i < total;
    int y = 0, am = 0;
    bool valid = false;
    if (ok) {
      const long b = i / HW, p = i - b * HW;
      const float* lp = logits + (size_t)b * NC * HW + p;
      const float l0 = lp[0], l1 = lp[HW], l2 = lp[2 * HW];
      y = load_label(labels, lbytes, i);
      valid = (unsigned)y < (unsigned)NC;
      float best = l0;
      if (l1 > best) { best = l1; am = 1; }
      if (l2 > best) { best = l2; am = 2; }
      const float e0 = expf(l0 - best), e1 = expf(l1 - best), e2 = expf(l2 - best);
      const float se = e0 + e1 + e2;
      const float lse = best + logf(se);
      if (valid) {
        const float wy = y == 0 ? w0 : (y == 1 ? w1 : w2);
        const float ly = y == 0 ? l0 : (y == 1 ? l1 : l2);
        lsum += (double)(wy * (lse - ly));
        if (dlogits != nullptr) {
          const float inv = 1.0f / se;
          const float s = wy * grad_scale;
          float* gp = dlogits + (size_t)b * NC * HW + p;
          gp[0] = s * (e0 * inv - (y == 0 ? 1.f : 0.f));
          gp[HW] = s * (e1 * inv - (y == 1 ? 1.f : 0.f));
          gp[2 * HW] = s * (e2 * inv - (y == 2 ? 1.f : 0.f));
        }
      } else {
        // A label outside [0, 3): nn.CrossEntropyLoss raises for it in the reference (losses.py:36).  A kernel cannot raise and
        // this ABI never synchronises, so the result is poisoned instead of silently dropping the pixel: the loss and this
        // pixel's gradient become NaN, which reaches every weight in the same step.
        lsum += (double)__builtin_nanf("");
        if (dlogits != nullptr) {
          float* gp = dlogits + (size_t)b * NC * HW + p;
          gp[0] = gp[HW] = gp[2 * HW] = __builtin_nanf("");
        }
      }
      if (pred != nullptr) pred[i] = am;
    }
    if (counts != nullptr) {
      const bool eq = ok && (am == y);
      const bool ne = ok && (am != y);
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        cnt[j] += wave_count(eq && y == j);          // tp: agree and gt == j
        cnt[3 + j] += wave_count(ne && am == j);     // fp: disagree and pred == j
        cnt[6 + j] += wave_count(ne && y == j);      // fn: disagree and gt == j
      }
    }
  }
  lsum = wave_sum(lsum);
  if (lane == 0) {
    s_loss[wave] = lsum;
#pragma unroll
    for (int k = 0; k < 9; ++k) s_cnt[wave][k] = cnt[k];  // wave-uniform already
  }
  __syncthreads();
  if (threadIdx.x == 0 && loss_sum != nullptr) atomicAdd(loss_sum, s_loss[0] + s_loss[1] + s_loss[2] + s_loss[3]);
  if (threadIdx.x < 9 && counts != nullptr) {
    const int k = threadIdx.x;
    const unsigned long long c = s_cnt[0][k] + s_cnt[1][k] + s_cnt[2][k] + s_cnt[3][k];
    if (c) atomicAdd(&counts[k], c);
  }
}

__global__ __launch_bounds__(256) void confusion_kernel(long n, const int64_t* __restrict__ pred,
                                                        const void* __restrict__ labels, int lbytes,
                                                        unsigned long long* counts) {
  __shared__ unsigned long long s_cnt[4][9];
  unsigned long long cnt[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) cnt[k] = 0ull;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long stride = (long)gridDim.x * blockDim.x;
  const long first = blockIdx.x * (long)blockDim.x + threadIdx.x;
  const long iters = (n + stride - 1) / stride;
  for (long it = 0; it < iters; ++it) {
    const long i = first + it * stride;
    const bool ok = i < n;
    const int am = ok ? (int)pred[i] : -1;
    const int y = ok ? load_label(labels, lbytes, i) : -2;
    const bool eq = ok && am == y, ne = ok && am != y;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      cnt[j] += wave_count(eq && y == j);
      cnt[3 + j] += wave_count(ne && am == j);
      cnt[6 + j] += wave_count(ne && y == j);
    }
  }
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < 9; ++k) s_cnt[wave][k] = cnt[k];
  __syncthreads();
  if (threadIdx.x < 9) {
    const int k = threadIdx.x;
    const unsigned long long c = s_cnt[0][k] + s_cnt[1][k] + s_cnt[2][k] + s_cnt[3][k];
    if (c) atomicAdd(&counts[k], c);
  }
}

}  // namespace dc

using namespace dc;

extern "C" int dc_wce_fused(int B, int H, int W, const float* logits_nchw, const void* labels, int label_dtype_bytes,
                            const float* class_weights, float grad_scale, double* loss_sum, float* dlogits,
                            int64_t* pred, int64_t* counts, void* stream) {
  DC_REQUIRE(logits_nchw && labels && class_weights && B > 0 && H > 0 && W > 0, "dc_wce_fused: bad argument");
  DC_REQUIRE(label_dtype_bytes == 1 || label_dtype_bytes == 4 || label_dtype_bytes == 8, "dc_wce_fused: labels must be uint8, int32 or int64");
  const long total = (long)B * H * W;
  long blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(wce_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, B, (long)H * W, logits_nchw, labels,
                     label_dtype_bytes, class_weights, grad_scale, loss_sum, dlogits, pred, (unsigned long long*)counts);
  DC_CHECK_LAUNCH();
  return 0;
}

extern "C" int dc_confusion_counts(long n, const int64_t* pred, const void* labels, int label_dtype_bytes,
                                   int64_t* counts, void* stream) {
  DC_REQUIRE(pred && labels && counts && n > 0, "dc_confusion_counts: bad argument");
  DC_REQUIRE(label_dtype_bytes == 1 || label_dtype_bytes == 4 || label_dtype_bytes == 8, "dc_confusion_counts: labels must be uint8, int32 or int64");
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(confusion_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, n, pred, labels,
                     label_dtype_bytes, (unsigned long long*)counts);
  DC_CHECK_LAUNCH();
  return 0;
}
