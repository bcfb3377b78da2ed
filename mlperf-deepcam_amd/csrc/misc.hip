// Image-pool branch helpers and the optimizers: all HBM-bound streaming kernels.
#include "common.h"

namespace dc {

// out[n][c] = scale * sum_p x[n,p,c]     (AdaptiveAvgPool2d((1,1)) with scale = 1/HW; its transpose with scale = 1)
// The per-sample vectors of the image-pool branch are ALWAYS fp32: its BatchNorm sees only B values per channel and a
// bf16 rounding of two nearly equal values can flip the sign of the normalised output (SURVEY hard part 6).
template <typename T>
__global__ __launch_bounds__(256) void hw_reduce_kernel(int HW, int C, const T* __restrict__ x, int ldx, float scale,
                                                        float* __restrict__ out) {
  constexpr int KPV = Elem<T>::kPerVec;
  __shared__ float red[16][16 * KPV];
  const int cgl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c0 = (blockIdx.x * 16 + cgl) * KPV;
  const int n = blockIdx.y;
  float a[KPV];
#pragma unroll
  for (int e = 0; e < KPV; ++e) a[e] = 0.f;
  // A thread walks HW/16 rows.  One 16-byte load in flight per thread made this a chain of ~216 memory latencies (86-147 us for
  // the image-pool branch's 48 x 72 maps, whatever the batch); UR loads are issued together and added in the same row order, so
  // the sums keep their bits.
  constexpr int UR = 8;
  if (c0 < C) {
    const T* base = x + (size_t)n * HW * ldx + c0;
    int p = rl;
    for (; p + 16 * (UR - 1) < HW; p += 16 * UR) {
      vec16 v[UR];
#pragma unroll
      for (int u = 0; u < UR; ++u) v[u] = ldg16(base + (size_t)(p + 16 * u) * ldx);
#pragma unroll
      for (int u = 0; u < UR; ++u) {
        float f[KPV];
        unpack(v[u], f, T());
#pragma unroll
        for (int e = 0; e < KPV; ++e) a[e] += f[e];
      }
    }
    for (; p < HW; p += 16) {
      float f[KPV];
      unpack(ldg16(base + (size_t)p * ldx), f, T());
#pragma unroll
      for (int e = 0; e < KPV; ++e) a[e] += f[e];
    }
  }
#pragma unroll
  for (int e = 0; e < KPV; ++e) red[rl][cgl * KPV + e] = a[e];
  __syncthreads();
  if (threadIdx.x < 16 * KPV) {
    const int c = blockIdx.x * 16 * KPV + threadIdx.x;
    if (c < C) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) s += red[r][threadIdx.x];
      out[(size_t)n * C + c] = s * scale;
    }
  }
}

// MODE 0: out[n,p,c]  = v[n,c]           (broadcast)
// MODE 1: out[n,p,c] += v[n,c] * scale   (avg-pool backward, accumulating)
// MODE 2: out[r,c]    = src[r,c]         (view copy; v is the source, HW rows per "n")
template <typename T, int MODE>
__global__ __launch_bounds__(256) void hw_ew_kernel(long rows, int HW, int C, const void* __restrict__ vsrc, int ldv,
                                                    float scale, T* __restrict__ out, int ldo) {
  constexpr int KPV = Elem<T>::kPerVec;
  const int ngroups = C / KPV;
  const long total = rows * ngroups;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(i % ngroups);
    const long r = i / ngroups;
    const int c0 = cg * KPV;
    if (MODE == 2) {
      stg16(out + (size_t)r * ldo + c0, ldg16(reinterpret_cast<const T*>(vsrc) + (size_t)r * ldv + c0));
    } else {
      const long n = r / HW;
      const float* v = reinterpret_cast<const float*>(vsrc) + (size_t)n * C + c0;   // fp32 per-sample vector
      float a[KPV];
#pragma unroll
      for (int e = 0; e < KPV; ++e) a[e] = v[e];
      if (MODE == 1) {
        float b[KPV];
        unpack(ldg16(out + (size_t)r * ldo + c0), b, T());
#pragma unroll
        for (int e = 0; e < KPV; ++e) a[e] = fmaf(a[e], scale, b[e]);
      }
      vec16 res;
      pack(res, a, T());
      stg16(out + (size_t)r * ldo + c0, res);
    }
  }
}

// ------------------------------------------------------------------------------------------------ optimizers
// torch.optim.Adam / AdamW single-tensor algorithm over the flat arena.
__global__ __launch_bounds__(256) void adam_kernel(int kind, long n, float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, const float* lr_dev,
                                                   float beta1, float beta2, float eps, float wd, const int* step_dev,
                                                   float grad_scale) {
  const float lr = *lr_dev;
  const int step = *step_dev;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  const long n4 = n >> 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float* pa = &pp.x;
    const float* ga = &gg.x;
    float* ma = &mm.x;
    float* va = &vv.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float gr = ga[e] * grad_scale;
      if (kind == DC_ADAM) gr = fmaf(wd, pa[e], gr);           // L2 penalty folded into the gradient
      else pa[e] *= (1.f - lr * wd);                            // decoupled decay
      ma[e] = ma[e] + (gr - ma[e]) * (1.f - beta1);             // lerp_
      va[e] = va[e] * beta2 + (1.f - beta2) * gr * gr;
      const float denom = sqrtf(va[e]) * inv_sqrt_bc2 + eps;
      pa[e] = pa[e] - step_size * (ma[e] / denom);
    }
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  // tail (n not a multiple of 4)
  const long tail0 = n4 << 2;
  const long ti = tail0 + blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (ti < n) {
    float gr = g[ti] * grad_scale;
    float pv = p[ti];
    if (kind == DC_ADAM) gr = fmaf(wd, pv, gr);
    else pv *= (1.f - lr * wd);
    const float mv = m[ti] + (gr - m[ti]) * (1.f - beta1);
    const float vv = v[ti] * beta2 + (1.f - beta2) * gr * gr;
    m[ti] = mv;
    v[ti] = vv;
    p[ti] = pv - step_size * (mv / (sqrtf(vv) * inv_sqrt_bc2 + eps));
  }
}

// Sum of squares of the scaled gradient arena, in two deterministic stages: SUMSQ_BLOCKS partial sums here, their fixed-order
// total in every workgroup of lamb_stage1_kernel (the first version added the partials with atomicAdd: the clip factor then
// differed in the last bit from run to run, and this network amplifies a 1e-7 perturbation into 1e-3 of the loss in ten steps).
constexpr int SUMSQ_BLOCKS = 1024;
__global__ __launch_bounds__(256) void sumsq_kernel(long n, const float* __restrict__ g, float scale, float* __restrict__ partial) {
  __shared__ float s[4];
  // 16-byte loads, two in flight per thread (the arena is 16-byte aligned; scalar loads ran at 2.3 TB/s), scalar tail
  float a = 0.f;
  const long n4 = n >> 2;
  const float4* __restrict__ g4 = reinterpret_cast<const float4*>(g);
  const long stride = (long)gridDim.x * blockDim.x;
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  for (; i + stride < n4; i += 2 * stride) {
    const float4 u = g4[i], w = g4[i + stride];
    float x;
    x = u.x * scale; a = fmaf(x, x, a);
    x = u.y * scale; a = fmaf(x, x, a);
    x = u.z * scale; a = fmaf(x, x, a);
    x = u.w * scale; a = fmaf(x, x, a);
    x = w.x * scale; a = fmaf(x, x, a);
    x = w.y * scale; a = fmaf(x, x, a);
    x = w.z * scale; a = fmaf(x, x, a);
    x = w.w * scale; a = fmaf(x, x, a);
  }
  for (; i < n4; i += stride) {
    const float4 u = g4[i];
    float x;
    x = u.x * scale; a = fmaf(x, x, a);
    x = u.y * scale; a = fmaf(x, x, a);
    x = u.z * scale; a = fmaf(x, x, a);
    x = u.w * scale; a = fmaf(x, x, a);
  }
  for (long j = (n4 << 2) + blockIdx.x * (long)blockDim.x + threadIdx.x; j < n; j += stride) {
    const float x = g[j] * scale;
    a = fmaf(x, x, a);
  }
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
}

// fixed-order sum of `count` floats by one 256-thread workgroup (double accumulation); every thread returns the total
__device__ inline double block_sum_fixed(const float* __restrict__ v, int count, int stride, double* sh) {
  double a = 0.0;
  for (int i = threadIdx.x; i < count; i += 256) a += (double)v[(size_t)i * stride];
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
  __syncthreads();
  const double t = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  __syncthreads();
  return t;
}

// LAMB works on fixed-size chunks that never straddle a tensor: a tensor of n elements owns ceil(n / LAMB_CHUNK) consecutive
// workgroups, so the 4.7 M-element ASPP kernels and the 728-element BatchNorm vectors get work in proportion (the first version
// gave every tensor 32 workgroups: the kernel lasted as long as the largest tensor took on 32 of the chip's 256 CUs).
constexpr int LAMB_CHUNK = 4096;

// chunk_prefix[t] = number of chunks of tensors 0..t-1 (ntensors + 1 entries).  One wave: every lane counts the chunks of a
// contiguous run of tensors, the 64 run totals are scanned across the wave, the lanes write their runs.  (One thread walking the
// ~300 tensors took 29 us on the step's critical path, in front of stage 1.)
__global__ __launch_bounds__(64) void lamb_plan_kernel(const int64_t* __restrict__ offs, int ntensors, int* __restrict__ chunk_prefix) {
  if (blockIdx.x != 0) return;
  const int lane = threadIdx.x;
  const int per = (ntensors + 63) / 64;
  const int t0 = lane * per, t1 = min(ntensors, t0 + per);
  int mine = 0;
  for (int t = t0; t < t1; ++t) mine += (int)((offs[t + 1] - offs[t] + LAMB_CHUNK - 1) / LAMB_CHUNK);
  int incl = mine;                       // inclusive scan over the 64 lanes
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(incl, d, 64);
    if (lane >= d) incl += up;
  }
  int acc = incl - mine;
  for (int t = t0; t < t1; ++t) {
    chunk_prefix[t] = acc;
    acc += (int)((offs[t + 1] - offs[t] + LAMB_CHUNK - 1) / LAMB_CHUNK);
  }
  if (lane == 63) chunk_prefix[ntensors] = incl;
}

// workgroup -> (tensor, element range); false when the workgroup is past the last chunk
__device__ inline bool lamb_locate(const int64_t* __restrict__ offs, const int* __restrict__ chunk_prefix, int ntensors, int& t,
                                   long& beg, long& end) {
  const int b = blockIdx.x;
  if (b >= chunk_prefix[ntensors]) return false;
  int lo = 0, hi = ntensors;   // invariant: chunk_prefix[lo] <= b < chunk_prefix[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (chunk_prefix[mid] <= b) lo = mid; else hi = mid;
  }
  t = lo;
  beg = offs[t] + (long)(b - chunk_prefix[t]) * LAMB_CHUNK;
  end = beg + LAMB_CHUNK < offs[t + 1] ? beg + LAMB_CHUNK : offs[t + 1];
  return true;
}

// stage 1: moments, update direction u (its own buffer: the gradient arena is read-only), per-CHUNK ||w||^2 and ||u||^2
// (part[chunk][2], no atomics)
__global__ __launch_bounds__(256) void lamb_stage1_kernel(const int64_t* __restrict__ offs, const int* __restrict__ chunk_prefix,
                                                          int ntensors, const float* __restrict__ p, const float* __restrict__ g,
                                                          float* __restrict__ uo, float* __restrict__ m, float* __restrict__ v, float beta1, float beta2,
                                                          float eps, float wd, const int* step_dev, float max_grad_norm,
                                                          float grad_scale, const float* __restrict__ gpartial, float* __restrict__ part) {
  __shared__ float s[2][4];
  __shared__ double sh[4];
  int t;
  long beg, end;
  if (!lamb_locate(offs, chunk_prefix, ntensors, t, beg, end)) return;
  const int step = *step_dev;
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  const float bc2 = (float)(1.0 - pow((double)beta2, (double)step));
  const float gnorm = (float)sqrt(block_sum_fixed(gpartial, SUMSQ_BLOCKS, 1, sh));
  const float clip = gnorm > max_grad_norm ? gnorm / max_grad_norm : 1.f;
  const float gs = grad_scale / clip;
  float wn = 0.f, un = 0.f;
  for (long i = beg + threadIdx.x; i < end; i += 256) {
    const float gr = g[i] * gs;
    const float pv = p[i];
    const float mv = m[i] + (gr - m[i]) * (1.f - beta1);
    const float vv = v[i] * beta2 + (1.f - beta2) * gr * gr;
    m[i] = mv;
    v[i] = vv;
    const float u = (mv / bc1) / (sqrtf(vv / bc2) + eps) + wd * pv;
    uo[i] = u;
    wn = fmaf(pv, pv, wn);
    un = fmaf(u, u, un);
  }
  wn = wave_sum(wn);
  un = wave_sum(un);
  if ((threadIdx.x & 63) == 0) {
    s[0][threadIdx.x >> 6] = wn;
    s[1][threadIdx.x >> 6] = un;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[2 * (size_t)blockIdx.x] = (s[0][0] + s[0][1]) + (s[0][2] + s[0][3]);
    part[2 * (size_t)blockIdx.x + 1] = (s[1][0] + s[1][1]) + (s[1][2] + s[1][3]);
  }
}

// lr * trust ratio of tensor t = blockIdx.x: the chunk sums of the tensor in a fixed order (at most ~1200 chunks)
__device__ inline float lamb_step_size(const int* __restrict__ chunk_prefix, int t, const float* lr_dev, const float* __restrict__ part, float wd,
                                       double* sh) {
  const int c0 = chunk_prefix[t], nc = chunk_prefix[t + 1] - c0;
  const float wn = (float)sqrt(block_sum_fixed(part + 2 * (size_t)c0, nc, 2, sh));
  const float un = (float)sqrt(block_sum_fixed(part + 2 * (size_t)c0 + 1, nc, 2, sh));
  // apex FusedLAMB (use_nvlamb = False, its default): the trust ratio applies only to tensors with non-zero weight decay
  const float ratio = (wd != 0.f && wn > 0.f && un > 0.f) ? wn / un : 1.f;
  return *lr_dev * ratio;
}

// one workgroup per tensor (stage 2's 14 000 workgroups each repeated the sum of their tensor's chunk norms: up to 1 152 pairs for the ASPP
// kernels, as long as the 16 KiB of parameters a workgroup then updates)
__global__ __launch_bounds__(256) void lamb_ratio_kernel(const int* __restrict__ chunk_prefix, int ntensors, const float* lr_dev,
                                                         const float* __restrict__ part, float wd, float* __restrict__ step_size) {
  __shared__ double sh[4];
  const float a = lamb_step_size(chunk_prefix, blockIdx.x, lr_dev, part, wd, sh);
  if (threadIdx.x == 0) step_size[blockIdx.x] = a;
}

// stage 2: the update of a chunk with its tensor's step size (step_size: lamb_ratio_kernel's; null: summed here, the same bits)
__global__ __launch_bounds__(256) void lamb_stage2_kernel(const int64_t* __restrict__ offs, const int* __restrict__ chunk_prefix,
                                                          int ntensors, float* __restrict__ p, const float* __restrict__ u,
                                                          const float* lr_dev, const float* __restrict__ part, float wd,
                                                          const float* __restrict__ step_size) {
  __shared__ double sh[4];
  int t;
  long beg, end;
  if (!lamb_locate(offs, chunk_prefix, ntensors, t, beg, end)) return;
  const float a = step_size != nullptr ? step_size[t] : lamb_step_size(chunk_prefix, t, lr_dev, part, wd, sh);
  for (long i = beg + threadIdx.x; i < end; i += 256) p[i] = fmaf(-a, u[i], p[i]);
}

static int ew_blocks2(long total) {
  long b = (total + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (int)b;
}

// ---- gradient payload for the all-reduce (dist.GradReducer, payload="bf16": SURVEY 5.8, 112.9 MB per step instead of 225.8) ----
// The fp32 gradient arena is the master copy.  A bucket is rounded to bf16 (round-to-nearest-even) into a send buffer, summed
// across ranks in bf16 by the collective, and widened back into the arena, which then holds the (bf16-rounded) SUM.
__global__ __launch_bounds__(256) void grad_pack_bf16_kernel(long n, const float* __restrict__ g, bf16* __restrict__ out) {
  const long nv = n >> 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
    const vec16 a = ldg16(g + i * 8), b = ldg16(g + i * 8 + 4);
    vec16 o;
    o.w[0] = pack2_bf16(__uint_as_float(a.w[0]), __uint_as_float(a.w[1]));
    o.w[1] = pack2_bf16(__uint_as_float(a.w[2]), __uint_as_float(a.w[3]));
    o.w[2] = pack2_bf16(__uint_as_float(b.w[0]), __uint_as_float(b.w[1]));
    o.w[3] = pack2_bf16(__uint_as_float(b.w[2]), __uint_as_float(b.w[3]));
    stg16(out + i * 8, o);
  }
  if (blockIdx.x == 0) {
    const long t = nv * 8 + threadIdx.x;
    if (t < n) out[t] = (bf16)g[t];
  }
}

__global__ __launch_bounds__(256) void grad_unpack_bf16_kernel(long n, const bf16* __restrict__ in, float* __restrict__ g) {
  const long nv = n >> 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
    float f[8];
    unpack(ldg16(in + i * 8), f, bf16());
    vec16 a, b;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a.w[e] = __float_as_uint(f[e]);
      b.w[e] = __float_as_uint(f[4 + e]);
    }
    stg16(g + i * 8, a);
    stg16(g + i * 8 + 4, b);
  }
  if (blockIdx.x == 0) {
    const long t = nv * 8 + threadIdx.x;
    if (t < n) g[t] = (float)in[t];
  }
}

}  // namespace dc

using namespace dc;

#define DISPATCH_T(dtype, ...)                      \
  do {                                              \
    if ((dtype) == DC_BF16) { typedef bf16 T; __VA_ARGS__; } \
    else { typedef float T; __VA_ARGS__; }          \
  } while (0)

extern "C" int dc_avgpool_fwd(int dtype, int N, int HW, int C, const void* x, int ldx, void* out, void* stream) {
  if (int e = dc_check_view(x, ldx, C, dtype, "dc_avgpool_fwd x")) return e;
  DC_REQUIRE(out && N > 0 && HW > 0, "dc_avgpool_fwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int kpv = dtype == DC_BF16 ? 8 : 4;
  dim3 grid(cdiv(C / kpv, 16), N);
  DISPATCH_T(dtype, hipLaunchKernelGGL(hw_reduce_kernel<T>, grid, dim3(256), 0, st, HW, C, (const T*)x, ldx, 1.0f / (float)HW, (float*)out));
  DC_CHECK_LAUNCH();
  return 0;
}

extern "C" int dc_sum_hw(int dtype, int N, int HW, int C, const void* dout, int lddo, void* g, void* stream) {
  if (int e = dc_check_view(dout, lddo, C, dtype, "dc_sum_hw dout")) return e;
  DC_REQUIRE(g && N > 0 && HW > 0, "dc_sum_hw: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int kpv = dtype == DC_BF16 ? 8 : 4;
  dim3 grid(cdiv(C / kpv, 16), N);
  DISPATCH_T(dtype, hipLaunchKernelGGL(hw_reduce_kernel<T>, grid, dim3(256), 0, st, HW, C, (const T*)dout, lddo, 1.0f, (float*)g));
  DC_CHECK_LAUNCH();
  return 0;
}

extern "C" int dc_broadcast_hw(int dtype, int N, int HW, int C, const void* v, void* out, int ldo, void* stream) {
  if (int e = dc_check_view(out, ldo, C, dtype, "dc_broadcast_hw out")) return e;
  DC_REQUIRE(v && N > 0 && HW > 0, "dc_broadcast_hw: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const long rows = (long)N * HW;
  const int kpv = dtype == DC_BF16 ? 8 : 4;
  const int blocks = ew_blocks2(rows * (C / kpv));
  DISPATCH_T(dtype, hipLaunchKernelGGL((hw_ew_kernel<T, 0>), dim3(blocks), dim3(256), 0, st, rows, HW, C, v, C, 1.f, (T*)out, ldo));
  DC_CHECK_LAUNCH();
  return 0;
}

extern "C" int dc_avgpool_bwd_add(int dtype, int N, int HW, int C, const void* g, void* dx, int lddx, void* stream) {
  if (int e = dc_check_view(dx, lddx, C, dtype, "dc_avgpool_bwd_add dx")) return e;
  DC_REQUIRE(g && N > 0 && HW > 0, "dc_avgpool_bwd_add: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const long rows = (long)N * HW;
  const int kpv = dtype == DC_BF16 ? 8 : 4;
  const int blocks = ew_blocks2(rows * (C / kpv));
  DISPATCH_T(dtype, hipLaunchKernelGGL((hw_ew_kernel<T, 1>), dim3(blocks), dim3(256), 0, st, rows, HW, C, g, C, 1.0f / (float)HW, (T*)dx, lddx));
  DC_CHECK_LAUNCH();
  return 0;
}

extern "C" int dc_copy_view(int dtype, long M, int C, const void* src, int lds, void* dst, int ldd, void* stream) {
  if (int e = dc_check_view(src, lds, C, dtype, "dc_copy_view src")) return e;
  if (int e = dc_check_view(dst, ldd, C, dtype, "dc_copy_view dst")) return e;
  hipStream_t st = (hipStream_t)stream;
  const int kpv = dtype == DC_BF16 ? 8 : 4;
  const int blocks = ew_blocks2(M * (C / kpv));
  DISPATCH_T(dtype, hipLaunchKernelGGL((hw_ew_kernel<T, 2>), dim3(blocks), dim3(256), 0, st, M, 1, C, src, lds, 1.f, (T*)dst, ldd));
  DC_CHECK_LAUNCH();
  return 0;
}

extern "C" int dc_grad_pack_bf16(long n, const float* g, void* out_bf16, void* stream) {
  DC_REQUIRE(g && out_bf16 && n > 0, "dc_grad_pack_bf16: bad argument");
  DC_REQUIRE((((uintptr_t)g | (uintptr_t)out_bf16) & 15) == 0, "dc_grad_pack_bf16: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(grad_pack_bf16_kernel, dim3(ew_blocks2(n / 8 + 1)), dim3(256), 0, (hipStream_t)stream, n, g, (bf16*)out_bf16);
  DC_CHECK_LAUNCH();
  return 0;
}

extern "C" int dc_grad_unpack_bf16(long n, const void* in_bf16, float* g, void* stream) {
  DC_REQUIRE(g && in_bf16 && n > 0, "dc_grad_unpack_bf16: bad argument");
  DC_REQUIRE((((uintptr_t)g | (uintptr_t)in_bf16) & 15) == 0, "dc_grad_unpack_bf16: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(grad_unpack_bf16_kernel, dim3(ew_blocks2(n / 8 + 1)), dim3(256), 0, (hipStream_t)stream, n, (const bf16*)in_bf16, g);
  DC_CHECK_LAUNCH();
  return 0;
}

extern "C" int dc_adam_step(int kind, long n, float* p, const float* g, float* m, float* v, const float* lr_dev,
                            float beta1, float beta2, float eps, float weight_decay, const int* step_dev,
                            float grad_scale, void* stream) {
  DC_REQUIRE(kind == DC_ADAM || kind == DC_ADAMW, "dc_adam_step: kind must be DC_ADAM or DC_ADAMW");
  DC_REQUIRE(p && g && m && v && lr_dev && step_dev && n > 0, "dc_adam_step: bad argument");
  DC_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "dc_adam_step: arenas must be 16-byte aligned");
  hipLaunchKernelGGL(adam_kernel, dim3(ew_blocks2(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, kind, n, p, g, m, v, lr_dev,
                     beta1, beta2, eps, weight_decay, step_dev, grad_scale);
  DC_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t dc_lamb_workspace_words(int ntensors, long n) {
  // [SUMSQ_BLOCKS] gradient partial sums | [ntensors + 1] chunk plan | [2 * max_chunks] per-chunk norms | pad to 4 | [n] update
  const size_t head = (size_t)SUMSQ_BLOCKS + (size_t)ntensors + 1 + 2 * (size_t)(n / LAMB_CHUNK + ntensors);
  return (head + 3) / 4 * 4 + (size_t)n;
}

extern "C" int dc_lamb_step(int ntensors, const int64_t* offsets_dev, long n, float* p, const float* g, float* m,
                            float* v, const float* lr_dev, float beta1, float beta2, float eps, float weight_decay,
                            const int* step_dev, float max_grad_norm, float grad_scale, float* workspace, void* stream) {
  DC_REQUIRE(ntensors > 0 && offsets_dev && p && g && m && v && lr_dev && step_dev && workspace && n > 0, "dc_lamb_step: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const long max_chunks = n / LAMB_CHUNK + ntensors;   // every tensor rounds up by less than one chunk
  DC_REQUIRE(max_chunks < (1L << 30), "dc_lamb_step: arena too large");
  DC_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)workspace) & 15) == 0, "dc_lamb_step: arenas and workspace must be 16-byte aligned");
  // workspace words: [SUMSQ_BLOCKS] gradient partial sums | [ntensors + 1] chunk plan | [2 * max_chunks] per-chunk norms |
  // (padded to a multiple of 4 words) [n] the update direction u.  The gradient arena is only read: a caller may keep using it
  // (gradient logging, accumulation) after the step, as with apex FusedLAMB.
  float* gpartial = workspace;
  int* chunk_prefix = reinterpret_cast<int*>(workspace + SUMSQ_BLOCKS);
  float* part = workspace + SUMSQ_BLOCKS + ntensors + 1;
  const size_t head = (size_t)SUMSQ_BLOCKS + (size_t)ntensors + 1 + 2 * (size_t)max_chunks;
  float* upd = workspace + (head + 3) / 4 * 4;
  hipLaunchKernelGGL(sumsq_kernel, dim3(SUMSQ_BLOCKS), dim3(256), 0, st, n, g, grad_scale, gpartial);
  DC_CHECK_LAUNCH();
  hipLaunchKernelGGL(lamb_plan_kernel, dim3(1), dim3(64), 0, st, offsets_dev, ntensors, chunk_prefix);
  DC_CHECK_LAUNCH();
  hipLaunchKernelGGL(lamb_stage1_kernel, dim3((unsigned)max_chunks), dim3(256), 0, st, offsets_dev, chunk_prefix, ntensors, p, g, upd,
                     m, v, beta1, beta2, eps, weight_decay, step_dev, max_grad_norm, grad_scale, gpartial, part);
  DC_CHECK_LAUNCH();
  // the per-tensor step sizes go where the gradient partial sums were (stage 1 was their last reader)
  float* step_size = ntensors <= SUMSQ_BLOCKS ? gpartial : nullptr;
  if (step_size != nullptr) {
    hipLaunchKernelGGL(lamb_ratio_kernel, dim3(ntensors), dim3(256), 0, st, chunk_prefix, ntensors, lr_dev, part, weight_decay, step_size);
    DC_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(lamb_stage2_kernel, dim3((unsigned)max_chunks), dim3(256), 0, st, offsets_dev, chunk_prefix, ntensors, p, upd,
                     lr_dev, part, weight_decay, step_size);
  DC_CHECK_LAUNCH();
  return 0;
}
