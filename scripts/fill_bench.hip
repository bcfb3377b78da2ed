// Micro-benchmark behind DESIGN.md's "what bounds the implicit-GEMM K loop": the LDS-DMA fill of a 256 x 256 GEMM tile in the
// access pattern of igemm256_kernel (a 512-thread workgroup gathers 256 pixel rows of a [M][ld] bf16 tensor and 256 weight rows
// per K stage), with and without the stage's MFMAs beside it, for 64-byte K rows (16 rows per wave-instruction: half an L2 line
// per row) and 128-byte K rows (8 rows per wave-instruction: whole lines).
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fill_bench scripts/fill_bench.hip && /tmp/fill_bench
//
// Every variant moves the same bytes per tile and issues the same number of LDS-DMA instructions; what changes is the row
// length (ROWB), the instructions per barrier interval (a 64-deep stage issues 4 + 4 per wave) and whether MFMAs run beside them.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(1))) const void* gas_ptr;
typedef __attribute__((address_space(3))) void* lds_ptr;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// ROWB: bytes of K per row per stage.  A "stage" is 32 KiB (ROWB 64) or 64 KiB (ROWB 128) of operands; the ring holds 128 KiB.
// DMA: 0 none, 1 LDS-DMA (global_load_lds, 64-bit per-lane address), 2 global_load_dwordx4 + ds_write_b128, 3 LDS-DMA as
// buffer_load ... offen lds (resource descriptor + 32-bit per-lane offset + scalar K offset).  NMFMA: MFMAs per wave per 32-deep K step (32 = the real kernel).
template <int ROWB, int DMA, int NMFMA, int READS, int WL = 0>
__global__ __launch_bounds__(512) void fill_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ w, int M, int C, int ld, int ntn,
                                                   float* __restrict__ out, unsigned long long* __restrict__ cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = 512 * ROWB;                 // both operands
  constexpr int NSTAGE = 131072 / STAGE;            // 4 or 2
  constexpr int IPW = STAGE / (8 * 1024);           // LDS-DMA instructions per wave per stage: 4 or 8
  constexpr int RPI = 1024 / ROWB;                  // rows per instruction: 16 or 8
  constexpr int LPR = ROWB / 16;                    // lanes per row: 4 or 8
  constexpr int KPS = ROWB / 2;                     // K elements per stage
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nwg = gridDim.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, xslot = blockIdx.x >> 3;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + xslot;
  const int nt = tile % ntn, mt = tile / ntn;
  const int lrow = lane / LPR, lslot = lane % LPR;
  // instruction i of this wave: operand (i < IPW/2 ? weights : pixels), rows ((i % (IPW/2)) * 8 + wave) * RPI + lrow
  const __bf16* src[IPW];
  unsigned voff[IPW];
#pragma unroll
  for (int i = 0; i < IPW; ++i) {
    const int r = ((i % (IPW / 2)) * 8 + wave) * RPI + lrow;
    if (i < IPW / 2) {
      int ch = nt * 256 + r;
      if (ch >= C) ch = C - 1;
      // WL = 1: weights stage-major, [K / KPS][C][KPS]: the rows of one stage are contiguous (a wave-instruction reads 1 KiB in one piece)
      src[i] = WL ? w + (size_t)ch * KPS + lslot * 8 : w + (size_t)ch * ld + lslot * 8;
      voff[i] = (unsigned)(((size_t)ch * ld + lslot * 8) * 2);
    } else {
      int m = mt * 256 + r;
      if (m >= M) m = M - 1;
      src[i] = x + (size_t)m * ld + lslot * 8;
      voff[i] = (unsigned)(((size_t)m * ld + lslot * 8) * 2);
    }
  }
#if defined(__HIP_DEVICE_COMPILE__)     // (the buffer-resource builtins do not exist in the host pass)
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((size_t)M * ld * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (int)((size_t)C * ld * 2), 0x00020000);
#endif
  const int steps = (C + KPS - 1) / KPS;
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 fa, fb;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    fa[i] = (__bf16)(0.001f * (lane + i));
    fb[i] = (__bf16)(0.002f * (lane - i));
  }
  const int steps_ = (C + KPS - 1) / KPS;
  auto issue = [&](int s, int i) {
    if constexpr (DMA == 0) return;
    int k = s * KPS;
    if (k + KPS > ld) k = ld - KPS;                // stay inside the row (timing only)
    char* dst = smem + (s % NSTAGE) * STAGE + (i * 8 + wave) * 1024;
    if constexpr (DMA == 1) {
      const size_t adv = (WL && i < IPW / 2) ? (size_t)(s < steps_ ? s : steps_ - 1) * C * KPS : (size_t)k;
      __builtin_amdgcn_global_load_lds((gas_ptr)(src[i] + adv), (lds_ptr)dst, 16, 0, 0);
    } else if constexpr (DMA == 3) {
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(i < IPW / 2 ? rw : rx, (lds_ptr)dst, 16, voff[i], k * 2, 0, 0);
#endif
    } else {
      const uint4 v = *reinterpret_cast<const uint4*>(src[i] + k);
      *reinterpret_cast<uint4*>(dst + lane * 16) = v;
    }
  };
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  // prologue: NSTAGE - 1 stages in flight
  for (int s = 0; s < NSTAGE - 1; ++s)
#pragma unroll
    for (int i = 0; i < IPW; ++i) issue(s, i);
  constexpr int SEGS = ROWB / 32;                   // 32-deep K steps per stage: 2 or 4 MFMA segments of NMFMA/2 each
  float sink = 0.f;
  for (int s = 0; s < steps; ++s) {
    // everything but the newest (NSTAGE-2) stages has landed -> stage s is readable
    if constexpr (DMA == 1 || DMA == 3) {
      if constexpr (NSTAGE == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    const char* cur = smem + (s % NSTAGE) * STAGE;
#pragma unroll
    for (int seg = 0; seg < SEGS; ++seg) {
      // the next stage's LDS-DMA instructions are spread over this stage's segments (slot (s-1) % NSTAGE is free: every wave passed the barrier)
      // (a 2-slot ring must issue early: the newest instructions need time to land before the next stage's wait)
      constexpr int ISEGS = NSTAGE == 2 ? SEGS / 2 : SEGS;
      if (seg < ISEGS) {
#pragma unroll
        for (int i = seg * (IPW / ISEGS); i < (seg + 1) * (IPW / ISEGS); ++i) issue(s + NSTAGE - 1, i);
      }
      if constexpr (READS > 0) {
#pragma unroll
        for (int r = 0; r < READS; ++r) {
          const uint4 v = *reinterpret_cast<const uint4*>(cur + ((seg * READS + r) * 512 + tid) % (STAGE / 16) * 16);
          fb[r & 7] = __builtin_bit_cast(__bf16, (unsigned short)(v.x ^ v.y ^ v.z ^ v.w));
        }
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int m = 0; m < NMFMA / 2; ++m) acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[m & 7], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int i = 0; i < 8; ++i) sink += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (sink == 12345.678f) out[blockIdx.x] = sink + smem[tid];
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int ROWB, int DMA, int NMFMA, int READS, int WL = 0>
void run(const char* name, const __bf16* x, const __bf16* w, int M, int C, int ld, float* out, unsigned long long* cyc, int nwg_override) {
  const int ntn = (C + 255) / 256, ntm = (M + 255) / 256;
  const int nwg = nwg_override > 0 ? nwg_override : ntn * ntm;
  auto k = fill_kernel<ROWB, DMA, NMFMA, READS, WL>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 131072 + 256));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(nwg), dim3(512), 131072 + 256, 0, x, w, M, C, ld, ntn, out, cyc);
  CK(hipDeviceSynchronize());
  const int reps = 20;
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(nwg), dim3(512), 131072 + 256, 0, x, w, M, C, ld, ntn, out, cyc);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps;
  std::vector<unsigned long long> h(nwg);
  CK(hipMemcpy(h.data(), cyc, nwg * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double mean = 0;
  for (auto v : h) mean += (double)v;
  mean /= nwg;
  const int ksteps32 = (C + 31) / 32;
  const double bytes = (double)nwg * ((C + ROWB / 2 - 1) / (ROWB / 2)) * 512.0 * ROWB;
  printf("%-44s wgs %4d  %8.1f us  fill %6.2f TB/s  in-kernel %7.0f cycles = %6.0f per 32-deep K step (MFMA alone: 1024)\n", name, nwg, us,
         DMA ? bytes / us * 1e-6 : 0.0, mean, mean / ksteps32);
}


// ---- one wave per SIMD: 256-thread workgroup, 256 x 384 tile (a wave owns 128 pixels x 192 channels: 8 x 12 MFMA tiles, 384
// accumulator registers of the SIMD's 512), 32-deep K steps, ring of 3 x 40 KiB.  Everything a step needs besides its 96 MFMAs
// (10 LDS-DMA issues, 20 ds_read_b128 for the NEXT operands) is interleaved between the MFMAs of one instruction stream.
template <int DMA, int READS, int NCB>
__global__ __launch_bounds__(256) void fill4_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ w, int M, int C, int ld, int ntn,
                                                    float* __restrict__ out, unsigned long long* __restrict__ cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TMp = 256, TNp = 32 * NCB, STAGE = (TMp + TNp) * 64, NST = 3, IPW = STAGE / 1024 / 4;   // 40 KiB, 10 per wave
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nwg = gridDim.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, xslot = blockIdx.x >> 3;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + xslot;
  const int nt = tile % ntn, mt = tile / ntn;
  const int lrow = lane >> 2, lslot = lane & 3;
  const __bf16* src[IPW];
#pragma unroll
  for (int i = 0; i < IPW; ++i) {
    const int r = (i * 4 + wave) * 16 + lrow;          // row of the 640-row stage image: [0,384) weights, [384,640) pixels
    if (r < TNp) {
      int ch = nt * TNp + r;
      if (ch >= C) ch = C - 1;
      src[i] = w + (size_t)ch * ld + lslot * 8;
    } else {
      int m = mt * TMp + r - TNp;
      if (m >= M) m = M - 1;
      src[i] = x + (size_t)m * ld + lslot * 8;
    }
  }
  const int steps = (C + 31) / 32;
  f32x4 acc[NCB][8];
#pragma unroll
  for (int i = 0; i < NCB; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[2], fb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    fa[0][e] = fa[1][e] = (__bf16)(0.001f * (lane + e));
#pragma unroll
    for (int j = 0; j < 8; ++j) fb[j][e] = (__bf16)(0.002f * (lane - e + j));
  }
  auto issue = [&](int s, int i) {
    if constexpr (DMA == 0) return;
    int k = s * 32;
    if (k + 32 > ld) k = ld - 32;
    char* dst = smem + (s % NST) * STAGE + (i * 4 + wave) * 1024;
    __builtin_amdgcn_global_load_lds((gas_ptr)(src[i] + k), (lds_ptr)dst, 16, 0, 0);
  };
  auto rd = [&](const char* base, int idx) {
    const uint4 v = *reinterpret_cast<const uint4*>(base + ((idx * 256 + tid) * 16) % STAGE);
    return __builtin_bit_cast(bf16x8, v);
  };
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < NST - 1; ++s)
#pragma unroll
    for (int i = 0; i < IPW; ++i) issue(s, i);
  for (int s = 0; s < steps; ++s) {
    // stage s+1 must have landed before this step's reads of it: only the newest stage may still be in flight
    if constexpr (DMA == 1) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const char* nxt = smem + ((s + 1) % NST) * STAGE;
#pragma unroll
    for (int i = 0; i < NCB; ++i) {
      // between the MFMAs of channel block i: the next block's weight fragment and one LDS-DMA issue of stage s+2; the pixel
      // fragments of the next step are re-read in place right after their last use (channel block 11)
      if constexpr (READS) fa[(i + 1) & 1] = rd(nxt, i);
      if (i < IPW) issue(s + NST - 1, i);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i & 1], fb[j], acc[i][j], 0, 0, 0);
        if constexpr (READS) {
          if (i == NCB - 1) fb[j] = rd(nxt, NCB + j);
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float sink = 0.f;
#pragma unroll
  for (int i = 0; i < NCB; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) sink += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (sink == 12345.678f) out[blockIdx.x] = sink + smem[tid];
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int DMA, int READS, int NCB>
void run4(const char* name, const __bf16* x, const __bf16* w, int M, int C, int ld, float* out, unsigned long long* cyc) {
  const int ntn = (C + 32 * NCB - 1) / (32 * NCB), ntm = (M + 255) / 256;
  const int nwg = ntn * ntm;
  auto k = fill4_kernel<DMA, READS, NCB>;
  const int lds = 3 * (256 + 32 * NCB) * 64 + 256;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(nwg), dim3(256), lds, 0, x, w, M, C, ld, ntn, out, cyc);
  CK(hipDeviceSynchronize());
  const int reps = 20;
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(nwg), dim3(256), lds, 0, x, w, M, C, ld, ntn, out, cyc);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps;
  std::vector<unsigned long long> h(nwg);
  CK(hipMemcpy(h.data(), cyc, nwg * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double mean = 0;
  for (auto v : h) mean += (double)v;
  mean /= nwg;
  const int ksteps32 = (C + 31) / 32;
  printf("%-44s wgs %4d  %8.1f us  in-kernel %7.0f cycles = %6.0f per 32-deep K step (MFMA alone: %d)  -> %6.0f TF/s algorithmic\n", name, nwg, us, mean,
         mean / ksteps32, 128 * NCB, 2.0 * M * C * (double)C / us * 1e-6);
}

// Does an out-of-range lane of buffer_load ... lds write zeros into LDS (usable as padding) or leave the bytes alone?
__global__ void oob_kernel(const float* src, int nbytes, float* out) {
  __shared__ __attribute__((aligned(16))) float lds[256];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = 123.f;
  __syncthreads();
  // lanes 0..31 in range, lanes 32..63 far out of range
  const unsigned off = threadIdx.x < 32 ? threadIdx.x * 16 : 0x40000000u + threadIdx.x * 16;
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)lds, 16, off, 0, 0, 0);
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}
void oob_test() {
  float *src, *out;
  CK(hipMalloc(&src, 4096));
  CK(hipMalloc(&out, 1024));
  std::vector<float> h(1024, 7.f), o(256);
  CK(hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(oob_kernel, dim3(1), dim3(64), 0, 0, src, 4096, out);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(o.data(), out, 1024, hipMemcpyDeviceToHost));
  printf("--- buffer_load ... lds, lanes 32..63 out of range: in-range lane dword %g, out-of-range lane dwords %g %g %g %g (123 = LDS untouched, 0 = zero-filled)\n",
         o[0], o[128], o[129], o[200], o[255]);
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 27648, C = argc > 2 ? atoi(argv[2]) : 728;
  // row stride in elements (third argument): the default pads rows to whole 128-byte lines; 728 / 736 are the strides the 728-channel
  // activations / weight images of the engine had before round 5's padding experiment
  const int ld = argc > 3 ? atoi(argv[3]) : (C + 63) / 64 * 64;
  __bf16 *x, *w;
  float* out;
  unsigned long long* cyc;
  CK(hipMalloc(&x, (size_t)M * ld * 2));
  CK(hipMalloc(&w, (size_t)C * ld * 2));
  CK(hipMalloc(&out, 1 << 20));
  CK(hipMalloc(&cyc, 1 << 20));
  std::vector<unsigned short> hx((size_t)M * ld);
  for (size_t i = 0; i < hx.size(); ++i) hx[i] = (unsigned short)(0x3c00 + (i * 2654435761u >> 20 & 0x3ff));
  CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(w, hx.data(), (size_t)C * ld * 2, hipMemcpyHostToDevice));
  printf("GEMM tile fill pattern: M %d, K = N = %d (ld %d), 256 x 256 tiles, 512 threads\n", M, C, ld);
  for (int nwg : {0, 256}) {
    printf("--- %s\n", nwg ? "one round: 256 workgroups" : "whole layer");
    run<64, 0, 32, 0>("MFMA only (32 per K step per wave)", x, w, M, C, ld, out, cyc, nwg);
    run<64, 1, 0, 0>("LDS-DMA only, 64-byte rows", x, w, M, C, ld, out, cyc, nwg);
    run<128, 1, 0, 0>("LDS-DMA only, 128-byte rows", x, w, M, C, ld, out, cyc, nwg);
    run<64, 2, 0, 0>("load + ds_write only, 64-byte rows", x, w, M, C, ld, out, cyc, nwg);
    run<128, 2, 0, 0>("load + ds_write only, 128-byte rows", x, w, M, C, ld, out, cyc, nwg);
    run<64, 3, 0, 0>("buffer LDS-DMA only, 64-byte rows", x, w, M, C, ld, out, cyc, nwg);
    run<128, 3, 0, 0>("buffer LDS-DMA only, 128-byte rows", x, w, M, C, ld, out, cyc, nwg);
    run<64, 3, 32, 0>("buffer LDS-DMA + MFMA, 64-byte rows", x, w, M, C, ld, out, cyc, nwg);
    run<128, 3, 32, 0>("buffer LDS-DMA + MFMA, 128-byte rows", x, w, M, C, ld, out, cyc, nwg);
    run<64, 1, 32, 0>("LDS-DMA + MFMA, 64-byte rows", x, w, M, C, ld, out, cyc, nwg);
    run<128, 1, 32, 0>("LDS-DMA + MFMA, 128-byte rows", x, w, M, C, ld, out, cyc, nwg);
    run<128, 1, 0, 0, 1>("LDS-DMA only, 128-byte rows, weights stage-major", x, w, M, C, ld, out, cyc, nwg);
    run<128, 1, 32, 0, 1>("LDS-DMA + MFMA, 128-byte rows, weights stage-major", x, w, M, C, ld, out, cyc, nwg);
    run<64, 1, 0, 0, 1>("LDS-DMA only, 64-byte rows, weights stage-major", x, w, M, C, ld, out, cyc, nwg);
    run<64, 1, 32, 0, 1>("LDS-DMA + MFMA, 64-byte rows, weights stage-major", x, w, M, C, ld, out, cyc, nwg);
    run<64, 1, 32, 6>("LDS-DMA + MFMA + 12 ds_read, 64-byte rows", x, w, M, C, ld, out, cyc, nwg);
    run<128, 1, 32, 6>("LDS-DMA + MFMA + 12 ds_read, 128-byte rows", x, w, M, C, ld, out, cyc, nwg);
  }
  oob_test();
  printf("--- one wave per SIMD, 256 x 384 tile, 4 waves\n");
  run4<0, 0, 12>("256x384: MFMA only (96 per K step per wave)", x, w, M, C, ld, out, cyc);
  run4<1, 0, 12>("256x384: LDS-DMA + MFMA", x, w, M, C, ld, out, cyc);
  run4<0, 1, 12>("256x384: ds_read + MFMA", x, w, M, C, ld, out, cyc);
  run4<1, 1, 12>("256x384: LDS-DMA + 20 ds_read + MFMA", x, w, M, C, ld, out, cyc);
  run4<0, 0, 8>("256x256: MFMA only (64 per K step per wave)", x, w, M, C, ld, out, cyc);
  run4<1, 0, 8>("256x256: LDS-DMA + MFMA", x, w, M, C, ld, out, cyc);
  run4<0, 1, 8>("256x256: ds_read + MFMA", x, w, M, C, ld, out, cyc);
  run4<1, 1, 8>("256x256: LDS-DMA + 16 ds_read + MFMA", x, w, M, C, ld, out, cyc);
  return 0;
}
