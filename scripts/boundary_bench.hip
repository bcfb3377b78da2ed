// What a dependent kernel boundary costs behind a kernel that has just written a large tensor, by the flavour of its stores.
// Hypothesis (MI355X guide, "boundary" row: + B / 6 TB/s when the predecessor leaves B bytes dirty in the eight L2s): a producer
// that writes 40 MB with plain stores leaves up to 32 MB dirty, which the end-of-kernel release has to write back before the
// next kernel may start; write-through (sc1) stores leave nothing.
//   hipcc --offload-arch=gfx950 -O3 -o boundary_bench scripts/boundary_bench.hip && ./boundary_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int MODE>
__device__ inline void st16(void* p, u32x4 v) {
  if (MODE == 0) *reinterpret_cast<u32x4*>(p) = v;
  else if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
  else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

// copy-like producer: reads src, writes dst (n16 16-byte vectors), grid-stride
template <int MODE>
__global__ __launch_bounds__(256) void producer(const u32x4* __restrict__ src, u32x4* __restrict__ dst, long n16) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n16; i += (long)gridDim.x * 256) {
    u32x4 v = src[i];
    v.x += 1;
    st16<MODE>(dst + i, v);
  }
}

__global__ void tiny(const unsigned* __restrict__ a, unsigned* __restrict__ b) { b[threadIdx.x] = a[threadIdx.x] + 1; }

// consumer that streams the whole tensor (does a dropped L2 line cost the reader anything?)
__global__ __launch_bounds__(256) void consumer(const u32x4* __restrict__ src, unsigned* __restrict__ out, long n16) {
  unsigned acc = 0;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n16; i += (long)gridDim.x * 256) {
    const u32x4 v = src[i];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE>
static float run(int reps, int kind, const u32x4* src, u32x4* dst, unsigned* small, long n16, hipStream_t st) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int blocks = 2048;
  auto body = [&]() {
    hipLaunchKernelGGL(producer<MODE>, dim3(blocks), dim3(256), 0, st, src, dst, n16);
    if (kind == 1) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, (const unsigned*)dst, small);
    if (kind == 2) hipLaunchKernelGGL(consumer, dim3(blocks), dim3(256), 0, st, (const u32x4*)dst, small, n16);
  };
  for (int i = 0; i < 5; ++i) body();
  CK(hipStreamSynchronize(st));
  std::vector<float> ts;
  for (int r = 0; r < 7; ++r) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) body();
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ts.push_back(ms * 1000.f / reps);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}

int main() {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const char* names[4] = {"plain", "nt", "sc1", "sc0 sc1"};
  for (long mb : {4L, 40L, 160L}) {
    const long bytes = mb << 20, n16 = bytes / 16;
    u32x4 *src, *dst;
    unsigned* small;
    CK(hipMalloc(&src, bytes));
    CK(hipMalloc(&dst, bytes));
    CK(hipMalloc(&small, 4096));
    CK(hipMemset(src, 1, bytes));
    const int reps = 100;
    printf("tensor %ld MB (us per iteration, median of 7 x %d)\n", mb, reps);
    printf("  %-8s %12s %18s %12s %22s\n", "stores", "producer", "producer+tiny", "boundary", "producer+full reader");
    for (int m = 0; m < 4; ++m) {
      float a, b, c;
      if (m == 0) { a = run<0>(reps, 0, src, dst, small, n16, st); b = run<0>(reps, 1, src, dst, small, n16, st); c = run<0>(reps, 2, src, dst, small, n16, st); }
      if (m == 1) { a = run<1>(reps, 0, src, dst, small, n16, st); b = run<1>(reps, 1, src, dst, small, n16, st); c = run<1>(reps, 2, src, dst, small, n16, st); }
      if (m == 2) { a = run<2>(reps, 0, src, dst, small, n16, st); b = run<2>(reps, 1, src, dst, small, n16, st); c = run<2>(reps, 2, src, dst, small, n16, st); }
      if (m == 3) { a = run<3>(reps, 0, src, dst, small, n16, st); b = run<3>(reps, 1, src, dst, small, n16, st); c = run<3>(reps, 2, src, dst, small, n16, st); }
      printf("  %-8s %12.2f %18.2f %12.2f %22.2f\n", names[m], a, b, b - a, c);
    }
    CK(hipFree(src));
    CK(hipFree(dst));
    CK(hipFree(small));
  }
  return 0;
}
