// Which property of a kernel makes the dependent boundary behind / in front of it expensive?  The step's kernel trace shows two kinds of
// boundary on one queue: ~2.5 us (next kernel's start stamp == previous end stamp) and ~7 us (5.8 us of idle between the stamps), the
// second kind around the MFMA GEMMs and the tiled depthwise kernels.  Factors tried one at a time on a synthetic streaming kernel A
// followed by a tiny dependent kernel B: dynamic LDS size, workgroup size, kernarg bytes, grid size, duration, LDS-DMA use, scratch.
//   hipcc --offload-arch=gfx950 -O3 -o boundary_factors scripts/boundary_factors.hip && ./boundary_factors
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int PAD>
struct Args {
  const u32x4* src;
  u32x4* dst;
  long n16;
  int use_lds, use_dma;
  unsigned pad[PAD];
};

extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

template <int PAD, int WG>
__global__ __launch_bounds__(WG) void kernA(const Args<PAD> a) {
  unsigned extra = 0;
  if (a.use_lds) {   // touch the allocation so it is not optimised away
    reinterpret_cast<unsigned*>(smem)[threadIdx.x] = threadIdx.x + a.pad[0];
    __syncthreads();
    extra = reinterpret_cast<unsigned*>(smem)[(threadIdx.x + 1) % WG];
  }
  if (a.use_dma) {
    typedef __attribute__((address_space(1))) const void* gptr;
    typedef __attribute__((address_space(3))) void* lptr;
    __builtin_amdgcn_global_load_lds((gptr)(a.src + threadIdx.x), (lptr)(smem + 4096 + (threadIdx.x / 64) * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    extra += reinterpret_cast<unsigned*>(smem + 4096)[threadIdx.x];
  }
  for (long i = blockIdx.x * (long)WG + threadIdx.x; i < a.n16; i += (long)gridDim.x * WG) {
    u32x4 v = a.src[i];
    v.x += extra;
    a.dst[i] = v;
  }
}

__global__ void tinyB(const unsigned* __restrict__ a, unsigned* __restrict__ b) { b[threadIdx.x] = a[threadIdx.x] + 1; }

struct Cfg {
  const char* name;
  int lds, wg512, bigargs, grid, use_dma, mb;
};

template <int PAD, int WG>
static void launchA(const Cfg& c, const u32x4* src, u32x4* dst, long n16, hipStream_t st) {
  Args<PAD> a;
  a.src = src; a.dst = dst; a.n16 = n16; a.use_lds = c.lds > 0; a.use_dma = c.use_dma;
  for (int i = 0; i < PAD; ++i) a.pad[i] = 0;
  static bool set = false;
  if (!set) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&kernA<PAD, WG>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; }
  hipLaunchKernelGGL((kernA<PAD, WG>), dim3(c.grid), dim3(WG), c.lds, st, a);
}

static void launch(const Cfg& c, const u32x4* src, u32x4* dst, long n16, hipStream_t st) {
  if (c.wg512) { if (c.bigargs) launchA<240, 512>(c, src, dst, n16, st); else launchA<1, 512>(c, src, dst, n16, st); }
  else { if (c.bigargs) launchA<240, 256>(c, src, dst, n16, st); else launchA<1, 256>(c, src, dst, n16, st); }
}

static float timeit(int reps, const std::vector<const Cfg*>& seq, bool withB, const u32x4* src, u32x4* dst, unsigned* small, hipStream_t st) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto body = [&]() {
    for (const Cfg* c : seq) {
      launch(*c, src, dst, ((long)c->mb << 20) / 16, st);
      if (withB) hipLaunchKernelGGL(tinyB, dim3(1), dim3(64), 0, st, (const unsigned*)dst, small);
    }
  };
  for (int i = 0; i < 5; ++i) body();
  CK(hipStreamSynchronize(st));
  std::vector<float> ts;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) body();
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ts.push_back(ms * 1000.f / reps);
  }
  std::sort(ts.begin(), ts.end());
  return ts[2];
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  const long bytes = 160L << 20;
  u32x4 *src, *dst; unsigned* small;
  CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes)); CK(hipMalloc(&small, 4096));
  CK(hipMemset(src, 1, bytes));
  //            name                        lds      wg512 bigargs grid  dma mb
  Cfg cfgs[] = {{"base 256thr 2048wg",          0,      0, 0, 2048, 0, 40},
                {"lds 16K",                 16384,      0, 0, 2048, 0, 40},
                {"lds 48K",                 49152,      0, 0, 2048, 0, 40},
                {"lds 64K",                 65536,      0, 0, 2048, 0, 40},
                {"lds 80K",                 81920,      0, 0, 2048, 0, 40},
                {"lds 128K",               131072,      0, 0, 2048, 0, 40},
                {"lds 160K",               163840,      0, 0, 2048, 0, 40},
                {"512thr",                      0,      1, 0, 1024, 0, 40},
                {"512thr lds 128K",        131072,      1, 0, 1024, 0, 40},
                {"512thr lds 128K 216wg",  131072,      1, 0,  216, 0, 40},
                {"512thr lds 128K 216wg dma", 131072,   1, 0,  216, 1, 40},
                {"1KB kernarg",                 0,      0, 1, 2048, 0, 40},
                {"512thr lds128K 216wg 1KB dma", 131072, 1, 1, 216, 1, 40},
                {"base 4MB",                    0,      0, 0, 2048, 0, 4},
                {"lds 128K 4MB",           131072,      0, 0, 2048, 0, 4},
                {"lds 48K dma",             49152,      0, 0, 2048, 1, 40}};
  const int n = sizeof(cfgs) / sizeof(cfgs[0]);
  printf("%-34s %10s %12s %10s\n", "kernel A", "A x N", "(A,tiny) x N", "delta");
  for (int i = 0; i < n; ++i) {
    std::vector<const Cfg*> s{&cfgs[i]};
    const float a = timeit(100, s, false, src, dst, small, st);
    const float b = timeit(100, s, true, src, dst, small, st);
    printf("%-34s %10.2f %12.2f %10.2f\n", cfgs[i].name, a, b, b - a);
  }
  // alternating two kinds of A (different LDS configuration) with no tiny kernel in between
  printf("alternating pairs (us per pair; sum of the two alone in brackets)\n");
  int pairs[][2] = {{0, 5}, {0, 9}, {2, 9}, {9, 9}, {0, 0}, {12, 0}, {12, 2}};
  for (auto& p : pairs) {
    std::vector<const Cfg*> s{&cfgs[p[0]], &cfgs[p[1]]}, s0{&cfgs[p[0]]}, s1{&cfgs[p[1]]};
    const float ab = timeit(100, s, false, src, dst, small, st);
    const float a = timeit(100, s0, false, src, dst, small, st), b = timeit(100, s1, false, src, dst, small, st);
    printf("  [%s] + [%s]: %8.2f (%8.2f)  extra %6.2f\n", cfgs[p[0]].name, cfgs[p[1]].name, ab, a + b, ab - a - b);
  }
  return 0;
}
