// How does the range check of `buffer_load_dwordx4 ... offen lds` (raw buffer, stride 0) treat the scalar offset and a "negative" vector offset?
//   hipcc --offload-arch=gfx950 -O2 -o bin_tmp/buffer_range_probe scripts/buffer_range_probe.hip && bin_tmp/buffer_range_probe
// The resource covers bytes [0, 1024) of a 4 KiB allocation of ones (as floats); a lane whose access the hardware calls out of range gets zeros.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr;
__global__ void probe(const float* src, int nrec, unsigned base_voff, unsigned soff, float* out) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 4];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = 123.f;
  __syncthreads();
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nrec, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)lds, 16, base_voff + threadIdx.x * 16, soff, 0, 0);
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out[threadIdx.x] = lds[threadIdx.x * 4];
}
int main() {
  float *src, *out;
  hipMalloc(&src, 8192); hipMalloc(&out, 256);
  std::vector<float> h(2048);
  for (int i = 0; i < 2048; ++i) h[i] = 1.f + i / 4;       // element value = 1 + (byte offset / 16)
  hipMemcpy(src, h.data(), 8192, hipMemcpyHostToDevice);
  struct { const char* what; int nrec; unsigned voff, soff; const float* base; } cases[] = {
    {"voffset 0.., soffset 0, records 1024     ", 1024, 0u, 0u, src},
    {"voffset 0.., soffset 512, records 1024   ", 1024, 0u, 512u, src},
    {"voffset -256.., soffset 512, records 1024", 1024, (unsigned)-256, 512u, src},
    {"voffset -256.., soffset 0, base + 1024    ", 1024, (unsigned)-256, 0u, src + 256},
  };
  for (auto& c : cases) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, c.base, c.nrec, c.voff, c.soff, out);
    float o[64]; hipMemcpy(o, out, 256, hipMemcpyDeviceToHost);
    printf("%s: lanes 0,8,15,16,24,31,32,40,63 ->", c.what);
    for (int l : {0, 8, 15, 16, 24, 31, 32, 40, 63}) printf(" %g", o[l]);
    printf("\n");
  }
  return 0;
}
