// LDS-tiled stride-1 depthwise kernels (dwtile.hip), called from the C ABI in dwconv.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace dc {

struct DwBnStats;   // dwtile_common.h
struct BnFinArgs;   // bn_fin.h
// the BatchNorm whose stored output relu(bn(y) + residual) is a depthwise layer's input: its backward sums ride along with that layer's data
// gradient (dwpipe.hip, PM_RES)
struct DwResStats {
  const void* y;
  int ldy;
  const float* mean;
  const float* invstd;
  int relu;
  float* slab;
  int sum_row;      // slab is a SUM ROW (bn_fin.h): double[2][C], zeroed by the caller, added to by every workgroup
};

constexpr int DWT_MAX_ROWS = 2048;   // most partial rows the weight-gradient slab may hold

int launch_dw_tile(int dtype, int dil, bool flip, const void* in, int ldin, const float* wp, const void* addend, int ldadd,
                   void* out, int ldout, int N, int H, int W, int C, hipStream_t st, const float* pscale = nullptr,
                   const float* pshift = nullptr, int prelu = 0, const DwBnStats* bnstats = nullptr, const BnFinArgs* fin = nullptr);
int dw_tile_rows(int dtype, int C, int N, int H, int W);   // pixel tiles of the stride-1 kernels = slab rows of the fused BN statistics
int launch_dw_tile_wgrad(int dtype, int dil, const void* x, int ldx, const void* dy, int lddy, float* slab, float* grad_w, int N,
                         int H, int W, int C, hipStream_t st, const float* pscale = nullptr, const float* pshift = nullptr, int prelu = 0);
size_t dw_tile_wgrad_workspace(int C, int N, int H, int W);
void dw_tile_set_tpb(int v);
void dw_tile_set_cg(int v);
int dw_tile_reduce(const float* slab, float* grad_w, int rows, int C, hipStream_t st);   // grad[c][t] = sum of slab rows (fp64, fixed order)

// persistent pipelined kernel (dwpipe.hip): bf16, >= 128 channels, image extents multiples of 8.  dw_pipe_rows: workgroups per channel
// block = slab rows of the sums that ride along with the data gradient; 0 = shape not served (or switched off: option "dw_pipe")
int dw_pipe_rows(int dtype, int C, int dil, int N, int H, int W);
int launch_dw_pipe(int dil, bool flip, const void* in, int ldin, const float* wp, const void* addend, int ldadd, void* out, int ldout, int N,
                   int H, int W, int C, hipStream_t st, const float* pscale, const float* pshift, int prelu, const DwBnStats* bnstats,
                   const DwResStats* res = nullptr);
void dw_pipe_set(int v);
void pw_bn_bwd_set(int v);       // pwbwd.hip: tuning switch "pw_bn_bwd"
void sep_fwd_set(int v);         // sepfwd.hip: tuning switch "sep_fwd"
bool dw_pipe_forward();

// stride-2 kernels (dwtile_s2.hip).  mode 0 forward, 1 data gradient (p1 = addend or null), 2 weight-gradient rows into the slab
int launch_dw_tile_s2(int dtype, int mode, int N, int Hi, int Wi, int C, const void* p0, int ld0, const float* wp, const void* p1, int ld1,
                      void* out, int ldout, float* slab, int* rows_out, hipStream_t st, const float* pscale = nullptr,
                      const float* pshift = nullptr, int prelu = 0, const DwBnStats* bnstats = nullptr);
int dw_tile_s2_dgrad_rows(int dtype, int C, int N, int Hi, int Wi);

}  // namespace dc
