// Shared by the weight-gradient kernels (wgrad.hip: 128 x 128 tile, 4 waves; wgrad384.hip: 256 x 384 tile, 8 waves).
#pragma once
#include "conv_geom.h"

namespace dc {

struct WgradParams {
  const void* x;   // gathered operand (forward input), channels -> ci
  const void* dy;  // output-side operand, channels -> co
  float* slab;     // [splits][taps][Co][Ci]
  GatherGeom g;    // forward geometry: Cin = ci extent, Cout = co extent
  int N, ldx, lddy;
  int M;           // pixels per phase
  int splits;
  int chunk;       // pixels per split (multiple of BP)
  // register-staged kernel (wgrad_kernel<bf16>) only: x is the raw output y of a convolution and the real operand is act(y * xscale + xshift),
  // the BatchNorm(+ReLU) output that was never stored (dc_head_bwd_bnin); formed in registers on the way to LDS
  const float* xscale = nullptr;
  const float* xshift = nullptr;
  int xrelu = 0;
};

// dc_conv_wgrad on a lazily applied BatchNorm output: the gathered operand is act(y * xscale + xshift) (stem_head.hip: the head's weight gradient)
int conv_wgrad_bnin(const dc_conv_desc* d, int N, int Hi, int Wi, const void* y, int ldy, const float* xscale, const float* xshift, int xrelu,
                    const void* dy, int lddy, void* workspace, size_t workspace_bytes, float* grad_w, void* stream);

// wgrad384.hip: pointwise, stride-1 3 x 3 and transposed layers on a 256 (co) x 384 (ci) tile; up to WG384_MAXL layers of one geometry per launch
// (group > 1: xs / dys / outs hold `group` pointers, entry 0 repeats p.x / p.dy / p.slab); outs[l] is layer l's [splits][tap][Co][Ci] slab -- the
// small kernel's layout, so wgrad_reduce_kernel and dc_fold_slabs serve both -- or, with splits == 1, its gradient tensor
constexpr int WG384_MAXL = 16;
bool wgrad384_eligible(const GatherGeom& g, int ldx, int lddy, long M);
bool wgrad384_is_tconv(const GatherGeom& g);
void wgrad384_plan(const GatherGeom& g, long M, int* splits, int* chunk, int group = 1);
void wgrad384_set_slots(int n);
void wgrad384_set_min_stages(int n);
int launch_wgrad384(const WgradParams& p, hipStream_t st, int group = 1, const void* const* xs = nullptr, const void* const* dys = nullptr,
                    float* const* outs = nullptr);

// thinconv.hip: all nine taps of the 16->32 (stride 2) and 32->64 stem convolutions in one pass over x and dy
bool thin_wgrad_eligible(const dc_conv_desc& d, int Hi, int Wi);
int thin_wgrad_splits(const dc_conv_desc& d, int N, int Hi, int Wi);
int launch_thin_wgrad(const dc_conv_desc& d, int N, int Hi, int Wi, const void* x, int ldx, const void* dy, int lddy, float* slab, hipStream_t st);

}  // namespace dc
