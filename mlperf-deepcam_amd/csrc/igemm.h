// Shared by the implicit-GEMM kernels (igemm.hip: 128 x 128 tile, 4 waves; igemm256.hip: 256 x 256 tile, 8 waves).
#pragma once
#include "conv_geom.h"

namespace dc {

// K stride (elements) of a packed weight row: rows of 64 and more elements start on a 128-byte boundary (728 -> 768), so that a 128-byte K piece of a
// row is ONE L2 line (scripts/fill_bench.hip: the LDS-DMA fill of a tile reads 15.3 TB/s from line-aligned rows, 12.6 from rows on 64-byte
// boundaries, 10.9 from rows on 16-byte boundaries); shorter rows keep 64-byte boundaries.  The pad is never read.  DC_WLD_ALIGN=32: the stride up to round 4.
#ifndef DC_WLD_ALIGN
#define DC_WLD_ALIGN 64
#endif
__host__ __device__ inline int weight_ld(int c) {
  const int a = c < 64 ? 32 : DC_WLD_ALIGN;
  return (c + a - 1) / a * a;
}


// Optional epilogue of a data-gradient launch whose output is the gradient w.r.t. a BatchNorm(+ReLU) output: the BatchNorm's
// backward sums (sum g, sum g * xhat; g = the stored gradient masked by the ReLU, recomputed as y * mscale + mshift > 0) are taken
// from the values on their way out, into the statistics slab, instead of a separate pass over dx and y (bn.hip, colred MODE 1).
struct BnBwdEpi {
  const void* y;        // BatchNorm input (raw conv output), same shape and dtype as the produced gradient; null: off
  int ldy;
  const float* mean;
  const float* invstd;
  const float* mscale;  // forward scale / shift (only read when relu)
  const float* mshift;
  int relu;
};

struct IgemmParams {
  const void* x;
  const void* w;
  void* y;
  const float* bias;
  float* slab;
  GatherGeom g;
  int N, ldx, ldy;
  int ldw;     // K stride of a packed weight row: weight_ld(Cin)
  const void* w_kn;   // the layer's OTHER packing, [k][n] rows of weight_ld(Cout) elements (pointwise layers: wb for the forward pass, wf for the
  int ldw_kn;         // data gradient); null when the caller passes one image only.  The 224 x 384 kernel (igemm224.hip) takes its weights from it.
  int M;       // END of this launch's pixel range (per phase); N*Qh*Qw when the launch covers the layer
  int m_beg;   // first pixel of this launch's range (a multiple of 256): the mixed plan of run_gather cuts a layer into a
               // 256-tile launch over [0, m_beg') and a 128-tile launch over [m_beg', M)
  int mtiles;  // 128-pixel tiles per phase of the WHOLE layer (= rows per phase of the statistics slab)
  int accumulate;
  const void* zero_page;   // 256 zero bytes in device memory (set by the 256-tile launcher)
  int phase_fast;          // 256-tile kernel: consecutive tiles walk the sub-pixel phases of one pixel tile (set by its launcher)
  // 256-tile kernel, grouped launch (dc_conv_fwd_dilated_group): `ngroup` "same" dilated 3x3 convolutions of ONE input share the
  // launch.  g holds the unit-dilation tap table (dy, dx in {-1,0,1}); member b gathers at dy*gdil[b], dx*gdil[b], multiplies
  // by gw[b] and writes gy[b] / gslab[b].  Member 0 is (w, y, slab) above with gdil[0].  ngroup <= 1: plain launch.
  static constexpr int MAXGROUP = 4;
  BnBwdEpi bst;            // 128- and 256-tile kernels (LDS epilogue): slab receives BatchNorm-backward sums instead of (sum, sum of squares)
  int ngroup;
  int gdil[MAXGROUP];
  const void* gw[MAXGROUP - 1];
  void* gy[MAXGROUP - 1];
  float* gslab[MAXGROUP - 1];
  // 256-tile kernel, split-K launch (few pixels, long K: the atrous ASPP branches at local batch 2 are 81 tiles of 576 K steps): `ksplit` > 1
  // workgroups share a tile, each walks a contiguous range of the K loop's channel-chunk groups and leaves its fp32 accumulators in
  // kslab[split][tile][256 px][256 ch]; igemm256_splitk_fold sums the splits in a fixed order, stores y and takes the BatchNorm sums.
  int ksplit;
  float* kslab;
  // 128-tile kernel: a 1 x 1 stride-2 data gradient written first-hand (the shortcut convs of the entry flow): only phase (0, 0) has a tap, and a
  // launch over that phase alone also stores the zeros of the other three beside its values -- contiguous output rows instead of every other
  // 128-byte line of every other image row (which wrote 56 MB in 165 us)
  int zfill;
};
// OUT32: the epilogue stores fp32 regardless of T (used by the classifier head, whose logits must not be rounded to bf16)

// 256 x 256 tile kernel (bf16 only).  Returns 0 after launching.
int launch_igemm256(const IgemmParams& p, hipStream_t st);
void igemm256_set_phase_fast(int v);
int igemm256_phase_fast_enabled();
// igemm256p.hip: persistent form (one workgroup per CU walks its tiles; the ring never drains between tiles)
bool igemm256p_eligible(const IgemmParams& p);
int launch_igemm256p(const IgemmParams& p, int workgroups, hipStream_t st);
// workgroups the 256-tile kernel would launch for this problem
inline long igemm256_tiles(const IgemmParams& p) {
  return (long)((p.g.Cout + 255) / 256) * ((p.M - p.m_beg + 255) / 256) * p.g.os * p.g.os * (p.ngroup > 1 ? p.ngroup : 1);
}
// split-K plan of a 256-tile launch: splits (1 = none) and the bytes of the partial-sum slab; launch + fold of a planned launch
int igemm256_splitk_plan(const IgemmParams& p, size_t* slab_bytes);
int launch_igemm256_splitk(const IgemmParams& p, int splits, void* ws, hipStream_t st);
void igemm256_set_splitk(int v);

// igemm384.hip: pointwise layers on a (32*npb) x 384 tile, one wave per SIMD (npb = 8 or 4 pixel blocks per wave)
bool pw384_eligible(const IgemmParams& p);
int launch_pw384(const IgemmParams& p, int npb, hipStream_t st);
inline long pw384_tiles(const IgemmParams& p, int npb) { return (long)((p.g.Cout + 383) / 384) * ((p.M + 32 * npb - 1) / (32 * npb)); }

// igemm224.hip: pointwise layers on a 224 x 384 tile with 32-deep weight stages from the [k][n] packing (p.w_kn) and three-deep operand rings
bool pw224_eligible(const IgemmParams& p);
int launch_pw224(const IgemmParams& p, int tn, hipStream_t st);      // tn: channels per tile, 384 or 192
inline long pw224_tiles(const IgemmParams& p, int tn = 384) { return (long)((p.g.Cout + tn - 1) / tn) * ((p.M + 223) / 224); }

// igemm192.hip: pointwise layers with few pixels (local batch 2) on a 128 x 192 tile, eight waves, four 64-deep ring stages
long pw192_tiles(const IgemmParams& p);
int launch_pw192(const IgemmParams& p, hipStream_t st);

// thinconv.hip: the thin 3x3 stem convolutions (forward and data gradient) without LDS staging of the pixel operand
bool thin_fwd_eligible(const GatherGeom& g, int dtype, int bias, int accumulate, int out32);
void thin_set_tile(int v);
int launch_thin_fwd(const GatherGeom& g, int N, const void* in, int ldin, const void* w, int ldw, void* out, int ldout, float* slab,
                    int slab_rows, hipStream_t st);

}  // namespace dc
