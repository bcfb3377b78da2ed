/* deepcam_hip.h -- C ABI of libdeepcam_hip.so: the MI355X (gfx950) kernels of the DeepCAM train step.
 *
 * The reference (azrael417/mlperf-deepcam) has no native code and no FFI: its hot path is the chain of
 * PyTorch/cuDNN/apex operator calls made by src/deepCam/train_hdf5_ddp.py:345-371.  Each entry point below
 * replaces one such operator call site (cited per function, paths relative to src/deepCam/), with plain
 * pointers and sizes only.  The Python host (mlperf-deepcam_amd/) binds them with ctypes; INTEGRATION.md
 * shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - All pointers are DEVICE pointers unless a parameter says "host".  `stream` is a hipStream_t.
 *   - Activations are channels-last: element (n,h,w,c) of a view lives at ptr[((n*H+h)*W+w)*ld + c];
 *     `ld` >= C lets a view be a channel slice of a wider buffer (this is how torch.cat disappears).
 *     ptr and ld*sizeof(elem) must be 16-byte aligned; C must be a multiple of 8 (true for every layer).
 *   - `dtype` selects the storage type of activations, activation gradients and packed GEMM weights:
 *     DC_BF16 (MFMA v_mfma_f32_16x16x32_bf16, fp32 accumulate) or DC_F32 (v_mfma_f32_16x16x4_f32, exact fp32).
 *     Master weights, gradients of weights, BN statistics, the loss and the optimizer state are always fp32.
 *   - Every function returns 0 on success.  On failure it returns non-zero and dc_last_error() describes it
 *     (bad shape/alignment => nothing was launched).  No function synchronises, allocates or frees.
 */
#ifndef DEEPCAM_HIP_H_
#define DEEPCAM_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { DC_F32 = 0, DC_BF16 = 1 };
enum { DC_ADAM = 0, DC_ADAMW = 1, DC_LAMB = 2 };

const char* dc_last_error(void);
int dc_version(void);
/* Tuning switches (A/B measurements inside one process).  "igemm_mode": 2 = LDS-DMA, 3-stage ring, counted waits (default);
 * 1 = LDS-DMA, 2 stages; 0 = register staging.  "igemm256": 0 = 128 x 128 tiles only, 1 = planner picks per layer (default),
 * 2 = 256 x 256 eight-wave kernel wherever eligible.  "wgrad_target_blocks": workgroups aimed at by the split-K planner.
 * "dw_tile" 0/1, "dw_wgrad_tpb", "bn_cgw", "bn_rows": depthwise / BatchNorm kernel variants (see the .hip files). */
int dc_set_option(const char* name, int value);
/* Every dc_set_option switch back to its default (the library's own table; also what a freshly loaded library holds). */
int dc_reset_options(void);

/* Streams for the host runtime.  The reference leaves stream management to PyTorch/apex (its DDP overlaps the all-reduce
 * on a side stream, train_hdf5_ddp.py:227,363); here the weight-gradient kernels run beside the backward chain on a stream
 * of the LOWEST priority (level < 0; 0 = default, > 0 = highest), created non-blocking.  The caller owns the stream. */
int dc_stream_create(int level, void** stream);
int dc_stream_destroy(void* stream);
int dc_stream_priority_range(int* least, int* greatest);
/* `to_stream` waits for everything enqueued on `from_stream` so far (the reference's loop relies on PyTorch's stream semantics for the
 * same order, train_hdf5_ddp.py:358-364: backward, all-reduce, optimizer). */
int dc_stream_fence(void* from_stream, void* to_stream);
int dc_memset_async(void* dst, int byte, size_t bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The launch list of a train step as a C object.  The reference's step is the Python loop body train_hdf5_ddp.py:345-371, a chain of
 * operator calls; here it is ~700 calls of THIS header per step.  A program records such a chain once (entry point name + argument
 * words in issue order, stream fences included) and dc_program_run issues it again from C, with no interpreter between two launches.
 *   words:  one 8-byte word per argument in declaration order -- ints / longs / size_t sign-extended, float as the bit pattern of the
 *           double of equal value, pointers as they are.  Pointers to HOST objects (dc_conv_desc, dc_fold_entry tables, pointer arrays)
 *           are recorded as addresses: the caller keeps those objects alive and unchanged while the program lives.
 *   slots:  NULL, or per argument -1 (use the word) or a slot index whose value dc_program_bind supplies before a run (a batch pointer).
 * Every int-returning entry point of this header except dc_program_* itself can be recorded.
 * ------------------------------------------------------------------------------------------------ */
int dc_program_create(void** program);
int dc_program_destroy(void* program);
int dc_program_append(void* program, const char* entry_point, int nargs, const long long* words, const int* slots);
int dc_program_bind(void* program, int slot, long long word);
int dc_program_len(void* program);
const char* dc_program_op_name(void* program, int index);
int dc_program_run(void* program, int* failed_op);   /* first failing call's code; *failed_op = its index or -1 */

/* ------------------------------------------------------------------------------------------------
 * Dense convolution family: nn.Conv2d (k=1 or 3, stride 1|2, dilation, zero padding) and
 * nn.ConvTranspose2d(k=3, stride=2, padding=1, output_padding=1), all lowered to one gather-form
 * implicit GEMM on MFMA.  Reference call sites: architecture/deeplab_xception.py:60,65 (pointwise),
 * :74 (skip), :149 (conv2), :291-292 (ASPP, atrous), :426,430,434 (1x1), :352-374 (decoder convs and
 * transposed convs).  Backward entry points replace what autograd dispatches at train_hdf5_ddp.py:363.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  int dtype;      /* DC_F32 | DC_BF16 */
  int k;          /* 1 or 3 */
  int stride;     /* 1 or 2 */
  int pad;
  int dil;
  int transposed; /* 1: ConvTranspose2d k3 s2 p1 op1 (stride/pad/dil fields ignored) */
  int cin;
  int cout;
} dc_conv_desc;


/* Output extent of the forward op for an Hi x Wi input. */
int dc_conv_out_hw(const dc_conv_desc* d, int Hi, int Wi, int* Ho, int* Wo);

/* Pack the fp32 master weight (PyTorch layout: [cout][cin][k][k], or [cin][cout][k][k] when transposed)
 * into the two GEMM operand layouts, both in `dtype`:  wf[tap][cout][cin'] (forward) and
 * wb[tap][cin][cout'] (data gradient), where the K extent of a row (cin', cout') is rounded up to 64 elements (32 below 64
 * channels) so that every bf16 row starts on a 128-byte boundary: a 128-byte K piece of a row is then ONE L2 line for the
 * LDS-DMA loaders (728 -> 768; 64-byte boundaries -- 736 -- filled 18 % slower, profiles/r05_fill_bench_align.txt).
 * dc_conv_packed_elems gives the element counts to allocate (and thereby the strides: wf_elems / (taps * cout)).  Either
 * output may be NULL. */
int dc_conv_packed_elems(const dc_conv_desc* d, size_t* wf_elems, size_t* wb_elems);
int dc_conv_pack_weights(const dc_conv_desc* d, const float* master, void* wf, void* wb, void* stream);

/* All layers in one launch.  table_dev: device array of `nentries` records {const float* master; void* wf; void* wb; int cin, cout,
 * taps, kind;} (40 bytes, natural alignment); kind 0 = conv, 1 = transposed conv (both as dc_conv_pack_weights), 2 = depthwise
 * (as dc_dwconv_pack_weights: cout = C, wf = fp32 [9][C]).  Dense entries are packed in `dtype`. */
int dc_pack_all(int dtype, const void* table_dev, int nentries, void* stream);

/* Rows of the per-tile BatchNorm partial-statistics slab that dc_conv_fwd writes when stat_slab != NULL:
 * slab is float[2][rows][cout] (sum, sum of squares of the STORED outputs). */
int dc_conv_stat_rows(const dc_conv_desc* d, int N, int Hi, int Wi);

/* y = conv(x, w) [+ bias];  accumulate != 0 adds into the existing y instead of overwriting. */
int dc_conv_fwd(const dc_conv_desc* d, int N, int Hi, int Wi, const void* x, int ldx, const void* wf,
                const float* bias, void* y, int ldy, float* stat_slab, int accumulate, void* stream);

/* dc_conv_fwd / dc_conv_dgrad given BOTH packed images of the layer (replaces the same call sites: F.conv2d of the `pointwise` convs at
 * architecture/deeplab_xception.py:60,65 and their autograd data gradient).  `wb` here / `wf` below is the image the plain entry point does not
 * take; its rows are [k][n] for the GEMM in question, so a pointwise layer's weight stages can be 32-deep slabs of whole L2 lines
 * (csrc/igemm224.hip: 224 x 384 tiles, three-deep operand rings).  Layers that kernel does not serve run exactly as through the plain entry
 * points.  Outputs are bit-identical to the plain calls; the BatchNorm partial sums come in another summation order (one slab row per
 * 224-pixel tile; with slab_rows = 0 the slab keeps dc_conv_stat_rows rows, the other rows written as zeros). */
int dc_conv_fwd_kn(const dc_conv_desc* d, int N, int Hi, int Wi, const void* x, int ldx, const void* wf, const void* wb,
                   const float* bias, void* y, int ldy, float* stat_slab, int slab_rows, int accumulate, void* stream);
/* Rows of the slab dc_conv_fwd_kn writes when slab_rows carries this value: one per 224-pixel tile where igemm224.hip serves the layer
 * (fewer, and none of them zeros), dc_conv_stat_rows elsewhere.  slab_rows = 0: the dc_conv_stat_rows layout whatever kernel runs.  The call
 * fails if slab_rows is not what the launch writes (the planner's answer changed between the query and the call). */
int dc_conv_stat_rows_kn(const dc_conv_desc* d, int N, int Hi, int Wi);
/* 1 where dc_conv_fwd_kn takes slab_rows = -1 for this layer (the pointwise tile kernels igemm224.hip and igemm192.hip): stat_slab is then a SUM ROW -- double[2][Cout] (sum, sum of squares), zeroed by
 * the caller -- that every tile of the launch adds its channel sums to with fp64 atomics, instead of one fp32 row per tile.  (The addends
 * are the fp32 sums a row slab would hold; their fp64 sum is exact, hence independent of the arrival order and equal to what dc_bn_finalize
 * computes from the rows.)  dc_bn_finalize, dc_bn_apply_fin and dc_dwconv_fwd_fin take such a slab with rows = -1: two numbers per channel to
 * read, so the consumer of the coefficients runs the finalize itself at any tensor size and the finalize launch leaves the dependent chain
 * (nn.BatchNorm2d in training mode behind every pointwise conv of the middle flow, deeplab_xception.py:62-66,104-119). */
int dc_conv_sum_row_kn(const dc_conv_desc* d, int N, int Hi, int Wi);
int dc_conv_dgrad_kn(const dc_conv_desc* d, int N, int Hi, int Wi, const void* dy, int lddy, const void* wb, const void* wf,
                     void* dx, int lddx, int accumulate, void* stream);

/* Same as dc_conv_fwd, but the result is stored as fp32 (y: float NHWC, ldy in floats) whatever d->dtype is:
 * used where the consumer must not see bf16-rounded sums (the classifier head's logits). */
int dc_conv_fwd_f32out(const dc_conv_desc* d, int N, int Hi, int Wi, const void* x, int ldx, const void* wf,
                       float* y, int ldy, void* stream);

/* The ASPP head (deeplab_xception.py:282-302; aspp2..aspp4 called at :445-447) runs three 3x3 convolutions of dilation 6 / 12 / 18
 * (padding == dilation, stride 1) over ONE input.  This entry point computes `count` (<= 4) such convolutions in one launch:
 * member b uses dilation dils[b] (host array), packed weights wfs[b], output view ys[b] (all with ldy) and statistics slab
 * stat_slabs[b] (NULL array: no statistics).  d->dil / d->pad are ignored, d->k must be 3 and d->stride 1.  Results are
 * bit-identical to `count` dc_conv_fwd calls (which is what runs when the fused kernel does not serve the dtype). */
int dc_conv_fwd_dilated_group(const dc_conv_desc* d, int N, int Hi, int Wi, int count, const int* dils, const void* x, int ldx,
                              const void* const* wfs, void* const* ys, int ldy, float* const* stat_slabs, void* stream);
/* The same launch with a caller-owned workspace: where dc_conv_fwd_dilated_group_workspace(...) > 0 (the launch would leave more than half of
 * the chip idle under a long K loop -- local batch 2: 81 tiles of 576 K steps) and `ws` holds that many bytes, every tile's K loop is split over
 * several workgroups (fp32 partial tiles in ws) and a second kernel sums the splits in a fixed order, stores the outputs and takes the
 * BatchNorm sums: another order of the K sum than the unsplit launch (not its bits), deterministic.  SURVEY section 2.3 K3: "needs split-K
 * to fill 256 CUs".  ws NULL or too small: the unsplit launch. */
size_t dc_conv_fwd_dilated_group_workspace(const dc_conv_desc* d, int N, int Hi, int Wi, int count, const int* dils);
int dc_conv_fwd_dilated_group_ws(const dc_conv_desc* d, int N, int Hi, int Wi, int count, const int* dils, const void* x, int ldx,
                                 const void* const* wfs, void* const* ys, int ldy, float* const* stat_slabs, void* ws, size_t ws_bytes,
                                 void* stream);

/* dx = conv_backward_data(dy, w).  Hi, Wi are the FORWARD input extents (= extents of dx). */
int dc_conv_dgrad(const dc_conv_desc* d, int N, int Hi, int Wi, const void* dy, int lddy, const void* wb,
                  void* dx, int lddx, int accumulate, void* stream);
/* The same data gradient when dx is the gradient w.r.t. a BatchNorm(+ReLU) output act(bn(y)) that feeds only this conv
 * (deeplab_xception.py:361-374 and the other conv -> BatchNorm -> ReLU -> conv chains): the BatchNorm's backward sums (sum g,
 * sum g*xhat; g = the stored dx masked by y*mscale + mshift > 0 when relu) are taken in the epilogue into slab[2][rows][C],
 * rows = dc_conv_dgrad_bnstats_rows (0: this layer is not served), so dc_bn_bwd_reduce's pass over dx and y is not needed.
 * dx is written, never accumulated. */
int dc_conv_dgrad_bnstats_rows(const dc_conv_desc* d, int N, int Hi, int Wi);
int dc_conv_dgrad_bnstats(const dc_conv_desc* d, int N, int Hi, int Wi, const void* dy, int lddy, const void* wb, void* dx,
                          int lddx, const void* y, int ldy, const float* mean, const float* invstd, const float* mscale,
                          const float* mshift, int relu, float* slab, void* stream);

/* grad_w (fp32, PyTorch master layout) = conv_backward_weight(x, dy).  Split over the pixel axis into fp32
 * partial slabs in `workspace` (dc_conv_wgrad_workspace bytes), then reduced deterministically. */
size_t dc_conv_wgrad_workspace(const dc_conv_desc* d, int N, int Hi, int Wi);
int dc_conv_wgrad(const dc_conv_desc* d, int N, int Hi, int Wi, const void* x, int ldx, const void* dy,
                  int lddy, void* workspace, size_t workspace_bytes, float* grad_w, void* stream);

/* The same for `count` layers of ONE geometry in one launch (count <= 16 where the 256 x 384 kernel serves the layer -- pointwise layers,
 * stride-1 3 x 3 layers, transposed convolutions: csrc/wgrad384.hip --, <= 4 on the 256 x 256 kernel): xs / dys / grad_ws are host arrays of
 * `count` device pointers (same ldx / lddy).  Replaces the per-layer conv_backward_weight calls autograd makes for the pointwise
 * convs of the middle-flow Blocks (deeplab_xception.py:69-122, 211-226): the layers share the launch, so each is cut into fewer,
 * longer pixel splits.  Results are bit-identical run to run (fixed-order reduction); layers the grouped kernel does not serve
 * are computed by `count` plain dc_conv_wgrad calls. */
size_t dc_conv_wgrad_group_workspace(const dc_conv_desc* d, int N, int Hi, int Wi, int count);
int dc_conv_wgrad_group(const dc_conv_desc* d, int N, int Hi, int Wi, int count, const void* const* xs, int ldx,
                        const void* const* dys, int lddy, void* workspace, size_t workspace_bytes,
                        float* const* grad_ws, void* stream);

/* Slab-only form of the two calls above: the split-K partial sums of `count` layers of one geometry (count > 1: the grouped launch
 * must serve the layer, dc_conv_wgrad_plan fails otherwise) are left in slabs[l] ([splits][taps][Cout][Cin] floats each,
 * dc_conv_wgrad_plan's slab_bytes) and NOT reduced: dc_fold_slabs adds the slabs of many layers in one launch.  `splits` must be the
 * value dc_conv_wgrad_plan returned (the call fails if tuning options changed the plan in between).  Same reference call site:
 * autograd's conv_backward_weight at train_hdf5_ddp.py:363. */
int dc_conv_wgrad_plan(const dc_conv_desc* d, int N, int Hi, int Wi, int count, int* splits, size_t* slab_bytes);
int dc_conv_wgrad_partial(const dc_conv_desc* d, int N, int Hi, int Wi, int count, const void* const* xs, int ldx,
                          const void* const* dys, int lddy, float* const* slabs, int splits, void* stream);

/* grad (fp32, PyTorch master layout) = fixed-order sum of the partial slabs of n layers, ONE launch per 24 entries (host array; the
 * table travels in the kernel arguments).  kind DC_FOLD_CONV: slab [splits][taps][co][ci] -> grad [co][ci][taps] (Conv2d.weight);
 * DC_FOLD_CONVT: -> grad [ci][co][taps] (ConvTranspose2d.weight); DC_FOLD_DW: slab [splits = rows][9][co = C] -> grad [C][1][3][3]
 * (taps = 9, ci = 1; the rows dc_dwconv_dgrad_bnstats_wgrad / dc_dwconv_dgrad_wgrad leave).  Bit-identical to the per-layer
 * reductions inside dc_conv_wgrad / dc_dwconv_wgrad_reduce whatever the grouping.  Replaces the accumulation step of autograd's
 * conv_backward_weight for all layers of an Xception block at once (train_hdf5_ddp.py:363). */
enum { DC_FOLD_CONV = 0, DC_FOLD_CONVT = 1, DC_FOLD_DW = 2 };
typedef struct dc_fold_entry {
  const float* slab;
  float* grad;
  int kind, splits, taps, co, ci;
} dc_fold_entry;
int dc_fold_slabs(const dc_fold_entry* entries, int n, void* stream);

/* grad_bias[c] = sum over M rows of dy (the one biased conv, deeplab_xception.py:366). */
int dc_colsum(int dtype, long M, int C, const void* dy, int lddy, float* out, void* workspace, void* stream);
size_t dc_colsum_workspace(long M, int C);

/* ------------------------------------------------------------------------------------------------
 * Depthwise 3x3 ("SeparableConv2d_same.conv1" after fixed_padding): deeplab_xception.py:45-51,58-59,63-64.
 * Zero padding of `dil` on every side is implicit.  Forward and data-gradient take the weights repacked as
 * fp32 [9][C] (`wp`, from dc_dwconv_pack_weights); the weight gradient comes back in the master layout [C][1][3][3].
 * ------------------------------------------------------------------------------------------------ */
int dc_dwconv_pack_weights(int C, const float* master, float* packed, void* stream);
/* pscale/pshift (both or neither; fp32[C]) fuse the PRECEDING BatchNorm (+ReLU when prelu != 0) into the load: x is then the
 * raw conv output and every in-bounds element is read as act(x*pscale[c] + pshift[c]); padding stays zero. */
int dc_dwconv_fwd(int dtype, int C, int stride, int dil, int N, int Hi, int Wi, const void* x, int ldx,
                  const float* wp, void* y, int ldy, const float* pscale, const float* pshift, int prelu, void* stream);
/* dc_dwconv_fwd through a training-mode BatchNorm(+ReLU) whose dc_bn_finalize it runs itself (SeparableConv2d_same after bn + relu inside a
 * Block, deeplab_xception.py:83-98, at small batch): x is the raw output of the convolution in front of that BatchNorm, slab[2][rows][C]
 * its partial sums with rows <= dc_bn_bwd_apply_fin_max_rows().  Every workgroup sums the slab for its own channels while its halo tile
 * travels (order and bits of dc_bn_finalize); the workgroups of pixel tile 0 store scale / shift / save_mean / save_invstd and update
 * the running statistics.  The slab is read with 16-byte loads: 16-byte aligned (as are dc_bn_apply_fin's and dc_bn_bwd_apply_fin's).
 * dc_dwconv_fwd_fin_ok: 1 when the shape is served (stride 1, dilation 1 or 2, the tiled kernel). */
int dc_dwconv_fwd_fin_ok(int dtype, int C, int stride, int dil, int N, int Hi, int Wi);
int dc_dwconv_fwd_fin(int dtype, int C, int stride, int dil, int N, int Hi, int Wi, const void* x, int ldx, const float* w, void* y, int ldy,
                      int prelu, long count, const float* slab, int rows, const float* gamma, const float* beta, float* running_mean,
                      float* running_var, int64_t* num_batches_tracked, float momentum, float eps, float* scale, float* shift,
                      float* save_mean, float* save_invstd, void* stream);
/* dx = dw_backward_data(dy) [+ addend]  (addend: same shape as dx, e.g. the residual branch's gradient) */
int dc_dwconv_dgrad(int dtype, int C, int stride, int dil, int N, int Hi, int Wi, const void* dy, int lddy,
                    const float* wp, const void* addend, int ldadd, void* dx, int lddx, void* stream);
/* Data gradient that also takes the backward statistics of the BatchNorm(+ReLU) whose (never materialised) output fed this
 * depthwise conv: dx as dc_dwconv_dgrad (no addend), and slab[2][rows][C] = per-tile partial sums of g and g*xhat with
 * g = dx masked by (ybn*mscale + mshift > 0) when relu, xhat = (ybn - save_mean)*save_invstd.  rows =
 * dc_dwconv_dgrad_bnstats_rows(...) (0: shape not served, use dc_dwconv_dgrad + dc_bn_bwd_reduce); finish with dc_bn_bwd_finalize.
 * Replaces one full read of dx and ybn per such BatchNorm (autograd of nn.BatchNorm2d after deeplab_xception.py:65). */
int dc_dwconv_dgrad_bnstats_rows(int dtype, int C, int stride, int dil, int N, int Hi, int Wi);
int dc_dwconv_dgrad_bnstats(int dtype, int C, int stride, int dil, int N, int Hi, int Wi, const void* dy, int lddy,
                            const float* wp, void* dx, int lddx, const void* ybn, int ldybn, const float* save_mean,
                            const float* save_invstd, const float* mscale, const float* mshift, int relu, float* slab, void* stream);
/* dc_dwconv_dgrad_bnstats that ALSO takes this depthwise layer's weight gradient (reference: SeparableConv2d_same.conv1,
 * deeplab_xception.py:54-66, whose .grad autograd fills at train_hdf5_ddp.py:363) from the same dy window: the layer's forward input
 * is the never-stored act(ybn*mscale + mshift), recomputed from the ybn the statistics read anyway.  wslab: dc_dwconv_dgrad_wgrad_rows
 * rows (0: shape not served -- stride 2, large images) of [9][C] floats; dc_dwconv_wgrad_reduce adds the rows into grad_w (master
 * layout [C][1][3][3]) in a fixed order and may run on another stream once the producing call has been ordered before it. */
int dc_dwconv_dgrad_wgrad_rows(int dtype, int C, int stride, int dil, int N, int Hi, int Wi);
int dc_dwconv_dgrad_bnstats_wgrad(int dtype, int C, int stride, int dil, int N, int Hi, int Wi, const void* dy, int lddy,
                                  const float* w_packed, void* dx, int lddx, const void* ybn, int ldybn, const float* save_mean,
                                  const float* save_invstd, const float* mscale, const float* mshift, int relu, float* slab,
                                  float* wslab, void* stream);
/* The same when this layer is not the only reader of that BatchNorm's output: dx = conv^T(dy) + addend (what the other readers left; may alias
 * dx), the sums are those of the complete gradient -- the caller makes this layer the LAST writer (stride 1). */
int dc_dwconv_dgrad_bnstats_wgrad_add(int dtype, int C, int stride, int dil, int N, int Hi, int Wi, const void* dy, int lddy,
                                      const float* w, const void* addend, int ldadd, void* dx, int lddx, const void* ybn, int ldybn,
                                      const float* save_mean, const float* save_invstd, const float* mscale, const float* mshift, int relu,
                                      float* slab, float* wslab, void* stream);
/* The same fusion where the layer's forward input is a stored tensor x, or act(x*pscale + pshift) of one: data gradient (written, or
 * added onto `addend`: the first separable conv of an Xception block, whose input gradient joins the shortcut's) plus weight-gradient
 * rows; no BatchNorm sums. */
int dc_dwconv_dgrad_wgrad(int dtype, int C, int stride, int dil, int N, int Hi, int Wi, const void* dy, int lddy,
                          const float* w_packed, const void* addend, int ldadd, void* dx, int lddx, const void* x, int ldx,
                          const float* pscale, const float* pshift, int prelu, float* wslab, void* stream);
/* dc_dwconv_dgrad_wgrad where the stored input x is itself relu(bn(ybn) + residual), the output of the previous Xception block
 * (deeplab_xception.py:99-122: `x += skip`, then the next block's in-place ReLU), and this data gradient is the LAST contribution to d(x):
 * that BatchNorm's backward sums (sum g, sum g*xhat; g = the stored dx masked by x > 0 when relu) are taken on the way out into
 * slab[2][rows][C], so dc_bn_bwd_reduce's pass over dx, ybn and x is not needed.  rows = dc_dwconv_dgrad_wgrad_bnres_rows (also the
 * rows of wslab; 0: not served -- use dc_dwconv_dgrad_wgrad and dc_bn_bwd_reduce). */
int dc_dwconv_dgrad_wgrad_bnres_rows(int dtype, int C, int stride, int dil, int N, int Hi, int Wi);
int dc_dwconv_dgrad_wgrad_bnres(int dtype, int C, int stride, int dil, int N, int Hi, int Wi, const void* dy, int lddy,
                                const float* w_packed, const void* addend, int ldadd, void* dx, int lddx, const void* x, int ldx,
                                float* wslab, const void* ybn, int ldybn, const float* save_mean, const float* save_invstd, int relu,
                                float* slab, void* stream);
/* The two data gradients above that take BatchNorm sums, with the sums added to a SUM ROW (see dc_conv_sum_row_kn): `slab` is double[2][C]
 * (sum g, sum g*xhat), zeroed by the caller; every workgroup of the persistent kernel adds its two sums per channel with fp64 atomics (exact,
 * hence the bits of the row slab's column sums in any arrival order).  dc_bn_bwd_finalize and dc_bn_bwd_apply_fin take the row with rows = -1,
 * so the BatchNorm-backward apply runs the finalize itself at any tensor size.  dc_dwconv_dgrad_sum_row_ok: 1 where both are served. */
int dc_dwconv_dgrad_sum_row_ok(int dtype, int C, int stride, int dil, int N, int Hi, int Wi);
int dc_dwconv_dgrad_bnstats_wgrad_sum(int dtype, int C, int stride, int dil, int N, int Hi, int Wi, const void* dy, int lddy,
                                      const float* w_packed, void* dx, int lddx, const void* ybn, int ldybn, const float* save_mean,
                                      const float* save_invstd, const float* mscale, const float* mshift, int relu, float* slab,
                                      float* wslab, void* stream);
int dc_dwconv_dgrad_wgrad_bnres_sum(int dtype, int C, int stride, int dil, int N, int Hi, int Wi, const void* dy, int lddy,
                                    const float* w_packed, const void* addend, int ldadd, void* dx, int lddx, const void* x, int ldx,
                                    float* wslab, const void* ybn, int ldybn, const float* save_mean, const float* save_invstd, int relu,
                                    float* slab, void* stream);
int dc_dwconv_wgrad_reduce(int C, int rows, const float* wslab, float* grad_w, void* stream);
size_t dc_dwconv_wgrad_workspace(int C, int N, int Hi, int Wi, int stride);
int dc_dwconv_wgrad(int dtype, int C, int stride, int dil, int N, int Hi, int Wi, const void* x, int ldx,
                    const void* dy, int lddy, void* workspace, float* grad_w, const float* pscale, const float* pshift,
                    int prelu, void* stream);

/* ------------------------------------------------------------------------------------------------
 * BatchNorm2d (train and eval) fused with ReLU and the residual add: deeplab_xception.py:86,92,97,
 * 146-147,150,180-186,293-302,353-371,427,431,435 and the `x += skip` at :120.
 * Statistics are fp32 per-tile partials combined in fp64.
 * ------------------------------------------------------------------------------------------------ */
int dc_bn_stat_rows(long M);
int dc_bn_stats(int dtype, long M, int C, const void* x, int ldx, float* slab, void* stream);
/* slab[2][rows][C] -> batch mean / biased var -> scale = gamma*rsqrt(var+eps), shift = beta - mean*scale;
 * running_mean/var (momentum, unbiased var) and num_batches_tracked (int64) are updated when non-NULL.
 * The slab is CONSUMED: above 1024 rows it is folded in two stages and the first stage leaves its fp64 results in the slab itself
 * (same for dc_bn_bwd_finalize). */
int dc_bn_finalize(int C, long count, float* slab, int rows, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                   float eps, float* scale, float* shift, float* save_mean, float* save_invstd, void* stream);
/* eval mode: scale/shift from the running statistics */
int dc_bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* running_mean,
                      const float* running_var, float eps, float* scale, float* shift, void* stream);
/* out = act(y*scale + shift [+ residual]),  act = ReLU when relu != 0 */
int dc_bn_apply(int dtype, long M, int C, const void* y, int ldy, const float* scale, const float* shift,
                const void* residual, int ldr, int relu, void* out, int ldo, void* stream);
/* dc_bn_finalize + dc_bn_apply in one launch (nn.BatchNorm2d in training mode, deeplab_xception.py:104-119 block outputs at small batch):
 * for a SHORT slab, rows <= dc_bn_bwd_apply_fin_max_rows(), every block of the apply kernel sums the slab for its own channels -- the order
 * and the bits of dc_bn_finalize -- and block row 0 stores scale / shift / save_mean / save_invstd and updates the running statistics.
 * The slab is read, not consumed. */
int dc_bn_apply_fin(int dtype, long M, int C, long count, const void* y, int ldy, const float* slab, int rows, const float* gamma,
                    const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                    float* scale, float* shift, float* save_mean, float* save_invstd, const void* residual, int ldr, int relu, void* out,
                    int ldo, void* stream);
/* backward, step 1: g = dout * (out > 0 if relu), partial sums of g and g*xhat -> slab[2][rows][C].
 * relu == 2: the activation was fused into its consumer and never stored; the mask is recomputed as
 * y*mscale[c] + mshift[c] > 0 (the forward scale/shift vectors), `out` is ignored. */
int dc_bn_bwd_reduce(int dtype, long M, int C, const void* dout, int lddo, const void* y, int ldy,
                     const void* out, int ldout, int relu, const float* save_mean, const float* save_invstd,
                     float* slab, const float* mscale, const float* mshift, void* stream);
/* dc_bn_bwd_reduce with the two sums added to a SUM ROW (see dc_conv_sum_row_kn): `slab` is double[2][C], zeroed by the caller; every block of the
 * pass adds its share with fp64 atomics (at most 2 048 per channel).  dc_bn_bwd_finalize / dc_bn_bwd_apply_fin take it with rows = -1: the
 * two-stage fold of a long slab and the finalize launch leave the chain. */
int dc_bn_bwd_reduce_sum(int dtype, long M, int C, const void* dout, int lddo, const void* y, int ldy,
                     const void* out, int ldout, int relu, const float* save_mean, const float* save_invstd,
                     float* slab, const float* mscale, const float* mshift, void* stream);
/* step 2: dgamma, dbeta (fp32, written to the gradient arena) */
int dc_bn_bwd_finalize(int C, float* slab, int rows, float* dgamma, float* dbeta, void* stream);
/* step 3: dy = gamma*invstd*(g - dbeta/count - xhat*dgamma/count);  g is also stored when g_out != NULL */
int dc_bn_bwd_apply(int dtype, long M, int C, long count, const void* dout, int lddo, const void* y, int ldy,
                    const void* out, int ldout, int relu, const float* gamma, const float* save_mean,
                    const float* save_invstd, const float* dgamma, const float* dbeta, void* dy, int lddy,
                    void* g_out, int ldg, const float* mscale, const float* mshift, void* stream);
/* dc_bn_bwd_finalize + dc_bn_bwd_apply in ONE launch for a short slab (rows <= dc_bn_bwd_apply_fin_max_rows(): the slabs the persistent
 * depthwise data-gradient kernels leave for the 728-channel layers have 42 rows): every block sums the slab rows of its own channels in
 * dc_bn_bwd_finalize's order (same bits), the first row of blocks stores dgamma / dbeta.  One kernel and one dependent kernel boundary less per
 * BatchNorm of the middle flow (autograd's batch_norm_backward at train_hdf5_ddp.py:363). */
int dc_bn_bwd_apply_fin_max_rows(void);
int dc_bn_bwd_apply_fin(int dtype, long M, int C, long count, const void* dout, int lddo, const void* y, int ldy,
                        const void* out, int ldout, int relu, const float* gamma, const float* save_mean,
                        const float* save_invstd, const float* slab, int rows, float* dgamma, float* dbeta, void* dy, int lddy,
                        void* g_out, int ldg, const float* mscale, const float* mshift, void* stream);

/* SeparableConv2d_same.forward as ONE operator (deeplab_xception.py:62-66: depthwise 3x3 "same", then pointwise 1x1) for the entry flow's thin
 * layers: d [N,H,W,Cin] = depthwise(act(x)) -- act = the producer's BatchNorm(+ReLU), applied on load when pscale / pshift are given, as in
 * dc_dwconv_fwd, whose bits d has -- and y [N,H,W,Cout] = pointwise(d), with the BatchNorm partial sums of the stored y in slab
 * [2][slab_rows][Cout] (NULL: none): the kernel fills dc_sepconv_fwd_rows(...) rows, one per workgroup, and zeroes the caller's other rows.
 * Served (rows > 0): bf16, stride 1, dilation 1, Cout 128, Cin 64 / 128, H % 8 == 0, W % 16 (Cin 128) or 32 (Cin 64) == 0, at least 65 536
 * pixels -- block 1 of the entry flow, where both halves are HBM-bound and d would otherwise be read back by a second launch.  wdw: the packed
 * depthwise taps (dc_dwconv_pack_weights), wf: the packed pointwise forward operand (dc_conv_pack_weights). */
int dc_sepconv_fwd_rows(int dtype, int Cin, int Cout, int stride, int dil, int N, int H, int W);
int dc_sepconv_fwd(int dtype, int Cin, int Cout, int N, int H, int W, const void* x, int ldx, const float* pscale, const float* pshift,
                   int prelu, const float* wdw, void* d, int ldd, const void* wf, void* y, int ldy, float* slab, int slab_rows, void* stream);

/* dc_bn_bwd_apply of a BatchNorm + dc_conv_dgrad + the weight gradient of the pointwise (1x1, stride 1) conv in front of it in ONE pass over
 * (dout, y, x): dy = the BatchNorm's input gradient (relu 2: dout masked by y * mscale + mshift > 0; relu 0: no mask) is formed in registers,
 * rounded to bf16 as dc_bn_bwd_apply stores it, and feeds both products from LDS; it is never written.  dx [M][Cin] = dy . W (wb: the packed
 * data-gradient operand of dc_conv_pack_weights); wslab [rows][Cout][Cin] fp32 = one partial weight gradient per workgroup, rows =
 * dc_pw_bn_bwd_rows(...) (0: shape not served -- bf16, Cout 128 with Cin 64 / 128 or Cout 256 with Cin 128 / 256, at least 65 536 pixels: the entry flow's first two blocks, whose 226 - 453 MB
 * tensors make all three passes HBM-bound), summed by dc_fold_slabs (DC_FOLD_CONV, splits = rows, taps = 1).  dgamma / dbeta: the finished
 * parameter gradients (dc_bn_bwd_finalize).  wslab_rows: the rows the caller allocated wslab (and planned its fold) for; the call fails if that is
 * not what dc_pw_bn_bwd_rows returns for the shape now.  Replaces autograd's batch_norm_backward + conv2d backward of SeparableConv2d_same.pointwise
 * (deeplab_xception.py:62-66, 84-101; train_hdf5_ddp.py:363). */
int dc_pw_bn_bwd_rows(int dtype, int Cin, int Cout, long M);
int dc_pw_bn_bwd(int dtype, long M, int Cin, int Cout, long count, const void* dout, int lddo, const void* y, int ldy, int relu,
                 const float* gamma, const float* save_mean, const float* save_invstd, const float* dgamma, const float* dbeta,
                 const float* mscale, const float* mshift, const void* x, int ldx, const void* wb, void* dx, int lddx, float* wslab, int wslab_rows,
                 void* stream);

/* ------------------------------------------------------------------------------------------------
 * Entry stem: Conv2d(16->32, k3, s2, p1) reading the caller's NCHW fp32 batch directly
 * (deeplab_xception.py:145,197).  Output NHWC `dtype`.  Also writes the BN partial-statistics slab.
 * ------------------------------------------------------------------------------------------------ */
int dc_stem_stat_rows(int N, int H, int W);
int dc_stem_fwd(int dtype, int N, int Cin, int H, int W, const float* x_nchw, const float* w /*[32][Cin][3][3]*/,
                void* y, int ldy, float* stat_slab, void* stream);
size_t dc_stem_wgrad_workspace(int N, int Cin, int H, int W);
int dc_stem_wgrad(int dtype, int N, int Cin, int H, int W, const float* x_nchw, const void* dy, int lddy,
                  void* workspace, float* grad_w, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Classifier head: ConvTranspose2d(256->n_classes=3, k3, s2, p1, op1), deeplab_xception.py:374,382.
 * Reads NHWC `dtype`, writes the NCHW fp32 logits the reference API returns.  Runs as GEMMs on the MFMA kernels:
 * P[pixel][co,tap] = x . W (27 products per input pixel, fp32) followed by the sub-pixel tap combination.
 * ------------------------------------------------------------------------------------------------ */
size_t dc_head_workspace(int dtype, int N, int Cin, int Hi, int Wi);
int dc_head_fwd(int dtype, int N, int Cin, int Hi, int Wi, const void* x, int ldx,
                const float* w /*[Cin][3][3][3]*/, float* logits_nchw, void* workspace, void* stream);
/* dc_head_fwd with the loss pass of dc_wce_fused on the logits while they are in registers (reference: upsample.last_deconv,
 * deeplab_xception.py:374,382, followed by fp_loss, utils/losses.py:35-50, argmax train_hdf5_ddp.py:406 and the IoU counts of
 * utils/utils.py:32-60): same logits, loss, gradient, predictions and counts bit for bit as the two calls.  logits_nchw may be
 * null (bf16, Cin = 256: the fused kernel then never stores them); the loss arguments are dc_wce_fused's. */
int dc_head_fwd_loss(int dtype, int N, int Cin, int Hi, int Wi, const void* x, int ldx, const float* w, float* logits_nchw,
                     void* workspace, const void* labels, int label_dtype_bytes, const float* class_weights, float grad_scale,
                     double* loss_sum, float* dlogits, int64_t* pred, int64_t* counts, void* stream);
/* dx = d(loss)/dx and grad_w (master layout) from the NCHW fp32 logit gradient, in one call (they share the gathered
 * 27-tap gradient image).  workspace: dc_head_workspace bytes, 256-byte aligned. */
int dc_head_bwd(int dtype, int N, int Cin, int Hi, int Wi, const void* x, int ldx, const float* dlogits_nchw,
                const float* w, void* dx, int lddx, float* grad_w, void* workspace, void* stream);
/* dc_head_bwd when x = act(bn(bn_y)) (upsample.deconv3: ConvTranspose2d -> BatchNorm2d -> ReLU, deeplab_xception.py:371-373): the
 * data gradient also leaves that BatchNorm's backward sums in bn_slab[2][cdiv(N*Hi*Wi, 128)][Cin] (see dc_conv_dgrad_bnstats). */
int dc_head_bwd_bnstats(int dtype, int N, int Cin, int Hi, int Wi, const void* x, int ldx, const float* dlogits_nchw,
                        const float* w, void* dx, int lddx, float* grad_w, void* workspace, const void* bn_y, int bn_ldy,
                        const float* bn_mean, const float* bn_invstd, const float* bn_mscale, const float* bn_mshift,
                        int bn_relu, float* bn_slab, void* stream);

/* The head on a BatchNorm(+ReLU) output that is never stored (upsample.deconv3's BatchNorm2d + ReLU in front of last_deconv,
 * deeplab_xception.py:371-374): y is the raw output of the convolution in front of that BatchNorm, scale / shift its forward
 * coefficients (dc_bn_finalize / dc_bn_eval_coeffs); the head forms act(y * scale + shift) while loading (fp32 fma, rounded to bf16:
 * the bits dc_bn_apply would have stored), so dc_bn_apply + dc_head_fwd[_loss] and these calls give the same logits bit for bit.
 * bf16, Cin = 256 only (the fused head kernel). */
int dc_head_fwd_bnin(int dtype, int N, int Cin, int Hi, int Wi, const void* y, int ldy, const float* scale, const float* shift,
                     int relu, const float* w, float* logits_nchw, void* workspace, void* stream);
int dc_head_fwd_loss_bnin(int dtype, int N, int Cin, int Hi, int Wi, const void* y, int ldy, const float* scale, const float* shift,
                          int relu, const float* w, float* logits_nchw, void* workspace, const void* labels, int label_dtype_bytes,
                          const float* class_weights, float grad_scale, double* loss_sum, float* dlogits, int64_t* pred,
                          int64_t* counts, void* stream);
/* dc_head_bwd for that case: the weight gradient forms the head's input from y itself (register-staged 128-tile kernel); bn_slab != NULL
 * also leaves the BatchNorm's backward sums there as dc_head_bwd_bnstats does (bn_mean / bn_invstd then required).
 * parts: 3 = everything; 1 = the gathered gradient image + the data gradient (+ sums); 2 = the weight gradient alone, which reads the
 * image a parts = 1 call left in `workspace` -- the caller may issue it on another stream ordered behind that call (the engine's
 * weight-gradient stream), since nothing on the chain waits for grad_w. */
int dc_head_bwd_bnin(int dtype, int N, int Cin, int Hi, int Wi, const void* y, int ldy, const float* scale, const float* shift,
                     int relu, const float* dlogits_nchw, const float* w, void* dx, int lddx, float* grad_w, void* workspace,
                     const float* bn_mean, const float* bn_invstd, float* bn_slab, int parts, void* stream);
/* dx == NULL in dc_head_bwd_bnin (bf16, 256 channels, bn_slab given): the head's data gradient -- 906 MB at local batch 8, read once, by that
 * BatchNorm's backward -- is not stored; the sums are taken all the same.  After dc_bn_bwd_finalize(bn_slab) this second pass forms the
 * gradient again (one MFMA step from the gathered image still in `workspace`) and writes the BatchNorm INPUT's gradient
 * dy = gamma*invstd*(g - dbeta/count - xhat*dgamma/count), bit-equal to the stored-dx path + dc_bn_bwd_apply(relu = 2): the pair moves
 * 2.9 GB instead of 4.6 (deeplab_xception.py:371-374 backward). */
int dc_head_bwd_bnin_apply(int dtype, int N, int Cin, int Hi, int Wi, const void* y, int ldy, const float* scale, const float* shift,
                           int relu, const float* gamma, const float* bn_mean, const float* bn_invstd, const float* dgamma,
                           const float* dbeta, long count, void* dy, int lddy, void* workspace, void* stream);

/* NCHW fp32 (the layout train_hdf5_ddp.py:348 hands over) -> NHWC `dtype`: the one layout pass of the step. */
int dc_nchw_to_nhwc(int dtype, int N, int C, int H, int W, const float* x_nchw, void* out, int ldo, void* stream);

/* Input pipeline (data/cam_hdf5_dataset.py:96-102,122-129): the dataset files are HWC fp32, already channels-last.
 * out[p][j] = scale[j] * (x_hwc[p][channels[j]] - shift[j])  (channels == NULL: identity), written as NHWC `dtype`.
 * Replaces the reference's numpy transpose + normalise on the host; with it the step has no layout pass at all. */
int dc_input_normalize_hwc(int dtype, long npix, int Cfile, int C, const int* channels, const float* x_hwc,
                           const float* shift, const float* scale, void* out, int ldo, void* stream);

/* The same arithmetic written as the reference's NCHW fp32 batch [N][C][HW] (any C >= 1): for --channels subsets
 * (train_hdf5_ddp.py:561) whose count is not a multiple of the 16-byte vector, which run dc_stem_fwd on NCHW fp32 input. */
int dc_input_normalize_hwc_to_nchw(int N, long HW, int Cfile, int C, const int* channels, const float* x_hwc,
                                   const float* shift, const float* scale, float* out_nchw, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Loss and metric: utils/losses.py:28-52 (fp_loss == plain mean of weighted CE), torch.max(.,1)[1]
 * (train_hdf5_ddp.py:406,458) and utils/utils.py:32-60 (compute_score) in one pass over the logits.
 *   loss_sum   double[1]   += sum over pixels of w[y]*(lse - logit[y])          (caller zeroes, divides)
 *   dlogits    fp32 NCHW   = w[y]*(softmax - onehot)*grad_scale                 (NULL to skip)
 *   pred       int64 [B,H,W] first-max argmax                                   (NULL to skip)
 *   counts     int64[9]    += tp[3], fp[3], fn[3]                               (NULL to skip)
 * label_dtype_bytes: 1 (uint8), 4 (int32) or 8 (int64) labels.  A label outside [0,3) -- for which the reference's
 * nn.CrossEntropyLoss raises -- makes loss_sum and that pixel's dlogits NaN (nothing here synchronises, so nothing can raise).
 * ------------------------------------------------------------------------------------------------ */
int dc_wce_fused(int B, int H, int W, const float* logits_nchw, const void* labels, int label_dtype_bytes,
                 const float* class_weights /*device float[3]*/, float grad_scale, double* loss_sum,
                 float* dlogits, int64_t* pred, int64_t* counts, void* stream);
/* compute_score on existing prediction / label maps (int64), counts int64[9] += */
int dc_confusion_counts(long n, const int64_t* pred, const void* labels, int label_dtype_bytes, int64_t* counts,
                        void* stream);

/* ------------------------------------------------------------------------------------------------
 * Image-pool branch helpers (deeplab_xception.py:425,449-450): AdaptiveAvgPool2d((1,1)), the 1x1 ->
 * HxW "bilinear" broadcast and their backward passes.  The per-sample vectors ([N][C]: out, g, v below) are ALWAYS
 * fp32 whatever `dtype` the NHWC side uses: the branch's BatchNorm sees only N values per channel.
 * ------------------------------------------------------------------------------------------------ */
int dc_avgpool_fwd(int dtype, int N, int HW, int C, const void* x, int ldx, void* out /*[N][C]*/, void* stream);
/* dx[n,p,c] += g[n,c] / HW */
int dc_avgpool_bwd_add(int dtype, int N, int HW, int C, const void* g, void* dx, int lddx, void* stream);
/* out[n,p,c] = v[n,c] */
int dc_broadcast_hw(int dtype, int N, int HW, int C, const void* v, void* out, int ldo, void* stream);
/* g[n,c] = sum_p dout[n,p,c] */
int dc_sum_hw(int dtype, int N, int HW, int C, const void* dout, int lddo, void* g, void* stream);
/* dst = src (NHWC view copy, e.g. gradient slices) */
int dc_copy_view(int dtype, long M, int C, const void* src, int lds, void* dst, int ldd, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Optimizers over one flat fp32 arena: torch.optim.Adam / AdamW (train_hdf5_ddp.py:213-216) and LAMB
 * (apex FusedLAMB, :218; this project's own definition).  `step` is the 1-based step count.
 * `lr` is read from a device float so that a captured hipGraph follows the LR schedule.
 * ------------------------------------------------------------------------------------------------ */
int dc_adam_step(int kind, long n, float* p, const float* g, float* m, float* v, const float* lr_dev,
                 float beta1, float beta2, float eps, float weight_decay, const int* step_dev, float grad_scale,
                 void* stream);
/* LAMB: tensor t occupies [offsets[t], offsets[t+1]) of the arena (device int64[ntensors+1]).
 * workspace: dc_lamb_workspace_words(ntensors, n) 4-byte words (partial sums of the global norm, chunk plan, per-chunk
 * norms and the n-element update direction).  p, g, m, v and workspace must be 16-byte aligned.  `g` is only read (as apex FusedLAMB leaves .grad intact).
 * As apex with use_nvlamb=False, the per-tensor trust ratio is applied only when weight_decay != 0.
 * No atomics: the update is bit-identical from run to run. */
size_t dc_lamb_workspace_words(int ntensors, long n);
int dc_lamb_step(int ntensors, const int64_t* offsets_dev, long n, float* p, const float* g, float* m, float* v,
                 const float* lr_dev, float beta1, float beta2, float eps, float weight_decay,
                 const int* step_dev, float max_grad_norm, float grad_scale, float* workspace, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Gradient payload of the data-parallel all-reduce (train_hdf5_ddp.py:227,363: apex / torch DDP reduce the gradients the
 * optimizer reads; SURVEY 5.8: 225.8 MB fp32 or 112.9 MB bf16 per step).  The fp32 arena stays the master copy:
 * dc_grad_pack_bf16 rounds a bucket (round-to-nearest-even) into the send buffer the collective sums in bf16,
 * dc_grad_unpack_bf16 widens the reduced buffer back into the arena.  n elements; both pointers 16-byte aligned.
 * ------------------------------------------------------------------------------------------------ */
int dc_grad_pack_bf16(long n, const float* g, void* out_bf16, void* stream);
int dc_grad_unpack_bf16(long n, const void* in_bf16, float* g, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The gradient all-reduce itself, inside the library (csrc/comm.cpp; SURVEY section 8b: grad_allreduce_{enqueue,wait} over a raw RCCL
 * communicator).  Replaces the reducer of apex / torch DistributedDataParallel: train_hdf5_ddp.py:227 (construction: one communicator per
 * process) and :363 (loss.backward(): bucketed all-reduces overlapped with the rest of backward).
 *   dc_comm_unique_id        rank 0: 128 bytes (ncclGetUniqueId) that the host program hands to every rank by any out-of-band channel
 *   dc_comm_create           every rank: ncclCommInitRank on the current device + a communication stream of the library's own.
 *                            world == 1 with id128 == NULL: no RCCL at all (enqueue / wait are no-ops)
 *   dc_comm_adopt            wrap an ncclComm_t the caller already has (not destroyed by dc_comm_destroy)
 *   dc_comm_create_callback  transport = a host function (tests: torch.distributed over gloo); sync_stream != 0: the compute stream is
 *                            synchronised before every call back (device buffers)
 *   dc_grad_allreduce_enqueue  the communication stream waits for everything enqueued on compute_stream so far, then sums
 *                            buf[0 .. count) (DC_F32 or DC_BF16) across the ranks IN PLACE on it; returns at once
 *   dc_grad_allreduce_wait   compute_stream waits for every collective enqueued so far (the optimizer goes behind it)
 * enqueue / wait take word-sized arguments only, so dc_program_append records them like any kernel launch.
 * ------------------------------------------------------------------------------------------------ */
typedef int (*dc_allreduce_callback)(void* ctx, void* buf, size_t count, int dtype);
int dc_comm_unique_id(void* id128);
int dc_comm_create(const void* id128, int rank, int world, void** comm);
int dc_comm_adopt(void* nccl_comm, int rank, int world, void** comm);
int dc_comm_create_callback(dc_allreduce_callback fn, void* ctx, int rank, int world, int sync_stream, void** comm);
int dc_comm_destroy(void* comm);
/* transport: 0 none (single rank), 1 RCCL, 2 callback; enqueued: calls of dc_grad_allreduce_enqueue so far */
int dc_comm_info(void* comm, int* rank, int* world, int* transport, long* enqueued);
int dc_grad_allreduce_enqueue(void* comm, void* buf, size_t count, int dtype, void* compute_stream);
int dc_grad_allreduce_wait(void* comm, void* compute_stream);

#ifdef __cplusplus
}
#endif
#endif /* DEEPCAM_HIP_H_ */
