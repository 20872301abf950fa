/*
 * osi.h — C ABI of libosi_hip.so: the MI355X (gfx950) implementation of the open-set ImageNet training hot path.
 *
 * The reference (AIML-IfI/openset-imagenet) is pure Python and has no operator / plugin / FFI layer of its own
 * (SURVEY.md §8b); the arithmetic it runs lives in torch / torchvision calls. Each entry point below therefore cites
 * the reference CALL SITE whose arithmetic it replaces (paths relative to /root/reference).
 *
 * Conventions
 *   - plain C: raw device pointers, ints, floats, one opaque stream handle (a hipStream_t); no torch types.
 *   - every function returns OSI_OK (0) or a negative OSI_ERR_* code; nothing throws across the ABI.
 *   - no allocation, no host synchronisation, no implicit stream: the caller owns every buffer (workspace sizes
 *     come from the matching *_workspace query) and passes the stream to launch on. Launch functions are
 *     re-entrant and safe to capture into a hipGraph.
 *   - activations are fp32 NHWC ([B][H][W][C], C contiguous); conv weights are fp32 KRSC ([Cout][R][S][Cin]) —
 *     byte-identical to a torch OIHW tensor kept in channels_last strides, which is how the Python module owns them.
 *   - all pointers must be 16-byte aligned; channel counts must be multiples of 4 (64 for conv GEMM dimensions).
 */
#ifndef OSI_H
#define OSI_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OSI_OK 0
#define OSI_ERR_ARG (-1)    /* shape / pointer / alignment precondition violated; nothing was launched */
#define OSI_ERR_LAUNCH (-2) /* the HIP runtime refused a launch or an attribute */
#define OSI_ERR_STATE (-3)  /* executor called out of order (backward before forward, wrong batch, ...) */

typedef void* osi_stream_t; /* hipStream_t */

int osi_abi_version(void);            /* bumped on any signature change */
const char* osi_build_arch(void);     /* "gfx950" */
const char* osi_strerror(int code);
/* Process-wide development knobs (A/B measurements; every default is the measured optimum). Launch functions only READ them and never
 * consult the environment. Unknown name or a value outside the knob's range -> OSI_ERR_ARG (nothing is changed). Not to be changed while
 * launches are in flight; an executor (osi_resnet50_create) sizes its workspace for the values in force at create and refuses to run
 * (OSI_ERR_STATE) once a plan-relevant one (marked *) differs.
 *   name             range         default  meaning
 *   wgrad_tile *     0 | 64        0        64 forces 64x64 weight-gradient tiles; 0 = 128-wide wherever the channel counts allow
 *   wgrad_blocks *   1 .. 2^20     2048     split-K footprint budget of one weight-gradient launch, in 64x64-workgroup units
 *   wgrad_nst        1 | 2         1        LDS stages of the per-tap weight-gradient kernel
 *   wgrad_group *    0 .. 2        2        weight-gradient block -> XCD mapping: 0 plain 2-D grid, 1 the taps of a cell share an XCD, 2 whole K splits do
 *   wgrad3 *         0 .. 2        2        all-taps 3x3 weight-gradient kernel: 0 never, 1 stride-1 layers, 2 stride-2 layers too
 *   wgrad3_blocks *  1 .. 2^20     768      workgroups per launch its split-K plan aims for
 *   fwd_wide         0 | 1         0        A/B: 64x128 forward tiles wherever Cout % 128 == 0
 *   fwd_rows         0 .. 2        1        1x1 stride-1 forward convolutions with Cin = 64 on the persistent row walker (weight tile resident in LDS,
 *                                           next row tile prefetched) when the launch has >= 8 row tiles per CU; 2 = every eligible shape (Cin = 64 | 128)
 *   fwd_w3           0 | 1         1        3x3 stride-1 forward convolutions stage ONE activation window per tap row and 32-channel slice (column-
 *                                           padded coordinates, the three taps of the row run from it) instead of one 64-row tile per tap
 *   dgrad_w3         0 | 1         1        the same row windows for the in-block fused 3x3 stride-1 input gradients (dY rows per tap row)
 *   dgrad_wide       0 | 1         0        A/B: 64x128 input-gradient tiles wherever Cin % 128 == 0
 *   fwd_wino *       0 | 1         1        the executor runs its 3x3 stride-1 forward convolutions in the Winograd F(2x2,3x3) form (osi_conv_fwd_wino)
 *   dgrad_wino *     0 | 1         1        the same for the in-block fused 3x3 stride-1 input gradients (osi_conv_dgrad_fused_wino)
 *   wgrad_wino *     0 | 1         1        the executor's 3x3 stride-1 weight gradients in the Winograd F(3x3,2x2) form (osi_conv_wgrad_wino)
 *   wino_wide        0 | 1         1        Winograd forward / input gradient: units of 32 tiles x 128 channels (instead of 64 x 64) where the channel count allows
 *   wino_streamk     0 .. 3        2        Winograd forms: the units of the ragged last round are cut along K over all workgroups (fix-up pass): 0 = whole
 *                                           units only, 1 = forward and input gradient, 2 = forward only, 3 = input gradient only
 *   bn_grid          1 .. 2^20     1024     grid cap of the BatchNorm stream kernels
 *   bn_grid_bwd      1 .. 2^20     1024     the same for the backward apply kernels
 *   bn_single_p      1 .. 2^20     128      BatchNorm partials merged by ONE 256-thread launch up to this many row tiles
 *   bn_wide_p        0 .. 2048     2048     ... by ONE 1024-thread launch up to this many (0 = two levels above bn_single_p)
 *   tail_split *     0 | 1         1        forward / input-gradient launches split the tiles of their ragged last round along K
 *   tail_cus *       0 .. 4096     0        CU count the tail plan balances for; 0 = the device's (minus dp_reserved_cus). Affects ONLY that
 *                                           plan: the stem weight-gradient grid always follows the hardware CU count
 *   tail_smax *      1 .. 64       8        most K splits a remainder tile is cut into
 *   tail_mint *      1 .. 4096     16       fewest K tiles (of 32) a split keeps
 *   tail_gain *      0 .. 100      8        least modelled gain of a launch, in percent, for its ragged round to be split
 *   tail_qmax *      0 .. 4096     8        most full rounds a launch may have and still be split
 *   stem_direct *    0 | 1         1        conv1 runs the direct kernels (forward, weight gradient, fused tail) where the geometry allows
 *   dp_reserved_cus * 0 .. 128     0        data parallel: CUs' worth of wave slots the tail plan leaves to co-resident communication kernels */
int osi_set_tuning(const char* name, int value);
int osi_get_tuning(const char* name, int* value);

/* ---- convolution (torchvision.models.resnet50 body constructed at openset_imagenet/model.py:17, run at model.py:37) --- */
typedef struct {
    int B, H, W, Cin; /* input  [B][H][W][Cin]   */
    int Ho, Wo, Cout; /* output [B][Ho][Wo][Cout] */
    int R, S, stride, pad;
} osi_conv_desc;

enum { OSI_TILE_AUTO = 0, OSI_TILE_128x128 = 1, OSI_TILE_128x64 = 2, OSI_TILE_64x128 = 3, OSI_TILE_64x64 = 4,
       OSI_TILE_64x64_S1 = 5, OSI_TILE_64x128_S1 = 6, OSI_TILE_128x128_S1 = 7, OSI_TILE_128x64_S1 = 8 /* _S1: single-buffered LDS */ };

/* y = conv2d(x, w), bias-free. The 7x7 stem is described with Cin = 4 (image staged by osi_nchw3_to_nhwc4) and takes
 * weights packed by osi_stem_weight_pack ([Cout][56 taps][4]). Otherwise Cin % 32 == 0, Cout % 64 == 0. */
int osi_conv_fwd(const osi_conv_desc* d, const float* x, const float* w, float* y, int tile, osi_stream_t stream);
/* Same convolution, and the epilogue also emits per-row-tile BatchNorm partials (mean, M2 per channel, *P tiles of
 * *rows_per_block rows) into pstats, to be finished by osi_bn_finalize_stats — the statistics never re-read y from HBM. */
size_t osi_conv_fwd_bnstats_workspace(const osi_conv_desc* d);
int osi_conv_fwd_bnstats(const osi_conv_desc* d, const float* x, const float* w, float* y, int tile, float* pstats,
                         size_t pstats_bytes, int* P, int* rows_per_block, osi_stream_t stream);
/* Forward convolution whose INPUT is the previous layer's BatchNorm + ReLU applied on the fly: the A operand is
 * relu(x * in_scale[c] + in_shift[c]) computed in the loader (x = the producer's pre-BN output, in_scale / in_shift [Cin] from
 * osi_bn_finalize_stats / osi_bn_eval_coeffs), zero padding applied AFTER the activation as conv2d pads the activation tensor.
 * Replaces conv2d(relu(bn(x)), w) of the Bottleneck (conv2 / conv3) without the activation ever touching HBM. pstats may be NULL
 * (then P / rows_per_block are ignored); tiles: OSI_TILE_AUTO, OSI_TILE_64x64_S1, OSI_TILE_64x128_S1. */
int osi_conv_fwd_act(const osi_conv_desc* d, const float* x, const float* in_scale, const float* in_shift, const float* w, float* y,
                     int tile, float* pstats, size_t pstats_bytes, int* P, int* rows_per_block, osi_stream_t stream);
/* 1x1 stride-1 convolution whose input is a whole bottleneck OUTPUT recomputed in the loader: A = relu(x * in_scale[c] + in_shift[c]
 * + res) with x = conv3's pre-BN output and res = the identity shortcut (same shape). conv1 of the next bottleneck can start
 * without waiting for the block-output pass (which still materialises the tensor for the later consumers, on another stream).
 * The value is bit-identical to what osi_bn_apply_relu_mask writes (one fma, one add, max). */
int osi_conv_fwd_act2(const osi_conv_desc* d, const float* x, const float* in_scale, const float* in_shift, const float* res,
                      const float* w, float* y, int tile, float* pstats, size_t pstats_bytes, int* P, int* rows_per_block,
                      osi_stream_t stream);
/* Winograd F(2x2, 3x3) forms of the two calls above for 3x3 / stride 1 / pad 1 convolutions (conv2 of the bottlenecks without a stride;
 * csrc/conv_wino.hip): 2.25x fewer multiplies, every product and sum still fp32 (exact-f32 MFMA), error against fp64 below the direct
 * kernels' on the network's shapes. `ws` = osi_conv_wino_workspace(d) bytes for the transformed weights (rebuilt by every call: the
 * weights change every step). osi_conv_wino_eligible: 1 when the shape is taken (Cin % 16 == 0 and Cout % 64 == 0 forward, swapped for the
 * input gradient; forward: every 16-tile statistics group must hold the same number of pixels — H, W even and B * H/2 * W/2 % 16 == 0, or
 * whole images per group as at 7 x 7).
 * osi_conv_fwd_wino = osi_conv_fwd_act (in_scale / in_shift given) or osi_conv_fwd_bnstats (NULL): *P = ceil(tiles / 16) partials of
 * *rows_per_block pixels for osi_bn_finalize_stats; pstats may be NULL.
 * osi_conv_dgrad_fused_wino = osi_conv_dgrad_fused restricted to the executor's "in-block" fusion: f->scale0 / shift0 / y0 required (gate
 * recomputed), no addend / relu_mask / y1 / pool mode; f->partials optional ([3][*P][Cin], planes 0 and 1 written). */
int osi_conv_wino_eligible(const osi_conv_desc* d, int input_gradient);
size_t osi_conv_wino_workspace(const osi_conv_desc* d);
/* The transformed weights of a convolution can be built AHEAD of its launches (they are the same for the forward and the backward pass of a
 * step): osi_conv_wino_transform_weights writes them into `u` (osi_conv_wino_weights_bytes(d) bytes; input_gradient = 1: the flipped /
 * transposed form of the input gradient), and the `_pre` calls take `u` instead of the raw weights plus the shared stream-K slab
 * (osi_conv_wino_slab_bytes() bytes, the same for every convolution). The executor does this for all its 3x3 layers on the side stream at
 * the start of a forward pass. */
size_t osi_conv_wino_weights_bytes(const osi_conv_desc* d);
size_t osi_conv_wino_slab_bytes(void);
int osi_conv_wino_transform_weights(const osi_conv_desc* d, const float* w, int input_gradient, float* u, size_t u_bytes, osi_stream_t stream);
int osi_conv_fwd_wino_pre(const osi_conv_desc* d, const float* x, const float* in_scale, const float* in_shift, const float* u, float* y,
                          void* slab, size_t slab_bytes, float* pstats, size_t pstats_bytes, int* P, int* rows_per_block, osi_stream_t stream);
int osi_conv_fwd_wino(const osi_conv_desc* d, const float* x, const float* in_scale, const float* in_shift, const float* w, float* y,
                      void* ws, size_t ws_bytes, float* pstats, size_t pstats_bytes, int* P, int* rows_per_block, osi_stream_t stream);
/* Inference forms (validate() / get_arrays(), openset_imagenet/train.py:142-234: model.eval(), BatchNorm on running statistics). In eval
 * mode a BatchNorm's scale / shift are known before its convolution is launched, so the convolution's epilogue applies them — plus the
 * Bottleneck's shortcut and ReLU (torchvision Bottleneck.forward under model.py:37) — and the pre-BN tensor is never written:
 *     out = [relu](conv2d(x, w) * scale[n] + shift[n] [+ residual])          one fma, one add, one max per element
 * (the roundings of osi_bn_apply: bit-identical to osi_conv_fwd followed by osi_bn_apply on the same accumulators). residual: same
 * shape as out, may be NULL, must not alias out. ws: osi_conv_fwd_epilogue_workspace(d) bytes (slab of a K-split tail; may be 0 / NULL:
 * the launch is then single-pass). Not for the stem (its tail is osi_bn_relu_maxpool_fwd). Cin % 32 == 0, Cout % 64 == 0. */
typedef struct {
    const float* scale;     /* [Cout], e.g. from osi_bn_eval_coeffs */
    const float* shift;     /* [Cout] */
    const float* residual;  /* [B][Ho][Wo][Cout] or NULL */
    int relu;
} osi_conv_epilogue;
size_t osi_conv_fwd_epilogue_workspace(const osi_conv_desc* d);
int osi_conv_fwd_epilogue(const osi_conv_desc* d, const float* x, const float* w, float* out, const osi_conv_epilogue* e, void* ws,
                          size_t ws_bytes, osi_stream_t stream);
/* The Winograd form of the same for the shapes osi_conv_wino_eligible(d, 0) takes; e->residual must be NULL (conv2 of a Bottleneck has
 * no shortcut); u / slab as osi_conv_fwd_wino_pre. */
int osi_conv_fwd_wino_epilogue_pre(const osi_conv_desc* d, const float* x, const float* u, float* out, const osi_conv_epilogue* e, void* slab,
                                   size_t slab_bytes, osi_stream_t stream);
/* dx (+)= conv2d_input_grad(dy, w). accumulate = 1 adds into dx (skip-connection sum); accumulate = 2 ("sparse", ABI 4) writes
 * only the input pixels some filter tap reaches and leaves every other element of dx UNTOUCHED (a stride-2 1x1 convolution reaches
 * the pixels with even h and even w: a quarter of the tensor) — for a consumer that knows the pattern, see
 * osi_dgrad_fusion.addend_stride. Cout % 32 == 0, Cin % 64 == 0. */
int osi_conv_dgrad(const osi_conv_desc* d, const float* dy, const float* w, float* dx, int accumulate, int tile,
                   osi_stream_t stream);
/* Input gradient with a fused epilogue: dx = relu_mask . (conv2d_input_grad(dy, w) + addend), i.e. the gradient already passed
 * through the ReLU that produced this conv's input (bitmask written by osi_bn_apply_relu_mask); when `partials` is set the
 * epilogue also emits, per row tile, the column sums of g, g*xhat0 and (y1 != NULL) g*xhat1 (xhat = (y - mean)*invstd) — the
 * BatchNorm-backward reductions of the layer(s) that produced that input — as partials[3][*P][Cin], to be finished by
 * osi_bn_backward_fused. addend may be dx. */
typedef struct {
    const void* relu_mask;  /* may be NULL: no mask */
    const float* y0;        /* pre-BN tensor of the consumer BatchNorm, same shape as dx */
    const float* mean0;     /* its batch mean [Cin] */
    const float* invstd0;   /* its batch 1/sqrt(var + eps) [Cin] */
    const float* y1;        /* second consumer (downsample branch) or NULL */
    const float* mean1;
    const float* invstd1;
    float* partials;        /* NULL: no reductions */
    size_t partials_bytes;  /* >= osi_conv_dgrad_fused_workspace(d) */
    const float* scale0;    /* alternative gate when relu_mask == NULL: the producer's activation relu(y0 * scale0 + shift0) was */
    const float* shift0;    /* never stored (osi_conv_fwd_act consumed y0 directly), so the gate is recomputed: on where > 0 */
    /* Pool mode (ABI 3; stride-1 convs, relu_mask = scale0 = NULL): this conv's input is the output of the stem's fused
     * bn -> ReLU -> max-pool 3x3 / 2 (osi_bn_relu_maxpool_fwd), so dx is the gradient w.r.t. the POOLED activation. pool_idx = that
     * kernel's arg-max bytes; y0 / mean0 / invstd0 describe the BatchNorm BEFORE the pool, y0 being [B][pool_H][pool_W][Cin]. dx is
     * written unmasked; the partials become the BatchNorm-backward reductions of the max-pool-scattered, ReLU-gated gradient
     * (sum g and sum g * xhat(arg-max pixel) per row tile): bn1's dgamma / dbeta need no pass over the 112 x 112 tensor of their own.
     * Finish them with osi_bn_backward_reduce. */
    const void* pool_idx;
    int pool_H, pool_W;
    /* ABI 4. 0 / 1: `addend` is dense. 2: `addend` holds values only at pixels with even h AND even w (what osi_conv_dgrad with
     * accumulate = 2 wrote for a stride-2 1x1 convolution); every other pixel of it is never read and counts as zero. */
    int addend_stride;
} osi_dgrad_fusion;
size_t osi_conv_dgrad_fused_workspace(const osi_conv_desc* d);
int osi_conv_dgrad_fused(const osi_conv_desc* d, const float* dy, const float* w, float* dx, const float* addend,
                         const osi_dgrad_fusion* f, int tile, int* P, osi_stream_t stream);
int osi_conv_dgrad_fused_wino(const osi_conv_desc* d, const float* dy, const float* w, float* dx, const osi_dgrad_fusion* f, void* ws,
                              size_t ws_bytes, int* P, osi_stream_t stream);
int osi_conv_dgrad_fused_wino_pre(const osi_conv_desc* d, const float* dy, const float* u, float* dx, const osi_dgrad_fusion* f, void* slab,
                                  size_t slab_bytes, int* P, osi_stream_t stream);
/* dw = conv2d_weight_grad(dy, x), deterministic split-K through `ws` (size from osi_conv_wgrad_workspace). The stem writes
 * the packed [Cout][224] form; osi_stem_grad_unpack converts to [Cout][7][7][3]. */
size_t osi_conv_wgrad_workspace(const osi_conv_desc* d);
int osi_conv_wgrad(const osi_conv_desc* d, const float* dy, const float* x, float* dw, void* ws, size_t ws_bytes,
                   osi_stream_t stream);
/* The same with x replaced by relu(x * in_scale[c] + in_shift[c]) in the loader (the conv's input activation was never stored). */
int osi_conv_wgrad_act(const osi_conv_desc* d, const float* dy, const float* x, const float* in_scale, const float* in_shift, float* dw,
                       void* ws, size_t ws_bytes, osi_stream_t stream);
/* Winograd F(3x3, 2x2) form of the two calls above for 3x3 / stride 1 / pad 1 convolutions with Cin % 64 == 0 and Cout % 64 == 0
 * (csrc/conv_wino.hip): the sum over output tiles is taken in the transformed domain (16 multiplies per tile and (cout, cin) pair
 * instead of 36), deterministic (split-K over the tile axis into partial slabs, fixed-order reduce). in_scale / in_shift may be NULL.
 * ws: osi_conv_wgrad_wino_workspace(d) bytes; 0 = the shape is not taken. */
size_t osi_conv_wgrad_wino_workspace(const osi_conv_desc* d);
int osi_conv_wgrad_wino(const osi_conv_desc* d, const float* dy, const float* x, const float* in_scale, const float* in_shift, float* dw,
                        void* ws, size_t ws_bytes, osi_stream_t stream);
int osi_stem_weight_pack(const float* w_krsc3, float* w_packed, int Cout, osi_stream_t stream);
/* Direct form of the stem weight gradient (7x7 / stride 2 / pad 3, Cout = 64, Ho % 8 == 0, Wo % 16 == 0 — the conv1 of
 * torchvision's resnet50 under model.py:17 at any image size that is a multiple of 16 x 32): writes the gradient in the PARAMETER
 * layout [64][7][7][3] directly (no packed form, no unpack pass), deterministic (one partial per workgroup, fixed-order reduce).
 * osi_stem_wgrad_direct_workspace returns 0 for a geometry it does not take (use osi_conv_wgrad + osi_stem_grad_unpack then);
 * osi_stem_wgrad_direct returns OSI_ERR_ARG for it. x4 = the NHWC4 image the forward read. */
size_t osi_stem_wgrad_direct_workspace(const osi_conv_desc* d);
int osi_stem_wgrad_direct(const osi_conv_desc* d, const float* dy, const float* x4, float* dw_krsc3, void* ws, size_t ws_bytes,
                          osi_stream_t stream);
/* The same gradient with the whole stem tail fused into the operand loader: dY = BatchNorm backward of the ReLU-gated max-pool
 * scatter of gpool (the gradient w.r.t. the pooled activation) is built per tile in LDS from gpool, the arg-max bytes of
 * osi_bn_relu_maxpool_fwd and the stem's conv output y, and never written to memory. dgamma / dbeta: the stem BatchNorm's reductions
 * (osi_bn_relu_maxpool_bwd with dy = NULL computes exactly those). ws: osi_stem_wgrad_fused_workspace(d) bytes. */
size_t osi_stem_wgrad_fused_workspace(const osi_conv_desc* d);
int osi_stem_wgrad_fused(const osi_conv_desc* d, const float* gpool, const void* pool_idx, const float* y, const float* x4,
                         const float* gamma, const float* mean, const float* invstd, const float* dgamma, const float* dbeta,
                         float* dw_krsc3, void* ws, size_t ws_bytes, osi_stream_t stream);
int osi_stem_grad_unpack(const float* g_packed, float* g_krsc3, int Cout, osi_stream_t stream);

/* ---- BatchNorm2d in training mode + ReLU + residual (torchvision Bottleneck under model.py:37; train() at train.py:125) --- */
size_t osi_bn_workspace(int M, int C);
/* batch statistics of y[M][C]: mean, invstd = 1/sqrt(biased var + eps), scale = gamma*invstd, shift = beta - mean*scale;
 * running stats updated with the unbiased variance when running_mean/var are non-NULL. */
int osi_bn_train_stats(const float* y, int M, int C, const float* gamma, const float* beta, float eps, float momentum,
                       float* running_mean, float* running_var, float* mean, float* invstd, float* scale, float* shift,
                       void* ws, size_t ws_bytes, osi_stream_t stream);
/* second half of osi_bn_train_stats for partials produced elsewhere (osi_conv_fwd_bnstats): pstats = [P][C] means then [P][C] M2 */
int osi_bn_finalize_stats(float* pstats, size_t pstats_bytes, int P, int rows_per_block, int M, int C, const float* gamma,
                          const float* beta, float eps, float momentum, float* running_mean, float* running_var, float* mean,
                          float* invstd, float* scale, float* shift, osi_stream_t stream);
/* eval mode (validate(), train.py:142-196): scale/shift from the running statistics */
int osi_bn_eval_coeffs(const float* running_mean, const float* running_var, const float* gamma, const float* beta, float eps,
                       int C, float* scale, float* shift, osi_stream_t stream);
/* the same for up to OSI_BN_MULTI_MAX BatchNorm layers in one launch (the executor's inference forward: all 53 at once) */
#define OSI_BN_MULTI_MAX 64
typedef struct {
    const float *running_mean, *running_var, *gamma, *beta;
    float *scale, *shift;
    int C;
} osi_bn_eval_layer;
int osi_bn_eval_coeffs_multi(const osi_bn_eval_layer* layers, int n, float eps, osi_stream_t stream);
/* out = [relu](y*scale + shift [+ residual]) */
int osi_bn_apply(const float* y, const float* residual, const float* scale, const float* shift, float* out, int M, int C,
                 int relu, osi_stream_t stream);
/* g = dout * (act > 0) (act NULL: g = dout); dgamma, dbeta; dy = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)).
 * gmasked (optional) receives g, the gradient that continues along the skip connection. dy may alias dout. */
size_t osi_bn_backward_workspace(int M, int C);
int osi_bn_backward(const float* dout, const float* act, const float* y, const float* mean, const float* invstd,
                    const float* gamma, float* dy, float* gmasked, float* dgamma, float* dbeta, int M, int C, void* ws,
                    size_t ws_bytes, osi_stream_t stream);

/* Bitmask forms: the forward writes one bit per element ((BN(+res)) > 0; osi_bn_relu_mask_bytes of storage) next to the ReLU
 * output, and the backward takes that mask instead of re-reading the activation tensor in both of its passes. */
size_t osi_bn_relu_mask_bytes(int M, int C);
int osi_bn_apply_relu_mask(const float* y, const float* residual, const float* scale, const float* shift, float* out,
                           void* relu_mask, int M, int C, osi_stream_t stream);
/* Block output of a Bottleneck with a projection shortcut in one pass: out = relu(y * scale + shift + (res_y * res_scale +
 * res_shift)) — bn3(conv3) + downsample BatchNorm(downsample conv) + ReLU — plus the ReLU bitmask; the shortcut's normalised
 * tensor is never stored. Bit-identical to osi_bn_apply on the shortcut followed by osi_bn_apply_relu_mask. */
int osi_bn_apply_relu_mask2(const float* y, const float* scale, const float* shift, const float* res_y, const float* res_scale,
                            const float* res_shift, float* out, void* relu_mask, int M, int C, osi_stream_t stream);
int osi_bn_backward_relu_mask(const float* dout, const void* relu_mask, const float* y, const float* mean, const float* invstd,
                              const float* gamma, float* dy, float* gmasked, float* dgamma, float* dbeta, int M, int C, void* ws,
                              size_t ws_bytes, osi_stream_t stream);

/* BatchNorm backward when g (already ReLU-masked) and its reductions come from osi_conv_dgrad_fused: psum_g / psum_gx are
 * [P][C] row-tile partials of sum g and sum g*xhat. Finishes dbeta = sum g, dgamma = sum g*xhat and applies
 * dy = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)). dy may alias g. ws: osi_bn_backward_workspace(M, C) bytes. */
int osi_bn_backward_fused(const float* g, const float* y, const float* mean, const float* invstd, const float* gamma,
                          const float* psum_g, const float* psum_gx, int P, float* dy, float* dgamma, float* dbeta, int M, int C,
                          void* ws, size_t ws_bytes, osi_stream_t stream);

/* ---- pooling / layout (ResNet.maxpool, ResNet.avgpool, flatten; image batch of train.py:128) --- */
int osi_nchw3_to_nhwc4(const float* x_nchw, float* y_nhwc4, int B, int H, int W, osi_stream_t stream);
/* uint8 [B][H][W][3] -> fp32 [B][H][W][4] = value / 255 (ToTensor(), train.py:263), horizontally flipped where flip[b] != 0
 * (RandomHorizontalFlip, train.py:262; flip may be NULL), 4th channel zero: the input pipeline's last mile on the device. */
int osi_u8hwc3_to_nhwc4(const unsigned char* x_u8_nhwc, const unsigned char* flip, float* y_nhwc4, int B, int H, int W,
                        osi_stream_t stream);
/* The device side of the reference's input transform Compose([Resize(256), RandomCrop(224) | CenterCrop(224), RandomHorizontalFlip,
 * ToTensor]) (train.py:259-268) after decode + resize: canvas = uint8 [B][Hc][Wc][3] holding the resized image (or a window of it
 * that contains the crop), crop_xy = int32 [B][2] top-left corner (x0, y0) of the HxW crop inside the canvas (NULL = (0, 0); read-only:
 * the kernel clamps each corner into the valid range in registers), flip[b] != 0 mirrors the CROPPED image (NULL = none);
 * y = fp32 [B][H][W][4], value / 255, 4th channel zero. */
int osi_u8_crop_flip_to_nhwc4(const unsigned char* canvas_u8, const int* crop_xy, const unsigned char* flip, float* y_nhwc4, int B, int Hc,
                              int Wc, int H, int W, osi_stream_t stream);
/* idx: B*Ho*Wo*C bytes (argmax position 0..8 per element) */
int osi_maxpool3x3s2_fwd(const float* x, float* y, void* idx, int B, int H, int W, int C, osi_stream_t stream);
int osi_maxpool3x3s2_bwd(const float* dy, const void* idx, float* dx, int B, int H, int W, int C, osi_stream_t stream);
/* The ResNet stem tail (bn1 -> relu -> maxpool, torchvision ResNet.forward under model.py:37) as one pass each way:
 *   fwd: pooled = maxpool3x3s2(relu(y*scale + shift)); the post-ReLU activation is never stored. idx bytes as above with bit 7 =
 *        "window maximum > 0" (the ReLU gate of the pixel the gradient will return to).
 *   bwd: gpool = dJ/dpooled -> dgamma, dbeta and dy = BatchNorm backward of the pool-scattered, ReLU-gated gradient, gathered on
 *        the fly (no [B][H][W][C] gradient tensor). ws: osi_bn_backward_workspace(B*H*W, C) bytes. */
/* The reduction half of osi_bn_backward_fused alone: merges per-row-tile partials (psum_g, psum_gx: [P][C], e.g. from a pool-mode
 * osi_conv_dgrad_fused) into dgamma / dbeta. M = elements per channel of the BatchNorm (c1 = dbeta / M, c2 = dgamma / M are left in ws
 * for callers that want them). ws: osi_bn_backward_workspace(M, C) bytes. */
int osi_bn_backward_reduce(const float* psum_g, const float* psum_gx, int P, float* dgamma, float* dbeta, int M, int C, void* ws,
                           size_t ws_bytes, osi_stream_t stream);
int osi_bn_relu_maxpool_fwd(const float* y, const float* scale, const float* shift, float* pooled, void* idx, int B, int H, int W,
                            int C, osi_stream_t stream);
int osi_bn_relu_maxpool_bwd(const float* gpool, const void* idx, const float* y, const float* mean, const float* invstd,
                            const float* gamma, float* dy, float* dgamma, float* dbeta, int B, int H, int W, int C, void* ws,
                            size_t ws_bytes, osi_stream_t stream);
int osi_avgpool_fwd(const float* x, float* y, int B, int HW, int C, osi_stream_t stream);
int osi_avgpool_bwd(const float* dy, float* dx, int B, int HW, int C, osi_stream_t stream);

/* ---- head: resnet_base.fc (model.py:19-20) and logits (model.py:23-26) --- */
int osi_linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int K, int O, osi_stream_t stream);
int osi_linear_bwd(const float* dy, const float* x, const float* w, float* dx, int dx_accumulate, float* dw, float* db, int B,
                   int K, int O, osi_stream_t stream);

/* ---- losses (losses.py:16-29; train.py:343; train.py:344-347; objectosphere term: SURVEY.md §8 a9) --- */
enum { OSI_LOSS_ENTROPIC = 0, OSI_LOSS_SOFTMAX = 1, OSI_LOSS_GARBAGE = 2 };
/* loss (1 float) and dlogits = dJ/dlogits in one launch. features != NULL adds alpha/B * sum r_i^2 and writes dfeatures.
 * dlogits may be NULL (validation: loss only). */
int osi_loss_fwd_bwd(int mode, const float* logits, const long long* target, int B, int C, float unk_weight,
                     long long ignore_index, const float* class_weights, const float* features, int F, float xi, float alpha,
                     float* loss, float* dlogits, float* dfeatures, osi_stream_t stream);
int osi_softmax(const float* logits, float* out, int B, int C, osi_stream_t stream); /* train.py:177 */
/* validation confidences of metrics.py:8-42 accumulated on the device across batches: acc4 (double[4], caller-zeroed) +=
 * {sum known score[y], #known, sum negatives (1 + offset - max score[:last_valid_class]), #negatives}.
 * last_valid_class: 0 = all columns (Python None), negative = Python negative slice end (-1: drop the background column). */
int osi_confidence_accumulate(const float* logits, const long long* target, int B, int C, float offset, long long unknown_class,
                              int last_valid_class, double* acc4, osi_stream_t stream);
/* the same sums from a matrix of softmax SCORES, i.e. metrics.confidence(scores, target_labels, offset, unknown_class,
 * last_valid_class) itself (metrics.py:8-42) */
int osi_confidence_from_scores(const float* scores, const long long* target, int B, int C, float offset, long long unknown_class,
                               int last_valid_class, double* acc4, osi_stream_t stream);

/* Open-Set Classification Rate curve, util.calculate_oscr (util.py:90-122), on device-resident scores[N][C] (f32 or f64) and
 * int64 labels. Outputs: taus[0 .. totals[0]) = the distinct target-class scores of the known samples in ascending order;
 * for the totals[0] - 1 thresholds tau = taus[u] (the largest is dropped, util.py:114): ccr_count[u] = #{known, argmax == label,
 * score[label] > tau}, fpr_count[u] = #{label == unk_label, max score > tau}; totals = {#distinct, #known, #unk_label}.
 * The curve is ccr_count / totals[1], fpr_count / totals[2] in float64 (integer counts: bit-identical to the reference).
 * taus / ccr_count / fpr_count hold N entries each, totals 3; ws: osi_oscr_workspace(N) bytes. */
size_t osi_oscr_workspace(int N);
int osi_oscr_f32(const float* scores, const long long* gt, int N, int C, long long unk_label, void* ws, size_t ws_bytes,
                 float* taus, long long* ccr_count, long long* fpr_count, long long* totals, osi_stream_t stream);
int osi_oscr_f64(const double* scores, const long long* gt, int N, int C, long long unk_label, void* ws, size_t ws_bytes,
                 double* taus, long long* ccr_count, long long* fpr_count, long long* totals, osi_stream_t stream);

/* ---- optimizer + arena utilities (train.py:356-359 construction, train.py:127,139 zero_grad/step) --- */
int osi_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double lr, double beta1,
                  double beta2, double eps, long long step, float grad_scale, osi_stream_t stream);
int osi_sgd_step(float* param, const float* grad, float* momentum_buf, size_t n, float lr, float momentum, int first_step,
                 float grad_scale, osi_stream_t stream);
int osi_fill_f32(float* p, size_t n, float value, osi_stream_t stream);
int osi_scale_f32(float* p, size_t n, float s, osi_stream_t stream);
int osi_i64_add(long long* p, int n, long long inc, osi_stream_t stream);

/* ---- whole-network executor: ResNet50.forward (model.py:28-39) and its autograd backward (train.py:138) ---------------
 * One call enqueues the full forward (or a range of backward stages) on `stream`. The caller owns three flat arenas whose
 * layout is reported by the *_info queries:
 *   params   fp32   all 161 weight tensors in nn.Module registration order (conv KRSC, BN gamma/beta, fc, logits)
 *   grads    fp32   same layout as params
 *   buffers  fp32   BN running_mean / running_var (53 x 2)      nbt: int64[53] num_batches_tracked
 * and a workspace of osi_resnet50_workspace_bytes() holding activations, saved statistics and scratch. */
typedef struct osi_resnet50* osi_resnet50_t;

int osi_resnet50_create(osi_resnet50_t* out, int B, int H, int W, int fc_dim, int out_features, int logit_bias);
void osi_resnet50_destroy(osi_resnet50_t net);
int osi_resnet50_num_tensors(osi_resnet50_t net);             /* parameter tensors */
/* name: reference state_dict key ("resnet_base.layer1.0.conv1.weight"); shape in torch order (OIHW for convs), ndim <= 4;
 * offset/numel in floats inside the params/grads arena */
int osi_resnet50_tensor_info(osi_resnet50_t net, int i, char* name, int name_cap, int* ndim, int* shape, size_t* offset,
                             size_t* numel);
size_t osi_resnet50_param_floats(osi_resnet50_t net);         /* arena length, multiple of 4 */
int osi_resnet50_num_bn(osi_resnet50_t net);
/* BN layer j: prefix ("resnet_base.layer1.0.bn1"), channel count, offsets of running_mean / running_var in `buffers` */
int osi_resnet50_bn_info(osi_resnet50_t net, int j, char* prefix, int cap, int* C, size_t* rm_offset, size_t* rv_offset);
size_t osi_resnet50_buffer_floats(osi_resnet50_t net);
size_t osi_resnet50_workspace_bytes(osi_resnet50_t net);
int osi_resnet50_num_stages(osi_resnet50_t net);              /* backward stages (gradient buckets), head first */
/* the fixed geometry the executor was created for: batch, image height, image width (any pointer may be NULL) */
int osi_resnet50_geometry(osi_resnet50_t net, int* B, int* H, int* W);
/* floats [lo, hi) of the grads arena that are final once backward stage s has run */
int osi_resnet50_stage_grad_range(osi_resnet50_t net, int s, size_t* lo, size_t* hi);

/* Optional input staging from a uint8 [B][H][W][3] batch (osi_u8hwc3_to_nhwc4 into the executor's input buffer inside
 * `workspace`); the next osi_resnet50_forward on that workspace passes image = NULL (OSI_ERR_STATE without a staged input). */
int osi_resnet50_stage_input_u8(osi_resnet50_t net, const unsigned char* images_u8_nhwc, const unsigned char* flip,
                                void* workspace, osi_stream_t stream);
/* Alternative to staging: bind an NHWC4 fp32 batch [B][H][W][4] that already lives in device memory (e.g. written by
 * osi_u8_crop_flip_to_nhwc4 on a copy stream, one batch ahead); the next osi_resnet50_forward with image = NULL reads it in place
 * and that step's stem weight gradient reads it again, so it must stay untouched until the step's backward has run. */
int osi_resnet50_bind_input_nhwc4(osi_resnet50_t net, const float* x_nhwc4);
/* image: [B][3][H][W] fp32 NCHW as the reference feeds it (or NULL after osi_resnet50_stage_input_u8 /
 * osi_resnet50_bind_input_nhwc4). training != 0: batch statistics + running-stat update. */
int osi_resnet50_forward(osi_resnet50_t net, const float* params, float* buffers, long long* nbt, const float* image,
                         void* workspace, float* logits, float* features, int training, osi_stream_t stream);
/* runs backward stages [stage_lo, stage_hi) given dJ/dlogits and (optionally, may be NULL) dJ/dfeatures */
int osi_resnet50_backward(osi_resnet50_t net, const float* params, float* grads, void* workspace, const float* dlogits,
                          const float* dfeatures, int stage_lo, int stage_hi, osi_stream_t stream);

/* Data-parallel hand-off (ABI 5). With option "stage_join" = 0 a staged osi_resnet50_backward call (stage_hi < stages) does NOT make
 * `stream` wait for the side stream's weight gradients (the last stage always does); instead the caller makes its COMMUNICATION stream
 * wait for everything the finished stages produced: osi_resnet50_grads_ready(net, main, waiter) records the progress of `main` (the
 * stream the backward calls were issued on) and of the executor's side stream and makes `waiter` wait for both — `main` itself waits
 * for nothing and goes on with the next stage while the collective runs (reference intent: config/train.yaml:18,35-39). */
int osi_resnet50_grads_ready(osi_resnet50_t net, osi_stream_t main_stream, osi_stream_t waiter_stream);

/* Weight gradients on a low-priority side stream, overlapped with dgrad / BatchNorm backward (default on; joined back into
 * `stream` at the end of every backward call unless "stage_join" = 0). enable = 0 serialises everything on the caller's stream. */
int osi_resnet50_set_overlap(osi_resnet50_t net, int enable);
/* Per-executor switches: "overlap" (= osi_resnet50_set_overlap), "fwd_fork" (projection shortcut of the forward pass on the side
 * stream, default 1), "fwd_recompute" (conv1 of a bottleneck recomputes the previous identity-shortcut block output in its loader and
 * that block's output pass runs beside it on the side stream; default 0: measured no faster), "side_priority_normal" (side stream at default instead of lowest priority; only before the first training
 * call, else OSI_ERR_STATE), "stage_join" (default 1; see osi_resnet50_grads_ready), "stagger", "stem_fused", "stem_pool_stats", "ds_sparse",
 * "stem_wgrad_main" (A/B switches of the backward schedule), "eval_fused" (default 1: a forward with training = 0 runs
 * the inference forms — every BatchNorm + shortcut + ReLU in its convolution's epilogue, no pre-BN tensor, no block-output pass, no
 * bitmask, one coefficient launch for all 53 BatchNorms; 0 = the training topology on running statistics, kept for A/B; same bits).
 * Unknown name -> OSI_ERR_ARG. */
int osi_resnet50_set_option(osi_resnet50_t net, const char* name, int value);

/* Optional HIP-event instrumentation of the executor (bench.py's roofline leg): one event after every op on the launch
 * stream, attributed to a kernel class. profile_read synchronises on the last event — call it outside timed regions. */
enum { OSI_PROF_START = 0, OSI_PROF_CONV_FWD = 1, OSI_PROF_CONV_DGRAD = 2, OSI_PROF_CONV_WGRAD = 3, OSI_PROF_BN_FWD = 4,
       OSI_PROF_BN_BWD = 5, OSI_PROF_OTHER = 6, OSI_PROF_NCLASS = 7 };
int osi_resnet50_profile(osi_resnet50_t net, int enable);
int osi_resnet50_profile_read(osi_resnet50_t net, double* ms_per_class, int* ops_per_class);
/* osi_resnet50_profile(net, 2) = timeline mode: the side-stream overlap stays on and each op's completion event remembers its
 * stream. timeline_read synchronises, then fills completion times (ms since the first event), classes and stream flags of up to
 * `cap` ops and resets the log. Dev instrumentation (tools/timeline.py): how far the weight gradients trail the critical path. */
int osi_resnet50_timeline_read(osi_resnet50_t net, double* t_ms, int* cls, int* on_side, int cap, int* count);


/* ---- debug: the non-smooth decisions of the latest forward (TEST INFRASTRUCTURE — nothing on the product path calls these) ----
 * The reference's backward (j.backward(), openset_imagenet/train.py:138) differentiates through 49 ReLUs (torchvision's stem ReLU +
 * three per Bottleneck) and one max-pool arg-max (model.py:17); a whole-network gradient comparison is only tight when both sides
 * take the same branch at every one of them. Gate i of osi_resnet50_debug_num_gates() = 1 + 3 x 16, in forward order: 0 = the
 * stem ReLU as the max-pool output sees it (shape [B][64][Hp][Wp]: the gate of the pooled element, all the backward ever uses),
 * then per bottleneck the ReLU after bn1, after bn2 and the block-output ReLU. Each is read from what the backward kernels
 * themselves consume (stored bitmask / the gate recomputed from the pre-BN tensor / the arg-max byte). Output: one byte (0 / 1)
 * per element in NCHW order [B][C][H][W]; pool_argmax_nchw (gate 0 only, may be NULL): int32 flat index h * Ws + w into the
 * stem's [Hs][Ws] plane, the convention of torch.nn.functional.max_pool2d(return_indices=True). OSI_ERR_STATE before any forward. */
int osi_resnet50_debug_num_gates(osi_resnet50_t net);
int osi_resnet50_debug_gate_shape(osi_resnet50_t net, int i, int* C, int* H, int* W);
int osi_resnet50_debug_gate(osi_resnet50_t net, void* workspace, int i, unsigned char* gate_nchw, int* pool_argmax_nchw,
                            osi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* OSI_H */
