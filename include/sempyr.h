/* sempyr.h - C ABI of libsempyr.so, the MI355X (gfx950) kernel library under the Semantic-Pyramid
 * GAN training step.
 *
 * The reference (/root/reference) is pure PyTorch and has NO FFI layer of its own; its arithmetic is
 * the stock torch ops called from models.py / lossfunction.py.  This header is therefore the boundary
 * a maintainer binds underneath those modules: every entry point names the reference call site(s)
 * whose arithmetic it replaces.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - Plain C: pointers + sizes only, no torch types.  All pointers are DEVICE pointers borrowed for
 *     the call (caller-allocated, e.g. torch tensors via data_ptr()); nothing is allocated or freed,
 *     no call synchronises.  `stream` is a hipStream_t passed as void*.
 *   - Return 0 on success, negative sp_status otherwise; sp_last_error_string() gives the reason
 *     (thread-local).  Nothing throws.
 *   - Activations are NHWC ("channels-last"): element (n,h,w,c) at ((n*H+h)*W+w)*ld + c, where ld is
 *     the channel pitch in elements.  `dtype` selects the HBM storage type of activations and packed
 *     weights (SP_F32 / SP_BF16 / SP_F16); accumulation is always fp32.  Master weights, biases, statistics,
 *     gradients of parameters and loss scalars are always fp32.
 *   - Packed conv weights: [Cout][kh*kw][cin_p] (K-contiguous per output channel), cin_p = Cin rounded
 *     up to a 16-byte multiple.  The same kernel computes the input gradient from the "dgrad" packing
 *     [Cin][flipped taps][cout_p].
 */
#ifndef SEMPYR_H
#define SEMPYR_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SP_VERSION 1

typedef void* sp_stream_t;

enum sp_status { SP_OK = 0, SP_ERR_INVALID = -1, SP_ERR_LAUNCH = -2, SP_ERR_UNSUPPORTED = -3 };
enum sp_dtype { SP_F32 = 0, SP_BF16 = 1, SP_F8 = 2 /* OCP e4m3 operands, see sp_conv_params: x_scale .. y8_amax; 16-bit outputs bf16 */,
                SP_F16 = 3 /* IEEE half-precision storage, v_mfma_f32_16x16x32_f16, fp32 accumulate: accepted wherever SP_BF16 is (same
                            * layouts, same kernels compiled for the other 16-bit type) - BASELINE.json config 5's "fp16 activations" */,
                SP_F8_F16 = 4 /* sp_conv2d_igemm only: SP_F8 operands with fp16 (not bf16) 16-bit outputs */ };
enum sp_act { SP_ACT_NONE = 0, SP_ACT_LRELU = 1, SP_ACT_RELU = 2, SP_ACT_TANH = 3 };

int sp_version(void);

/* Knobs for tests and A/B measurements.  The library reads NO environment variables: every switch goes through this call
 * (the Python binding forwards SP_* environment variables of the same names at load time, _lib.py).  value < 0 restores the
 * built-in default.  Except SP_TUNE_DETERMINISTIC none of them changes results beyond fp summation order.
 *   SP_TUNE_CONV_TALL        0 = never use conv3x3_tall_kernel, 1 = where its tiles fill the chip (default), 2 / 3 = force the 16- / 8-row form
 *   SP_TUNE_IGEMM_DMA        0 = never use the LDS-DMA igemm kernel, 1 = small-spatial 3x3 layers (default), 2 = everywhere
 *   SP_TUNE_WGRAD_ROWS       0 = never use the row-walker 3x3 weight-gradient kernel, 1 = maps >= 32 wide and 16 x 16 maps (default),
 *                            2 = maps >= 32 wide only, 3 = 8 x 8 maps too
 *   SP_TUNE_DETERMINISTIC    1 = the reductions behind gradients, parameters and normalisation statistics run in a fixed
 *                            order (per-split partial slabs + ordered reduce instead of fp32 atomics): those tensors are
 *                            bit-identical run to run (tests/test_gpu_step.py).  NOT covered: the scalar loss VALUES (fp64
 *                            atomics over blocks: equal to ~1e-16 relative, not bit for bit) and the <dW, W> dot of
 *                            sp_conv2d_wgrad_fused; a row-walker launch whose workspace is too small for its slabs does not
 *                            happen in this mode (the ordered per-tap kernels take the layer; sp_conv2d_wgrad_accum_pooled,
 *                            which has no other kernel, reports an error) - no silent fall-back to atomics.  0 = atomics where they are
 *                            faster; default (-1): on for SP_F32 storage (the parity mode), off for SP_BF16
 *   SP_TUNE_SPLITK_TARGET / _MINSTEPS     split-K plan of the small-spatial 3x3 forward / input-gradient layers (384 / 6)
 *   SP_TUNE_CONV1X1_DIRECT   0 = 1x1 layers on the tiled igemm kernel (default 1: direct kernel)
 *   SP_TUNE_CONV_SHORT       0 = Cout > 64 layers on the register-staged halo kernel (default 1: 8-row tall kernel)
 *   SP_TUNE_WGRAD9_BLOCKS, _WGRAD_BLOCKS, _WGRAD_MINSTEPS, _WGRAD_SMALL_M, _WGRAD_K1_TILE64   per-tap weight-gradient plan
 *   SP_TUNE_WGRAD_ROWS_THIN, _WGRAD_ROWS_BLOCKS, _WGRAD_ROWS_SLABS                            row-walker plan
 *   SP_TUNE_CONV_STAGGER     0 = both halves of a tall-kernel block issue their LDS-DMA at the head of a stage; 1 = the second half
 *                            issues a third of the way into its MFMA stream (default: 1 for the 16-row 128-co tile, 0 otherwise)
 *   SP_TUNE_CONV1X1_SPLITK   a bf16 1x1 layer (128 <= Cin <= 768) splits K over the four waves of a block while it has at most
 *                            this many (32-pixel x 64-channel) blocks (default 320: the 2x2 .. 8x8 maps at batch 20; 0 = never)
 *   SP_TUNE_WGRAD1X1         block target of the streaming weight-gradient kernel of the bf16 1x1 layers (default 256; 0 = those
 *                            layers stay on the per-tap kernel)
 *   SP_TUNE_CONV_CIN8        0 = the 3x3 layers with an 8-channel input (the padded RGB images) stay on the generic kernels (default 1)
 *   SP_TUNE_CONV_THINCO      0 = the 3x3 layers with at most 4 output channels (input gradients towards the images) stay on the
 *                            generic kernels (default 1)
 *   SP_TUNE_CONV_PP          bf16 3x3 layers with Cout > 64: 0 = conv3x3_tall_kernel (round 2's lockstep schedule), 1 = the ping-pong
 *                            schedule of conv_pp.hip, tile height by round count (default), 8 / 16 = force that tile height
 *   SP_TUNE_WGRAD_PP         0 = the 3x3 weight gradient of wide maps stays on the 4-wave row walker (default 1: its ping-pong form)
 *   SP_TUNE_IGEMM_TILE       output tile of the LDS-DMA igemm on small-spatial 3x3 layers with Cout > 64: 0 = 64 co x 64 px, 1 = 128 x 128,
 *                            2 = 128 co x 64 px, 3 = 64 co x 128 px; default: 0, or 2 where 64 x 64 tiles make 1 - 2 rounds of the chip and K is long
 *   SP_TUNE_CONV_PPW         16-bit 3x3 layers with Cout > 64 on 16 x 32-pixel patches: the ping-pong kernel with 64 co x 4 rows per wave
 *                            (conv_ppw.hip) where the round count favours 16-row items; 0 = off (tall<2,16> / the 8-row form), 2 = wherever eligible,
 *                            3 = like 1 with the 16-row form priced WITH the K-split of its last round
 *   SP_TUNE_LINEAR_KS        K range per block of the split-K MFMA linear kernel: 1024 / 512 / 256 / 128 (default: the widest that yields 256 blocks)
 *   SP_TUNE_BN_ITERS         pixels per thread of the elementwise BatchNorm passes (grid sizing; default 2)
 *   SP_TUNE_CONV_PP_SPLIT    0 = the ping-pong 3x3 kernels never split the work items of their last, partial round along K (default 1: they
 *                            do where sp_conv_params.workspace holds the partial tiles - sp_conv2d_workspace() says how many bytes - and
 *                            sp_conv_params.split_sync the counters; 2 = only launches of at least one full round of the 256 blocks; 3 (tests) = like
 *                            1, and the closing piece of an item stores and counts like every other piece - the hand-over's re-read
 *                            path, which otherwise runs only when a block is delayed, on every split launch)
 *   SP_TUNE_CONV_PP_PRIO     bit 0: s_setprio 1 around every MFMA segment of the ping-pong kernel (default 1); bit 1: static priority 1
 *                            for the second-dispatched half of the block */
enum { SP_TUNE_CONV_TALL = 0, SP_TUNE_IGEMM_DMA = 1, SP_TUNE_WGRAD_ROWS = 2, SP_TUNE_DETERMINISTIC = 3,
       SP_TUNE_SPLITK_TARGET = 4, SP_TUNE_SPLITK_MINSTEPS = 5, SP_TUNE_CONV1X1_DIRECT = 6, SP_TUNE_CONV_SHORT = 7,
       SP_TUNE_WGRAD9_BLOCKS = 8, SP_TUNE_WGRAD_BLOCKS = 9, SP_TUNE_WGRAD_MINSTEPS = 10, SP_TUNE_WGRAD_SMALL_M = 11,
       SP_TUNE_WGRAD_K1_TILE64 = 12, SP_TUNE_WGRAD_ROWS_THIN = 13, SP_TUNE_WGRAD_ROWS_BLOCKS = 14, SP_TUNE_WGRAD_ROWS_SLABS = 15,
       SP_TUNE_CONV_STAGGER = 16, SP_TUNE_CONV1X1_SPLITK = 17, SP_TUNE_WGRAD1X1 = 18, SP_TUNE_CONV_CIN8 = 19, SP_TUNE_CONV_THINCO = 20, SP_TUNE_CONV_PP = 21, SP_TUNE_CONV_PP_PRIO = 22, SP_TUNE_WGRAD_PP = 23, SP_TUNE_BN_ITERS = 24, SP_TUNE_IGEMM_TILE = 25, SP_TUNE_CONV_PPW = 26, SP_TUNE_LINEAR_KS = 27, SP_TUNE_CONV_PP_SPLIT = 28, SP_TUNE_COUNT = 29 };
int sp_set_tuning(int32_t key, int32_t value);
const char* sp_last_error_string(void);
/* Name of the kernel (route) the last sp_conv2d_igemm / sp_conv2d_wgrad* call of THIS thread launched ("" before the first one):
 * lets a profiler-less caller attribute its event timings to kernels (bench.py's per-route table).  Static strings. */
const char* sp_last_route(void);

/* ------------------------------------------------------------------------------------------------
 * Convolution as NHWC implicit GEMM on MFMA (3x3 stride 1 pad 1, or 1x1).
 * Replaces nn.Conv2d forward at models.py:34,55,58,232-243,299,303,309,312,394,398,403,439,443,448
 * and the torchvision VGG-16 `features` convs (models.py:176,201-202); with the dgrad packing it is
 * also the input-gradient pass autograd runs for them at model_wrapper.py:160,188.
 *   y = act( (conv(x, w) + bias) * slope(mask_src) + res1 + res2 )
 *   slope(m) = m > 0 ? 1 : mask_neg_slope   (only when mask_src != NULL; used to fold the derivative
 *   of a LeakyReLU/ReLU that precedes the convolution into its input-gradient pass)
 * ---------------------------------------------------------------------------------------------- */
typedef struct sp_conv_params {
    const void* x;          /* [n][h][w][cin_p]                      dtype */
    const void* w;          /* [cout][ksize*ksize][cin_p]            dtype */
    const float* bias;      /* [cout] or NULL                                */
    void* y;                /* [n][h][w][ldy] (cout valid channels)   dtype */
    const void* res1;       /* like y, or NULL                               */
    const void* res2;       /* like y, or NULL                               */
    const void* mask_src;   /* like y, or NULL                               */
    float mask_neg_slope;
    int32_t n, h, w_, cin_p, cout, ldy, ksize, act, dtype;
    void* workspace;        /* optional fp32 scratch of sp_conv2d_workspace() bytes: lets small-spatial layers split K, and the 16-bit 3x3
                             * ping-pong kernel split the items of its last partial round (conv_pp.hip, "tail split"); NULL = never */
    int64_t workspace_bytes;
    int32_t pool2;          /* 2: y is [n][h/2][w/2][ldy] and y = act(maxpool2x2(conv) + bias) - conv -> ReLU -> nn.MaxPool2d(2) of
                             * the frozen VGG-16 stages (models.py:158-216) when the unpooled tensor is not needed (no-grad
                             * pass); no residuals, act NONE / ReLU / LeakyReLU.
                             * 1: y, res1, res2 are [n][h/2][w/2][ldy] and y = act(avgpool2x2(conv) + bias + res1 + res2): the
                             * nn.AvgPool2d(2) that follows the second convolution of a discriminator block (models.py:407-417,
                             * 452-462) rides in the epilogue.  3x3, cout > 32 and a multiple of 16, h % 8 == 0, w % 32 == 0,
                             * ldy % 8 == 0, no mask_src; anything else is rejected (SP_ERR_INVALID) */
    int32_t in_up2;         /* 1: x is [n][h/2][w/2][cin_p] and the convolution runs over 1/4 x its nearest-neighbour x2 expansion,
                             * i.e. over the gradient of a 2x2 average pooling that is never written out (input-gradient pass of
                             * a pool2 layer).  3x3, cout > 32, h % 8 == 0, w % 32 == 0 (SP_ERR_INVALID otherwise) */
    /* dtype == SP_F8 (BASELINE.json config 5: fp8 MFMA implicit-GEMM path; the frozen VGG-16 pyramid's 3x3 layers with
     * cout > 64, w % 32 == 0, h % 8 == 0, models.py:183-216): x and w hold OCP e4m3 bytes (cin_p a multiple of 16), the MFMA
     * is v_mfma_f32_16x16x32_fp8_fp8, accumulation fp32:
     *   v = act( conv(x, w) * x_scale[0] * w_scale[co] + bias[co] ), optionally 2x2 max-pooled (pool2 = 2),
     * stored as bf16 into y (may be NULL) and / or re-quantised, q = sat(v * y8_inv_scale[0]), as e4m3 into y8 (may be NULL;
     * same [n][h][w][ldy] geometry) for the next layer of the chain; max|v| is merged into y8_amax[0] (atomic max on the fp32
     * bit pattern, v >= 0 after ReLU) for the delayed scaling of the next call.  act NONE / ReLU, no residuals, no mask_src,
     * no in_up2, pool2 0 / 2, cout % 16 == 0.  All scale pointers are DEVICE pointers (no host round trip). */
    const float* x_scale;   /* [1]    dequantisation scale of x                      (SP_F8 only) */
    const float* w_scale;   /* [cout] dequantisation scale per output channel        (SP_F8 only) */
    void* y8;               /* e4m3 output or NULL                                   (SP_F8 only) */
    const float* y8_inv_scale; /* [1] quantisation scale of y8 (1 / its dequantisation scale), required with y8 */
    float* y8_amax;         /* [1] running max of v, or NULL                         (SP_F8 only) */
    /* Two-group batch (SP_F32 / SP_BF16): img_scale != NULL multiplies the fp32 accumulator of image n by img_scale[n >= img_split]
     * BEFORE bias / mask / residuals / activation:  y = act((conv(x, w) * img_scale[g(n)] + bias) * slope(mask_src) + res1 + res2).
     * It lets the discriminator's D(real) and D(fake) passes of one step (model_wrapper.py:153-155) - same weight_orig, but a
     * different spectral-norm sigma each, since every forward advances the power iteration (models.py:128-135) - run as ONE launch
     * over 2B images with the packing W / sigma_real and img_scale = {1, sigma_real / sigma_fake}; the same launch form computes the
     * input gradient (dgrad packing).  DEVICE pointer to 2 floats, both > 0.  NULL: no scaling (img_split ignored). */
    const float* img_scale;
    int32_t img_split;      /* images [0, img_split) use img_scale[0], the rest img_scale[1] */
    int32_t reserved_;
    int64_t split_pix_;     /* set by the library (first OUTPUT pixel index of the second group); callers leave it 0 */
    /* Fused 1x1 tail (16-bit storage, 3x3, cout == 64, h % 16 == 0, w % 32 == 0, ldy % 8 == 0, no pooling; SP_ERR_UNSUPPORTED elsewhere):
     *   tail_y[n,h,w,o] = tail_act( sum_c tail_w[o][c] * y[n,h,w,c] + tail_bias[o] ),  o < tail_cout <= 4
     * computed in the epilogue from the (16-bit rounded) y of this launch - the generator's last two layers, conv3x3 -> LeakyReLU ->
     * conv1x1 -> tanh (models.py:55-61), in one launch.  y may then be NULL (a forward pass that never looks at the 64-channel tensor
     * again): the 168 MB tensor of the 256 x 256 layer is neither written nor read back.  tail_w: [tail_cout][64] in the storage type
     * (the forward packing of the 1x1 layer), tail_y pitch tail_ld elements. */
    const void* tail_w;
    const float* tail_bias;
    void* tail_y;
    int32_t tail_cout, tail_act, tail_ld, reserved2_;
    /* pool2 == 2 only (conv -> ReLU -> MaxPool2d(2), the VGG-16 stages of /root/reference/models.py:183-216 in the pass WITH gradient):
     * where to leave the window position of every pooled element, so that the unpooled tensor never reaches HBM and the pooling's
     * backward (sp_maxpool2_bwd_idx) needs neither it nor a recomputation.  One uint32 per (pooled pixel, group of 16 channels),
     * [n * (h/2) * (w/2)][cout / 16]: bits 2c .. 2c+1 = 2 * row + column of the FIRST maximum (scan order, as torch's max_pool2d
     * routes its gradient) of channel 16 g + c, taken over the values as the storage type holds them - exactly the element
     * sp_maxpool2_bwd would pick from the stored unpooled tensor.  NULL: not recorded. */
    uint32_t* pool_idx;
    /* Round 6: counters of the 3x3 ping-pong kernels' K-split of their last partial round of work items (conv_pp.hip / conv_ppw.hip).
     * NULL: the launch never splits.  Otherwise SP_CONV_SPLIT_SYNC_BYTES of device memory that is ZERO when first handed to the
     * library and is left zero by every launch that used it (the kernels clean up behind themselves): one such area per STREAM
     * makes concurrent launches on different streams safe - the library keeps no device-side state of its own.  Needs
     * `workspace` for the partial tiles (sp_conv2d_workspace). */
    int32_t* split_sync;
} sp_conv_params;
#define SP_CONV_SPLIT_SYNC_BYTES 8192
int sp_conv2d_igemm(const sp_conv_params* p, sp_stream_t stream);
/* Bytes of fp32 scratch sp_conv2d_igemm wants in sp_conv_params.workspace to split the K loop of this shape over several
 * blocks (3x3 layers of small spatial extent: too few output tiles to fill 256 CUs); 0 = it would not split.  The scratch
 * holds one partial [n*h*w][cout] slab per split; it needs no initialisation and carries nothing between calls. */
int sp_conv2d_workspace(int32_t n, int32_t h, int32_t w_, int32_t cin_p, int32_t cout, int32_t ksize, int32_t dtype,
                        int64_t* bytes_out);

/* Weight gradient of the same convolution (autograd of nn.Conv2d at model_wrapper.py:160,188):
 *   dw[co][tap][ci] (+)= sum_{n,h,w} dy[n,h,w,co] * x[n,h+dr,w+ds,ci]      fp32, layout [cout][taps][cin_p]
 * dw is zeroed by the call, then accumulated with split-K partial sums. */
int sp_conv2d_wgrad(const void* x, const void* dy, float* dw, int32_t n, int32_t h, int32_t w_,
                    int32_t cin_p, int32_t cout, int32_t ld_dy, int32_t ksize, int32_t dtype,
                    sp_stream_t stream);

/* Same, plus by-products that cost no extra pass over the activations: dbias[cout] = sum_{n,h,w} dy (the bias
 * gradient; NULL to skip) and dot[0] = <dw, w_packed> where w_packed is the forward packing (W/sigma) of this layer,
 * i.e. the inner product the spectral-norm backward needs (NULL/NULL to skip).  dw, dbias, dot are zeroed by the call. */
int sp_conv2d_wgrad_fused(const void* x, const void* dy, float* dw, float* dbias, const void* w_packed, float* dot,
                          float* workspace, int64_t workspace_floats, int32_t n, int32_t h, int32_t w_, int32_t cin_p,
                          int32_t cout, int32_t ld_dy, int32_t ksize, int32_t dtype, sp_stream_t stream);
/* sp_conv2d_wgrad_fused without its fill: ACCUMULATES into dw [cout][taps][cin_p] and, if given, dbias [cout]; the caller
 * has zero-filled them (one fill for a whole network's gradient arena).  workspace (may be NULL): fp32 scratch of
 * sp_conv2d_wgrad_workspace() floats for per-block partial tiles - with it the split-K blocks merge through plain stores
 * and one reduce pass instead of serialised fp32 atomics. */
int sp_conv2d_wgrad_accum(const void* x, const void* dy, float* dw, float* dbias, float* workspace, int64_t workspace_floats,
                          int32_t n, int32_t h, int32_t w_, int32_t cin_p, int32_t cout, int32_t ld_dy, int32_t ksize,
                          int32_t dtype, sp_stream_t stream);
/* sp_conv2d_wgrad_accum for a pool2 layer: dy is the gradient at the POOLED resolution [n][h/2][w/2][ld_dy] and stands for
 * 1/4 x its nearest-neighbour x2 expansion (n, h, w_ describe x).  Only the shapes of the row-walking kernel (bf16, 3x3,
 * w % 32 == 0, h % 2 == 0: sp_conv2d_wgrad_workspace() > 0); SP_ERR_INVALID otherwise. */
int sp_conv2d_wgrad_accum_pooled(const void* x, const void* dy, float* dw, float* dbias, float* workspace, int64_t workspace_floats,
                                 int32_t n, int32_t h, int32_t w_, int32_t cin_p, int32_t cout, int32_t ld_dy, int32_t ksize,
                                 int32_t dtype, sp_stream_t stream);
int sp_conv2d_wgrad_workspace(int32_t n, int32_t h, int32_t w_, int32_t cin_p, int32_t cout, int32_t ksize,
                              int32_t dtype, int64_t* floats_out);
/* sp_conv2d_wgrad_accum for a two-group batch (sp_conv_params.img_scale; the discriminator's D(real) | D(fake) pass): images
 * [0, split) accumulate into (dw_a, dbias_a), images [split, n) into (dw_b, dbias_b) - the spectral-norm backward of each forward
 * needs ITS weight gradient.  dy_pooled != 0: dy is at the pooled resolution as in sp_conv2d_wgrad_accum_pooled.  Where the row-walking
 * kernel takes the layer and the group boundary falls between two of its blocks this is ONE launch over all n images plus one
 * reduce pass per group (half the partial-tile traffic of two launches); otherwise the two groups run one after the other.
 * workspace: sp_conv2d_wgrad_workspace() floats for n images. */
int sp_conv2d_wgrad_accum_pair(const void* x, const void* dy, float* dw_a, float* dbias_a, float* dw_b, float* dbias_b, float* workspace,
                               int64_t workspace_floats, int32_t n, int32_t split, int32_t h, int32_t w_, int32_t cin_p, int32_t cout,
                               int32_t ld_dy, int32_t ksize, int32_t dy_pooled, int32_t dtype, sp_stream_t stream);
/* Deferred slab reductions.  The streaming weight-gradient kernels of the 1x1 and 8-channel 3x3 layers split the pixels over blocks,
 * leave one partial tile per split in the workspace and a second, tiny launch adds the partial tiles to dW in a fixed order.  Nothing
 * reads dW before the end of a backward pass (model_wrapper.py:160,188: the optimizer step), so a caller may collect those launches:
 *   sp_wgrad_reduce_defer(1)        from now on sp_conv2d_wgrad_accum* QUEUE that reduction instead of launching it (a layer whose dW
 *                                    is already queued keeps its own launch);
 *   sp_wgrad_reduce_flush(1, s)     ONE launch per 48 queued reductions on stream s (descriptors by value: capturable), the same
 *                                    arithmetic in the same order - bit-identical to the separate launches;
 *   sp_wgrad_reduce_flush(0, s)     drops the queue (a pass that was abandoned); sp_wgrad_reduce_pending() = its length;
 *   sp_wgrad_reduce_defer(0)        back to immediate reductions (what is queued stays queued until the flush).
 * The caller keeps every workspace alive until the flush and flushes on the stream the weight-gradient launches ran on.  The queue is
 * host state of the process (not thread-safe: one thread drives the passes of a device, as torch's autograd does). */
int sp_wgrad_reduce_defer(int32_t on);
int sp_wgrad_reduce_flush(int32_t run, sp_stream_t stream);
int sp_wgrad_reduce_pending(void);

/* ------------------------------------------------------------------------------------------------
 * Skinny linear layers: y = act(x W^T + bias + res), batch rows of any pitch, weights packed [n][kp]
 * (kp = K rounded up to 8, zero padded).  Replaces nn.Linear at models.py:28,128,132,356,359 and the
 * VGG-16 classifier (models.py:210-213); with the transposed packing [K][np] it is the dgrad pass.
 * ---------------------------------------------------------------------------------------------- */
int sp_linear_fwd(const void* x, int32_t ldx, const void* w_packed, int32_t kp, const float* bias,
                  const void* res, void* y, int32_t ldy, int32_t batch, int32_t k, int32_t n, int32_t act,
                  int32_t dtype, sp_stream_t stream);
/* Same contract with an fp32 scratch of sp_linear_workspace() floats supplied by the caller: large bf16 matrices
 * (batch <= 32) take the MFMA split-K path (weight panels streamed by hundreds of blocks; every K-split
 * stores its partial [batch][n] slab, a finalize pass sums them and applies bias / residual / activation).  The scratch
 * needs no initialisation.  scratch == NULL: sp_linear_fwd. */
int sp_linear_workspace(int32_t batch, int32_t k, int32_t n, int32_t dtype, int64_t* floats_out);
int sp_linear_fwd_ws(const void* x, int32_t ldx, const void* w_packed, int32_t kp, const float* bias,
                     const void* res, void* y, int32_t ldy, int32_t batch, int32_t k, int32_t n, int32_t act,
                     int32_t dtype, float* scratch, sp_stream_t stream);
/* dw[n][kp] = sum_b dy[b][n] x[b][k] (fp32), dbias[n] = sum_b dy[b][n] (may be NULL). */
int sp_linear_wgrad(const void* x, int32_t ldx, const void* dy, int32_t ld_dy, float* dw, int32_t kp,
                    float* dbias, int32_t batch, int32_t k, int32_t n, int32_t dtype, sp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Spectral normalisation of every layer of one network, batched (torch.nn.utils.spectral_norm, the
 * forward-pre-hook behind models.py:28,34,55,58,128,132,135,232-243,299-315,356-360,393-404,438-449):
 * one power iteration (in place on u, v) when power_iter != 0, sigma = u.(W v), then W/sigma is written
 * in the packings the conv / linear kernels read.  The table lives in DEVICE memory.
 *   scratch (fp32, per call, needs no initialisation): per layer at scratch_off: v-snapshot[cols], s[rows], u-snapshot[rows],
 *   {sigma, 1/sigma, -, -}; at part_off: ceil(rows/128) x cols floats (W^T u is formed as per-row-slab partial sums that are
 *   added in a fixed order: the forward pass is bit-reproducible)
 *   pack_arena (per call): per layer fwd packing at fwd_off, dgrad packing at dgrad_off (byte offsets, -1 = none)
 *   max_pack_elems: >= the element count of the largest packing and >= 1024 * ceil(cin_p/32) * ceil(cout_p/32) of every
 *   layer (the packing kernel moves 32 x 32 x taps tiles; this bounds its 2-D grid).  taps <= 9.
 *   pack_blocks > 0: the packing kernel runs on a 1-D grid of exactly that many blocks and layer i owns blocks
 *   [pack_block0[i], pack_block0[i+1]) - ceil(cin_p/32)*ceil(cout_p/32) of them (kind 1: ceil(rows*cols/1024)).  The 2-D grid
 *   (pack_blocks == 0) launches max-tiles x n_layers blocks, 90 % of which exit at once for a network with one big layer.
 * ---------------------------------------------------------------------------------------------- */
typedef struct sp_sn_layer {
    const float* w;        /* weight_orig viewed [rows][cols] (OIHW flattened, cols = cin*taps)   */
    float* u;              /* weight_u [rows]  (updated in place)                                  */
    float* v;              /* weight_v [cols]  (updated in place)                                  */
    int64_t scratch_off;   /* in floats                                                             */
    int64_t part_off;      /* in floats: ceil(rows/128) x cols partial sums of W^T u (power iteration only; summed in order) */
    int64_t fwd_off;       /* bytes; layout [rows][taps][cin_p] dtype, or plain fp32 [rows][cols] if kind==1 */
    int64_t dgrad_off;     /* bytes; layout [cin][flipped taps][cout_p] dtype                       */
    int32_t rows, cols, cin, taps, cin_p, cout_p, kind, pack_block0;
} sp_sn_layer;
int sp_sn_forward(const sp_sn_layer* table_dev, int32_t n_layers, int32_t max_rows, int32_t max_cols,
                  int64_t max_pack_elems, float* scratch, int64_t scratch_floats, void* pack_arena,
                  int32_t power_iter, int32_t dtype, int32_t pack_blocks, sp_stream_t stream);
/* Two forwards of one network whose activations run as ONE two-group batch (sp_conv_params.img_scale; the discriminator's
 * D(real) / D(fake), model_wrapper.py:153-155): forward a packed W / sigma_a, forward b (one more power iteration) has sigma_b.
 * out[2 i] = 1, out[2 i + 1] = sigma_a / sigma_b for layer i - the per-group accumulator scales of a launch that uses a's
 * packing.  scratch_a / scratch_b: the scratch buffers of the two sp_sn_forward calls (same table). */
int sp_sn_pair_scales(const sp_sn_layer* table_dev, int32_t n_layers, const float* scratch_a, const float* scratch_b, float* out,
                      sp_stream_t stream);
/* The same backward for every layer of a network in one call (two launches).  Offsets are in floats: dw_off / dot_off
 * into `arena` (the caller zero-fills the arena once per backward pass; the weight-gradient kernels accumulate the
 * dW slots, this call the dots), scratch_off into the scratch of the matching sp_sn_forward call, grad_off into
 * `grads` (out, [rows][cols] per layer).  A layer whose dW slot was never written yields a zero gradient.
 * max_elems: largest rows*cols of the table.  accumulate_from (NULL, or a buffer laid out like `grads`, `grads` itself
 * included): the result is added to it - the gradients of a second forward through the same network (D(real) and D(fake),
 * model_wrapper.py:150-160) land on the first one's without one autograd addition per parameter.  bias_grads (NULL to
 * skip): the layers' bias-gradient slots of the arena are copied (accumulate_from == NULL) or added into it.
 * dot_partials: n_layers x 512 floats of scratch (no initialisation): per-block partial sums of <dW, W>, added in block order. */
typedef struct sp_sn_bwd_layer {
    const float* w;        /* weight_orig [rows][cols] */
    int64_t dw_off, dot_off, scratch_off, grad_off;
    int32_t rows, cols, cin, taps, cin_p, plain;
    int32_t db_off, bias_off;  /* bias gradient: arena[db_off .. +rows) -> bias_grads[bias_off .. +rows) (floats) */
} sp_sn_bwd_layer;
int sp_sn_backward_batched(const sp_sn_bwd_layer* table_dev, int32_t n_layers, int64_t max_elems, float* arena,
                           const float* scratch, float* grads, const float* accumulate_from, float* bias_grads,
                           float* dot_partials, sp_stream_t stream);
/* The same with every gradient it writes (weights and biases) multiplied by grad_scale: the SP_F16 mode's static loss scale comes
 * off where the (fp32) parameter gradients are formed, without a pass of its own over the buffer. */
int sp_sn_backward_batched_scaled(const sp_sn_bwd_layer* table_dev, int32_t n_layers, int64_t max_elems, float* arena,
                                  const float* scratch, float* grads, const float* accumulate_from, float* bias_grads,
                                  float* dot_partials, float grad_scale, sp_stream_t stream);
/* ... and with the factor read from device memory (grad_scale_dev[0]) when the kernel runs: the SP_F16 mode's DYNAMIC loss scale
 * (sp_loss_scale_update) - a captured graph keeps working while the scale moves. */
int sp_sn_backward_batched_dscaled(const sp_sn_bwd_layer* table_dev, int32_t n_layers, int64_t max_elems, float* arena,
                                   const float* scratch, float* grads, const float* accumulate_from, float* bias_grads,
                                   float* dot_partials, const float* grad_scale_dev, sp_stream_t stream);

/* One-off packing of a frozen, non-normalised fp32 weight (the VGG-16 pyramid, models.py:176-181) into the
 * same two packings.  chw_c > 0: the input-feature index is permuted from NCHW-flatten (c*chw_hw + s) to
 * NHWC (s*chw_c + c) order (models.py:208 flattens NCHW; the kernels keep NHWC). */
int sp_pack_weight(const float* w, int32_t rows, int32_t cols, int32_t cin, int32_t taps, int32_t cin_p,
                   int32_t cout_p, int32_t chw_c, int32_t chw_hw, void* fwd, void* dgrad, int32_t dtype,
                   sp_stream_t stream);
/* Backward of one layer: grad[rows][cols] = (dwsn - <dwsn, W/sigma> u v^T) / sigma with the u, v, sigma
 * snapshots of the forward whose scratch slice is passed.  dwsn is fp32 in the forward packing
 * (plain != 0: [rows][cols]).  dot_tmp: one float; dot_ready = 1: it already holds <dwsn, w_orig>;
 * dot_ready = 2: it holds <dwsn, w_orig / sigma> (delivered by sp_conv2d_wgrad_fused); dot_ready = 3: it is
 * already zero-filled (sp_conv2d_wgrad_fused given `dot` without `w_packed`) and the dot product is formed here. */
int sp_sn_backward(const float* dwsn, const float* w_orig, const float* layer_scratch, int32_t rows, int32_t cols,
                   int32_t cin, int32_t taps, int32_t cin_p, int32_t plain, float* dot_tmp, int32_t dot_ready,
                   float* grad, sp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * (Conditional) BatchNorm, training mode (nn.BatchNorm2d at models.py:53,484; ConditionalBatchNorm.forward
 * models.py:491-506).  sp_bn_stats: batch mean / invstd (+ running-stat update with the unbiased variance;
 * training == 0 derives them from the running statistics instead).  sp_bn_apply:
 *   y = act(scale[n,c] * (x - mean[c]) * invstd[c] + bias[n,c]),  (scale,bias) = (gamma,beta) or, when emb != NULL,
 *   the two halves of emb[cls[n]] (row = [scale(C) | bias(C)]).  partials: 1024*2*C floats of scratch (per-block
 *   partial sums, added in fp64 by a second kernel: deterministic, no atomics).
 * ---------------------------------------------------------------------------------------------- */
int sp_bn_stats(const void* x, int32_t n, int64_t hw, int32_t c, float* partials, float eps, float momentum,
                float* running_mean, float* running_var, int32_t training, float* mean_out, float* invstd_out,
                int32_t dtype, sp_stream_t stream);
int sp_bn_apply(const void* x, void* y, int32_t n, int64_t hw, int32_t c, const float* mean, const float* invstd,
                const float* gamma, const float* beta, const float* emb, const int64_t* cls, int32_t act,
                int32_t dtype, sp_stream_t stream);
/* sp_bn_apply fused with the bilinear x2 upsampling (align_corners=True) that follows CBN -> LeakyReLU in a generator block
 * (models.py:296-298): y is [n][2h][2w][c]; every output pixel normalises + activates its four source pixels on the fly. */
/* The same for a batch of TWO groups - images [0, split) and [split, n) belong to two forwards of the network (round 5: the generator's
 * two forwards of an iteration in one pass, model_wrapper.py:144-151,165-172) - each group normalised by ITS OWN batch statistics, in
 * one launch set: mean2 / invstd2 are [2][c]; the running statistics take the two batches one after the other, group `first_group`
 * first (the forward the reference runs first).  Training mode.  cls: n class indices. */
int sp_bn_stats_pair(const void* x, int32_t n, int32_t split, int64_t hw, int32_t c, float* partials, float eps, float momentum,
                     float* running_mean, float* running_var, int32_t first_group, float* mean2, float* invstd2, int32_t dtype,
                     sp_stream_t stream);
int sp_bn_apply_pair(const void* x, void* y, int32_t n, int32_t split, int64_t hw, int32_t c, const float* mean2, const float* invstd2,
                     const float* gamma, const float* beta, const float* emb, const int64_t* cls, int32_t act, int32_t dtype,
                     sp_stream_t stream);
int sp_bn_apply_upsample2_pair(const void* x, void* y, int32_t n, int32_t split, int32_t h, int32_t w_, int32_t c, const float* mean2,
                               const float* invstd2, const float* gamma, const float* beta, const float* emb, const int64_t* cls,
                               int32_t act, int32_t dtype, sp_stream_t stream);   /* sp_bn_apply_upsample2 on a batch of two groups */
int sp_bn_apply_upsample2(const void* x, void* y, int32_t n, int32_t h, int32_t w_, int32_t c, const float* mean,
                          const float* invstd, const float* gamma, const float* beta, const float* emb,
                          const int64_t* cls, int32_t act, int32_t dtype, sp_stream_t stream);
/* dy is the gradient w.r.t. the (activated) output; partials: 1024*2*c floats, c_tmp: 2*c floats of scratch.
 * Parameter gradients: dgamma/dbeta [c] (plain) or demb [num_classes][2c] (conditional; zeroed by the call). */
int sp_bn_backward(const void* dy, const void* x, void* dx, int32_t n, int64_t hw, int32_t c, const float* mean,
                   const float* invstd, const float* gamma, const float* beta, const float* emb, const int64_t* cls,
                   int32_t act, float* partials, float* c_tmp, float* dgamma, float* dbeta, float* demb,
                   int32_t num_classes, int32_t dtype, sp_stream_t stream);

/* BatchNorm (+ activation) of u = bilinear x2 (align_corners=True) of a stored tensor x [n][h][w][c], without materialising u: the
 * generator's final block, UpsamplingBilinear2d -> BatchNorm2d -> LeakyReLU (models.py:52-54).  Same statistics / running-stat /
 * gradient semantics as sp_bn_stats / sp_bn_apply / sp_bn_backward applied to u [n][2h][2w][c] (u takes the storage type's rounding,
 * as if it had been written); y and dy are [n][2h][2w][c]; sp_bn_backward_up2 writes du = d loss / d u [n][2h][2w][c] - the caller
 * folds it back with sp_upsample2_bwd.  partials: 1024*2*c floats, c_tmp: 2*c floats of scratch.  c / (16-byte group) <= 256. */
int sp_bn_stats_up2(const void* x, int32_t n, int32_t h, int32_t w_, int32_t c, float* partials, float eps, float momentum,
                    float* running_mean, float* running_var, float* mean_out, float* invstd_out, int32_t dtype, sp_stream_t stream);
int sp_bn_apply_up2(const void* x, void* y, int32_t n, int32_t h, int32_t w_, int32_t c, const float* mean, const float* invstd,
                    const float* gamma, const float* beta, const float* emb, const int64_t* cls, int32_t act, int32_t dtype,
                    sp_stream_t stream);
int sp_bn_backward_up2(const void* dy, const void* x, void* du, int32_t n, int32_t h, int32_t w_, int32_t c, const float* mean,
                       const float* invstd, const float* gamma, const float* beta, const float* emb, const int64_t* cls,
                       int32_t act, float* partials, float* c_tmp, float* dgamma, float* dbeta, float* demb, int32_t num_classes,
                       int32_t dtype, sp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Pooling / resampling (NHWC, c % 4 == 0).
 *   avg-pool 2x2 (models.py:406,451); optional second output y_act = act(y)
 *   max-pool 2x2/2 (models.py:245; VGG features models.py:203), relu != 0 fuses the preceding ReLU
 *   adaptive avg-pool (models.py:126 and the VGG 8->7 avgpool models.py:206), act_in fuses a preceding activation
 *   bilinear x2 align_corners=True (nn.UpsamplingBilinear2d, models.py:52,298,308)
 * ---------------------------------------------------------------------------------------------- */
int sp_avgpool2_fwd(const void* x, void* y, void* y_act, int32_t act, int32_t n, int32_t h, int32_t w_, int32_t c,
                    int32_t dtype, sp_stream_t stream);
int sp_avgpool2_bwd(const void* dy, void* dx, int32_t n, int32_t h, int32_t w_, int32_t c, int32_t dtype,
                    sp_stream_t stream);
/* y_act = act(x) and y_pool = avgpool2x2(x) in one pass over x (the two consumers of a discriminator block's input,
 * models.py:452-462: LeakyReLU -> conv and AvgPool -> 1x1 conv); backward: dx = act'(x) * d_act + expand(d_pool) / 4, either
 * gradient may be NULL.  act: NONE / LeakyReLU(0.2) / ReLU. */
int sp_act_avgpool2_fwd(const void* x, void* y_act, void* y_pool, int32_t act, int32_t n, int32_t h, int32_t w_, int32_t c,
                        int32_t dtype, sp_stream_t stream);
int sp_act_avgpool2_bwd(const void* d_act, const void* d_pool, const void* x, void* dx, int32_t act, int32_t n, int32_t h,
                        int32_t w_, int32_t c, int32_t dtype, sp_stream_t stream);
int sp_maxpool2_fwd(const void* x, void* y, int32_t n, int32_t h, int32_t w_, int32_t c, int32_t relu, int32_t dtype,
                    sp_stream_t stream);
int sp_maxpool2_bwd(const void* dy, const void* x, void* dx, int32_t n, int32_t h, int32_t w_, int32_t c,
                    int32_t relu, int32_t dtype, sp_stream_t stream);
/* Backward of conv -> ReLU -> MaxPool2d(2) from what the fused epilogue left behind (sp_conv_params.pool_idx): dx [n][h][w][c] gets
 * dy [n][h/2][w/2][c] at the recorded window position where the pooled value y is positive, zero elsewhere - bit for bit what
 * sp_maxpool2_bwd(relu = 1) computes from the unpooled tensor.  c % 16 == 0. */
int sp_maxpool2_bwd_idx(const void* dy, const void* y, const uint32_t* idx, void* dx, int32_t n, int32_t h, int32_t w_, int32_t c,
                        int32_t dtype, sp_stream_t stream);
int sp_adaptive_avgpool_fwd(const void* x, void* y, int32_t n, int32_t h, int32_t w_, int32_t c, int32_t oh,
                            int32_t ow, int32_t act_in, int32_t dtype, sp_stream_t stream);
int sp_adaptive_avgpool_bwd(const void* dy, const void* x, void* dx, int32_t n, int32_t h, int32_t w_, int32_t c,
                            int32_t oh, int32_t ow, int32_t act_in, int32_t dtype, sp_stream_t stream);
int sp_upsample2_fwd(const void* x, void* y, int32_t n, int32_t h, int32_t w_, int32_t c, int32_t dtype,
                     sp_stream_t stream);
int sp_upsample2_bwd(const void* dy, void* dx, int32_t n, int32_t h, int32_t w_, int32_t c, int32_t dtype,
                     sp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * SAGAN attention core (torch.bmm + softmax + torch.bmm at models.py:262-270): o = softmax(q k^T) v, no
 * 1/sqrt(d) scale.  q [b][n][d], k [b][nk][d], v [b][nk][dv], o [b][n][dv]; lse [b][n] fp32 saved for backward.
 * Backward: dk_f32 / dv_f32 are fp32 scratch of sp_attention_bwd_slabs() slabs of [b][nk][d] resp. [b][nk][dv] floats (one
 * partial sum per query block - 64 queries on the bf16 MFMA path, 32 or 16 on the fp32 path - added in block order by the
 * call: no atomics, no initialisation needed); dk / dv receive the result.
 * ---------------------------------------------------------------------------------------------- */
int sp_attention_bwd_slabs(int32_t n, int32_t nk, int32_t d, int32_t dv, int32_t dtype, int64_t* slabs_out);
int sp_attention_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int32_t batch, int32_t n,
                     int32_t nk, int32_t d, int32_t dv, int32_t dtype, sp_stream_t stream);
int sp_attention_bwd(const void* q, const void* k, const void* v, const void* dout, const float* lse, void* dq,
                     float* dk_f32, float* dv_f32, void* dk, void* dv_out, int32_t batch, int32_t n, int32_t nk,
                     int32_t d, int32_t dv, int32_t dtype, sp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Glue: image ingest (NCHW/NHWC 3-channel source -> padded NHWC, optional affine = kornia.normalize at
 * models.py:195-197; scale3/shift3 are HOST pointers to 3 floats), mask application (models.py:78,80,94),
 * activations, attention residual (models.py:274), NCHW-flatten <-> NHWC permutation (models.py:83,208),
 * per-channel sums (bias gradients).
 * ---------------------------------------------------------------------------------------------- */
int sp_ingest_image(const void* src, int32_t src_dtype, int64_t sn, int64_t sc, int64_t sh, int64_t sw, void* dst,
                    int32_t n, int32_t c, int32_t h, int32_t w_, int32_t cp, const float* scale3,
                    const float* shift3, int32_t dtype, sp_stream_t stream);
int sp_ingest_image_bwd(const void* dy, int32_t cp, void* dsrc, int32_t c, int64_t pixels, const float* scale3,
                        int32_t dtype, sp_stream_t stream);
int sp_mask_concat(const void* feat, const float* mask, void* out, int64_t pixels, int32_t c, int32_t cp,
                   int32_t dtype, sp_stream_t stream);
int sp_mask_mul_2d(const void* feat, int32_t ldf, const float* mask, void* out, int32_t ldo, int32_t batch,
                   int32_t k, int32_t dtype, sp_stream_t stream);
int sp_act_fwd(const void* x, void* y, int64_t numel, int32_t act, int32_t dtype, sp_stream_t stream);
/* dz = dy * act'(.) evaluated from the POST-activation y; pitch c -> cp (zero padded). */
int sp_act_bwd(const void* dy, const void* y, void* dz, int64_t pixels, int32_t c, int32_t cp, int32_t act,
               int32_t dtype, sp_stream_t stream);
int sp_scale_add(const void* a, const void* b, const float* g, void* y, int64_t numel, int32_t dtype,
                 sp_stream_t stream);
/* y[row][c] = (x[row][c] - bias[c]) * (sig_a[0] * sig_b[1]) + bias[c]: the output of a spectral-normalised layer re-expressed under
 * the sigma of ANOTHER forward of the same weights - conv(x, W / sigma_b) + b from conv(x, W / sigma_a) + b without running the layer
 * again.  The generator's masked-feature mappings (models.py:78-94) see the same pyramid, masks and weight_orig in both generator
 * forwards of a step (model_wrapper.py:147-151,168-172); only the power iteration has advanced.  sig_a / sig_b: DEVICE pointers to
 * the {sigma, 1 / sigma} pairs in the two sp_sn_forward scratches; rows x c elements, pitches ldx / ldy; bias may be NULL. */
int sp_rescale_bias(const void* x, void* y, int64_t rows, int32_t c, int32_t ldx, int32_t ldy, const float* bias, const float* sig_a,
                    const float* sig_b, int32_t dtype, sp_stream_t stream);
/* dg[0] = <dy, a>: per-block partial sums in `partials` (512 floats of scratch), added in block order by a second kernel */
int sp_scale_add_bwd(const void* dy, const void* a, const float* g, void* da, float* dg, float* partials, int64_t numel,
                     int32_t dtype, sp_stream_t stream);
int sp_permute_chw_hwc(const void* src, void* dst, int32_t batch, int32_t c, int32_t hw, int32_t to_hwc,
                       int32_t dtype, sp_stream_t stream);
/* out[c] = sum over pixels; partials: 512 * c floats of scratch (per-block partial rows, added in block order) */
int sp_channel_sum(const void* x, int32_t ld, int64_t pixels, int32_t c, float* out, float* partials, int32_t dtype,
                   sp_stream_t stream);
int sp_f64_to_f32(const double* src, float* dst, int32_t n, sp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Discriminator head (models.py:149-155; pred is the (B,B,F) tensor the reference produces) and the
 * losses (lossfunction.py:137,164 LSGAN; :31-68 semantic reconstruction, one call per pyramid level,
 * accumulating into acc; :92-110 diversity).  gout = device pointer to the upstream gradient scalar.
 * sp_dhead_bwd: scratch = `batch` floats (per-sample sums handed from the first kernel to the second; no initialisation; the
 * call keeps no state of its own, so it is re-entrant per stream); demb [num_classes][f] is zeroed by the call.
 * ---------------------------------------------------------------------------------------------- */
int sp_dhead_fwd(const void* x, int32_t ldx, const float* emb_sn, const int64_t* cls, const float* wc,
                 const float* bc, float* pred, int32_t batch, int32_t f, int32_t dtype, sp_stream_t stream);
int sp_dhead_bwd(const float* dpred, const void* x, int32_t ldx, const float* emb_sn, const int64_t* cls,
                 const float* wc, void* dx, int32_t lddx, float* demb, int32_t num_classes, float* dwc, float* dbc,
                 float* scratch, int32_t batch, int32_t f, int32_t dtype, sp_stream_t stream);
int sp_sqerr_loss_fwd(const float* p, int64_t numel, float target, double* acc_tmp, float* loss, sp_stream_t stream);
int sp_sqerr_loss_bwd(const float* p, int64_t numel, float target, const float* gout, float* dp, sp_stream_t stream);
int sp_rec_loss_fwd(const void* real, int32_t ld_real, const void* fake, int32_t ld_fake, const float* mask,
                    int32_t n, int32_t h, int32_t w_, int32_t c, double* acc, int32_t dtype, sp_stream_t stream);
int sp_rec_loss_bwd(const void* real, int32_t ld_real, const void* fake, int32_t ld_fake, const float* mask,
                    const float* gout, void* dfake, int32_t ld_dfake, int32_t n, int32_t h, int32_t w_, int32_t c,
                    int32_t dtype, sp_stream_t stream);
int sp_div_loss_fwd(const void* img, int64_t half_elems, const float* z, int64_t half_z, double* acc_tmp,
                    float* out2, int32_t dtype, sp_stream_t stream);
int sp_div_loss_bwd(const void* img, int64_t half_elems, const float* fwd_out2, const float* gout, void* dimg,
                    int32_t dtype, sp_stream_t stream);
/* One-launch forms of the loss plumbing (a small launch costs ~5 us of queue time whatever it computes; the three generator losses
 * used to take 36 launches).  `weight`: the factor the caller multiplies the loss by (model_wrapper.py:183-186's w_rec / w_div) - the
 * value written is weight * loss and the backward forms scale by it, so no elementwise launch is needed around them.  `acc`: fp64
 * accumulators (1 for the reconstruction loss, 2 for the diversity loss) that are ZERO on entry and left zero: allocate and clear
 * once, pass to every call of one stream.
 * sp_rec_loss_fwd_levels / _bwd_levels: all pyramid levels (<= 8) of lossfunction.py:31-68 in one launch each; a level with
 * h = w_ = 1 is a 2-D (fully connected) level with c features per row.  sp_sqerr_loss_fwd itself runs as one block up to 2^18 elements. */
typedef struct sp_rec_level {
    const void* real; const void* fake; const float* mask;
    void* dfake;                                   /* backward only */
    int32_t ld_real, ld_fake, ld_dfake, n, h, w_, c, reserved_;
} sp_rec_level;
int sp_rec_loss_fwd_levels(const sp_rec_level* levels, int32_t n_levels, double* acc, float* loss, float weight,
                           int32_t dtype, sp_stream_t stream);
int sp_rec_loss_bwd_levels(const sp_rec_level* levels, int32_t n_levels, const float* gout, float weight,
                           int32_t dtype, sp_stream_t stream);
int sp_div_loss_fwd_w(const void* img, int64_t half_elems, const float* z, int64_t half_z, double* acc,
                      float* out2, float weight, int32_t dtype, sp_stream_t stream);

/* fp8 (OCP e4m3) helpers of the SP_F8 convolution path (BASELINE.json config 5).
 * sp_quantize_fp8: q[i] = e4m3(sat(x[i] * inv_scale[0])) for a bf16 / fp32 tensor of `numel` elements (numel % 16 == 0);
 *   amax (may be NULL): max|x| merged into amax[0].
 * sp_pack_weight_fp8: conv weight [cout][cin][3][3] fp32 (OIHW, the frozen VGG-16 filters) -> e4m3 [cout][9][cin_p] in the
 *   forward packing of sp_conv2d_igemm with ONE scale per output channel, w_scale[co] = max|w[co]| / 448 (pad channels zero).
 * sp_fp8_update_scales: delayed scaling - for each of n slots: if amax[i] > 0: scale[i] = margin * amax[i] / 448,
 *   inv_scale[i] = 1 / scale[i]; amax[i] = 0.  One launch, no host sync. */
int sp_quantize_fp8(const void* x, void* q, int64_t numel, const float* inv_scale, float* amax, int32_t dtype, sp_stream_t stream);
int sp_pack_weight_fp8(const float* w, int32_t cout, int32_t cin, int32_t cin_p, void* out, float* w_scale, sp_stream_t stream);
int sp_fp8_update_scales(float* amax, float* scale, float* inv_scale, int32_t n, float margin, sp_stream_t stream);

/* kornia.normalize_min_max(image[None], min_val=-1, max_val=1) of the reference's data pipeline (/root/reference/data.py:53), for a
 * whole batch on the device: y = (hi - lo) * (x - min) / (max - min + eps) + lo in fp32, operation for operation as torch evaluates
 * it (bit-identical).  x, y: [batch][channels][hw] contiguous fp32 (NCHW, what TVF.to_tensor yields); per_channel != 0: min / max per
 * (image, channel) plane - kornia's definition (x.view(B, C, -1).min(-1)); 0: per image over all channels.  eps: kornia uses 1e-6. */
int sp_minmax_normalize(const float* x, float* y, int32_t batch, int32_t channels, int64_t hw, float lo, float hi, float eps,
                        int32_t per_channel, sp_stream_t stream);

/* A batch of training masks generated on the device: misc.get_masks_for_training
 * (/root/reference/misc.py:13-68) for every sample of the batch in one launch - stage ~ choice([0..6, 0, 1]) counted from the deep
 * end, with probability p_random_mask (reference: 0.3) and 0 < stage < 6 a 0/1 shape map on the next finer level expanded
 * nearest-neighbour to all finer levels, everything deeper than the stage zero, exact 0.0f / 1.0f.  The shape map holds 1 - 4
 * shapes of the four kinds skimage.draw.random_shapes draws (rectangle, circle, triangle, ellipse), each inside a random bounding box.  The seven outputs are the
 * reference's list order, fp32, [batch][1][S][S] contiguous / [batch][4096] / [batch][365].  All randomness is integer
 * arithmetic on splitmix64(seed, sample, draw): the same (seed, batch) gives the same masks on every device, and the oracle
 * restates the generator bit for bit (tests/test_gpu_next_rows.py). */
int sp_training_masks(float* m128, float* m64, float* m32, float* m16, float* m8, float* m4096, float* m365,
                      int32_t batch, uint64_t seed, float p_random_mask, sp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Multi-tensor Adam: torch.optim.Adam (main.py:64-65; .step() at model_wrapper.py:162,190) for every parameter of
 * a network in one launch.  The host splits the fp32 tensors into chunks (<= 65536 elements each); step_size =
 * lr / (1 - beta1^step) and inv_sqrt_bc2 = 1 / sqrt(1 - beta2^step) are per parameter (torch keeps a step count
 * per parameter).  amsgrad / maximize are not supported (the reference uses the defaults).
 * ---------------------------------------------------------------------------------------------- */
typedef struct sp_adam_chunk {
    float* p;              /* parameter (updated in place) */
    const float* g;        /* gradient */
    float* m;              /* exp_avg (updated in place) */
    float* v;              /* exp_avg_sq (updated in place) */
    int32_t n;             /* elements in this chunk */
    float step_size;
    float inv_sqrt_bc2;
    float reserved;
} sp_adam_chunk;
/* hyper-parameters are doubles: torch derives (1 - beta) from the Python float and only then rounds to fp32 */
int sp_adam_multi(const sp_adam_chunk* chunks_dev, int32_t n_chunks, double beta1, double beta2, double eps,
                  double weight_decay, sp_stream_t stream);

/* x[i] *= factor for an fp32 buffer (16-byte aligned), in place: takes the static loss scale of the SP_F16 storage mode off a
 * network's flat gradient buffer before the optimizer step (the activation gradients of this network - means over 5e4 ... 5e6
 * elements - would be subnormal in fp16 without it; the reference trains in fp32, model_wrapper.py:160,188). */
int sp_scale_f32(float* x, int64_t numel, float factor, sp_stream_t stream);
int sp_scale_f32_dev(float* x, int64_t numel, const float* factor_dev, sp_stream_t stream);   /* factor_dev[0] read on the device */

/* Dynamic loss scaling of the SP_F16 storage mode, entirely on the device (no host sync, graph-safe; torch.cuda.amp.GradScaler's
 * policy - the reference trains in fp32 and needs none, model_wrapper.py:160-162,188-190).  fp16 activation gradients overflow to
 * inf above 65504 and one such step would poison Adam's moments for good, so per optimizer step:
 *   sp_check_finite        found[0] = 1 if x holds any inf / NaN (x: a network's flat fp32 gradient buffer, 16-byte aligned; found is
 *                          only ever set here - sp_loss_scale_update clears it)
 *   sp_adam_multi_guarded  sp_adam_multi that does NOTHING when skip_if_nonzero[0] != 0 (NULL: plain sp_adam_multi)
 *   sp_loss_scale_update   state = {scale, 1 / scale, clean steps, found, skipped steps}: found -> scale *= backoff (>= 1), clean = 0,
 *                          skipped += 1; else clean += 1 and after `interval` clean steps scale *= growth (<= 2^24); found = 0. */
int sp_check_finite(const float* x, int64_t numel, float* found, sp_stream_t stream);
int sp_adam_multi_guarded(const sp_adam_chunk* chunks_dev, int32_t n_chunks, double beta1, double beta2, double eps,
                          double weight_decay, const float* skip_if_nonzero, sp_stream_t stream);
int sp_loss_scale_update(float* state, float growth, float backoff, int32_t interval, sp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SEMPYR_H */
