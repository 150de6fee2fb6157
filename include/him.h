/*
 * him.h -- C ABI of libhim_hip.so: the MI355X (gfx950 / CDNA4) kernels under the mask2image
 * (layout-to-image GAN) training hot path of xcyan/neurips18_hierchical_image_manipulation.
 *
 * The reference has NO native/FFI boundary (it is 100 % Python on stock torch.nn modules); the
 * boundary this header defines sits directly beneath the reference's Python operator surface.  Each
 * entry point names the reference call sites it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - plain C, `extern "C"`; raw DEVICE pointers + sizes; no torch / hip types in signatures
 *     (`stream` is a hipStream_t passed as void*; NULL = the null stream).
 *   - every call is asynchronous on `stream`, never synchronises, never allocates: the caller owns
 *     all memory, including the scratch `ws` whose size the matching *_ws() query returns.
 *   - all tensors are fp32, NCHW, contiguous unless a (ptr, total-channels, first-channel) "channel
 *     slice" triple says otherwise.
 *   - return 0 on success, a negative HIM_E_* code on failure; him_last_error() (thread local) says why.
 */
#ifndef HIM_H_
#define HIM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HIM_OK 0
#define HIM_E_INVALID (-1)   /* bad descriptor / argument */
#define HIM_E_WORKSPACE (-2) /* ws too small */
#define HIM_E_LAUNCH (-3)    /* HIP launch / runtime error */
#define HIM_E_UNSUPPORTED (-4)

/* activations fused into epilogues / norm kernels */
#define HIM_ACT_NONE 0
#define HIM_ACT_RELU 1  /* nn.ReLU        (models/Pix2Pix_NET.py:70) */
#define HIM_ACT_LRELU 2 /* nn.LeakyReLU(0.2) (models/Discriminator_NET.py:73) */
#define HIM_ACT_TANH 3  /* nn.Tanh        (models/Pix2Pix_NET.py:91) */
#define HIM_ACT_SIGMOID 4 /* nn.Sigmoid    (models/MaskTwoStreamConvSwitch_NET.py:22,206: object-mask head) */

#define HIM_PAD_ZERO 0    /* Conv2d(padding=p) */
#define HIM_PAD_REFLECT 1 /* nn.ReflectionPad2d(p) followed by Conv2d(padding=0) */

const char* him_version(void);
const char* him_arch(void); /* "gfx950" */
const char* him_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Conv2d family: fp32-MFMA (v_mfma_f32_32x32x2_f32) implicit GEMM, LDS-staged tiles.
 * Replaces nn.Conv2d (+ the nn.ReflectionPad2d in front of it, + bias, + the activation behind it)
 * at models/Pix2Pix_NET.py:74,78-79,91; models/layer_util.py:333-378 (ResnetBlock);
 * models/Discriminator_NET.py:69-93; models/layer_util.py:380-411 (Vgg19 conv3+ReLU).
 * ------------------------------------------------------------------------------------------- */
/* ---------------------------------------------------------------------------------------------
 * Algorithm selection.  The library holds NO mutable global state and reads NO environment variable on any compute or
 * size-query path: which kernel family a descriptor runs on -- and therefore its workspace size, its panel layout and
 * its fp32 summation order -- is a pure function of the descriptor, whose `algo` member carries the caller's overrides.
 * A zero-filled HimAlgo selects the defaults (what `him_algo_resolve` writes back); every entry point is re-entrant
 * across host threads, also with different HimAlgo values (tests/test_ops_gpu.py::test_library_is_reentrant).
 * Use ONE HimAlgo for the size query, the panel build and the launches of a layer.
 * (torch has no counterpart: cuDNN / MIOpen pick algorithms behind `torch.backends.cudnn.benchmark`; reference call
 * sites = every nn.Conv2d / nn.ConvTranspose2d of the path, see "Conv2d family" below.)
 * ------------------------------------------------------------------------------------------- */
#define HIM_TILE_DEFAULT 0   /* 64x128 (MxN) tiles: best inside the multi-stream training step */
#define HIM_TILE_128x128 1
#define HIM_TILE_128x128_8W 2
#define HIM_TILE_128x256 3   /* batched Winograd GEMM only */
#define HIM_TILE_64x128 4
#define HIM_TILE_64x64 5
#define HIM_TILE_MIXED 6     /* 128x128 when the launch has >= 512 (or 200..256) such tiles, else 128x64 */
#define HIM_TILE_128x64 7

#define HIM_ALGO_NO_SPLITK (1u << 0)        /* no split-K of few-tile conv / data-gradient launches */
#define HIM_ALGO_NO_DFOLD (1u << 1)         /* reflect-pad-1 3x3 data gradient through the padded gradient + fold pass */
#define HIM_ALGO_WINO_PADDED_DGRAD (1u << 2)/* Winograd reflect data gradient on the padded grid instead of folded border tiles */
#define HIM_ALGO_NO_SMALL_WIN (1u << 3)     /* tiny-M "same"-window weight gradient kernel off */
#define HIM_ALGO_NO_FEWOUT_TILED (1u << 4)  /* LDS-tiled VALU kernel for 2..4 output channels off */
#define HIM_ALGO_NO_FEWIN_TILED (1u << 5)   /* LDS-tiled VALU kernel for <= 4 reduction channels off */
#define HIM_ALGO_NO_FEWCH_MFMA (1u << 6)    /* MFMA weight gradient of the few-channel 5x5 / 7x7 layers off */
#define HIM_ALGO_GENERIC_CONV (1u << 7)     /* generic implicit-GEMM kernels instead of the buffer-load fast path */
#define HIM_ALGO_NO_RESBLOCK_FUSED (1u << 8)/* him_resblock_supported() answers 0 */
#define HIM_ALGO_NO_BGEMM (1u << 10)        /* batched Winograd GEMMs on the conv kernel instead of the LDS-DMA GEMM kernel */
#define HIM_ALGO_NO_ONEHOT_RLE (1u << 11)   /* one-hot stem weight gradient per pixel (round-1 kernel) instead of per run of equal class */
#define HIM_ALGO_NO_FEWIN_FOLD (1u << 12)    /* reflection-padded few-channel data gradient (generator head) through the padded
                                               gradient + reflect_fold pass instead of the fold inside the tiled kernel */
#define HIM_ALGO_NO_FEWIN_REFLECT (1u << 14) /* forward of a reflection-padded conv with <= 4 input channels (the dense channels of a
                                               generator stem: 3 -> 64, 7x7) on the generic implicit-GEMM kernel (K = 147 is too
                                               ragged for it: 0.55 ms at C2) instead of the tiled few-channel kernel (0.3 ms) */
#define HIM_ALGO_NO_WINO_FUSED2 (1u << 15)   /* fused Winograd launches on the round-2 kernel (one workgroup per spatial block, all
                                               waves producer + consumer) instead of the persistent wave-specialised kernel
                                               (csrc/him_wino_fused2.inc, round 6) */
#define HIM_ALGO_NO_BGEMM_PERSISTENT (1u << 16) /* F(4x4) GEMMs with K <= 256 on one workgroup per tile instead of the persistent
                                               form of the batched GEMM (csrc/him_bgemm.inc: bgemm_p_kernel, round 6) */
#define HIM_ALGO_WINO4_TRAIN_FWD (1u << 13)  /* OPT-IN, reduced-work variant: the FORWARD of trainable 3x3 s1 p1
                                               layers with >= wino4_min_c channels (the ResnetBlock stack) as Winograd
                                               F(4x4,3x3) -- 1.78x fewer multiplies than F(2x2), ~3e-6 instead of 5e-7 relative
                                               rounding per convolution; data / weight gradients stay F(2x2).  Never the
                                               default: reported as its own bench line (bench.py --variant) */
#define HIM_ALGO_FROZEN_WEIGHTS (1u << 9)   /* the layer's weights never change (VGG19 of the perceptual loss,
                                               models/layer_util.py:380-411): forward / data gradient may use Winograd
                                               F(4x4,3x3), whose 36-position panel is built once per run */

typedef struct HimAlgo {
  int wino_min_c;       /* 3x3 s1 p1 layers with Cin and Cout >= this run as separate-transform Winograd F(2x2,3x3);
                           0 = default (256); < 0: EVERY Winograd form off (all convolutions in the direct form) */
  int wino_fused_min_c; /* lower end of the fused single-launch Winograd kernel's channel range; 0 = default (64); < 0: off */
  int wino_fused_max_c; /* upper end; 0 = default (255) */
  int wino4_min_c;      /* FROZEN_WEIGHTS layers with Cin and Cout >= this AND at least 64 output tiles of 4x4 in the batch
                           (B*H*W >= 1024) run as F(4x4,3x3); 0 = default (128: VGG conv2_2 upwards; 256 until
                           the last session of round 5 -- in the step -0.34 ms, profiles/r05_ab_log.txt); < 0: off */
  int ksplit_max;       /* cap of the split-K factor; 0 = default (4; 8 until the last session of round 5: -0.24 ms in the step) */
  int tile_wb, tile_nb; /* HIM_TILE_*: batched Winograd GEMMs / direct-form convolutions */
  int wino_tblock;      /* threads per workgroup of the Winograd transform kernels: 64 (default), 128, 256 */
  int wgrad_splits;     /* fast weight-gradient kernel: split count; 0 = automatic */
  unsigned disable;     /* HIM_ALGO_* bits */
  int wino_fused_chunk; /* reduction channels per K-chunk of the fused Winograd kernel: 0 = default, 8 = the whole 160 KB of a CU's
                           LDS per workgroup (fastest alone: 0.54 ms at VGG conv1_2), 4 = 80 KB (0.69 ms alone, but the
                           workgroup shares its CU with the other streams' kernels inside the training step) */
  int wgrad_tile;       /* fast weight-gradient kernel: 0 = default (128x128 where M > 64 and C % 128 == 0: 74 KB of LDS, two
                           workgroups per CU), 1 = 64-row tiles (64x128: 55 KB), 2 = 64x64 tiles (37 KB: four per CU) */
} HimAlgo;
/* out = in with every 0 replaced by the default it selects (in == NULL: all defaults). */
void him_algo_resolve(const HimAlgo* in, HimAlgo* out);
/* Development aid for tools/: zero-fills *a, then applies the HIM_* environment overrides (HIM_NO_WINOGRAD,
 * HIM_WINO_MIN_C, HIM_NO_WINO_FUSED, HIM_WINO_FUSED_MIN_C / _MAX_C, HIM_WINO4_MIN_C, HIM_KSPLIT_MAX, HIM_NO_SPLITK,
 * HIM_GCONV_TILE[_WB|_NB], HIM_WINO_TBLOCK, HIM_WGRAD_SPLITS, HIM_NO_DFOLD, HIM_WINO_PADDED_DGRAD, HIM_NO_SMALL_WIN,
 * HIM_NO_FEWOUT_TILED, HIM_NO_FEWIN_TILED, HIM_NO_FEWCH_MFMA, HIM_GENERIC_CONV, HIM_NO_RESBLOCK_FUSED, HIM_NO_BGEMM, HIM_NO_ONEHOT_RLE,
 * HIM_NO_WINO_FUSED2, HIM_NO_BGEMM_PERSISTENT).  The ONLY place
 * the library reads the environment; nothing else calls it. */
void him_algo_from_env(HimAlgo* a);

typedef struct HimConv2d {
  int B, Cin, H, W;   /* input  (B,Cin,H,W) */
  int Cout, KH, KW;   /* weight (Cout,Cin,KH,KW) */
  int stride, pad;    /* symmetric stride / padding */
  int pad_mode;       /* HIM_PAD_ZERO | HIM_PAD_REFLECT */
  int OH, OW;         /* output (B,Cout,OH,OW); must equal (H+2p-K)/s+1 */
  int act;            /* epilogue activation applied by *_fwd */
  float slope;        /* LeakyReLU negative slope */
  HimAlgo algo;       /* kernel selection overrides; zero-filled = defaults */
} HimConv2d;

/* y = act(conv(x, w) + bias); bias may be NULL.  ws holds the tap-major regrouped weight tile stream. */
size_t him_conv2d_fwd_ws(const HimConv2d* d);
int him_conv2d_fwd(const HimConv2d* d, const float* x, const float* w, const float* bias, float* y, void* ws,
                   size_t ws_bytes, void* stream);
/* dx = conv^T(dy, w); dy is the gradient w.r.t. the PRE-activation output (see him_act_bwd). */
size_t him_conv2d_bwd_data_ws(const HimConv2d* d);
int him_conv2d_bwd_data(const HimConv2d* d, const float* dy, const float* w, float* dx, void* ws,
                        size_t ws_bytes, void* stream);
/* dw (+)= x (*) dy, dbias (+)= sum dy (dbias may be NULL).  accumulate!=0 adds into dw/dbias
 * (this is how the flat gradient arena receives several contributions per step). */
size_t him_conv2d_bwd_weight_ws(const HimConv2d* d);
int him_conv2d_bwd_weight(const HimConv2d* d, const float* x, const float* dy, float* dw, float* dbias,
                          int accumulate, void* ws, size_t ws_bytes, void* stream);

/* nn.ConvTranspose2d(k3,s2,p1,op1)+bias(+act): models/Pix2Pix_NET.py:89-90,185-188.
 * Computed as stride-phase sub-convolutions (no zero insertion). weight is (Cin,Cout,KH,KW). */
typedef struct HimDeconv2d {
  int B, Cin, H, W;
  int Cout, KH, KW;
  int stride, pad, out_pad;
  int OH, OW; /* (H-1)*s - 2p + K + out_pad */
  int act;
  float slope;
  HimAlgo algo;
} HimDeconv2d;
size_t him_deconv2d_fwd_ws(const HimDeconv2d* d);
int him_deconv2d_fwd(const HimDeconv2d* d, const float* x, const float* w, const float* bias, float* y,
                     void* ws, size_t ws_bytes, void* stream);
size_t him_deconv2d_bwd_data_ws(const HimDeconv2d* d);
int him_deconv2d_bwd_data(const HimDeconv2d* d, const float* dy, const float* w, float* dx, void* ws,
                          size_t ws_bytes, void* stream);
size_t him_deconv2d_bwd_weight_ws(const HimDeconv2d* d);
int him_deconv2d_bwd_weight(const HimDeconv2d* d, const float* x, const float* dy, float* dw, float* dbias,
                            int accumulate, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * box2mask building blocks (second hot path, SURVEY 8 row a18: models/MaskTwoStreamConvSwitch_NET.py,
 * models/layer_util.py:128-250 ConvResnetBlock / DeconvResnetBlock, models/mask_losses.py).
 *   batchnorm: nn.BatchNorm2d(affine=True) of get_norm_layer('batch') (layer_util.py:19-21), training mode = batch
 *     statistics + running-statistics update (momentum, unbiased variance), eval mode = running statistics; fused with
 *     the activation behind it and an optional residual add.  gamma/beta may be NULL (affine=False).
 *     bwd: dgamma/dbeta (+)= when accumulate != 0; dx may be NULL.
 *   act_fwd: stand-alone nn.ReLU / LeakyReLU / Tanh / Sigmoid (the blocks start with an activation).
 *   upsample2: nn.Upsample(scale_factor=2, mode='bilinear') (layer_util.py:189,206); align_corners selects the
 *     torch >= 0.4 default (0) or the torch 0.3.1 behaviour (1).
 *   logsoftmax: nn.LogSoftmax(dim=1) over the channel axis (MaskTwoStreamConvSwitch_NET.py:21,196).
 *   masked_nll: MaskReconLoss = NLLLoss2d(ignore_index) with the positions where mask < 0.5 ignored
 *     (mask_losses.py:12-27); out2 = {loss, #valid positions}.   bce_mean: nn.BCELoss() (TwoStreamAE_mask.py:53).
 * -------------------------------------------------------------------------------------------*/
size_t him_batchnorm_ws(int C);
int him_batchnorm_fwd(const float* x, const float* residual, const float* gamma, const float* beta, float* run_mean,
                      float* run_var, float* y, float* save_mean, float* save_rstd, int B, int C, int hw, float eps,
                      float momentum, int training, int act, float slope, void* ws, size_t ws_bytes, void* stream);
int him_batchnorm_bwd(const float* x, const float* gamma, const float* beta, const float* save_mean,
                      const float* save_rstd, const float* dy, float* dx, float* dgamma, float* dbeta, int B, int C,
                      int hw, int training, int act, float slope, int accumulate, void* ws, size_t ws_bytes,
                      void* stream);
int him_act_fwd(const float* x, float* y, size_t n, int act, float slope, void* stream);
int him_upsample2_fwd(const float* x, float* y, int planes, int H, int W, int align_corners, void* stream);
int him_upsample2_bwd(const float* dy, float* dx, int planes, int H, int W, int align_corners, void* stream);
int him_logsoftmax_fwd(const float* x, float* y, int B, int C, int hw, void* stream);
int him_logsoftmax_bwd(const float* y, const float* dy, float* dx, int B, int C, int hw, void* stream);
/* MaskTwoStreamConv_NET.py:213-221 (the generator the parser builds without --no_comb): the context stream's logits gated
 * by the object stream's probability, comb = (1 - p) * ctx + p * obj with p, obj (B,1,hw) broadcast over ctx's C channels
 * (products and sum rounded one by one: torch's values).  bwd: dctx (B,C,hw), dp and dobj (B,1,hw). */
int him_gate_comb_fwd(const float* ctx, const float* p, const float* obj, float* out, int B, int C, int hw, void* stream);
int him_gate_comb_bwd(const float* ctx, const float* p, const float* obj, const float* dout, float* dctx, float* dp,
                      float* dobj, int B, int C, int hw, void* stream);
size_t him_mask_loss_ws(void);
int him_masked_nll_fwd(const float* logp, const float* label, const float* mask, float* out2, int B, int C, int hw,
                       void* ws, size_t ws_bytes, void* stream);
int him_masked_nll_bwd(const float* label, const float* mask, const float* g, const float* count, float* dlogp, int B,
                       int C, int hw, void* stream);
int him_bce_mean_fwd(const float* p, const float* t, size_t n, float* out, void* ws, size_t ws_bytes, void* stream);
int him_bce_mean_bwd(const float* p, const float* t, size_t n, const float* g, float* dp, void* stream);
/* The ADE recipe of box2mask (scripts/train_box2mask_ade.sh: --add_dilated_layers --lr_control):
 *   space_to_batch: y[(b*d+py)*d+px][c][i][j] = x[b][c][i*d+py][j*d+px] (inverse != 0: the reverse copy).  The dilated
 *     bias-free conv3x3 of DilatedResnetBlock (models/layer_util.py:254-293, dilation = padding = d) is the plain pad-1
 *     conv on the d*d phase images, so it runs on the same MFMA / Winograd kernels as every other 3x3 layer.
 *   lr_control: models/Discriminator_NET.py:190-211 on device scalars, out2 = {g_lr, d_lr} (no host read-back). */
int him_space_to_batch(const float* x, float* y, int B, int C, int H, int W, int d, int inverse, void* stream);
int him_lr_control(const float* loss_d_real, const float* loss_d_fake, float margin, float* out2, void* stream);
/* box2mask condition, object half (models/TwoStreamAE_mask.py:127-152): dst[b][c0+c][px] = (c == cls[b]) ? mask[b][px] : 0
 * for c in [0, NC); cls = one class id per sample as a float on the device. */
int him_class_mask(const float* mask, const float* cls, float* dst, int B, int NC, int Ctot, int c0, int hw, void* stream);

/* ---------------------------------------------------------------------------------------------
 * One-hot stems: conv over [one-hot(label) | dense channels] evaluated from the label ids.  `label` is the
 * (B,1,H,W) float id map (data/segmentation_dataset.py:82), channels [0, n_onehot) of x are its one-hot encoding
 * (him_onehot; reference encode_input models/pix2pixHD_condImg_model.py:150-155) and are never read; channels
 * [n_onehot, Cin) are ordinary dense inputs.  Replaces nn.Conv2d(input_nc, ngf, 7) of models/Pix2Pix_NET.py:74-76
 * (GlobalGenerator stem) and :137-140 (two-stream label encoder stem).  stride 1, odd square kernel, pad = K/2
 * (zero or reflect) -- or, round 5, 4x4 stride 2 zero-padded (see below) --, Cout % 16 == 0; *_ws returns 0 when the
 * descriptor is not eligible (use him_conv2d_*).
 * No data gradient: the stem input is data.
 * -------------------------------------------------------------------------------------------*/
size_t him_conv2d_onehot_fwd_ws(const HimConv2d* d, int n_onehot);
int him_conv2d_onehot_fwd(const HimConv2d* d, const float* label, int n_onehot, const float* x, const float* w,
                          const float* bias, float* y, void* ws, size_t ws_bytes, void* stream);
size_t him_conv2d_onehot_bwd_weight_ws(const HimConv2d* d, int n_onehot);
int him_conv2d_onehot_bwd_weight(const HimConv2d* d, const float* label, int n_onehot, const float* x,
                                 const float* dy, float* dw, float* dbias, int accumulate, void* ws, size_t ws_bytes,
                                 void* stream);
/* Round 5.  (1) The same pair for the first PatchGAN convolution of scale 0 -- nn.Conv2d(input_nc, ndf, 4, stride 2,
 * padding 2) of models/Discriminator_NET.py:71-74 applied to cat(one-hot | cond image | real-or-fake image)
 * (pix2pixHD_condImg_model.py:149-152,176-186): 4x4, stride 2, zero padding is eligible too (*_ws != 0); forward = 16
 * table lookups per output pixel, weight gradient in the run-length form (one difference of two prefix sums of a dy row per
 * (run of equal class, tap), output columns ceil((x0 - tw) / 2) .. ceil((x1 - tw) / 2)).  (2) *_dense: the dense channels
 * arrive as their OWN tensor xdense (B, Cin - n_onehot, H, W) (NULL when Cin == n_onehot): the (B, Cin, H, W) concatenation
 * -- 147 MB of one-hot at 512x256 bs 8, written and re-read by three discriminator passes per step -- never exists.
 * Workspace sizes are those of the plain pair. */
int him_conv2d_onehot_fwd_dense(const HimConv2d* d, const float* label, int n_onehot, const float* xdense, const float* w,
                                const float* bias, float* y, void* ws, size_t ws_bytes, void* stream);
int him_conv2d_onehot_bwd_weight_dense(const HimConv2d* d, const float* label, int n_onehot, const float* xdense,
                                       const float* dy, float* dw, float* dbias, int accumulate, void* ws,
                                       size_t ws_bytes, void* stream);
/* (3) The weight gradient in two independently launchable parts, for callers that own two streams: HIM_ONEHOT_PART_IDS = the
 * label-id channels' slice of dw (run-length kernel: LDS-bound, one workgroup per CU, 1.1 ms at the generator stem of
 * 512x256 bs 8), HIM_ONEHOT_PART_DENSE = the dense channels' slice of dw (an MFMA weight gradient) + dbias.  The parts write
 * disjoint elements of dw and use disjoint regions of ws; on two streams they run next to each other (the stem's weight
 * gradient is the LAST kernel chain of the generator's backward, netG.model[1] of models/Pix2Pix_NET.py:74, and what the
 * next step's generator forward waits for).  x: the (B, Cin, H, W) concatenation, or the dense channels alone when
 * x_is_dense != 0.  parts == IDS | DENSE is the one-call form above. */
#define HIM_ONEHOT_PART_IDS 1
#define HIM_ONEHOT_PART_DENSE 2
int him_conv2d_onehot_bwd_weight_part(const HimConv2d* d, const float* label, int n_onehot, const float* x, int x_is_dense,
                                      const float* dy, float* dw, float* dbias, int accumulate, void* ws, size_t ws_bytes,
                                      int parts, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Weight panels.  The MFMA kernels read the weights regrouped (forward: [Cout][Cin/16][KH][KW][16]; data
 * gradient: the transpose, per stride phase).  The plain entry points above rebuild that panel inside the
 * workspace on EVERY launch (75 MB of HBM traffic for a 1024x1024x3x3 ResnetBlock conv); weights only change
 * once per optimizer step, so a trainer builds each panel once after the Adam update (on a side stream, under
 * the other network's backward) and calls the *_panel variants.  torch.nn has no counterpart: cuDNN/MIOpen
 * hide the same transform inside their "find"/workspace logic (reference call sites: every nn.Conv2d /
 * nn.ConvTranspose2d of models/Pix2Pix_NET.py:63-135, models/Discriminator_NET.py:62-125,
 * models/layer_util.py:340-378,400-440).
 *   kind HIM_PANEL_FWD      -> consumed by him_conv2d_fwd_panel      / him_deconv2d_fwd_panel
 *   kind HIM_PANEL_BWD_DATA -> consumed by him_conv2d_bwd_data_panel / him_deconv2d_bwd_data_panel
 * *_panel_bytes returns 0 when that kernel reads the raw weights (Cout <= 4 heads, Cin < 16 stems):
 * use the plain entry point.  Workspace sizes are those of the plain entry points.
 * -------------------------------------------------------------------------------------------*/
/* 3x3 stride-1 pad-1 convs with Cin and Cout >= HimAlgo.wino_min_c channels (default 256) run as Winograd F(2x2,3x3):
 * transforms + ONE batched fp32-MFMA GEMM over the 16 transform positions (2.25x fewer multiplies; fp32 rounding differs
 * from the direct form at the 1e-6 level).  The data and weight gradients of those layers use the same scheme.  Workspace
 * / panel sizes follow the descriptor's HimAlgo: a cached panel belongs to the HimAlgo it was built with. */

/* Stage 2 of the Winograd convolution on its own: c[z][m][n] = sum_k a[z][m][k] * b[z][k][n] for the 16 transform
 * positions z (a: [16][M][K] weight/gradient panels, b: [16][K][N], c: [16][M][N]; K % 16 == 0, N % 128 == 0).
 * This is the launch the roofline in bench.py is measured on (fp32 MFMA, 2*16*M*K*N executed FLOP).  algo: tile_wb is
 * read (NULL = defaults). */
int him_winograd_gemm(const float* a, const float* b, float* c, int M, int K, int N, const HimAlgo* algo, void* stream);

/* ---------------------------------------------------------------------------------------------
 * ResnetBlock  out = x + IN(conv3x3(refpad(relu(IN(conv3x3(refpad(x)))))))  with both InstanceNorms fused into the
 * Winograd transforms of the two convolutions: reference models/layer_util.py:333-378 (ResnetBlock.build_conv_block /
 * forward) with norm_layer = InstanceNorm2d(affine=False) (:19-26), as built by GlobalGenerator
 * (models/Pix2Pix_NET.py:82-84).  Statistics are reduced in the output transform, the normalisation + ReLU is applied on
 * load by the next input transform (the normalised tensor is never written), the tail norm + residual in the second
 * output transform; backward applies the ReLU gate + InstanceNorm backward in the data gradient's output transform.
 * Supported: the separate-transform Winograd range (C above the fused kernel's, C % 128 == 0), even planes >= 4x4 with
 * H*W <= 4096; him_resblock_supported() tells.  panel*: HIM_PANEL_FWD / HIM_PANEL_BWD_DATA panels of the 3x3 reflect-pad
 * conv descriptor (him_conv2d_panel_build).  stat*: [mean(B*C) | rstd(B*C)].  y1 / y2: raw conv outputs (+bias).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  int B, C, H, W;
  float eps;
  HimAlgo algo;
} HimResBlock;
int him_resblock_supported(const HimResBlock* d);
size_t him_resblock_ws(const HimResBlock* d);
size_t him_resblock_bwd_weight_ws(const HimResBlock* d);
int him_resblock_fwd(const HimResBlock* d, const float* x, const void* panel1, const float* bias1, const void* panel2,
                     const float* bias2, float* y1, float* stat1, float* y2, float* stat2, float* out, void* ws,
                     size_t ws_bytes, void* stream);
int him_resblock_bwd_data(const HimResBlock* d, const float* g_out, const float* y1, const float* stat1, const float* y2,
                          const float* stat2, const void* panel1_bwd, const void* panel2_bwd, float* dy2, float* dy1,
                          float* dx, void* ws, size_t ws_bytes, void* stream);
int him_resblock_bwd_weight(const HimResBlock* d, int which, const float* src, const float* stat, const float* dy, float* dw,
                            int accumulate, void* ws, size_t ws_bytes, void* stream);

#define HIM_PANEL_FWD 0
#define HIM_PANEL_BWD_DATA 1
size_t him_conv2d_panel_bytes(const HimConv2d* d, int kind);
/* 1 when the data gradient of `d` reads the SAME panel as its forward (the separate-transform Winograd layers: the batched
 * GEMM reads the forward panel transposed with mirrored transform positions) -- a trainer then keeps one panel per weight:
 * HIM_PANEL_FWD serves him_conv2d_bwd_data_panel / him_resblock_bwd_data as well (a HIM_PANEL_BWD_DATA panel built for
 * such a layer has the same content). */
int him_conv2d_bwd_data_shares_fwd_panel(const HimConv2d* d);
/* Which regrouping the panel of (d, kind) holds: 0 none (raw weights), 1 implicit-GEMM panel, 2 Winograd F(2x2,3x3) for the
 * separate-transform pipeline, 3 the fused Winograd kernel's chunked panel, 4 Winograd F(4x4,3x3) (frozen weights).  A pure
 * function of the descriptor, NOT of the weight alone: one nn.Conv2d weight called with another batch / plane size may need
 * another layout (e.g. VGG conv5_1 of models/layer_util.py:380-411 on the last, smaller batch of an epoch leaves F(4x4)),
 * so a cache of built panels keys on (kind, layout, him_conv2d_panel_bytes, HimAlgo). */
int him_conv2d_panel_layout(const HimConv2d* d, int kind);
int him_conv2d_panel_build(const HimConv2d* d, int kind, const float* w, void* panel, size_t panel_bytes,
                           void* stream);
int him_conv2d_fwd_panel(const HimConv2d* d, const float* x, const void* panel, const float* bias, float* y,
                         void* ws, size_t ws_bytes, void* stream);
int him_conv2d_bwd_data_panel(const HimConv2d* d, const float* dy, const void* panel, float* dx, void* ws,
                              size_t ws_bytes, void* stream);
/* Separate-transform Winograd layers (the 1024-channel ResnetBlock convolutions, models/layer_util.py:333-378): the
 * transformed input V = B^T x B of the forward, [16][Cin][tiles] floats, is also the column operand of the layer's weight
 * gradient dU = dM V^T.  A trainer that keeps it between forward and backward (67 MB per conv at config C2; this build is
 * sized for 288 GB) saves the weight gradient's own transposed input transform, and the weight-gradient GEMM reads both
 * operands K-contiguous.  *_keep_bytes: 0 when the descriptor has no such tensor (use the plain entry points).
 * him_conv2d_fwd_panel_keep == him_conv2d_fwd_panel that leaves V in `keep` (NULL: not kept);
 * him_conv2d_bwd_weight_kept == him_conv2d_bwd_weight with `keep` in place of x (same workspace size). */
size_t him_conv2d_fwd_keep_bytes(const HimConv2d* d);
int him_conv2d_fwd_panel_keep(const HimConv2d* d, const float* x, const void* panel, const float* bias, float* y, float* keep,
                              void* ws, size_t ws_bytes, void* stream);
int him_conv2d_bwd_weight_kept(const HimConv2d* d, const float* keep, const float* dy, float* dw, float* dbias, int accumulate,
                               void* ws, size_t ws_bytes, void* stream);
/* Data gradient GATED by the ReLU that produced this layer's input x: dx[i] = x[i] > 0 ? dgrad(dy)[i] : 0  -- the
 * activation backward of the PREVIOUS layer (models/layer_util.py:380-411 Vgg19: conv -> ReLU -> conv chains) done in
 * this launch's epilogue instead of a separate pass over dx.  `panel` (him_conv2d_panel_build, kind BWD_DATA) or `w`. */
int him_conv2d_bwd_data_gated(const HimConv2d* d, const float* dy, const float* w, const void* panel, const float* x,
                              float* dx, void* ws, size_t ws_bytes, void* stream);
size_t him_deconv2d_panel_bytes(const HimDeconv2d* d, int kind);
int him_deconv2d_panel_build(const HimDeconv2d* d, int kind, const float* w, void* panel, size_t panel_bytes,
                             void* stream);
int him_deconv2d_fwd_panel(const HimDeconv2d* d, const float* x, const void* panel, const float* bias, float* y,
                           void* ws, size_t ws_bytes, void* stream);
int him_deconv2d_bwd_data_panel(const HimDeconv2d* d, const float* dy, const void* panel, float* dx, void* ws,
                                size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * InstanceNorm2d(affine=False, eps) fused with the activation behind it and, for the second half of a
 * ResnetBlock, the residual add.  Replaces get_norm_layer('instance') models/layer_util.py:19-26 and
 * the `x + conv_block(x)` of models/layer_util.py:376-378.  Per-(n,c) reductions are wave64 shuffles.
 *   fwd: mean/rstd[planes] written; y = act((x-mean)*rstd) (+ residual when residual!=NULL)
 *   bwd: dz = dy*act'(xhat); dx = rstd*(dz - mean(dz) - xhat*mean(dz*xhat))
 * ------------------------------------------------------------------------------------------- */
int him_instnorm_fwd(const float* x, const float* residual, float* y, float* mean, float* rstd,
                     int planes, int hw, float eps, int act, float slope, void* stream);
int him_instnorm_bwd(const float* x, const float* mean, const float* rstd, const float* dy, float* dx,
                     int planes, int hw, int act, float slope, void* stream);

/* Conv2d -> InstanceNorm2d(affine=False) [-> ReLU / LeakyReLU] [+ residual] as ONE call: the encoder / PatchGAN block of
 * models/Pix2Pix_NET.py:74-92 and models/Discriminator_NET.py:64-96 (`nn.Conv2d`, `norm_layer`, activation in sequence).
 *   y_raw (B,Cout,OH,OW) = conv(x) + bias (kept: the InstanceNorm backward reads it), z = act((y_raw - mean) * rstd)
 *   [+ residual], mean / rstd [B*Cout].  `d->act` must be HIM_ACT_NONE (the activation follows the norm).
 * Few-tile layers that the forward launches with split-K (him_conv2d_in_act_fused(d) == 1: the PatchGAN blocks, the
 * generator's last down-convolutions) hand the raw split-K slabs to the InstanceNorm kernel, which sums them in the finish
 * pass's order, adds the bias, writes y_raw, reduces the statistics and writes z in one pass: the split-K finish launch
 * and one read of y_raw disappear, every output is bit-identical to him_conv2d_fwd + him_instnorm_fwd.  All other
 * descriptors run exactly those two launches behind this entry.  `panel` (HIM_PANEL_FWD) or `w`; workspace =
 * him_conv2d_fwd_ws(d).  Backward: him_instnorm_bwd(y_raw, ...) then the conv's him_conv2d_bwd_*. */
int him_conv2d_in_act_fused(const HimConv2d* d);
int him_conv2d_in_act_fwd(const HimConv2d* d, const float* x, const float* w, const void* panel, const float* bias,
                          float* y_raw, const float* residual, float* z, float* mean, float* rstd, float eps, int act,
                          float slope, void* ws, size_t ws_bytes, void* stream);

/* dz = dy * act'(.) expressed through the activation OUTPUT y (ReLU/LeakyReLU/Tanh epilogues). */
int him_act_bwd(const float* y, const float* dy, float* dz, size_t n, int act, float slope, void* stream);
/* out = a + b  (gradient fan-in, residual adds) */
int him_add(const float* a, const float* b, float* out, size_t n, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Input encoding: models/pix2pixHD_condImg_model.py:144-174 (+ :285-291 get_edges,
 * models/pix2pixHD_condImgColor_model.py:147-186).  All write into a channel slice
 * [c0, c0+n) of a (B,Ctot,H,W) destination so torch.cat never materialises.
 * ------------------------------------------------------------------------------------------- */
int him_onehot(const float* label, float* dst, int B, int label_nc, int Ctot, int c0, int hw, void* stream);
/* nn.AvgPool2d(3, 2, 1, count_include_pad=False) applied to one-hot(label) (models/Discriminator_NET.py:31-32,47-58 on the
 * label channels of pix2pixHD_condImg_model.py:176-186's discriminator input) evaluated from the ids: class counts of the
 * 3x3 window / valid pixels, written to channels [c0, c0 + label_nc) of a (B, Ctot, OH, OW) destination.  Bit-identical to
 * him_avgpool3s2_fwd(him_onehot(label)); the full-resolution one-hot tensor is never written. */
int him_onehot_pool3s2(const float* label, float* dst, int B, int label_nc, int Ctot, int c0, int H, int W, int OH, int OW,
                       void* stream);
/* Compact inputs: label / instance maps travel as uint8 ids (1 byte per pixel over PCIe instead of the
 * float maps of data/segmentation_dataset.py:82) and are widened on the device; get_masked_image of
 * data/base_dataset.py:342-357 for a whole batch: bbox[b] = (wmin, hmin, wmax, hmax) floats on the device;
 * mask (B,1,H,W), masked_object = mask*image, masked_context = (1-mask)*image + mask*cls2fill (any output may be NULL). */
int him_u8_to_f32(const unsigned char* src, float* dst, size_t n, void* stream);
int him_masked_image(const float* image, const float* bbox, float* mask, float* masked_object, float* masked_context,
                     int B, int C, int H, int W, float cls2fill, void* stream);
int him_edges(const float* inst, float* dst, int B, int H, int W, int Ctot, int c0, void* stream);
/* per-image masked mean colour, times noise (B,3) (NULL = 1), clamped to [-1,1]  -> emb (B,3) */
int him_masked_mean(const float* image, const float* obj_mask, const float* noise, float* emb, int B,
                    int hw, void* stream);
/* dst[:, c0:c0+3] = emb[b,c] * mask[b]   (encode_global_embedding) */
int him_tile_embed(const float* emb, const float* mask, float* dst, int B, int Ctot, int c0, int hw,
                   void* stream);

/* dst[:, cd0:cd0+n] = f(mask) * src[:, cs0:cs0+n]; mask (B,1,HW) or NULL;
 * mask_mode 0: f=1, 1: f=mask, 2: f=1-mask.  Implements torch.cat / slicing / `x*mask.repeat(...)`
 * (pix2pixHD_condImg_model.py:165-166,176-186,229-233) and their backward passes. */
int him_copy_channels(const float* src, int Csrc, int cs0, float* dst, int Cdst, int cd0, int n, int B,
                      int hw, const float* mask, int mask_mode, int accumulate, void* stream);
/* out = (1-m)*a + m*b with a, b channel slices: output gate models/Pix2Pix_NET.py:96-99,242-245 and the
 * two-stream fusion :215-217.  m is (B,1,HW). */
int him_blend(const float* a, int Ca, int ca0, const float* b, int Cb, int cb0, const float* m, float* out,
              int B, int C, int hw, void* stream);

/* nn.AvgPool2d(3, stride=2, padding=1, count_include_pad=False): models/Discriminator_NET.py:31-32. */
int him_avgpool3s2_fwd(const float* x, float* y, int planes, int H, int W, int OH, int OW, void* stream);
int him_avgpool3s2_bwd(const float* dy, float* dx, int planes, int H, int W, int OH, int OW, void* stream);
/* nn.MaxPool2d(k, k): VGG19 2x2 pools (models/layer_util.py:383) and the 2^n mask pool
 * (models/Pix2Pix_NET.py:134). bwd routes to the first maximum in row-major window order. */
int him_maxpool_fwd(const float* x, float* y, int planes, int H, int W, int k, void* stream);
int him_maxpool_bwd(const float* x, const float* dy, float* dx, int planes, int H, int W, int k, void* stream);
/* the same for a pool that follows a ReLU whose backward is folded in: windows with a non-positive maximum pass nothing */
int him_maxpool_relu_bwd(const float* x, const float* dy, float* dx, int planes, int H, int W, int k, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Losses: models/losses.py:40-50 (LSGAN MSE vs a constant), nn.L1Loss pairs of
 * pix2pixHD_condImg_model.py:235-251 and models/losses.py:75-82.  Deterministic two-stage reductions.
 *   *_fwd : out[0] = mean(...)     (device scalar)
 *   *_bwd : d(in) (+)= g[0] * d mean / d in     (g = device scalar upstream gradient)
 * ------------------------------------------------------------------------------------------- */
size_t him_reduce_ws(size_t n);
int him_l1_mean_fwd(const float* a, const float* b, size_t n, float* out, void* ws, size_t ws_bytes,
                    void* stream);
/* him_l1_mean_bwd: `accumulate` bit 0 = add into da, bit 1 = gate by a > 0 (a is a ReLU output whose activation
 * backward is folded into this pass) */
int him_l1_mean_bwd(const float* a, const float* b, size_t n, const float* g, float* da, int accumulate,
                    void* stream);
/* Several L1 terms at once (feature matching: 12 pairs, VGG: 5): out[i] = mean|a[i] - b[i]|, da[i] = g[i] * d/da[i]
 * (NULL da[i]: skipped).  a / b / n / da are HOST arrays of npairs entries; out and g are device vectors.  Work split and
 * summation order per pair are those of him_l1_mean_fwd / _bwd (bit-identical results), three launches per 16 pairs. */
size_t him_l1_multi_ws(int npairs);
int him_l1_multi_fwd(const float* const* a, const float* const* b, const size_t* n, int npairs, float* out, void* ws,
                     size_t ws_bytes, void* stream);
int him_l1_multi_bwd(const float* const* a, const float* const* b, const size_t* n, int npairs, const float* g,
                     float* const* da, int accumulate, void* stream);
/* Scalar loss arithmetic on device scalars (pix2pixHD_condImg_model.py:218-251: the per-scale GAN-loss sums, `* lambda_feat`;
 * train_mask2image.py:68-76: (D_fake + D_real) * 0.5, G_GAN + G_GAN_Feat + G_VGG):  out[0] = scale * sum_i weights[i] * terms[i][0],
 * summed left to right, every product and sum rounded to fp32 (the reference's chain of one-element tensor ops, in ONE
 * launch).  terms / dterms / weights are HOST arrays of n <= 8 entries; bwd: dterms[i][0] = g[0] * scale * weights[i]. */
int him_lincomb_fwd(const float* const* terms, const float* weights, int n, float scale, float* out, void* stream);
int him_lincomb_bwd(const float* g, const float* weights, int n, float scale, float* const* dterms, void* stream);
int him_mse_const_fwd(const float* x, size_t n, float target, float* out, void* ws, size_t ws_bytes,
                      void* stream);
int him_mse_const_bwd(const float* x, size_t n, float target, const float* g, float* dx, int accumulate,
                      void* stream);

/* ---------------------------------------------------------------------------------------------
 * torch.optim.Adam (betas=(beta1,0.999), eps 1e-8, no weight decay) over a flat parameter arena:
 * pix2pixHD_condImg_model.py:135-139 + the .step() calls of train_mask2image.py:80,86.
 * `step` is the 1-based step count.  lr / betas / eps are DOUBLES: 1 - beta2, the bias corrections and lr / bc1 are
 * derived in double and rounded to fp32 once, as torch.optim.Adam does with its python-float hyper-parameters.
 * ------------------------------------------------------------------------------------------- */
int him_adam_step(float* p, const float* g, float* m, float* v, size_t n, double lr, double beta1,
                  double beta2, double eps, int step, void* stream);
int him_fill(float* p, size_t n, float value, void* stream);
int him_scale(float* p, size_t n, float s, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Spectral norm power iteration, Ip = 1: models/sn_utils.py:8-25,62-67.
 *   v = l2n(u W), u' = l2n(W v^T), sigma = u' W v^T, with NOTHING detached: the backward
 *   differentiates through both normalisations.
 *   fwd: v_out[cols], u_out[rows], sigma_out[1]; scratch t[rows], s[cols] live in ws.
 *   bwd: dW (+)= g_sigma * d sigma / dW   given the saved u (input), v_out, u_out.
 * ------------------------------------------------------------------------------------------- */
size_t him_sn_ws(int rows, int cols);
int him_sn_power_iter_fwd(const float* W, const float* u, int rows, int cols, float* v_out, float* u_out,
                          float* sigma_out, void* ws, size_t ws_bytes, void* stream);
int him_sn_power_iter_bwd(const float* W, const float* u, const float* v_out, const float* u_out,
                          const float* sigma, const float* g_sigma, int rows, int cols, float* dW,
                          int accumulate, void* ws, size_t ws_bytes, void* stream);
/* out = W / sigma[0] ; and the W-side of its backward: dW (+)= dWbar/sigma, dsigma = -sum(dWbar*W)/sigma^2 */
int him_div_scalar_fwd(const float* W, const float* sigma, float* out, size_t n, void* stream);
int him_div_scalar_bwd(const float* W, const float* sigma, const float* dout, float* dW, float* dsigma,
                       size_t n, int accumulate, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Loader-side pixel work: what the reference's Dataset does per sample with PIL / torchvision on the
 * host -- data/base_dataset.py:243-268 get_transform_fn (select_region crop -> Image.resize NEAREST | BICUBIC ->
 * FLIP_LEFT_RIGHT -> ToTensor -> Normalize), data/segmentation_dataset.py:86-131 (get_masked_image x2, the instance
 * mask) -- for a whole batch on the device.  The crop windows are raw bytes inside ONE staging buffer `base`
 * (sample b at byte offset off[b], pitch[b] pixels per row); the per-row / per-column tables come from the host
 * (neurips18_hierchical_image_manipulation_amd/data/resample.py) and live in the same buffer.  Results are bit-identical to Pillow 12.2's.
 *   him_data_nearest:   dst[b][y][x] = src_b[ytab[b][y]][xtab[b][x]]; src_kind 0 u8 / 1 u16 / 2 i32;
 *                       dst_kind 0 f32 value / 1 f32 value/255 / 2 u8 / 3 i32.  A flip is folded into xtab.
 *   him_data_bicubic_h: tmp[b][r][i][c] = clip8((2^21 + sum_k weights[b][i][k] * src_b[r][first[b][i]+k][c]) >> 22),
 *                       interleaved RGB bytes, rows[b] source rows, tmp (B, maxrows, W, 3) bytes.
 *   him_data_bicubic_v: the same down the columns of tmp, then flip[b], /255 and (t-.5)/.5 -> dst (B,3,H,W) f32.
 *   him_data_region_masks: boxes (B,8) = input window, output window as (wmin,hmin,wmax,hmax) ints; fill (B) = the
 *                       class written into the input window; inst_id (B,2) = (selected?, id); inst_kind 0 f32 / 1 i32.
 *                       Writes mask_in, mask_object_in, mask_context_in, mask_out, mask_object_out and (if not NULL)
 *                       mask_object_inst, all (B,1,H,W) f32.
 * ------------------------------------------------------------------------------------------- */
int him_data_nearest(const void* base, const long long* off, const int* pitch, const int* xtab, const int* ytab,
                     int src_kind, void* dst, int dst_kind, int B, int H, int W, void* stream);
int him_data_bicubic_h(const void* base, const long long* off, const int* pitch, const int* rows, const int* first,
                       const int* count, const int* weights, int ksize, unsigned char* tmp, int maxrows, int B, int W,
                       void* stream);
int him_data_bicubic_v(const unsigned char* tmp, int maxrows, const int* first, const int* count, const int* weights,
                       int ksize, const int* flip, float* dst, int normalize, int B, int H, int W, void* stream);
int him_data_region_masks(const float* label, const void* inst, int inst_kind, const int* boxes, const float* fill,
                          const int* inst_id, float* mask_in, float* obj_in, float* ctx_in, float* mask_out,
                          float* obj_out, float* inst_mask, int B, int H, int W, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HIM_H_ */
