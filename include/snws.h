/* libsnws - C ABI of the MI355X (gfx950) normalized-conv hot path.
 *
 * Drop-in boundary for the weight-standardized Conv2d -> SwitchNorm2d -> ReLU stack, its resampling
 * skips, and the Evaluator confusion matrix of the reference (paths below are relative to the
 * reference repository root).  Plain pointers and sizes only; no torch types.
 *
 * Conventions
 *  - Every function returns 0 on success or a negative snws_status; snws_last_error() gives the text
 *    (thread-local).  No exception crosses this boundary.
 *  - All data pointers are DEVICE pointers valid on `stream` (a hipStream_t passed as void*; NULL = the
 *    null stream).  Kernels are enqueued asynchronously; nothing here synchronises or allocates.
 *  - Activations are NHWC ("channels last"): element (n,y,x,c) lives at ((n*H+y)*W+x)*ld + c where the
 *    pixel stride `ld` (in elements) may exceed C so that a tensor can be a channel slice of a wider
 *    concat buffer.  The logical shape seen by the reference API (N,C,H,W) is unchanged.
 *  - dtype: SNWS_F32 or SNWS_BF16 for activations; statistics, coefficients and master weights are f32.
 */
#ifndef SNWS_H
#define SNWS_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { SNWS_OK = 0, SNWS_ERR_INVALID = -1, SNWS_ERR_UNSUPPORTED = -2, SNWS_ERR_HIP = -3 } snws_status;
typedef enum { SNWS_F32 = 0, SNWS_BF16 = 1, SNWS_F64 = 2, SNWS_I64 = 3, SNWS_I32 = 4, SNWS_U8 = 5 } snws_dtype;

int snws_version(void);
const char* snws_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Evaluator confusion matrix - replaces Evaluator._generate_matrix, utils/metrics.py:40-45
 * (and, for the *_argmax form, the logits D2H + np.argmax of train.py:294-296).
 *
 * counts: int64[num_class*num_class + 2], ACCUMULATED into (caller zeroes it at Evaluator.reset()):
 *   counts[num_class*gt + pred] += 1 for every pixel whose label passes (gt >= 0) & (gt < num_class),
 *   with the label truncated toward zero like astype('int'); rows = ground truth, cols = prediction.
 *   counts[C*C]   += pixels whose index C*gt+pred fell at or beyond C*C   (np.bincount would grow -> reshape raises)
 *   counts[C*C+1] += pixels whose index was negative                      (np.bincount raises)
 * gt_dtype: SNWS_F32 | SNWS_F64 | SNWS_I64 | SNWS_I32 | SNWS_U8; pred_dtype: SNWS_I64 | SNWS_I32 | SNWS_U8.
 * num_class <= 64. */
int snws_confusion_hist(const void* gt, int gt_dtype, const void* pred, int pred_dtype, int64_t n_pixels,
                        int num_class, int64_t* counts, void* stream);

/* Same, with pred = argmax over the channel axis of NHWC logits (first maximum wins, a NaN counts as
 * the maximum, as numpy.argmax): logits (n_pixels, ld) of logit_dtype SNWS_F32 | SNWS_BF16. */
int snws_confusion_hist_argmax(const void* gt, int gt_dtype, const void* logits, int logit_dtype, int64_t ld,
                               int64_t n_pixels, int num_class, int64_t* counts, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Weight standardisation - replaces Conv2dws.forward lines modeling/weightstd.py:13-17 and their autograd.
 *
 * w: f32 (O, I, kh, kw) master weight (PyTorch layout).  n = I*kh*kw must be >= 2.
 * what_oihw (nullable): f32 (O, I, kh, kw) standardized weight.
 * what_ohwi (nullable): f32 (O, kh, kw, I) standardized weight for the direct (generic) conv kernels.
 * pack_fwd / pack_dgrad (nullable): bf16 MFMA operand images for snws_conv_mfma_*; sizes from
 *   snws_conv_pack_bytes().  pack_dgrad holds the spatially flipped, in/out-transposed filter. */
int snws_ws_fwd(const float* w, int O, int I, int kh, int kw, float* what_oihw, float* what_ohwi, void* pack_fwd,
                void* pack_dgrad, void* stream);

/* The same layouts for a filter used as is (no standardisation) - the plain nn.Conv2d classification head,
 * modeling/UNet3p_SNws.py:93. */
int snws_conv_pack(const float* w, int O, int I, int kh, int kw, float* w_ohwi, void* pack_fwd, void* pack_dgrad,
                   void* stream);

/* dw (O,I,kh,kw) f32 = gradient w.r.t. the master weight given g = dL/d(what) stored as (O, kh, kw, ldg)
 * with ldg >= I (the wgrad kernels' output layout).  accumulate != 0: dw += (a module used at several places
 * in one step); 0: dw is overwritten. */
int snws_ws_bwd(const float* w, const float* g_ohwi, int64_t ldg, float* dw, int O, int I, int kh, int kw,
                int accumulate, void* stream);

/* Bytes of one MFMA operand image for a conv with these channel counts / kernel size. */
int64_t snws_conv_pack_bytes(int c_in, int c_out, int kh, int kw);

/* ---------------------------------------------------------------------------------------------
 * Convolution - replaces F.conv2d at modeling/weightstd.py:18 and its autograd (dgrad, wgrad, bias grad).
 *
 * Generic direct kernels: any stride / padding / dilation / groups, dtype f32 or bf16, f32 accumulate.
 *   x (N,H,W,ldx) -> y (N,OH,OW,ldy);  weight f32 (O, kh, kw, I/groups);  bias f32 (O) or NULL. */
typedef struct {
  int N, H, W, C_in, C_out, kh, kw, stride, pad, dil, groups;
  int64_t ldx, ldy; /* pixel strides (elements) of the input-side and output-side tensors */
} snws_conv_desc;

int snws_conv_out_size(int in, int k, int stride, int pad, int dil);
int snws_conv_direct_fwd(const snws_conv_desc* d, int dtype, const void* x, const float* w_ohwi, const float* bias,
                         void* y, void* stream);
/* dx (N,H,W,ldx) = dgrad; every element of the C_in channels is written. */
int snws_conv_direct_dgrad(const snws_conv_desc* d, int dtype, const void* dy, const float* w_ohwi, void* dx,
                           void* stream);
/* dw_ohwi f32 (O, kh, kw, I/groups) overwritten; dbias f32 (O) overwritten when non-NULL. */
int snws_conv_direct_wgrad(const snws_conv_desc* d, int dtype, const void* x, const void* dy, float* dw_ohwi,
                           float* dbias, void* stream);

/* MFMA implicit-GEMM kernels: bf16 NHWC, f32 accumulate, kernel 3x3 (pad 1) or 1x1 (pad 0), stride 1,
 * dilation 1, groups 1, C_in a multiple of 16, C_out a multiple of 8 (dgrad: C_out a multiple of 16).
 * Returns SNWS_ERR_UNSUPPORTED otherwise.
 *   fwd:   y = conv(x, pack_fwd) + bias.  When stat_part != NULL it also emits per-tile partial moments
 *          of the ROUNDED output for SwitchNorm: stat_part[((n*T + t)*C_out + c)*2 + {0,1}] = sum and sum of
 *          squares of (y - bias[c]) over tile t's valid pixels, T = snws_conv_mfma_tiles(H, W).
 *   dgrad: the same kernel run on dy with pack_dgrad (desc as for fwd; reads C_out channels, writes C_in).
 *   wgrad: dw_ohwi f32 (O, kh, kw, C_in) and dbias (O) are ACCUMULATED into with float atomics (zero them first);
 *          the order of the adds varies from run to run - snws_conv_mfma_wgrad_det is the reproducible form. */
int snws_conv_mfma_tiles(int H, int W);
int snws_conv_mfma_fwd(const snws_conv_desc* d, const void* x, const void* pack_fwd, const float* bias, void* y,
                       float* stat_part, void* stream);
int snws_conv_mfma_dgrad(const snws_conv_desc* d, const void* dy, const void* pack_dgrad, void* dx, void* stream);
/* dx += dgrad (same operands as snws_conv_mfma_dgrad): for an input tensor with several consumers, every consumer
 * after the first adds its share of the gradient in place instead of writing a tensor that a separate add kernel
 * would have to fold in (autograd's accumulation of modeling/UNet3p_SNws.py:127-156's shared skips). */
int snws_conv_mfma_dgrad_acc(const snws_conv_desc* d, const void* dy, const void* pack_dgrad, void* dx, void* stream);
/* dgrad whose epilogue also runs pass 1 of the SwitchNorm backward (what snws_sn_bwd_reduce computes) for the
 * SwitchNorm(+ReLU) layers that produced this convolution's input: the gradient this launch writes, dx, IS their
 * incoming gradient when the convolution is their only consumer (autograd of modeling/snorm.py:41-81 chained to
 * F.conv2d, modeling/weightstd.py:18), so the tile that is being stored is reduced on the way out instead of being
 * read back.  A segment = one such layer occupying input channels [c_begin, c_begin + c_count) of the convolution
 * (several when the input is a channel concatenation): x / ldx its bf16 input, ctx its snws_sn_ctx_floats(N, c_count)
 * forward coefficients, beta its bias, relu whether a ReLU follows it; part receives [N][T][c_count][2] partial
 * sums, T = snws_conv_mfma_tiles(H, W), in the layout snws_sn_bwd_mix_a / snws_bn_bwd_pack take.
 * c_begin must be a multiple of 64, c_count of 8, segments must not overlap, C_in <= 1024. */
typedef struct snws_sn_bwd_seg {
  int c_begin, c_count;
  const void* x;
  int64_t ldx;
  const float* ctx;
  const float* beta;
  float* part;
  int relu;
} snws_sn_bwd_seg;
int snws_conv_mfma_dgrad_sn(const snws_conv_desc* d, const void* dy, const void* pack_dgrad, void* dx,
                            const snws_sn_bwd_seg* segs, int n_segs, void* stream);
int snws_conv_mfma_wgrad(const snws_conv_desc* d, const void* x, const void* dy, float* dw_ohwi, float* dbias,
                         void* stream);
/* Bit-reproducible wgrad (autograd of F.conv2d, modeling/weightstd.py:18, w.r.t. the filter and the bias): the same
 * kernels, but every workgroup stores its partial filter tile to `workspace` with plain stores and a second kernel
 * sums the partials in a fixed order.  dw_ohwi (O, kh, kw, C_in) is OVERWRITTEN (no zero-fill needed); dbias (O),
 * when non-NULL, is ADDED to by one thread per channel (zero it first, or let it hold an earlier use's gradient).
 * workspace: snws_conv_mfma_wgrad_workspace_bytes(d) bytes of device memory, 16-byte aligned, contents irrelevant. */
int64_t snws_conv_mfma_wgrad_workspace_bytes(const snws_conv_desc* d);
int snws_conv_mfma_wgrad_det(const snws_conv_desc* d, const void* x, const void* dy, float* dw_ohwi, float* dbias,
                             void* workspace, int64_t workspace_bytes, void* stream);

/* float32 MFMA kernels (v_mfma_f32_32x32x2_f32: f32 operands, f32 accumulation - exactly an fmaf chain): the float32
 * tier of the same convolution (F.conv2d at modeling/weightstd.py:18 and its autograd) for 3x3 (pad 1) / 1x1 (pad 0),
 * stride 1, dilation 1, groups 1 and ANY channel counts; SNWS_ERR_UNSUPPORTED otherwise (use the direct kernels).
 *   fwd:   y = conv(x, w_ohwi) + bias with w_ohwi f32 (O, kh, kw, I) as snws_ws_fwd / snws_conv_pack write it; stat_part
 *          (nullable) [N][T][C_out][2], T = snws_conv_f32_tiles(H, W): per 8x32-pixel tile the sum and the sum of
 *          squares of (y - bias[c]) over the tile's valid pixels - the layout snws_sn_finalize takes with shift = bias.
 *   dgrad: dx = the same kernel run on dy with w_ihwo_flipped f32 (I, kh, kw, O) from snws_conv_f32_flip.
 *   wgrad: dw_ohwi f32 (O, kh, kw, I) OVERWRITTEN, dbias (O, nullable) ADDED to; per-workgroup partial tiles in
 *          `workspace` (snws_conv_f32_wgrad_workspace_bytes(d) bytes, 16-byte aligned) summed in a fixed order. */
int snws_conv_f32_tiles(int H, int W);
int snws_conv_f32_fwd(const snws_conv_desc* d, const float* x, const float* w_ohwi, const float* bias, float* y,
                      float* stat_part, void* stream);
int snws_conv_f32_flip(const float* w_ohwi, float* w_ihwo_flipped, int O, int I, int kh, int kw, void* stream);
int snws_conv_f32_dgrad(const snws_conv_desc* d, const float* dy, const float* w_ihwo_flipped, float* dx, void* stream);
int64_t snws_conv_f32_wgrad_workspace_bytes(const snws_conv_desc* d);
int snws_conv_f32_wgrad(const snws_conv_desc* d, const float* x, const float* dy, float* dw_ohwi, float* dbias,
                        void* workspace, int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * SwitchNorm - replaces SwitchNorm2d.forward, modeling/snorm.py:41-81, fused with the ReLU that follows it
 * in ConvSnRelu (modeling/UNet3p_SNws.py:29-33), and its autograd.
 *
 * One invocation works on x (N, HW, C) with pixel stride ldx.  Small f32 side buffers:
 *   part   (N, T, C, 2)  per-tile partial sums of (x - shift) and (x - shift)^2
 *   shift  (N, C) or (C) the per-channel offset the partial sums were taken around
 *   ctx    snws_sn_ctx_floats(N, C) floats: [mean_in NC][var_in NC][mean NC][rstd NC][scale NC][ln 2N][bn 2C]
 *   pack   (2C+1): [sum_n mean_in, sum_n (var_in+mean_in^2), N_local] - the data-parallel all-reduce payload
 *          (the replacement for the _ChildMessage/_MasterMessage exchange of
 *          modeling/sync_batchnorm/batchnorm.py:24-25,80-101); the caller all-reduces it between
 *          snws_sn_mix_a and snws_sn_mix_b when training on several GPUs.
 */
int64_t snws_sn_ctx_floats(int N, int C);
int snws_sn_stats_tiles(int N, int64_t HW, int C);

/* Pass 1 (stand-alone form): partial moments of x.  part has T = snws_sn_stats_tiles(N,HW,C) tiles per image;
 * shift (N,C) is written with x's first pixel of each image. */
int snws_sn_stats(const void* x, int dtype, int N, int64_t HW, int C, int64_t ldx, float* part, float* shift,
                  void* stream);
/* Partial moments -> ctx.mean_in / ctx.var_in (unbiased, /(HW-1); snorm.py:45-46).  shift_per_image: 1 if shift is
 * (N,C), 0 if it is (C) (the conv epilogue's bias). */
int snws_sn_finalize(const float* part, int T, const float* shift, int shift_per_image, int N, int64_t HW, int C,
                     float* ctx, void* stream);
/* LN moments (snorm.py:48-50) into ctx.ln and the local BN sums into pack. */
int snws_sn_mix_a(float* ctx, float* pack, int N, int C, void* stream);
/* Mix the three branches (snorm.py:52-77), update the running buffers (training && using_bn; snorm.py:56-63),
 * write ctx.mean / ctx.rstd / ctx.scale.  mean_weight/var_weight have 3 (using_bn) or 2 entries.
 * pack is read after the caller's all-reduce: pack[2C] is the GLOBAL batch size. */
int snws_sn_mix_b(float* ctx, const float* pack, const float* gamma, const float* mean_weight,
                  const float* var_weight, float* running_mean, float* running_var, int N, int C, float eps,
                  float momentum, int training, int using_movavg, int using_bn, void* stream);
/* y = act((x - ctx.mean) * ctx.scale + beta [+ residual]), act = ReLU if relu else identity (snorm.py:79-81).
 * residual (nullable) has pixel stride ldr and y's dtype. */
int snws_sn_apply(const void* x, int64_t ldx, void* y, int64_t ldy, const void* residual, int64_t ldr, int dtype,
                  const float* ctx, const float* beta, int N, int64_t HW, int C, int relu, void* stream);

/* Backward.  ctxb: snws_sn_ctxb_floats(N,C) floats of scratch/coefficients. */
int64_t snws_sn_ctxb_floats(int N, int C);
/* Pass 1: per-tile partial sums of dy' and dy'*xhat (dy' = dy masked by the ReLU), part (N,T,C,2).
 * When y_for_mask != NULL the mask is (y_for_mask > 0) (needed with a residual add); otherwise it is recomputed
 * from x. */
int snws_sn_bwd_reduce(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* y_for_mask, int64_t ldy,
                       int dtype, const float* ctx, const float* gamma, const float* beta, int N, int64_t HW, int C,
                       int relu, float* part, void* stream);
/* Partials -> dm, dv per (n,c); per-image and per-channel sums; dgamma/dbeta (C) overwritten (accumulate == 0) or
 * added to (accumulate != 0); packb (2C) = [sum_n dm, sum_n dv] is the backward all-reduce payload. */
int snws_sn_bwd_mix_a(const float* part, int T, const float* ctx, const float* gamma, float* ctxb, float* packb,
                      float* dgamma, float* dbeta, int N, int C, int accumulate, void* stream);
/* After the caller's all-reduce of packb: coefficients A,B,K of dx = A*dy' + B*x + K into ctxb, and the
 * gradients of mean_weight / var_weight (3 or 2 entries).  n_global = global batch size.
 * dx_sum (nullable, C floats): sum of dx over (n, pixels) per channel in closed form - the gradient
 * of a per-channel bias added before the normalisation (the conv bias of a ConvSnRelu block).
 * accumulate: bit 0 - add to dmean_weight / dvar_weight instead of overwriting them; bit 1 - the same for dx_sum. */
int snws_sn_bwd_mix_b(const float* ctx, float* ctxb, const float* packb, const float* mean_weight,
                      const float* var_weight, const float* running_mean, const float* running_var,
                      float* dmean_weight, float* dvar_weight, float* dx_sum, int N, int64_t HW, int C,
                      float n_global, int training, int using_bn, int accumulate, void* stream);
/* Pass 2: dx = A*dy' + B*x + K. */
int snws_sn_bwd_apply(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* y_for_mask, int64_t ldy,
                      void* dx, int64_t lddx, int dtype, const float* ctx, const float* ctxb, const float* beta,
                      int N, int64_t HW, int C, int relu, void* stream);
/* Pass 2 of the residual block's second SwitchNorm (relu(SwitchNorm(conv) + residual),
 * modeling/UNet3p_res_ocr_SNws.py:152-159): as snws_sn_bwd_apply with relu = 1 and the mask taken from the block
 * output y_for_mask (required), and the masked incoming gradient dy' = [y > 0] * dy - the gradient of the residual
 * input - is stored to dres in the same pass (replaces autograd's threshold-backward of `out += residual; relu`). */
int snws_sn_bwd_apply_res(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* y_for_mask, int64_t ldy,
                          void* dx, int64_t lddx, void* dres, int64_t lddres, int dtype, const float* ctx,
                          const float* ctxb, const float* beta, int N, int64_t HW, int C, void* stream);

/* Batched forms: up to SNWS_SN_BATCH_MAX INDEPENDENT SwitchNorm calls of the same N, HW, C and dtype in one launch per
 * pass - the branch blocks of a UNet3+ decoder scale (modeling/UNet3p_SNws.py:127-156: five ConvSnRelu blocks whose
 * outputs meet only in the torch.cat that follows).  Item i is exactly the call of the single form with the item's
 * pointers; the results are bit-identical to `count` single calls.  The calls of one launch must not share what they
 * write: running buffers (snws_sn_mix_b_batched, training with using_bn) and gradient buffers (the backward forms) must
 * differ between items, or SNWS_ERR_INVALID is returned - uses of ONE module belong to successive launches, in the
 * module's call order.  The apply passes need 16-byte aligned tensors, pixel strides and channel counts
 * (SNWS_ERR_UNSUPPORTED otherwise: fall back to the single calls).
 *   forward : snws_sn_fwd_a_batched (= snws_sn_finalize + snws_sn_mix_a) -> [all-reduce of the packs]
 *             -> snws_sn_mix_b_batched -> snws_sn_apply_batched (no residual)
 *   backward: [snws_sn_bwd_reduce or a dgrad epilogue per item] -> snws_sn_bwd_a_batched (= snws_sn_bwd_mix_a)
 *             -> [all-reduce of the packbs] -> snws_sn_bwd_mix_b_batched -> snws_sn_bwd_apply_batched (mask from x) */
#define SNWS_SN_BATCH_MAX 8
typedef struct snws_sn_fwd_item {
  const float* part;   /* (N, T, C, 2) partial moments */
  const float* shift;  /* (N, C) or (C) */
  float* ctx;
  float* pack;         /* 2C + 1 floats */
  const float* gamma;
  const float* beta;
  const float* mean_weight;
  const float* var_weight;
  float* running_mean; /* nullable unless using_bn */
  float* running_var;
  const void* x;       /* apply pass only */
  void* y;
  int64_t ldx, ldy;
  int T, shift_per_image;
} snws_sn_fwd_item;
typedef struct snws_sn_bwd_item {
  const float* part;   /* (N, T, C, 2) partial sums of dy', dy' * xhat */
  const float* ctx;
  const float* gamma;
  const float* beta;
  const float* mean_weight;
  const float* var_weight;
  float* ctxb;
  float* packb;        /* 2C floats */
  float* dgamma;
  float* dbeta;
  float* dmean_weight;
  float* dvar_weight;
  float* dx_sum;       /* nullable */
  const void* dy;      /* apply pass only */
  const void* x;
  void* dx;
  int64_t lddy, ldx, lddx;
  int T;
  int accumulate_a;    /* `accumulate` of snws_sn_bwd_mix_a */
  int accumulate_b;    /* `accumulate` of snws_sn_bwd_mix_b */
  int reserved;
} snws_sn_bwd_item;
int snws_sn_fwd_a_batched(const snws_sn_fwd_item* items, int count, int N, int64_t HW, int C, void* stream);
int snws_sn_mix_b_batched(const snws_sn_fwd_item* items, int count, int N, int C, float eps, float momentum,
                          int training, int using_movavg, int using_bn, void* stream);
int snws_sn_apply_batched(const snws_sn_fwd_item* items, int count, int dtype, int N, int64_t HW, int C, int relu,
                          void* stream);
int snws_sn_bwd_a_batched(const snws_sn_bwd_item* items, int count, int N, int C, void* stream);
int snws_sn_bwd_mix_b_batched(const snws_sn_bwd_item* items, int count, int N, int64_t HW, int C, float n_global,
                              int training, int using_bn, void* stream);
int snws_sn_bwd_apply_batched(const snws_sn_bwd_item* items, int count, int dtype, int N, int64_t HW, int C, int relu,
                              void* stream);

/* SwitchNorm forward in one call that allocates nothing and takes no side buffers: snws_sn_stats -> snws_sn_finalize
 * -> snws_sn_mix_a -> snws_sn_mix_b -> snws_sn_apply with part / shift / ctx / pack held in a static device array of
 * the library (snws_sn_noalloc_max_floats() floats = 16 MiB, resident on every GPU the library runs on until the
 * process ends).  No counterpart in the reference: it exists for callers that cannot allocate device memory - code
 * under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  Single-GPU statistics only (there is
 * no point at which the caller could all-reduce `pack`); the coefficients are not returned, so it serves inference and
 * forward-only use (training = 1 still updates the running buffers exactly as snws_sn_mix_b does).
 * Returns SNWS_ERR_INVALID when snws_sn_noalloc_floats(N, HW, C) exceeds the array.  Every call on one GPU shares
 * that GPU's array: calls that overlap in time on one GPU (different streams or threads) race; calls on one stream
 * are ordered and safe.
 * snws_sn_noalloc_max_floats() is also the warm-up: the first use on a device looks the array's address up with
 * hipGetSymbolAddress, which is not a capturable stream operation - call it once per device, with that device
 * current, BEFORE beginning a stream capture that contains snws_sn_apply_noalloc (without a GPU it only returns the
 * size). */
int64_t snws_sn_noalloc_max_floats(void);
int64_t snws_sn_noalloc_floats(int N, int64_t HW, int C);
int snws_sn_apply_noalloc(const void* x, int64_t ldx, void* y, int64_t ldy, const void* residual, int64_t ldr, int dtype,
                          const float* gamma, const float* beta, const float* mean_weight, const float* var_weight,
                          float* running_mean, float* running_var, int N, int64_t HW, int C, float eps, float momentum,
                          int training, int using_movavg, int using_bn, int relu, void* stream);

/* Wide 1x1 convolution of an NHWC bf16 tensor as one GEMM on the matrix cores: y[m][n] = sum_k x[m][k] * w[n][k] + bias[n]
 * over all M = N*H*W pixels - the plain nn.Conv2d 1x1 layers of the OCR head (modeling/UNet3p_res_ocr_SNws.py:39-73,
 * 118-123,269-274: F.conv2d with a (C_out, C_in, 1, 1) filter), their input gradient (the same product with w = the
 * transposed filter), and the 1x1 half of conv(upsample(x)) (modeling/UNet3p_SNws.py:127-156).  w: bf16 [N][K] row-major;
 * bias: f32 [N] or NULL; x / y: pixel strides ldx >= K / ldy >= N (multiples of 8).  snws_gemm1x1_supported(N, K): N >= 64
 * and a multiple of 8, K a multiple of 64; other shapes return SNWS_ERR_UNSUPPORTED (use snws_conv_mfma_fwd). */
int snws_gemm1x1_supported(int N, int K);
int snws_gemm1x1(const void* x, int64_t ldx, const void* w, const float* bias, void* y, int64_t ldy, int64_t M, int N, int K,
                 void* stream);
/* y += the same product (no bias; the f32 sum of the stored bf16 value and the new one, rounded once): the second half of
 * a 1x1 convolution over torch.cat([a, b], 1) computed as w[:, :Ca] a + w[:, Ca:] b without building the concatenation
 * (modeling/UNet3p_res_ocr_SNws.py:130), and a further contribution to an input gradient with several contributors
 * (autograd's sum over the consumers of one tensor). */
int snws_gemm1x1_acc(const void* x, int64_t ldx, const void* w, void* y, int64_t ldy, int64_t M, int N, int K, void* stream);
/* The same product (accumulate = 0: y = x w^T + bias; != 0: y += x w^T, no bias) whose store epilogue also hands a
 * normalisation layer that follows - the nn.BatchNorm2d after every 1x1 nn.Conv2d of the OCR head,
 * modeling/UNet3p_res_ocr_SNws.py:39-73,118-123 - the per-tile moments of the values it stored:
 * stat_part[((img * T + t) * N + c) * 2 + {0, 1}] = sum and sum of squares of y over the 256 pixels of tile t of image
 * img, T = snws_gemm1x1_stat_tiles(HW) (-1: HW is not a multiple of 256 - then use snws_sn_stats).  Feed it to
 * snws_sn_finalize with an all-zero shift (shift_per_image = 0).  M = images * HW. */
int snws_gemm1x1_stat_tiles(int64_t HW);
int snws_gemm1x1_stats(const void* x, int64_t ldx, const void* w, const float* bias, void* y, int64_t ldy, int64_t M, int N,
                       int K, int accumulate, int64_t HW, float* stat_part, void* stream);

/* The two attention steps of the OCR head as fused streaming kernels (bf16 NHWC pixel tensors, float32 soft-max, every
 * small product on the matrix cores; per-class results leave as per-wave partial rows to be summed with snws_partial_sum).
 *   SpatialGather_Module.forward, modeling/UNet3p_res_ocr_SNws.py:21-28:
 *     ctx[b][k][c] = sum_p softmax_p(scale * probs[b][p][k]) * feats[b][p][c]          (C = 512 feature channels, K <= 16)
 *     snws_ocr_gather_stats: per-run (max, sum exp) pairs, stat_part [B][snws_ocr_tiles(HW, 0)][16][2]
 *     snws_ocr_gather_fwd:   ctx_part [B][snws_ocr_tiles(HW, 1)][16][C]
 *     snws_ocr_gather_bwd:   dfeats = P * dctx, dprobs = scale * P * (feats . dctx - dot), dot[b][k] = <ctx, dctx>;
 *                            dctx bf16 [B][16][C] (rows >= K zero), dprobs bf16 [B][HW][16]
 *     snws_ocr_gather_bwd_acc: the same with dfeats += (the feature map has other consumers whose shares of its gradient
 *                            are already in dfeats: autograd's sum over consumers, done in place)
 *   _ObjectAttentionBlock.forward, modeling/UNet3p_res_ocr_SNws.py:87-100 (between f_pixel / f_object / f_down and f_up):
 *     out[b][p][c] = sum_k softmax_k(scale * q[b][p] . key[b][k]) * val[b][k][c]       (C = 256 key channels, K <= 16)
 *     snws_ocr_attend_fwd; key / val: bf16 [B][16][C], rows >= K zero
 *     snws_ocr_attend_bwd:   dq, and dkey_part / dval_part [B][snws_ocr_tiles(HW, 2)][16][C]
 * Other channel counts return SNWS_ERR_UNSUPPORTED. */
int snws_ocr_tiles(int64_t HW, int kind);
int snws_ocr_gather_stats(const void* probs, int64_t ldp, float* stat_part, int B, int64_t HW, int K, float scale, void* stream);
int snws_ocr_gather_fwd(const void* feats, int64_t ldf, const void* probs, int64_t ldp, const float* stat_part, float* ctx_part,
                        int B, int64_t HW, int C, int K, float scale, void* stream);
int snws_ocr_gather_bwd(const void* feats, int64_t ldf, const void* probs, int64_t ldp, const float* stat_part, const void* dctx,
                        const float* dot, void* dfeats, int64_t lddf, void* dprobs, int B, int64_t HW, int C, int K, float scale,
                        void* stream);
int snws_ocr_gather_bwd_acc(const void* feats, int64_t ldf, const void* probs, int64_t ldp, const float* stat_part,
                            const void* dctx, const float* dot, void* dfeats, int64_t lddf, void* dprobs, int B, int64_t HW,
                            int C, int K, float scale, void* stream);
int snws_ocr_attend_fwd(const void* q, int64_t ldq, const void* key, const void* val, void* out, int64_t ldo, int B, int64_t HW,
                        int C, int K, float scale, void* stream);
int snws_ocr_attend_bwd(const void* q, int64_t ldq, const void* dout, int64_t lddo, const void* key, const void* val, void* dq,
                        int64_t lddq, float* dkey_part, float* dval_part, int B, int64_t HW, int C, int K, float scale,
                        void* stream);

/* The same two attention steps in float32, for any channel count and up to 64 classes - the float32 parity tier of
 * UNet3p_res_ocr_SNws (the reference computes them with F.softmax + torch.matmul in float32,
 * modeling/UNet3p_res_ocr_SNws.py:26-27,93-98).  Plain f32 vector arithmetic, fixed-order reductions.  All pixel tensors
 * (B, HW, channels) float32 with a pixel stride; class scores (B, HW, K) with stride ldp; region tensors (B, K, C) dense.
 *   snws_ocr32_gather_fwd: stat [B][K][2] (max, sum exp) and ctx [B][K][C]
 *   snws_ocr32_gather_bwd: dfeats (+= when accumulate) and dprobs (B, HW, K) with stride lddp, dot[b][k] = <ctx, dctx>
 *   snws_ocr32_attend_fwd / _bwd: key, val [B][K][C]; dkey_part / dval_part [B][snws_ocr32_runs(HW)][K][C], to be summed
 *   over the runs (snws_partial_sum_batched). */
int snws_ocr32_runs(int64_t HW);
int snws_ocr32_gather_fwd(const float* feats, int64_t ldf, const float* probs, int64_t ldp, float* stat, float* ctx, int B,
                          int64_t HW, int C, int K, float scale, void* stream);
int snws_ocr32_gather_bwd(const float* feats, int64_t ldf, const float* probs, int64_t ldp, const float* stat,
                          const float* dctx, const float* dot, float* dfeats, int64_t lddf, float* dprobs, int64_t lddp, int B,
                          int64_t HW, int C, int K, float scale, int accumulate, void* stream);
int snws_ocr32_attend_fwd(const float* q, int64_t ldq, const float* key, const float* val, float* out, int64_t ldo, int B,
                          int64_t HW, int C, int K, float scale, void* stream);
int snws_ocr32_attend_bwd(const float* q, int64_t ldq, const float* dout, int64_t lddo, const float* key, const float* val,
                          float* dq, int64_t lddq, float* dkey_part, float* dval_part, int B, int64_t HW, int C, int K,
                          float scale, void* stream);

/* The last block of the network fused with the classification head - SwitchNorm + ReLU of the final convolution
 * (modeling/snorm.py:41-81, modeling/UNet3p_SNws.py:29-33) followed by the 1x1 nn.Conv2d head
 * (modeling/UNet3p_SNws.py:93,158; modeling/UNet_SNws.py:93,151) and their autograd - without storing the normalised
 * activation or its gradient.  bf16, C = 64 or 320 (SNWS_ERR_UNSUPPORTED otherwise: run the unfused entry points),
 * at most 16 classes.  head_w: bf16 (16, C), rows >= n_class zero; head_b: f32 (16) or NULL; logits / dlogits: bf16
 * (N*HW, 16), pixel stride 16, channels >= n_class of dlogits zero.
 *   snws_sn_apply_head:       logits = head_w * act((x - ctx.mean) * ctx.scale + beta) + head_b   (ctx from snws_sn_mix_b)
 *   snws_sn_head_bwd_reduce:  pass 1 of the SwitchNorm backward with dy = head_w^T * dlogits rebuilt on the fly:
 *                             part [N][T][C][2], T = snws_sn_head_tiles(HW, C, 1), as snws_sn_bwd_mix_a takes it; plus the
 *                             head's gradient as per-wave partials hw_part [N*T][16][C] and hb_part [N*T][16]: sum them
 *                             with snws_partial_sum (fixed order: bit-reproducible)
 *   snws_sn_head_bwd_apply:   pass 2: dx = A * dy' + B * x + K (ctxb from snws_sn_bwd_mix_b), dy' rebuilt the same way
 *   snws_partial_sum:         out[i] (+)= sum_p part[p * len + i] */
int snws_sn_head_tiles(int64_t HW, int C, int backward);
int snws_sn_apply_head(const void* x, int64_t ldx, const float* ctx, const float* beta, const void* head_w,
                       const float* head_b, void* logits, int N, int64_t HW, int C, int relu, void* stream);
int snws_sn_head_bwd_reduce(const void* dlogits, const void* x, int64_t ldx, const float* ctx, const float* beta,
                            const void* head_w, float* part, float* hw_part, float* hb_part, int N, int64_t HW, int C,
                            int relu, void* stream);
int snws_sn_head_bwd_apply(const void* dlogits, const void* x, int64_t ldx, const float* ctx, const float* ctxb,
                           const float* beta, const void* head_w, void* dx, int64_t lddx, int N, int64_t HW, int C,
                           int relu, void* stream);
int snws_partial_sum(const float* part, int n_parts, int len, float* out, int accumulate, void* stream);
/* the same for `batch` independent entries: part [batch][n_parts][len] -> out [batch][len] */
int snws_partial_sum_batched(const float* part, int batch, int n_parts, int len, float* out, int accumulate, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Batch normalisation on the same streaming kernels - replaces nn.BatchNorm2d (+ nn.ReLU) of the OCR head
 * (modeling/UNet3p_res_ocr_SNws.py:52-79,124,257,263) and _SynchronizedBatchNorm's statistics
 * (modeling/sync_batchnorm/batchnorm.py:45-65,80-115).
 *
 * forward:  snws_sn_stats -> snws_sn_finalize -> snws_bn_pack -> [all-reduce pack over ranks: synchronised BN]
 *           -> snws_bn_coeff -> snws_sn_apply(relu)
 * backward: snws_sn_bwd_reduce -> snws_bn_bwd_pack (also gives d gamma, d beta of this rank) -> [all-reduce packb]
 *           -> snws_bn_bwd_coeff -> snws_sn_bwd_apply
 * pack  [2C+1]: sum_n mean_n(c), sum_n E_n[x^2](c), N     packb [2C]: sum dy'(c), sum dy'*xhat(c)
 * snws_bn_coeff: mean / biased variance from pack (training) or the running buffers; inv_std = (var + eps)^-1/2, or
 *   max(var, eps)^-1/2 with clamp_eps (batchnorm.py:104); running buffers (may be NULL) take momentum * the batch
 *   mean / UNBIASED variance; gamma may be NULL (= 1).  count = global N*HW for snws_bn_bwd_coeff. */
int snws_bn_pack(const float* ctx, int N, int64_t HW, int C, float* pack, void* stream);
int snws_bn_coeff(float* ctx, const float* pack, const float* gamma, float* running_mean, float* running_var, int N,
                  int64_t HW, int C, float eps, float momentum, int training, int clamp_eps, void* stream);
int snws_bn_bwd_pack(const float* part, int T, int N, int C, float* packb, float* dgamma, float* dbeta, void* stream);
int snws_bn_bwd_coeff(const float* ctx, float* ctxb, const float* packb, int N, int C, float count, int training,
                      void* stream);
/* nn.Dropout2d after BatchNorm2d + ReLU (conv_bn_dropout, modeling/UNet3p_res_ocr_SNws.py:118-124) folded into the
 * normalisation passes: post (N x C floats, >= 0) is the per-(image, channel) feature-dropout factor (0 or 1/(1-p)).
 *   forward:  ... -> snws_bn_coeff -> snws_sn_apply_post:      y = post * act((x - mean) * scale + beta)
 *   backward: snws_sn_bwd_reduce (on the unscaled gradient) -> snws_bn_bwd_pack_post (scales image n's partial sums by
 *             post[n]) -> snws_bn_bwd_coeff -> snws_sn_bwd_apply_post: dx = A * post * dy' + B * x + K */
int snws_sn_apply_post(const void* x, int64_t ldx, void* y, int64_t ldy, int dtype, const float* ctx, const float* beta,
                       const float* post, int N, int64_t HW, int C, int relu, void* stream);
int snws_sn_bwd_apply_post(const void* dy, int64_t lddy, const void* x, int64_t ldx, void* dx, int64_t lddx, int dtype,
                           const float* ctx, const float* ctxb, const float* beta, const float* post, int N, int64_t HW,
                           int C, int relu, void* stream);
int snws_bn_bwd_pack_post(const float* part, int T, int N, int C, const float* post, float* packb, float* dgamma,
                          float* dbeta, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Full-scale skip resampling - replaces nn.MaxPool2d(k,k,ceil_mode=True) / nn.Upsample(scale, bilinear,
 * align_corners=False) / torch.cat at modeling/UNet3p_SNws.py:49-55,127-156 and their autograd.
 *
 * mode 0: copy, 1: max-pool factor k (window clipped at the border, out = ceil(in/k)), 2: bilinear up x k.
 * src (N,H,W,lds) C channels -> dst (N,OH,OW,ldd) written at the destination's channel slice (dst pointer already
 * offset).  bwd with accumulate != 0 adds (+=) into dsrc, which the caller zeroes or reuses across consumers; with
 * accumulate == 0 it writes every element of dsrc (no zero-fill needed). */
int snws_resample_fwd(const void* src, int64_t lds, void* dst, int64_t ldd, int dtype, int N, int H, int W, int C,
                      int mode, int k, void* stream);
int snws_resample_bwd(const void* ddst, int64_t ldd, const void* src, int64_t lds, void* dsrc, int64_t ldds,
                      int dtype, int N, int H, int W, int C, int mode, int k, int accumulate, void* stream);
/* dst += src over NHWC tensors of `pixels` pixels x C channels (f32 or bf16, f32 accumulation): the in-place sum of
 * a gradient with several contributors, for contributors whose own kernels cannot add on the fly (no counterpart in
 * the reference: autograd's implicit accumulation). */
int snws_add_inplace(void* dst, int64_t ldd, const void* src, int64_t lds, int dtype, int64_t pixels, int C, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Convolution of a bilinearly up-sampled map without materialising it - replaces the pairs
 * Conv2dws(nn.Upsample(scale k, bilinear)(x)) of modeling/UNet3p_SNws.py:131,138-139,145-147,152-155 (and the 1x1
 * conv after nn.Upsample of modeling/UNet_SNws.py:122-123).  The channel contraction commutes with the
 * interpolation:  conv(up_k(x))(p) = b + sum_t [p+t inside] * up_k(Z_t)(p+t),  Z_t = W_t x.
 *   Z  (N, h, w, ldz): ksize*ksize*C_out channels at the SOURCE resolution, tap-major (channel t*C_out + o),
 *      produced by a 1x1 convolution (snws_conv_mfma_fwd / snws_conv_direct_fwd) with the filter re-laid as
 *      (ksize*ksize*C_out, C_in, 1, 1).
 *   fwd: y (N, h*k, w*k, ldy) = bias + sum over taps of the bilinear interpolation of Z_t at p+t (zero outside).
 *   bwd: dZ (N, h, w, lddz) overwritten with the adjoint applied to dy (a gather; no atomics).
 * ksize 1 or 3; dtype of Z, y, dy, dZ: SNWS_F32 or SNWS_BF16 (f32 accumulation). */
int snws_upconv_fwd(const void* Z, int64_t ldz, void* y, int64_t ldy, const float* bias, int dtype, int N, int h,
                    int w, int C_out, int k, int ksize, void* stream);
int snws_upconv_bwd(const void* dy, int64_t lddy, void* dZ, int64_t lddz, int dtype, int N, int h, int w, int C_out,
                    int k, int ksize, void* stream);
/* dZ += the same gradient: for a tap map Z consumed at several decoder scales (modeling/UNet3p_SNws.py:131-155 up-samples
 * hd5 / hd4 / hd3 for every finer scale), whose later consumers add their share into the first one's tensor instead of
 * leaving the sum to autograd.  Only the k = 2 matrix-core kernel adds in place (bf16, 3x3, C_out a multiple of 64):
 * SNWS_ERR_UNSUPPORTED otherwise - then call snws_upconv_bwd into a temporary and snws_add_inplace. */
int snws_upconv_bwd_acc(const void* dy, int64_t lddy, void* dZ, int64_t lddz, int dtype, int N, int h, int w, int C_out,
                        int k, int ksize, void* stream);
/* snws_upconv_fwd that also emits what the SwitchNorm after the block (modeling/UNet3p_SNws.py:29-33) needs:
 * stat_part [N][T][C_out][2], T = snws_upconv_stat_tiles(h*k, w*k): per 8x32 output strip the sum and the sum of
 * squares of (y - bias[c]) over the strip's valid pixels, y as stored (rounded) - the layout snws_sn_finalize takes
 * with shift = bias.  bf16, ksize 3, k in {2,4,8,16}, C_out a multiple of 64 only (else SNWS_ERR_UNSUPPORTED: run
 * snws_upconv_fwd and snws_sn_stats). */
int snws_upconv_stat_tiles(int H, int W);
/* Which kernel family runs the staged bf16 3x3 forward (snws_upconv_fwd / snws_upconv_fwd_stats; no counterpart in the
 * reference): 0 = interpolation on the vector ALUs, 1 or 2 = as small matrix products on the matrix cores (the default;
 * the environment variable SNWS_UPCONV_MFMA overrides it at first use).  The two families give the same products and
 * differ only in the f32 summation order.  Returns the previous mode; any other argument only queries.  Process-wide. */
int snws_upconv_fwd_mode(int mode);
/* The same switch for snws_upconv_bwd with k = 2 (bf16, 3x3, C_out a multiple of 64): 1 (default; SNWS_UPCONV_BWD_MFMA
 * overrides it at first use) = matrix cores, 0 = the separable vector-ALU kernel. */
int snws_upconv_bwd_mode(int mode);
int snws_upconv_fwd_stats(const void* Z, int64_t ldz, void* y, int64_t ldy, const float* bias, int dtype, int N, int h,
                          int w, int C_out, int k, int ksize, float* stat_part, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused soft-max cross-entropy, forward + backward in one pass - replaces nn.CrossEntropyLoss(ignore_index=255)
 * at utils/loss.py:146-159 (called at train.py:252) and its autograd.
 *
 * logits (n_pixels, ld) NHWC, C classes, dtype SNWS_F32 | SNWS_BF16; target f32 labels (truncated toward zero like
 * .long()); pixels whose label equals ignore_index contribute nothing.
 * loss_part[snws_ce_blocks(n_pixels)]: per-block sums of (logsumexp - logit[target]) in a fixed order; the caller
 *   adds them and divides by (valid pixels * batch size) - both global under data parallelism.
 * dlogits (n_pixels, ldd), may be NULL: (softmax - onehot) * *scale for valid pixels, 0 otherwise; channels
 *   C..c_write-1 (the MFMA head's padding) are written as 0.  scale: device scalar 1/(valid * batch), or NULL = 1. */
int snws_ce_blocks(int64_t n_pixels);
int snws_ce_fwd_bwd(const void* logits, int dtype, int64_t ld, const float* target, int64_t n_pixels, int C,
                    int ignore_index, const float* scale, float* loss_part, void* dlogits, int64_t ldd, int c_write,
                    void* stream);
/* The same with per-class weights - nn.CrossEntropyLoss(weight=self.weight, ...) at utils/loss.py:148, the weights
 * train.py:55-67 builds under --use-balanced-weights.  class_weight: C device floats, or NULL (= snws_ce_fwd_bwd).
 * loss_part sums w[target] * (logsumexp - logit[target]); dlogits = w[target] * (softmax - onehot) * *scale; the
 * caller's normaliser is the (global) sum of w[target] over valid pixels instead of their count. */
int snws_ce_fwd_bwd_w(const void* logits, int dtype, int64_t ld, const float* target, int64_t n_pixels, int C,
                      int ignore_index, const float* scale, const float* class_weight, float* loss_part, void* dlogits,
                      int64_t ldd, int c_write, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Arg-max of one inference tile written into the full-disk prediction - replaces torch.max(outputs, 1) and
 * pred_arr[yoff:yoff+B, xoff:xoff+B] = tile at predict.py:196,248-253 (tiles are stitched in list order, a later
 * tile overwrites an earlier one where they overlap).
 * logits (h*w, ld) NHWC of one tile; pred int64 (H, W); the tile's top-left pixel lands at (y0, x0); pixels that
 * fall outside the disk are dropped.  First maximum wins, a NaN counts as the maximum (torch.max / numpy.argmax).
 * remap_background != 0 applies predict.py:207-213 (classes 0 and 10 -> 255). */
int snws_argmax_stitch(const void* logits, int dtype, int64_t ld, int h, int w, int C, int64_t* pred, int H, int W,
                       int y0, int x0, int remap_background, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Input-side tile preparation - replaces the numpy lines of CLOUDdataset.__getitem__,
 * dataloaders/datasets/cloud_dataset.py:127-134.
 *
 * snws_tile_minmax_scale: n_tiles tiles of elems_per_tile consecutive float32 values each (a (C, H, W) tile as
 *   gdal's ReadAsArray returns it, or a stack of them): NaN -> 0, then ((x - min) / (max - min)) * 255 with the
 *   tile's own minimum and maximum over ALL its channels, every operation rounded to float32 like numpy's
 *   (a constant tile gives NaN / inf exactly as numpy's 0/0 does).  y has x's layout, out_dtype SNWS_F32 or SNWS_BF16.
 *   part: 2 * n_tiles * snws_tile_minmax_blocks(elems_per_tile) floats of scratch; minmax_out (nullable): 2 * n_tiles
 *   floats [min, max] per tile.
 * snws_label_remap: dst[i] = float(src[i]) with NaN -> 255, then 10 -> 0, then 255 -> 0 (cloud_dataset.py:132-134).
 *   src_dtype: SNWS_F32 | SNWS_F64 | SNWS_I64 | SNWS_I32 | SNWS_U8. */
int snws_tile_minmax_blocks(int64_t elems_per_tile);
int snws_tile_minmax_scale(const float* x, void* y, int out_dtype, int n_tiles, int64_t elems_per_tile, float* part,
                           float* minmax_out, void* stream);
int snws_label_remap(const void* src, int src_dtype, float* dst, int64_t n, void* stream);

/* snws_nchw_to_nhwc_pad: the planar image batch a data loader delivers (N, C, H, W; train.py:241 `image.cuda()`, consumed by
 *   the first Conv2dws of the model, modeling/UNet3p_SNws.py:98) -> pixel-major (N, H, W, c_pad) in out_dtype with the
 *   channels C .. c_pad-1 zero: the layout the convolution kernels read (14 bands -> 16), one pass.  x: contiguous NCHW of
 *   in_dtype (SNWS_F32 | SNWS_BF16); y: N*H*W*c_pad elements of out_dtype, 16-byte aligned; c_pad: a multiple of 8 (bf16 out) / 4 (f32 out), C <= c_pad <= 48
 *   (the 256-pixel staging tile then fits the default dynamic-LDS limit). */
int snws_nchw_to_nhwc_pad(const void* x, int in_dtype, void* y, int out_dtype, int N, int C, int H, int W, int c_pad,
                          void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused AdamW over a flat float32 buffer - replaces torch.optim.AdamW(model.parameters(), lr, weight_decay)
 * (train.py:81) stepped once per iteration (train.py:254); `lr` is the poly-schedule value of that iteration
 * (utils/lr_scheduler.py:43-70).  param/grad/exp_avg/exp_avg_sq: n floats each, 16-byte aligned; step >= 1 is the
 * 1-based update count (bias correction); the gradient is read as grad * grad_scale.  In place. */
int snws_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                    float beta1, float beta2, float eps, float weight_decay, int64_t step, float grad_scale,
                    void* stream);

/* ---------------------------------------------------------------------------------------------
 * Collectives for hosts without their own - the exchange points of the data-parallel path as in-place all-reduces
 * over RCCL, loaded at run time.  Replaces the SyncMaster / SlavePipe rendez-vous of
 * modeling/sync_batchnorm/comm.py:39-43,71-90 (SwitchNorm / BN statistics: the `pack` buffers of snws_sn_mix_a and
 * snws_sn_bwd_mix_a), DataParallel's gradient reduction (train.py:85,253) and the per-replica confusion matrices
 * (train.py:301).  One communicator per process, created on the current HIP device.
 * snws_comm_unique_id: rank 0 fills 128 bytes, the host hands them to every rank (file, socket, MPI ...). */
typedef struct snws_comm snws_comm;
int snws_comm_unique_id(void* id128);
int snws_comm_init(const void* id128, int rank, int world, snws_comm** out);
int snws_comm_allreduce_f32(snws_comm* comm, float* buf, int64_t count, void* stream);              /* sum */
int snws_comm_allreduce_i64(snws_comm* comm, int64_t* buf, int64_t count, int op, void* stream);    /* op 0 sum, 1 max */
int snws_comm_destroy(snws_comm* comm);

#ifdef __cplusplus
}
#endif
#endif
