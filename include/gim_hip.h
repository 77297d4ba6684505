/* libgim_hip.so — C ABI of the MI355X (gfx950) kernels behind the GIM image training hot path.
 *
 * The reference (roymor1/OptimalStrategiesAgainstGenerativeAttacks) is pure Python/PyTorch and has NO
 * native layer: every entry point below replaces a torch primitive *call site* of the reference's hot
 * path (cited per function as file:line relative to the reference root).  The binding a maintainer
 * adds on the reference side is a ctypes stub, shown in INTEGRATION.md.
 *
 * Conventions
 *  - plain pointers and sizes only; all pointers are DEVICE pointers (fp32 unless noted); no torch types;
 *  - activations are NHWC ([N][H][W][C], C fastest); conv weights are [Cout][KH][KW][Cin]
 *    (= the reference's [Cout,Cin,KH,KW] parameter stored channels-last); linear weights are [out][in];
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*), allocates nothing, owns
 *    nothing and is safe to capture into a hipGraph; the library keeps NO mutable process-wide state and reads NO
 *    environment variables: everything a launch depends on is in its arguments (matrix path and launch overrides are
 *    fields of gim_conv_shape) or in the read-only per-shape launch table compiled into the library
 *    (csrc/conv_tune_table.inc).  Concurrent calls from several threads / streams are independent (gim_last_error()
 *    is thread-local);
 *  - spatial sizes H, W are powers of two (the reference's Encoder/EnvDecoder require img_size = 2^k,
 *    models/gim_img_models.py:30,72);
 *  - return 0 on success, negative GIM_E_* otherwise; gim_last_error() gives a message.
 */
#ifndef GIM_HIP_H
#define GIM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GIM_OK 0
#define GIM_E_BADARG (-1)
#define GIM_E_LAUNCH (-2)

const char* gim_last_error(void);
int gim_version(void);

/* Shape of one stride-1 'same' convolution (KH x KH, pad = (KH-1)/2) with the fused prologue
 * x~ = nearest_up2^ups( leaky_relu(x, pre_slope) ); pre_slope = 1 disables the activation.
 * H, W are the CONVOLUTION's spatial size (= size of x~); the stored input is [N][H>>ups][W>>ups][Cin].
 *   pool    = 1: the output is avgpool2(conv) ([N][H/2][W/2][Cout]; nn.AvgPool2d(2), models/model_blocks.py:502,509),
 *                computed as ONE stride-2 convolution with the (KH+1)^2-tap folded weights (16/36 of the FLOPs
 *                for KH = 3, 100/324 for KH = 9).  Needs ups = 0 and wfold = 1.
 *   wfold   = 1: `w` points to the folded weights F of gim_conv2d_fold_weights ([Cout][KH+1][KH+1][Cin]).  With
 *                ups = 1 the convolution of the upsampled image is computed in its sub-pixel form (4 output-parity
 *                classes, ((KH+1)/2)^2 taps each on the LOW-resolution x): same FLOP ratios as above.
 *   res_ups = 1: (forward) the residual is stored at half resolution and nearest-upsampled on the fly.
 *   tune_tile / tune_ksplit / tune_wgrad : launch overrides for tools/conv_autotune.py and the tuned-row parity tests; 0
 *             (the product's value) = the table row of this shape, else the heuristic.  tune_tile: 128 = 128x128, 641 =
 *             64x128, 1264 = 128x64, 64 = 64x64 output tile, 6432 = 64x64 with a 32-deep K step (forward / dgrad: channel counts that
 *             are multiples of 32, wgrad: 32 pixels per step on the vector path; otherwise as 64), tile code + 20000 (forward / dgrad of
 *             plain 3x3 layers, forward of pool-folded 3x3 layers) = the patch-resident K loop on that tile (the input patch of a tile
 *             stays in LDS for all taps; where the geometry does not allow it the tap-major loop runs), < 0 = ignore the table; tune_ksplit: split-K factor of
 *             fwd / dgrad; tune_wgrad: workgroup target of the wgrad pixel slicing.  Results never depend on them beyond
 *             the summation order.
 *   out_zeroed = 1: (gim_conv2d_fwd / _dgrad / _dgrad_t) the caller guarantees that the output buffer holds zeros.  Launches that
 *             split K over the grid combine their slices with float atomics and otherwise clear the output themselves (one
 *             memset per launch); a caller that hands out outputs from a zero-filled pool saves those launches.
 *   prec       : 0 = fp32 operands on the fp32 MFMA (the reference's arithmetic, every parity bound); 1 = operands rounded to fp16
 *                (nearest even, saturating) when they are staged into LDS, v_mfma_f32_32x32x16_f16 with fp32 accumulation - BASELINE
 *                config 5 "fp16 MFMA", opt-in - on eligible launches only (gathered channels % 32 == 0, >= 32 output channels,
 *                forward / gim_conv2d_dgrad_t / wgrad): gim_conv_launch_plan out[7] == 2 says a launch takes it.
 *   post_slope : (gim_conv2d_fwd only; 0 or 1 = none) LeakyReLU with this slope on the STORED output, y = lrelu(conv + bias +
 *             residual): the LeakyReLU that the reference applies in front of the NEXT conv (models/model_blocks.py:507), done
 *             once per element here instead of once per tap and output tile in that conv's K loop (which the caller then runs
 *             with pre_slope = 1; its dgrad keeps the real slope: the sign of an activated value is the sign of the raw one).
 *             Refused (GIM_E_BADARG) when the launch splits K - slices combine by addition; ask gim_conv_launch_plan first. */
typedef struct {
    int32_t N, H, W, Cin, Cout, KH, ups;
    float pre_slope;
    int32_t pool, wfold, res_ups;
    int32_t tune_tile, tune_ksplit, tune_wgrad;
    int32_t out_zeroed;
    float post_slope;
    int32_t prec;
} gim_conv_shape;

/* F[co][a][b][ci] = sum_{dh,dw in {0,1}} w[co][a-dh][b-dw][ci], a, b in [0, KH]  (out-of-range taps are zero). */
int gim_conv2d_fold_weights(const float* w, float* f, int Cout, int Cin, int KH, void* stream);
/* The folds of MANY convolutions in one launch (a model folds ~20 weight tensors once per optimizer step): `jobs` and `tab` in
 * device memory, tab = n_blocks x {job, chunk of 65536 folded elements}. */
typedef struct {
    const float* w;
    float* f;
    int32_t Cout, Cin, KH, reserved;
} gim_fold_job;
int gim_conv2d_fold_weights_batched(const gim_fold_job* jobs, const int32_t* tab, int n_blocks, void* stream);

/* y = conv(x~, w) / sigma + bias + residual.
 * Replaces F.conv2d behind nn.Conv2d + spectral_norm (models/model_blocks.py:492-495,522-526,744-750,
 * 792-793,836-840), the LeakyReLU in front of it (:505,507,764,768,810,856,860), nn.Upsample (:759,765,
 * 851,857), the residual adds (:512,771,813,863) and nn.Linear (H=W=KH=1; :86,89,786-789,830-833).
 * sigma (device scalar, may be NULL = 1), bias [Cout] (may be NULL), residual [N,H,W,Cout] (may be NULL). */
int gim_conv2d_fwd(const float* x, const float* w, const float* bias, const float* sigma, const float* residual,
                   float* y, const gim_conv_shape* s, void* stream);

/* dx~ = conv_transpose(dy, w) / sigma (autograd of the call sites above w.r.t. the input).
 *   plain / pool : dx is [N,H,W,Cin] (dy is [N,H,W,Cout], or [N,H/2,W/2,Cout] with pool = 1);
 *   ups without wfold: dx is at the upsampled resolution [N,H,W,Cin]; follow with gim_upsample2x_bwd;
 *   ups with wfold   : dx is the gradient of the LOW-resolution input [N,H/2,W/2,Cin] directly.
 * If mask_x != NULL (same shape as dx; not for ups without wfold) the result is multiplied by
 * leaky_relu'(mask_x) with slope pre_slope, i.e. it is the gradient w.r.t. the raw input x. */
int gim_conv2d_dgrad(const float* dy, const float* w, const float* sigma, const float* mask_x, float* dx,
                     const gim_conv_shape* s, void* stream);

/* gim_conv2d_dgrad of a PLAIN convolution with the gradient of a pooled branch of its input folded into the epilogue:
 * dx = leaky_relu'(mask_x) * conv_transpose(dy, w) / sigma + res_scale * nearest_up2(res_half), res_half [N, H/2, W/2, Cin].
 * ResBlockDown reads its input twice (models/model_blocks.py:497-514: conv_r1 on lrelu(x), conv_l1 on the raw x, both pooled later);
 * with res_scale = 0.25 res_half is the gradient w.r.t. avgpool2(x) and the sum is the block's whole input gradient (autograd's
 * add of the two branches).  mask_x and res_half required. */
int gim_conv2d_dgrad_res(const float* dy, const float* w, const float* sigma, const float* mask_x, const float* res_half,
                         float res_scale, float* dx, const gim_conv_shape* s, void* stream);

/* Split-K weight gradient: slabs[i] for i < n_slabs hold partial sums over disjoint pixel ranges of
 * dy^T * im2col(x~): layout [Cout][KH][KW][Cin] (plain), [Cout][KH+1][KH+1][Cin] (pool: gradient of F) or
 * [Cin][KH+1][KH+1][Cout] (ups + wfold: transposed gradient of F; no bias_slabs in this form);  bias_slabs[i] ([Cout] each, may be NULL) the matching partial
 * sums of dy over pixels (the bias gradient, produced from the dy tiles the kernel streams anyway).
 * n_slabs = gim_conv2d_wgrad_slabs(shape): deterministic (each pixel slice writes its own slab, gim_wgrad_finish adds them in
 * a fixed order); n_slabs = 1: the slices add into ONE slab with float atomics (cleared by the call; summation order varies
 * from run to run in the last bits). */
int gim_conv2d_wgrad_slabs(const gim_conv_shape* s);
int gim_conv2d_wgrad(const float* dy, const float* x, float* slabs, float* bias_slabs, int n_slabs, const gim_conv_shape* s,
                     void* stream);

/* Finish a weight gradient: g = sum_i slabs[i] (un-folded to [Cout][KH][KW][Cin] when fold != 0: fold = 1 for
 * pool slabs, 2 for ups + wfold slabs); db = sum_i bias_slabs[i] (if db != NULL);
 *   sigma == NULL : dw = g                                   (plain nn.Linear weight)
 *   else          : dw = g / sigma - (<g, w> / sigma^2) u v^T (autograd through torch spectral_norm's
 *                   weight = weight_orig / (u^T W v) with u, v constants).
 * u [Cout]; v [Cin*KH*KW] in the REFERENCE's flattening order (ci, kh, kw).  scratch: >= 512 floats, plus
 * Cout*(KH+1)^2*Cin floats when fold != 0.  acc_dw / acc_db (may be NULL): ADD the results into these buffers
 * (the optimizer's flat gradient bucket) instead of returning them in dw / db (dw then is scratch). */
int gim_wgrad_finish(const float* slabs, const float* bias_slabs, int n_slabs, const float* w, const float* sigma,
                     const float* u, const float* v, float* dw, float* db, float* scratch, int Cout, int Cin, int KH,
                     int fold, float* acc_dw, float* acc_db, void* stream);

/* One power iteration of torch.nn.utils.spectral_norm (n_power_iterations=1, eps=1e-12, dim=0):
 *   v <- normalize(W^T u); u <- normalize(W v); sigma = u^T W v.       (training = 1)
 *   sigma = u^T W v with the stored u, v.                               (training = 0)
 * u [Cout] and v [Cin*KH*KW] (reference order) are updated in place when training; u_out / v_out receive
 * copies of the vectors used for sigma (saved for backward). scratch: >= 9*Cin*KH*KW + Cout floats. */
int gim_spectral_sigma(const float* w, float* u, float* v, float* sigma, float* u_out, float* v_out, float* scratch,
                       int Cout, int Cin, int KH, int training, void* stream);

/* The same power iteration for ALL spectral-normed convs of a model in one call (4 launches per round instead
 * of 4 per conv call; the iterations do not depend on activations).  jobs / tab_cols / tab_rows are DEVICE arrays,
 * static per model:  tab_cols[i] = {job, 256-column block, row chunk r, rows per chunk} for every
 * (job, block, r < min(8, ceil(Cout/64))),  tab_rows[i] = {job, 4-row block}.  Results of job j land at
 * out_base + off_sigma (1 float), + off_u (Cout), + off_v (Cin*KH*KH); off_scratch needs 10*Cin*KH*KH + Cout floats. */
typedef struct {
    const float* w;
    float* u;
    float* v;
    int64_t off_sigma, off_u, off_v, off_scratch;
    int32_t Cout, Cin, KH, reserved;
} gim_sn_job;
int gim_spectral_sigma_batched(const gim_sn_job* jobs, int n_jobs, const int32_t* tab_cols, int n_col_blocks,
                               const int32_t* tab_rows, int n_row_blocks, float* out_base, int training, void* stream);

/* dgrad on TRANSPOSED weights: the input gradient of gim_conv2d_fwd as gim_conv2d_dgrad computes it (same shape struct, same
 * fused mask / 1/sigma / folds), but reading WT[Cin][KF][KF][Cout] (gim_conv2d_transpose_weights of the plain or folded
 * weights), whose rows are k-contiguous for this contraction: dgrad then runs the forward kernel's operand path (vector
 * weight loads also when the conv has <= 8 INPUT channels: the gradient w.r.t. images).  Needs Cout % 16 == 0.  (F.conv2d's backward w.r.t. the input, autograd of training/gim_img_training.py:164,176.) */
int gim_conv2d_transpose_weights(const float* w, float* wt, int Cout, int Cin, int KF, void* stream);
int gim_conv2d_dgrad_t(const float* dy, const float* wt, const float* sigma, const float* mask_x, float* dx,
                       const gim_conv_shape* shape, void* stream);

/* The same gradient for PLAIN convolutions (no pool / ups / fold) with <= 8 input channels, x-folded: J horizontally adjacent dx
 * pixels become the J * Cin output columns of one stride-(1, J) convolution of dy with K x (K + J - 1) taps - [N, H, W / J, J * Cin]
 * is [N, H, W, Cin] in memory - so that the 16-column MFMA tile carries 12 useful columns (Cin = 3: J = 4; Cin = 6: J = 2) instead of
 * 3 or 6.  gim_conv2d_xfold_weights builds WX[(j, ci)][a][u][co] = W[co][K-1-a][K-1-(u-j)][ci] (0 outside 0 <= u-j < K; J * Cin * K *
 * (K + J - 1) * Cout floats) once per weight update.  J: power of two dividing W, J * Cin <= 32; Cout % 16 == 0.
 * (F.conv2d's backward w.r.t. the input images, autograd of training/gim_img_training.py:164,176.) */
int gim_conv2d_xfold_weights(const float* w, float* wx, int Cout, int Cin, int KH, int J, void* stream);
int gim_conv2d_dgrad_xfold(const float* dy, const float* wx, const float* sigma, const float* mask_x, float* dx,
                           const gim_conv_shape* shape, int J, void* stream);

/* Row-contiguous form of the image layers (<= 8 input channels, plain stride-1 convolution, KH * Cin <= 64, >= 16 output channels):
 * in NHWC memory the KH taps of one tap ROW of an output pixel are KH * Cin contiguous floats, so on a zero-padded copy of the image
 * the convolution is an implicit GEMM over KH "taps" of CaP = KH * Cin rounded up to 16 "channels" with a pixel stride of Cin floats -
 * 16-byte vector loads where the generic path gathers scalars (the first conv of every encoder, models/gim_img_models.py:30-36 ->
 * models/model_blocks.py:493,505; the 9x9 6 -> 64 conv of the image-to-image module, models/gim_img_models.py:123).
 *   gim_pad_image: xp [N, H + 2 pad, W + 2 pad, C] = lrelu(x, slope) with a zero border (the conv's LeakyReLU is applied here, once).
 *   gim_conv2d_pack_rows_weights: wp [Cout][KH][CaP] = the rows of w [Cout][KH][KH][Cin], zero-padded.
 *   gim_conv2d_fwd_rows: y = conv / sigma + bias + residual (+ post_slope) from xp and wp; shape = the convolution's own shape.
 *   gim_conv2d_wgrad_rows_acc: dW slot [Cout][KH][CaP] += dy^T * rows(xp), bias slot += column sums of dy (pre-zeroed slots, float
 *     atomics); gim_wgrad_finish_batched un-pads the slot with gim_wgrad_job.fold = 3. */
int gim_pad_image(const float* x, float* xp, int N, int H, int W, int C, int pad, float slope, void* stream);
int gim_conv2d_pack_rows_weights(const float* w, float* wp, int Cout, int Cin, int KH, void* stream);
int gim_conv2d_fwd_rows(const float* xp, const float* wp, const float* bias, const float* sigma, const float* residual,
                        float* y, const gim_conv_shape* shape, void* stream);
int gim_conv2d_wgrad_rows_acc(const float* dy, const float* xp, float* acc, float* bias_acc, const gim_conv_shape* shape, void* stream);

/* Sub-pixel convolution to <= 4 output channels (the generator's last layer, 9x9 64 -> 3 behind nn.Upsample: models/gim_img_models.py:187-193
 * -> models/model_blocks.py:752-773): for KH = 5, 9, 13 the four output-parity classes read the same ((KH+1)/2)^2 window of the
 * low-resolution input, so they stack into ONE plain ((KH+1)/2)-tap convolution to 4 * Cout channels (gim_conv2d_fwd on wm, no bias)
 * followed by a depth-to-space copy that adds the bias.  Forward only: the gradients run the sub-pixel forms of gim_conv2d_dgrad / wgrad.
 *   gim_conv2d_pack_subpixel_weights: wm [4 Cout][(KH+1)/2][(KH+1)/2][Cin] from the folded taps wf [Cout][KH+1][KH+1][Cin]
 *     (gim_conv2d_fold_weights), row (2 py + px) * Cout + co = output parity (py, px) of channel co.
 *   gim_depth_to_space2: y [N, 2 Hs, 2 Ws, C][n][2 h + py][2 w + px][c] = y4 [N, Hs, Ws, 4 C][n][h][w][(2 py + px) * C + c] + bias[c],
 *     stored as lrelu(., post_slope) when post_slope != 1 (the activated-output convention of gim_conv_shape.post_slope). */
int gim_conv2d_pack_subpixel_weights(const float* wf, float* wm, int Cout, int Cin, int KH, void* stream);
int gim_depth_to_space2(const float* y4, const float* bias, float* y, int N, int Hs, int Ws, int C, float post_slope, void* stream);

/* Introspection (tests, tools/conv_autotune.py): the launch an entry point would make for `shape`, nothing is launched.
 *   kind 0 = gim_conv2d_fwd, 1 = gim_conv2d_dgrad, 2 = gim_conv2d_dgrad_t, 3 = gim_conv2d_wgrad_acc
 *   out[8] = {1 if a row of the compiled-in per-shape launch table matched, tile rows, tile columns, split-K factor
 *             (wgrad: pixel slices), grid x, y, z, loop form: 0 = tap-major K loop, 1 = patch-resident (plain 3x3 layers: the input
 *             patch of a tile stays in LDS for all nine taps; tune_tile + 20000 selects it explicitly), 2 = fp16 operands, 3 = direct
 *             kernel of the 3 -> 3 / 1 -> 1 image layers}; out[7] bits 8 and up: the share of the launch's K steps that is SKIPPED,
 *             in 1/1000 - the tap-major loop on maps of <= 16 pixels and >= 32 images orders its GEMM rows by pixel position and skips,
 *             per tile, the taps that fall into the zero padding for all of its rows (31 % of a 3x3 convolution on a 4x4 map).
 * A batch beyond the 32-bit buffer-offset range (the entry points then halve it) reports the plan of its last half. */
int gim_conv_launch_plan(const gim_conv_shape* shape, int kind, int32_t* out);

/* Accumulating weight gradient: ADDS the gradient of one convolution (raw, un-finished: dW, dF or G layout as in
 * gim_conv2d_wgrad) into `acc` / `bias_acc` with float atomics and clears nothing - the caller hands in zeroed (or
 * partially accumulated) buffers, typically slots of one arena cleared once per backward pass. */
int gim_conv2d_wgrad_acc(const float* dy, const float* x, float* acc, float* bias_acc, const gim_conv_shape* shape, void* stream);

/* Batched form of gim_wgrad_finish for ALL convolutions / linears of one backward pass (two launches): un-fold, spectral-norm
 * chain rule dW = G/sigma - <G,W>/sigma^2 u v^T (torch.nn.utils.spectral_norm backward; sigma == NULL: plain), ADD into
 * grad_w / grad_b.  tab: n_blocks x {job, chunk} with chunks of 4096 elements of Cout*K*K*Cin; tab_sn: the same for the
 * spectral-norm jobs only.  partial >= n_chunks floats; tmp >= Cout*K*K*Cin floats when fold != 0 and sigma != NULL.
 * Cout*(K+1)^2*Cin < 2^31 per job (32-bit index arithmetic on the device; the job table lives in device memory, so the
 * caller checks). */
typedef struct {
    const float* src;
    const float* bias_src;
    const float* w;
    const float* sigma;
    const float* u;
    const float* v;
    float* tmp;
    float* partial;
    float* grad_w;
    float* grad_b;
    int32_t Cout, Cin, K, fold;
    int32_t n_chunks, exclusive;   /* exclusive = 1: no other job of this call adds into grad_w (plain read-modify-write instead of float atomics) */
} gim_wgrad_job;
int gim_wgrad_finish_batched(const gim_wgrad_job* jobs, int n_jobs, const int32_t* tab, int n_blocks, const int32_t* tab_sn,
                             int n_blocks_sn, void* stream);

/* Column sums: out[c] = sum_r x[r][c]  (bias gradients; rows x C). scratch >= 256*C floats. */
int gim_colsum(const float* x, float* out, float* scratch, int64_t rows, int C, void* stream);

/* gim_colsum that ADDS into out (bias gradients straight into the optimizer's flat gradient bucket). */
int gim_colsum_acc(const float* x, float* out, float* scratch, int64_t rows, int C, void* stream);
/* Two short column sums in one launch: out_a[c] (+)= sum_r a[r][c], out_b[c] (+)= sum_r b[r][c]  (InstanceNorm affine
 * gradients from the per-image partials of gim_norm_bwd; accumulate != 0 adds into the outputs). */
int gim_colsum2(const float* a, const float* b, float* out_a, float* out_b, int rows, int C, int accumulate, void* stream);

/* Instance norm / AdaIN over H*W per (n, c) on NHWC data.
 *  mode 0: nn.InstanceNorm2d(affine=True), biased var, eps inside the sqrt (models/gim_img_models.py:126,
 *          models/model_blocks.py:747-748);  scale/shift are [C].
 *  mode 1: ada_in (models/model_blocks.py:611-630): unbiased std, eps added to the std; scale/shift are [N][C].
 * y = scale * (x - mean) / d + shift (+ residual).  stats [N][C][3] = {mean, 1/d, c2} saved for backward. */
int gim_norm_fwd(const float* x, const float* scale, const float* shift, const float* residual, float* y, float* stats,
                 int N, int HW, int C, int mode, float eps, void* stream);
/* The same with the output stored ACTIVATED, y = lrelu(scale * xhat + shift (+ residual), post_slope): the LeakyReLU the
 * reference applies between a norm and the conv behind it (models/model_blocks.py:764,768,810,856,860), done here once per
 * element so that that conv (gim_conv_shape.pre_slope = 1 for its forward and wgrad, the real slope for its dgrad mask) does
 * not redo it per tap.  The backward (gim_norm_bwd) reads x and stats only and is unchanged. */
int gim_norm_fwd_act(const float* x, const float* scale, const float* shift, const float* residual, float* y, float* stats,
                     int N, int HW, int C, int mode, float eps, float post_slope, void* stream);
/* dx, and per-(n,c) partials dscale_nc = sum dy*xhat, dshift_nc = sum dy  ([N][C] each). */
int gim_norm_bwd(const float* dy, const float* x, const float* scale, const float* stats, float* dx, float* dscale_nc,
                 float* dshift_nc, int N, int HW, int C, int mode, void* stream);

/* 2x2 average pool (nn.AvgPool2d(2), models/model_blocks.py:502,509) NHWC; H, W are the INPUT size. */
int gim_avgpool2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream);
/* The same pool on a tensor that its producer stored ACTIVATED (lrelu(x, in_slope), gim_conv_shape.post_slope / gim_norm_fwd_act /
 * gim_scale_add_fwd_act): LeakyReLU is inverted on the fly (x = a for a > 0, a / in_slope otherwise), so that the skip path of a
 * ResBlockDown (models/model_blocks.py:499-503, which reads the RAW block input) can share the activated copy its conv path reads. */
int gim_avgpool2_fwd_act(const float* x, float* y, int N, int H, int W, int C, float in_slope, void* stream);
/* Gradient fan-in of a tensor with several consumers in ONE pass (autograd's AccumulateGrad / input-buffer additions, one launch
 * per extra consumer): out = a + b (+ c) (+ d) (c, d may be NULL);  and the ResBlockDown form (models/model_blocks.py:499-510:
 * x feeds the conv path and, through AvgPool2d, the skip path): out = g + avgpool2_bwd(dy_pooled), g and out [N][H][W][C]. */
int gim_add_n(const float* a, const float* b, const float* c, const float* d, float* out, int64_t n, void* stream);
int gim_add_avgpool2_bwd(const float* g, const float* dy_pooled, float* out, int N, int H, int W, int C, void* stream);
int gim_avgpool2_bwd(const float* dy, float* dx, int N, int H, int W, int C, void* stream);
/* Backward of nearest x2 upsampling: dx[n,h,w,c] = sum of the 2x2 block of dy_up; optional
 * multiplication by leaky_relu'(mask_x) (slope).  H, W are the LOW-resolution size. */
int gim_upsample2x_bwd(const float* dy_up, const float* mask_x, float slope, float* dx, int N, int H, int W, int C, void* stream);

/* Global spatial max + LeakyReLU (nn.AdaptiveMaxPool2d((1,1)) + LeakyReLU(0.2), models/gim_img_models.py:53-57).
 * idx [N][C] int32 receives the arg-max pixel (first maximum in row-major scan, as torch). */
int gim_maxpool_lrelu_fwd(const float* x, float* y, int32_t* idx, int N, int HW, int C, float slope, void* stream);
int gim_maxpool_lrelu_bwd(const float* dy, const float* y, const int32_t* idx, float* dx, int N, int HW, int C, float slope, void* stream);

/* Attention probabilities in ONE kernel: P[b][i][j] = softmax over i of f[b][i][:] . g[b][j][:]  - the energy bmm and the
 * Softmax(dim=-2) of SelfAttention.forward (models/model_blocks.py:540-543) without the round trip of the T x T energy through
 * memory.  f, g [batch][T][K] (token-major, K contiguous), P [batch][T][T].  Built for T = 256, K = 16 (GIM_E_BADARG otherwise:
 * callers fall back to gim_bgemm + gim_softmax_dim1_fwd).  The backward is the unfused one (gim_softmax_dim1_bwd + gim_bgemm). */
int gim_attn_prob_fwd(const float* f, const float* g, float* P, int batch, int T, int K, void* stream);

/* Batched strided GEMM  C[b](i,j) = sum_k A[b](i,k) * B[b](k,j)  (torch.bmm in SelfAttention,
 * models/model_blocks.py:539-544 and their autograd).  Element strides; C is row-major [b][M][N]. */
int gim_bgemm(const float* A, const float* B, float* C, int batch, int M, int N, int K,
              int64_t sAb, int64_t sAi, int64_t sAk, int64_t sBb, int64_t sBk, int64_t sBj, void* stream);

/* Grouped form: n independent small products C = A * B (+ bias), each with its own operands, element strides and output, in ONE
 * launch - the 36 style projections nn.Linear(style_dim, C) of the generator's AdaIN blocks on one shared style matrix
 * (models/model_blocks.py:786-789,829-832 and their autograd: forward, input gradient, weight gradient = three launches instead
 * of 108).  `jobs` and `tiles` live in DEVICE memory; tiles = n_tiles x {job, row tile, column tile} of 64 x 64 outputs.
 *   flags bit 0: C += product (one writer per element);  bit 1: all jobs add into one pre-zeroed C with float atomics.
 * gim_colsum_grouped: C[j] (+)= sum_i A[i * sAi + j * sAk] per job (bias gradients; tiles = {job, 0, block of 64 columns}). */
typedef struct {
    const float* A;
    const float* B;
    float* C;
    const float* bias;           /* per output column, or NULL */
    int32_t M, N, K, ldc;        /* C is row-major with leading dimension ldc */
    int64_t sAi, sAk, sBk, sBj;  /* element strides of A(i, k) and B(k, j) */
    int32_t flags, reserved;
} gim_gemm_job;
int gim_bgemm_grouped(const gim_gemm_job* jobs, const int32_t* tiles, int n_tiles, void* stream);
int gim_colsum_grouped(const gim_gemm_job* jobs, const int32_t* tiles, int n_tiles, void* stream);

/* Softmax over dim -2 of [B][R][Ccols] (nn.Softmax(-2), models/model_blocks.py:528,540): columns sum to 1. */
int gim_softmax_dim1_fwd(const float* s, float* p, int B, int R, int Ccols, void* stream);
int gim_softmax_dim1_bwd(const float* dp, const float* p, float* ds, int B, int R, int Ccols, void* stream);

/* y = gamma * a + x   (models/model_blocks.py:548), gamma a device scalar.  Backward: da = gamma*dy,
 * dgamma = sum(dy * a) (device scalar).  scratch: >= 2048 floats. */
int gim_scale_add_fwd(const float* a, const float* x, const float* gamma, float* y, int64_t n, void* stream);
/* y = lrelu(gamma * a + x, post_slope): the SelfAttention output (models/model_blocks.py:548) stored activated for the block behind it. */
int gim_scale_add_fwd_act(const float* a, const float* x, const float* gamma, float* y, int64_t n, float post_slope, void* stream);
int gim_scale_add_bwd(const float* dy, const float* a, const float* gamma, float* da, float* dgamma, float* scratch,
                      int64_t n, void* stream);

/* tanh (models/gim_img_models.py:215) */
int gim_tanh_fwd(const float* x, float* y, int64_t n, void* stream);
int gim_tanh_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream);

/* Layout changes at the API boundary: NCHW <-> NHWC for [N][C][H*W]. */
int gim_nchw_to_nhwc(const float* x, float* y, int N, int C, int HW, void* stream);
int gim_nhwc_to_nchw(const float* x, float* y, int N, int C, int HW, void* stream);

/* Set statistics of x [B][t][D] (models/gim_basic_models.py:34,51; models/model_blocks.py:41-48):
 * mean over t and custom_std = sqrt(var_unbiased + 1e-8) (zeros when t == 1), written with row stride
 * `ld` (so they can land inside the [B][5120] head input, models/gim_img_models.py:297). std may be NULL. */
int gim_set_stats_fwd(const float* x, float* mean, float* std, int B, int t, int D, int64_t ld_mean, int64_t ld_std, void* stream);
/* dx = dmean/t + dstd * (x - mean) / ((t-1) * std);  dmean/dstd rows have stride ld_*; either may be NULL. */
int gim_set_stats_bwd(const float* x, const float* dmean, const float* dstd, float* dx, int B, int t, int D,
                      int64_t ld_dmean, int64_t ld_dstd, void* stream);

/* BCE-with-logits against a constant target, no reduction (training/gim_img_trainer.py:90-94):
 * loss = max(x,0) - x*t + log1p(exp(-|x|));  dx = dloss * (sigmoid(x) - t). */
int gim_bce_logits_fwd(const float* x, float* loss, float target, int n, void* stream);
int gim_bce_logits_bwd(const float* dloss, const float* x, float* dx, float target, int n, void* stream);

/* y[b][d] = scale * sum_j x[b][j][d]  (x.mean(1), models/gim_img_models.py:289-290,370-371; and the backward
 * of an expand over the sample dim) and its transpose y[b][j][d] = scale * x[b][d]. */
int gim_sum_dim1(const float* x, float* y, int B, int t, int D, float scale, void* stream);
int gim_repeat_dim1(const float* x, float* y, int B, int t, int D, float scale, void* stream);

/* Generator noise combine (models/gim_img_models.py:378-380): y[b][j] = env[b] + w[b][j] - mean_j w[b][j]
 * (mean term only if remove_mean).  With env == NULL it is also its own backward w.r.t. w. */
int gim_noise_combine(const float* env, const float* w, float* y, int B, int t, int D, int remove_mean, void* stream);

/* Channel concat of NHWC rows with a broadcast second operand (torch.cat((env_img, expanded_img), dim=2),
 * models/gim_img_models.py:367,385): a [R][Ca]; b holds R/(P*rep) images of P pixels x Cb channels, each
 * repeated for `rep` consecutive images of a.  gim_slice_channels is the backward w.r.t. a. */
int gim_concat2(const float* a, const float* b, float* y, int64_t R, int Ca, int Cb, int P, int rep, void* stream);
int gim_slice_channels(const float* dy, float* da, int64_t R, int Ca, int Cy, void* stream);

/* Second-order pieces of the R1 regulariser (training/utils.py:115-124: autograd.grad(create_graph=True) of the
 * authenticator output w.r.t. its input images, then backward through that gradient).  Convolutions need nothing
 * new (the adjoint of dgrad is the forward conv, its weight derivative is wgrad); these are the other adjoints. */
int gim_maxpool_gather(const float* gdx, const float* y, const int32_t* idx, float* gdy, int N, int HW, int C, float slope, void* stream);
int gim_softmax_dim1_bwd_dp(const float* gs, const float* dp, const float* p, float* gp, int B, int R, int Ccols, void* stream);
int gim_set_stats_bwd_bwd(const float* x, const float* dstd, const float* g, float* g_dmean, float* g_dstd, float* gx, int B, int t, int D,
                          int64_t ld_dstd, int64_t ld_out, void* stream);
int gim_lrelu_mask_mul(const float* g, const float* x, float slope, float* out, int64_t n, void* stream);
/* out[b] = sum_i x[b][i]^2 over L elements per episode, and dx = 2 * x * dout[b]. */
int gim_sqsum_rows_fwd(const float* x, float* out, int B, int64_t L, void* stream);
int gim_sqsum_rows_bwd(const float* x, const float* dout, float* dx, int B, int64_t L, void* stream);

/* Input pipeline: gather episodes from a resident uint8 NHWC image bank into float NCHW samples in [-1, 1] with an optional
 * horizontal flip per image - load_image / process_pil_image / RandomHorizontalFlip of data_handling/img_datasets.py:296-303,43-46
 * after the (offline) resize.  out[i] = (bank[idx[i]] / 255) * 2 - 1. */
int gim_episode_gather(const uint8_t* bank, const int32_t* idx, const uint8_t* flip, float* out, int n_out, int H, int W, int C,
                       void* stream);

/* ImgAttention mix (models/model_blocks.py:598-608; only with use_img_att): per pixel (P pixels, C channels, NHWC)
 * s1 = sum_c q1*k1, s2 = sum_c q2*k2, (a1, a2) = softmax(s1, s2), out = x1*a1 + v2*a2; att [P] keeps a1. */
int gim_img_att_mix_fwd(const float* q1, const float* k1, const float* q2, const float* k2, const float* x1, const float* v2,
                        float* out, float* att, int64_t P, int C, void* stream);
int gim_img_att_mix_bwd(const float* dout, const float* q1, const float* k1, const float* q2, const float* k2, const float* x1,
                        const float* v2, const float* att, float* dq1, float* dk1, float* dq2, float* dk2, float* dx1, float* dv2,
                        int64_t P, int C, void* stream);

/* Fused multi-tensor Adam on flat buffers (torch.optim.Adam form, no weight decay;
 * training/gim_img_trainer.py:50-58).  seg_end[n_seg] are exclusive element offsets of the parameter
 * groups, lr[n_seg] their learning rates (device arrays).  `step` is a device int32 incremented by the call
 * (graph-replay safe).  grad_scale multiplies the gradient first (1/world_size after an all-reduce sum). */
int gim_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const int64_t* seg_end, const float* lr,
                  int n_seg, float beta1, float beta2, float eps, float grad_scale, int32_t* step, void* stream);

/* Stream self-check: one wave busy for `usec` microseconds (1 .. 5000, constant 100 MHz wall clock) on `stream`.  The host
 * launches one per engine stream at the same moment and event-times the total: streams that share a HIP hardware queue serialize
 * (nn.DataParallel's one-thread-per-device streams of training/gim_img_training.py:406-411 have no such aliasing to check). */
int gim_spin(int usec, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GIM_HIP_H */
