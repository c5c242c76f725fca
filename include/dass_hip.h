/*
 * dass_hip.h -- C-ABI of libdass_hip.so, the MI355X (gfx950) device library under the
 * DeepLab-v3+ training / active-selection scoring hot path.
 *
 * The reference (nihalsid/deep-active-semantic-segmentation) has no FFI of its own: its
 * device work is whatever ATen dispatches for the Python call sites cited on each entry
 * point below.  These entry points are what the host-side mirror of the reference
 * surface (models.deeplab.DeepLab, utils.loss.SegmentationLosses, active_selection.*)
 * binds through ctypes -- see INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 on success, a DASS_ERR_* code otherwise; nothing here
 *     allocates, synchronises or touches the host: callers own every buffer.
 *   - `stream` is a hipStream_t passed as void*.
 *   - activations are NHWC ("pixel rows"): element (n,h,w,c) of a tensor lives at
 *     ptr + ((n*H + h)*W + w)*ld + c, where ld >= C is the pixel stride in ELEMENTS.
 *     A channel slice of a wider buffer is (ptr + c_off, ld = C_total).
 *   - dtype: DASS_F32 (f32 tensors, f32 MFMA), DASS_BF16 (storage bf16, f32 accumulate), or -- conv entry points
 *     only -- DASS_F32X3: f32 tensors exactly as DASS_F32, products formed on the bf16 MFMA as
 *     a_hi*b_hi + a_lo*b_hi + a_hi*b_lo with f32 accumulation (per-product error <= 2^-17, i.e. the size of the
 *     f32 accumulation rounding; SURVEY.md 7.1 "fp32-parity: fp32-MFMA or 3xbf16-split").
 *   - conv weights are "KRSC": w[k][r][s][c], i.e. an OIHW tensor in channels_last memory.
 *   - logits at the module boundary are NCHW f32, as the reference returns them.
 */
#ifndef DASS_HIP_H
#define DASS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DASS_OK 0
#define DASS_ERR_ARG 1      /* bad shape / alignment / null pointer */
#define DASS_ERR_LAUNCH 2   /* hipGetLastError() after a launch      */
#define DASS_ERR_UNSUPPORTED 3

#define DASS_F32 0
#define DASS_BF16 1
#define DASS_F32X3 2 /* conv2d_igemm / conv2d_igemm_stats / conv2d_wgrad only */
#define DASS_F32X6 3 /* same entry points: three-way bf16 split, six products -- f32-exact products */
#define DASS_F16X3 4 /* weight-operand format of the pre-split kernels in their two-part mode (dass_set_x3_parts(2)):
                        w * s as two f16 parts, s = a per-tensor power of two; three products per pair.  See "x3 rows". */

#define DASS_BF16X1 5 /* weight-operand format of the pre-split kernels in their ONE-part mode (dass_set_x3_parts(1), the "bf16x1" perf
                         engine): bf16(w), one product per pair -- what autocast-bf16 multiplies; NOT parity-grade */

#define DASS_ACT_NONE 0
#define DASS_ACT_RELU 1
#define DASS_ACT_RELU6 2

/* library / build identification */
int dass_version(void);
const char *dass_arch(void);

/* ---------------------------------------------------------------- convolution
 * Implicit-GEMM NHWC convolution on MFMA; replaces every nn.Conv2d(groups=1) forward on the
 * path (models/backbone/resnet.py:11-15,65,101-104; mobilenet.py:14,51-66; aspp.py:13,63,67;
 * decoder.py:25,29-36) and, with ustride>1 / flipped weights, their input gradients.
 *
 *   y[n,oh,ow,k] = act( scale[k]*( sum_{r,s,c} x[n,iy,ix,c]*in_scale[n,c]*w[k,r,s,c] ) + shift[k]
 *                       + residual[n,oh,ow,k] )
 *   iy = oh*stride - pad + r*dil ; when ustride>1 the tap contributes only if iy%ustride==0
 *   and then iy/=ustride (transposed-conv / dgrad addressing).  Same for ix.
 * scale, shift, residual, in_scale may be NULL.  C and ldx, K-offsets must keep 16-byte alignment
 * (C % 4 == 0 for f32, C % 8 == 0 for bf16).
 */
int dass_conv2d_igemm(const void *x, int64_t ldx, const void *w, void *y, int64_t ldy,
                      const float *scale, const float *shift,
                      const void *residual, int64_t ldr, const float *in_scale,
                      int N, int H, int W, int C, int OH, int OW, int K,
                      int R, int S, int stride, int pad, int dil, int ustride,
                      int act, int dtype, void *stream);

/* Forward conv for train-mode BN: same kernel, raw output (no epilogue), plus the batch statistics fused into the
 * producer: stat_partial[row][0][k] = sum over the row's pixel tile of y[.,k], [row][1][k] = sum of squares.
 * stat_partial must hold dass_conv2d_igemm_stats_rows(M) rows of 2*K floats; *stat_rows (HOST int) receives the
 * number of rows actually written, to be handed to dass_bn_finalize. */
int dass_conv2d_igemm_stats_rows(int64_t M);
int dass_conv2d_igemm_stats(const void *x, int64_t ldx, const void *w, void *y, int64_t ldy,
                            int N, int H, int W, int C, int OH, int OW, int K,
                            int R, int S, int stride, int pad, int dil, int dtype,
                            float *stat_partial, int *stat_rows, void *stream);

/* dw[k][r][s][c] (f32, KRSC, ld = C) = sum_pixels dy[n,oh,ow,k] * x[n,iy,ix,c]*in_scale[n,c];
 * dw is zeroed inside (hipMemsetAsync on `stream`) then accumulated with f32 atomics.
 * Replaces the weight gradient of the same nn.Conv2d sites. */
int dass_conv2d_wgrad(const void *x, int64_t ldx, const void *dy, int64_t lddy, float *dw,
                      int N, int H, int W, int C, int OH, int OW, int K,
                      int R, int S, int stride, int pad, int dil, int dtype, void *stream);
/* same, ACCUMULATING into dw (dw += gradient; no zero fill): the caller hands a zeroed slice of one arena it cleared
 * with a single memset for all layers of a backward pass, or a gradient buffer it is accumulating into */
int dass_conv2d_wgrad_acc(const void *x, int64_t ldx, const void *dy, int64_t lddy, float *dw,
                      int N, int H, int W, int C, int OH, int OW, int K,
                      int R, int S, int stride, int pad, int dil, int dtype, void *stream);

/* Network stems (resnet.py:65 7x7/s2, mobilenet.py:14 3x3/s2 over the 3-channel image): "row-tap" form of
 * the same implicit GEMM.  x is a DENSE NHWC image [N,H,W,Cin] (ld = Cin, no channel padding); a tap is a
 * whole kernel row, whose S*Cin input values are contiguous, so the reduction is R slabs of S*Cin (<= 32)
 * instead of R*S slabs of a padded channel.  w is KRSC (= [K][R][S*Cin]); dilation 1; f32 only. */
int dass_conv2d_rowtap(const void *x, const void *w, void *y, int64_t ldy,
                       int N, int H, int W, int Cin, int OH, int OW, int K,
                       int R, int S, int stride, int pad, int dtype, void *stream);
int dass_conv2d_rowtap_wgrad(const void *x, const void *dy, int64_t lddy, float *dw,
                             int N, int H, int W, int Cin, int OH, int OW, int K,
                             int R, int S, int stride, int pad, int dtype, void *stream);

/* KRSC f32 master weights -> device operand.  mode 0: cast/copy [K][R][S][Csrc] -> [K][R][S][Cdst]
 * (zero pad or truncate channels); mode 1: dgrad operand [C][R][S][K] with taps flipped.
 * dtype DASS_F32X6: the operand dass_conv2d_igemm(.., dtype = DASS_F32X6) REQUIRES as `w`: every weight split into
 * its three bf16 parts, [rows][R*S][ceil(red/32)][3][32] bf16 (rows/red = K/Cdst for mode 0, C/K for mode 1),
 * dass_weight_split_bytes(rows, R, S, red) bytes.  (DASS_F32X3 and DASS_F32 take the plain f32 operand.) */
int dass_weight_transform(const float *src, void *dst, int K, int R, int S, int Csrc, int Cdst,
                          int mode, int dtype, void *stream);
int64_t dass_weight_split_bytes(int rows, int R, int S, int red);
/* bytes of the pre-split operand of `dtype`: DASS_F32X6 (= dass_weight_split_bytes) or DASS_F16X3 (128 B per slab + trailer) */
int64_t dass_weight_operand_bytes(int rows, int R, int S, int red, int dtype);
/* the DASS_F32X6 transform of n weights in one launch (after an optimizer step every conv weight is stale at once):
 * desc = n x 8 int64 on the device {src ptr, dst ptr, K, R, S, Csrc, Cdst, mode}, start[w] = sum over earlier weights of
 * their tiles ceil(rows/32)*R*S*ceil(red/32) (32 operand rows x one 32-wide slab), total = the sum over all n. */
int dass_weight_split_batch(const void *desc, const int64_t *start, int n, int64_t total, void *stream);
/* the same table, DASS_F16X3 operands: zero the trailers, max |w| per tensor (the scale), split -- three launches */
int dass_weight_split_batch_f16(const void *desc, const int64_t *start, int n, int64_t total, void *stream);
/* the same table, DASS_BF16X1 operands (64 B per slab + trailer): one launch */
int dass_weight_split_batch_bf16(const void *desc, const int64_t *start, int n, int64_t total, void *stream);

/* depthwise 3x3 (MobileNetV2 InvertedResidual, mobilenet.py:49,59): w[c][3][3] f32 */
int dass_dwconv3x3_fwd(const void *x, int64_t ldx, const float *w, void *y, int64_t ldy,
                       int N, int H, int W, int C, int OH, int OW,
                       int stride, int pad, int dil, int dtype, void *stream);
int dass_dwconv3x3_bwd_data(const void *dy, int64_t lddy, const float *w, void *dx, int64_t lddx,
                            int N, int H, int W, int C, int OH, int OW,
                            int stride, int pad, int dil, int dtype, void *stream);
int dass_dwconv3x3_bwd_weight(const void *x, int64_t ldx, const void *dy, int64_t lddy, float *dw,
                              int N, int H, int W, int C, int OH, int OW,
                              int stride, int pad, int dil, int dtype, void *stream);
/* dass_dwconv3x3_fwd (f32, stride 1 or 2, dilation 1 or 2) that also adds every output channel's sum and sum of squares into sums
 * ([2][C] f64, zeroed by the caller): the batch statistics of the train-mode BN behind the conv (models/backbone/mobilenet.py:45-50)
 * without dass_channel_sums' pass over y.  DASS_ERR_UNSUPPORTED outside the specialisation: the caller runs the two passes. */
int dass_dwconv3x3_fwd_sums(const void *x, int64_t ldx, const float *w, void *y, int64_t ldy,
                            int N, int H, int W, int C, int OH, int OW, int stride, int pad, int dil, double *sums, void *stream);
/* dass_dwconv3x3_bwd_data (f32, stride 1, dilation 1 or 2, lddx == C) whose result dx [N*H*W][C] IS the gradient d_out of the
 * conv + BN (+ act) layer that produced the depthwise conv's input -- MobileNetV2's expand 1x1 of an inverted-residual block
 * (models/backbone/mobilenet.py:52-58).  The launch also adds THAT layer's BN-backward sums (sum dz, sum dz xhat, max |dz|; dz = dx *
 * act'(fma(bn_y, gate_scale, gate_shift))) into bn_sums ([2][C] f64 + C floats, zeroed by the caller) -- what dass_bn_bwd_reduce_sums
 * computes in a pass of its own over dx and bn_y [N*H*W][C]; the depthwise counterpart of dass_conv2d_x3_dgrad_bnstats.
 * DASS_ERR_UNSUPPORTED for any other stride / dilation / row pitch: the caller runs the two passes. */
int dass_dwconv3x3_bwd_data_bnstats(const void *dy, int64_t lddy, const float *w, void *dx, int64_t lddx,
                                    int N, int H, int W, int C, int OH, int OW, int stride, int pad, int dil,
                                    const float *bn_y, const float *bn_mean, const float *bn_invstd,
                                    const float *gate_scale, const float *gate_shift, int bn_act, double *bn_sums, void *stream);

/* ---------------------------------------------------------------- batch norm (+ReLU/ReLU6, residual, Dropout2d scale)
 * Replaces F.batch_norm at every batchnorm(...) site + the ReLU that follows it. */

/* number of partial rows dass_channel_stats / dass_bn_bwd_reduce write for M pixel rows */
int dass_stat_rows(int64_t M);
/* partial[row][0][k] = sum_m x[m,k], partial[row][1][k] = sum_m x[m,k]^2 over a slab of rows */
int dass_channel_stats(const void *x, int64_t ldx, int64_t M, int K, float *partial, int dtype, void *stream);
/* reduce partials in f64; count = M*rep ("rep" = how often each of the M rows is logically repeated,
 * used by the ASPP image-pool branch whose BN sits after a 1x1 -> HxW broadcast, aspp.py:79-81).
 * Writes mean, invstd, scale=gamma*invstd, shift=beta-mean*scale; updates running stats when
 * momentum >= 0 (running_var uses the unbiased variance, as torch). */
int dass_bn_finalize(const float *partial, int rows, int K, double count, double rep,
                     const float *gamma, const float *beta, float *running_mean, float *running_var,
                     float momentum, float eps, float *mean, float *invstd, float *scale, float *shift,
                     void *stream);
/* SyncBN (models/sync_batchnorm/batchnorm.py:48-125 semantics, one process per GPU): sums[0][k] = global sum x,
 * sums[1][k] = global sum x^2 after the RCCL all-reduce, count = global element count per channel.
 * clamp_var=1: invstd = clamp(biased_var, eps)^-1/2 as the reference's vendored SyncBN; 0: (var+eps)^-1/2. */
int dass_bn_finalize_sums(const float *sums, int K, double count, const float *gamma, const float *beta,
                          float *running_mean, float *running_var, float momentum, float eps, int clamp_var,
                          float *mean, float *invstd, float *scale, float *shift, void *stream);
/* BatchNorm over N ROWS of an [N][K] f32 matrix, each row standing for `rep` identical elements -- the ASPP image-pool
 * branch, where the reference broadcasts a [N,C,1,1] tensor to H x W and THEN applies BN (aspp.py:62-65,79-81).  Post-ReLU
 * inputs and N = batch size make sum / sum-of-squares partials cancel catastrophically, so both directions run two-pass
 * in f64, one thread per channel.  fwd: batch mean / biased var over the rows -> mean, invstd, scale, shift (+ running
 * statistics with the unbiased correction for count = N * rep).  bwd: g = gradient rows already summed over the copies. */
int dass_bn_rows_fwd(const float *x, int N, int K, double rep, const float *gamma, const float *beta,
                     float *running_mean, float *running_var, float momentum, float eps,
                     float *mean, float *invstd, float *scale, float *shift, void *stream);
int dass_bn_rows_bwd(const float *g, const float *x, const float *mean, const float *invstd, const float *gamma,
                     int N, int K, int train, float *dx, float *dgamma, float *dbeta, void *stream);
/* ---- train-mode BN without finalize launches (world size 1, dass_get_deterministic() == 0).  The batch statistics of a
 * layer live in `sums`: [2][K] f64 accumulators (sum, sum of squares of the raw conv output), zeroed by the caller;
 * the producing kernel adds to them with hardware f64 atomics and the apply kernel reads them directly:
 *   dass_conv2d_igemm_sums / dass_conv2d_x3_sums   the conv of dass_conv2d_igemm_stats / dass_conv2d_x3 (plain output);
 *   dass_channel_sums                              the standalone statistics pass (depthwise / stem producers);
 *   dass_bn_apply_train                            dass_bn_finalize + dass_scale_shift_act in one launch: every block
 *                                                  derives scale / shift (f64) into LDS, block 0 stores mean / invstd /
 *                                                  scale / shift for the backward and updates the running statistics
 *                                                  (F.batch_norm(training=True) semantics, momentum < 0: no update);
 *   dass_bn_bwd_reduce_sums / dass_bn_bwd_apply_sums   the backward pair of dass_bn_bwd_reduce(_gate) / _apply(_gate)
 *                                                  with sums = (sum dz, sum dz * xhat); dbeta_out / dgamma_out get them
 *                                                  rounded to f32.  out == NULL selects the gate-from-x form (f32).
 *   gates (nullable, uint8 [M][K/4])               dass_bn_apply_train stores the activation gate of every output (bit e of
 *                                                  byte [m][k/4] = channel k+e passes gradient); the backward pair then
 *                                                  reads that byte instead of the 16 bytes of `out` (layers with a
 *                                                  residual cannot re-derive the gate from the conv output alone).
 *   gates_bytes                                    size of that buffer: all three entry points return DASS_ERR_ARG when it is
 *                                                  smaller than M * K / 4 (a short buffer would be written / read out of
 *                                                  bounds by every row: the GPU memory fault of round 2).
 *   sums                                           backward: [2][K] f64 followed by K floats (zeroed like the rest) in which
 *                                                  dass_bn_bwd_reduce_sums keeps every channel's max |dz| (atomic max).
 *   residual_bound (dass_bn_apply_train, nullable) device float >= max |residual|; required when out3 is written in the
 *                                                  two-part x3 format (dass_set_x3_parts(2)) and a residual is added.
 * Arithmetic is that of the partial-row entry points (f32 inside a tile / slab, f64 across); only the order of the f64
 * additions is unspecified.  K <= 2048 for dass_bn_apply_train (DASS_ERR_UNSUPPORTED beyond). */
int dass_conv2d_igemm_sums(const void *x, int64_t ldx, const void *w, void *y, int64_t ldy, int N, int H, int W, int C,
                           int OH, int OW, int K, int R, int S, int stride, int pad, int dil, int dtype,
                           double *stat_sums, void *stream);
int dass_conv2d_x3_sums(const void *x3, const void *w3, void *y, int64_t ldy, int N, int H, int W, int C, int OH, int OW,
                        int K, int R, int S, int stride, int pad, int dil, double *stat_sums, void *workspace,
                        int64_t workspace_bytes, void *stream);
/* dass_conv2d_x3_dgrad_bnstats: the input-gradient launch of a stride-1 conv (flipped / transposed weight operand and
 * pad' = dil (R - 1) - pad, exactly as a dass_conv2d_x3 dgrad call; residual = gradient of a forked identity branch, nullable)
 * whose output dx [N*OH*OW][K] IS the gradient d_out of the conv + BN (+ act) layer that produced this conv's input.  Its
 * epilogue also adds THAT layer's BN-backward sums -- what dass_bn_bwd_reduce_sums computes in a pass of its own over dx and
 * the layer's conv output bn_y [M][K] -- into bn_sums ([2][K] f64 + K floats, zeroed by the caller):  dz = dx * gate, gate from
 * bn_gates (M * K / 4 bytes, bounds-checked against gates_bytes) or act'(fma(bn_y, gate_scale, gate_shift)).
 * *fused = 1: sums complete (the layer's backward goes straight to dass_bn_bwd_apply_sums); 0: this launch's schedule cannot
 * carry them (stream-K cut, tile without the whole-tile kernel) -- dx is complete, the caller runs dass_bn_bwd_reduce_sums.
 * (The reduce half of F.batch_norm's backward, models/sync_batchnorm/batchnorm.py:62-71 via autograd, without its own read of
 * the two tensors.) */
int dass_conv2d_x3_dgrad_bnstats(const void *x3, const void *w3, void *y, int64_t ldy, const void *residual, int64_t ldr,
                                 int N, int H, int W, int C, int OH, int OW, int K, int R, int S, int pad, int dil,
                                 const float *bn_y, const float *bn_mean, const float *bn_invstd, const float *gate_scale,
                                 const float *gate_shift, const void *bn_gates, int64_t gates_bytes, int bn_act,
                                 double *bn_sums, int *fused, void *workspace, int64_t workspace_bytes, void *stream);
int dass_channel_sums(const void *x, int64_t ldx, int64_t M, int K, double *sums, int dtype, void *stream);
int dass_bn_apply_train(const void *x, int64_t ldx, void *out, int64_t ldo, const double *sums, double count,
                        const float *gamma, const float *beta, float *running_mean, float *running_var, float momentum,
                        float eps, float *mean, float *invstd, float *scale, float *shift, const void *residual,
                        int64_t ldr, const float *nc_scale, int64_t M, int K, int64_t rows_per_image, int act, int dtype,
                        void *out3, void *gates, int64_t gates_bytes, const float *residual_bound, void *stream);
int dass_bn_bwd_reduce_sums(const void *dout, int64_t lddo, const void *out, int64_t ldo, const void *x, int64_t ldx,
                            const float *mean, const float *invstd, const float *gate_scale, const float *gate_shift,
                            const float *nc_scale, int64_t M, int K, int64_t rows_per_image, int act, double *sums,
                            const void *gates, int64_t gates_bytes, int dtype, void *stream);
int dass_bn_bwd_apply_sums(const void *dout, int64_t lddo, const void *out, int64_t ldo, const void *x, int64_t ldx,
                           const float *mean, const float *invstd, const float *gamma, const double *sums,
                           float *dbeta_out, float *dgamma_out, const float *gate_scale, const float *gate_shift,
                           const float *nc_scale, void *dx, int64_t lddx, void *dres, int64_t lddr, int64_t M, int K,
                           int64_t rows_per_image, double count, int act, const void *gates, int64_t gates_bytes, int dtype,
                           void *dx3, void *stream);
/* eval-mode BN folded to scale/shift from running stats */
int dass_bn_eval_scale_shift(const float *gamma, const float *beta, const float *running_mean,
                             const float *running_var, float eps, int K,
                             float *mean, float *invstd, float *scale, float *shift, void *stream);
/* out = act(x*scale[k] + shift[k] + residual) * nc_scale[n][k] ; n = m / rows_per_image */
/* out3 (nullable, f32 only): the same values additionally as x3 rows [M + 1][ceil(K/32)][192] (see dass_split3_rows;
 * K % 32 != 0: the caller zero-fills the buffer first); out may then be NULL when only the split form is consumed. */
int dass_scale_shift_act(const void *x, int64_t ldx, void *out, int64_t ldo,
                         const float *scale, const float *shift, const void *residual, int64_t ldr,
                         const float *nc_scale, int64_t M, int K, int64_t rows_per_image,
                         int act, int dtype, void *out3, void *stream);
/* backward of the above.  pass 1: partial[row][0][k]=sum dact, [1][k]=sum dact*xhat with
 * dact = dout*nc_scale*act'(out) and xhat=(x-mean)*invstd. */
int dass_bn_bwd_reduce(const void *dout, int64_t lddo, const void *out, int64_t ldo,
                       const void *x, int64_t ldx, const float *mean, const float *invstd,
                       const float *nc_scale, int64_t M, int K, int64_t rows_per_image,
                       int act, float *partial, int dtype, void *stream);
/* sums[0][k] = dbeta, sums[1][k] = dgamma (f64 reduction of the partial rows) */
int dass_bn_bwd_finalize(const float *partial, int rows, int K, float *dbeta, float *dgamma, void *stream);
/* pass 2: dx = gamma*invstd*(dact - train*(dbeta + xhat*dgamma)/count); dres = dact (may be NULL) */
int dass_bn_bwd_apply(const void *dout, int64_t lddo, const void *out, int64_t ldo,
                      const void *x, int64_t ldx, const float *mean, const float *invstd,
                      const float *gamma, const float *dbeta, const float *dgamma,
                      const float *nc_scale, void *dx, int64_t lddx, void *dres, int64_t lddr,
                      int64_t M, int K, int64_t rows_per_image, double count, int train,
                      int act, int dtype, void *dx3, void *stream);
/* (dx3, nullable, f32 only: dx additionally as x3 rows, the operand of the input-gradient conv; K % 32 == 0 or a
 * zero-filled buffer)
 * the two passes for a layer WITHOUT residual, f32: the activation gate is re-derived from the conv output as
 * act'(fma(x, gate_scale, gate_shift)) -- the forward's own affine (dass_scale_shift_act applies exactly this fma), so the
 * gate is bit-identical to the one of the stored output and `out` is not read (one HBM pass less in each). */
int dass_bn_bwd_reduce_gate(const void *dout, int64_t lddo, const void *x, int64_t ldx,
                            const float *mean, const float *invstd, const float *gate_scale, const float *gate_shift,
                            const float *nc_scale, int64_t M, int K, int64_t rows_per_image,
                            int act, float *partial, int dtype, void *stream);
int dass_bn_bwd_apply_gate(const void *dout, int64_t lddo, const void *x, int64_t ldx,
                           const float *mean, const float *invstd, const float *gamma, const float *dbeta, const float *dgamma,
                           const float *gate_scale, const float *gate_shift, const float *nc_scale,
                           void *dx, int64_t lddx, int64_t M, int K, int64_t rows_per_image, double count, int train,
                           int act, int dtype, void *dx3, void *stream);
/* column sums only: out[k] = sum_m x[m,k] (bias gradient of decoder.last_conv.7) */
int dass_colsum(const void *x, int64_t ldx, int64_t M, int K, float *partial, float *out, int dtype, void *stream);

/* ---------------------------------------------------------------- layout / pooling / interpolation */
/* NCHW f32 image -> NHWC (dtype), channels zero-padded to Cpad (stem input, resnet.py:83 / mobilenet.py:140) */
int dass_nchw_to_nhwc(const float *x, void *y, int N, int C, int H, int W, int Cpad, int dtype, void *stream);
int dass_nhwc_to_nchw(const void *x, int64_t ldx, float *y, int N, int C, int H, int W, int dtype, void *stream);
/* dst[m, 0:C] = src[m, 0:C] with independent pixel strides (concat / split, torch.cat at aspp.py:83, decoder.py:46) */
int dass_copy_channels(const void *src, int64_t lds, void *dst, int64_t ldd, int64_t M, int C, int dtype, void *stream);
/* dst[m,c] += src[m,c] */
int dass_add_channels(const void *src, int64_t lds, void *dst, int64_t ldd, int64_t M, int C, int dtype, void *stream);
/* dst = the sum of n (1..8) row tensors, one pass: the gradients of an activation with several consumers (autograd's own
 * accumulation would be n - 1 two-operand at::add launches); srcs / lds: host arrays of n device pointers / pixel strides */
int dass_sum_channels(const void *const *srcs, const int64_t *lds, int n, void *dst, int64_t ldd, int64_t M, int C, int dtype,
                      void *stream);
/* nn.MaxPool2d(3, 2, 1) (resnet.py:68): idx = winning tap 0..8 (uint8), first max wins */
int dass_maxpool3x3s2_fwd(const void *x, void *y, uint8_t *idx, int N, int H, int W, int C, int OH, int OW, int dtype, void *stream);
int dass_maxpool3x3s2_bwd(const void *dy, const uint8_t *idx, void *dx, int N, int H, int W, int C, int OH, int OW, int dtype, void *stream);
/* nn.AdaptiveAvgPool2d((1,1)) (aspp.py:62): y[n][c] f32-accumulated mean */
int dass_global_avgpool_fwd(const void *x, int64_t ldx, void *y, int N, int64_t HW, int C, int dtype, void *stream);
/* dx[n,hw,c] = dy[n,c]*mult  (broadcast; also the 1x1 -> HxW bilinear of aspp.py:80) */
int dass_broadcast_rows(const void *src, void *dst, int64_t ldd, int N, int64_t HW, int C, float mult, int dtype, void *stream);
/* dst[n][c] = sum_hw src[n,hw,c] (backward of the broadcast) */
int dass_reduce_rows(const void *src, int64_t lds, void *dst, int N, int64_t HW, int C, int dtype, void *stream);
/* F.interpolate(mode='bilinear', align_corners=True) (aspp.py:80, decoder.py:45, deeplab.py:59).
 * NHWC in; out NHWC (out_nchw=0, pixel stride ldy) or NCHW f32 (out_nchw=1). */
int dass_bilinear_fwd(const void *x, int64_t ldx, void *y, int64_t ldy, int N, int IH, int IW, int C,
                      int OH, int OW, int out_nchw, int dtype, void *stream);
/* gather-form backward (deterministic): dy NHWC (ld) or NCHW f32 */
int dass_bilinear_bwd(const void *dy, int64_t lddy, void *dx, int64_t lddx, int N, int IH, int IW, int C,
                      int OH, int OW, int dy_nchw, int dtype, void *stream);

/* ---------------------------------------------------------------- loss (utils/loss.py:39-51)
 * logits NCHW f32; target float32 (target_is_float=1, as the reference's dataloader yields) or int64.
 * ce_fwd writes per-block partial {sum w*nll, sum w}; ce_finalize reduces them (f64) into acc[2].
 * ce_bwd: dlogits = gscale * w[t]*(softmax - onehot) / acc[1]  (gscale read from device memory). */
int dass_ce_blocks(int64_t npix);
int dass_ce_fwd(const float *logits, const void *target, int target_is_float, const float *weight,
                int N, int C, int64_t HW, int ignore_index, float *partial, void *stream);
int dass_ce_finalize(const float *partial, int blocks, float *acc, void *stream);
int dass_ce_bwd(const float *logits, const void *target, int target_is_float, const float *weight,
                int N, int C, int64_t HW, int ignore_index, const float *acc, const float *gscale,
                float *dlogits, void *stream);

/* ---------------------------------------------------------------- acquisition scoring
 * (active_selection/mc_dropout.py:30-49,148-155,82-108; ceal.py:34-39,82-95,111-123,158-164;
 *  core_set.py:17-38,61-63) */
/* bilinear(align_corners) upsample of low-res NHWC logits fused with argmax over classes;
 * votes[n*vote_nstride + oy*OW + ox] (uint8) -- never materialises the full-size logits. */
int dass_upsample_argmax(const void *x, int64_t ldx, uint8_t *votes, int64_t vote_nstride,
                         int N, int IH, int IW, int C, int OH, int OW, int dtype, void *stream);
/* argmax over dim 1 of NCHW f32 logits (generic models; first max wins as torch.argmax) */
int dass_argmax_nchw(const float *logits, uint8_t *votes, int64_t vote_nstride, int N, int C, int64_t HW, void *stream);
/* per-image sums below go through fixed-order partials: partial must hold N*dass_score_blocks() floats */
int dass_score_blocks(void);
/* vote entropy: votes [N][T][HW] u8; label f32 [N][HW] or NULL; p_c = count_c/T;
 * e = -sum_c p_c*log2(p_c+1e-12); 0 where label<0 or >=num_classes.
 * entropy_map (nullable) [N][HW]; image_sum[n] = sum of the map (f64 finalize). */
int dass_vote_entropy(const uint8_t *votes, const float *label, int N, int T, int64_t HW, int num_classes,
                      float *entropy_map, float *partial, float *image_sum, void *stream);
/* softmax scores from NCHW f32 logits. mode 0: max prob (masked->1), 1: top1-top2 margin (masked->1),
 * 2: entropy log2 (masked->0).  map nullable; image_sum[n] = sum over pixels. */
int dass_softmax_scores(const float *logits, const float *label, int N, int C, int64_t HW, int num_classes,
                        int mode, float *map, float *partial, float *image_sum, void *stream);
/* weak labels: argmax u8 with 255 where label is outside [0,num_classes) (ceal.py:158-164) */
int dass_weak_labels(const float *logits, const float *label, int N, int C, int64_t HW, int num_classes,
                     uint8_t *out, void *stream);
/* F.avg_pool2d(feat, k, s) of NHWC features flattened channel-major: out[n][c*PH*PW + ph*PW + pw] f32 */
int dass_avgpool_features(const void *x, int64_t ldx, float *out, int N, int H, int W, int C,
                          int k, int s, int PH, int PW, int dtype, void *stream);
/* k-center (core_set.py:32-38): min_dist[i] = min(min_dist[i], ||f_i - f_center||_2), f64 distances;
 * the centre index is read from device memory so a whole greedy selection enqueues without a host sync.
 * first=1 overwrites min_dist. */
int dass_kcenter_update(const float *feat, int64_t n, int d, const int64_t *center, double *min_dist,
                        int first, void *stream);
/* first-max argmax (np.argmax, core_set.py:22): partial_val/partial_idx hold dass_argmax_blocks(n) entries */
int dass_argmax_blocks(int64_t n);
int dass_argmax_f64(const double *v, int64_t n, double *partial_val, int64_t *partial_idx,
                    int64_t *out_idx, double *out_val, void *stream);
/* valid r x r box sum of [N][H][W] maps -> [N][H-r+1][W-r+1] (conv2d with ones, mc_dropout.py:148);
 * tmp holds N*H*(W-r+1) floats */
int dass_box_sum(const float *maps, float *out, float *tmp, int N, int H, int W, int r, void *stream);
/* zero the rectangle [r0,r1)x[c0,c1) of map n (suppress_labeled_entropy, mc_dropout.py:110-121) */
int dass_zero_rect(float *maps, int n, int H, int W, int r0, int r1, int c0, int c1, void *stream);
/* global min/max then x = (x - min) * (1/(max-min)) in place (mc_dropout.py:152-155);
 * partial holds 2*dass_minmax_blocks(n) floats, out_min_max 2 floats (device) */
int dass_minmax_blocks(int64_t n);
int dass_minmax(const float *v, int64_t n, float *partial, float *out_min_max, void *stream);
int dass_affine_inplace(float *v, int64_t n, const float *min_max, void *stream);
/* greedy square NMS (mc_dropout.py:82-108) over [N][H][W] normalised score maps, entirely on device:
 * picks[i] = (image, row, col), count[0] = number of picks; imax/iarg are N-entry scratch. */
int dass_square_nms(float *maps, int N, int H, int W, int region, int max_picks, float *imax, int *iarg,
                    int *picks, int *count, void *stream);

/* greedy facility location of active_selection/max_subset.py:17-39 ("max representative samples"):
 * D[i][j] = ||a_i - b_j||_2 in f64; scores[j] = -sum_i min(mind[i], D[i][j]) (-inf for selected columns);
 * update: mind[i] = min(mind[i], D[i][*col]), selected[*col] = 1 (column index read from device memory). */
int dass_pairwise_dist_f64(const float *a, int64_t n, const float *b, int64_t m, int d, double *D, void *stream);
int dass_facility_scores(const double *D, int64_t n, int64_t m, const double *mind, const uint8_t *selected,
                         double *scores, void *stream);
int dass_facility_update(const double *D, int64_t n, int64_t m, const int64_t *col, double *mind,
                         uint8_t *selected, void *stream);

/* validation metrics (active_train.py:159-163 + utils/metrics.py:37-42): cm[gt*num_class + pred] += 1 over the
 * pixels with 0 <= gt < num_class; pred = argmax over classes of the NCHW f32 logits (first max), or the uint8
 * map `pred` when given (then logits may be NULL).  cm is int64 [num_class][num_class] on the device. */
int dass_confusion_accumulate(const float *logits, const uint8_t *pred, const float *target, int N, int C,
                              int64_t HW, int num_class, int64_t *cm, void *stream);

/* fused SGD(momentum, weight decay, nesterov=False) step over one flat f32 tensor
 * (torch.optim.SGD as built at active_train.py:60): g += wd*p; buf = mom*buf + g; p -= lr*buf */
int dass_sgd_step(float *p, const float *g, float *buf, int64_t n, float lr, float momentum,
                  float weight_decay, int first_step, void *stream);
/* the same update for n tensors at once (the whole optimizer step: active_train.py:60-66 builds one SGD over two lr
 * groups).  p, g, buf, numel, lr are HOST arrays of n device pointers / sizes / learning rates; tensors travel 64 per
 * launch inside the kernel argument.  A momentum buffer that starts at zero reproduces torch's first-step copy. */
int dass_sgd_step_multi(void *const *p, const void *const *g, void *const *buf, const int64_t *numel, const float *lr,
                        int n, float momentum, float weight_decay, void *stream);
/* The same update with {lr, momentum, weight_decay} read from DEVICE memory when the kernel runs (hyper: 3 floats, one triple for all n
 * tensors = one param group): a step captured into a hipGraph keeps following the poly learning-rate schedule the reference applies
 * every iteration (active_train.py:101 `self.scheduler(self.optimizer, i, epoch, ...)`), which by-value kernel arguments cannot.
 * Same arithmetic, bit for bit.  dass_set_floats writes up to 16 host floats to device memory through a kernel argument (no pinned
 * buffer whose reuse could race with an earlier, still queued copy). */
int dass_sgd_step_multi_dev(void *const *p, const void *const *g, void *const *buf, const int64_t *numel, int n,
                            const float *hyper, void *stream);
int dass_set_floats(float *dst, const float *vals, int n, void *stream);

/* ---- pre-split ("x3") operands: the pipelined form of the exact three-way bf16 split engine (csrc/conv_x3.hip).
 * An x3 tensor holds a [rows][C] f32 activation as rows x ceil(C/32) slabs of [3 parts][32 ch] bf16 (192 B; x = x0+x1+x2
 * exactly as DASS_F32X6 defines it, channels >= C zero) followed by ONE all-zero row (padded taps point there).
 * dass_x3_bytes: allocation size.  dass_split3_rows: f32 rows -> x3 rows; nc_scale (nullable) multiplies row m by
 * nc_scale[m / rows_per_image][c] first (Dropout2d mask of the producer, aspp.py:89 / decoder.py:35). */
int64_t dass_x3_bytes(int64_t rows, int C);
/* Operand format of the pre-split kernels and of every x3 buffer written from now on (process-wide, like
 * dass_set_deterministic): 3 (default) = three bf16 parts, six products per pair ("bf16x6"); 2 = "f16x3": x * s as TWO f16
 * parts (11 + 1 + 11 significant bits; s = a power of two per tensor with bound * s in [2^14, 2^15), bound >= max |x|), three
 * products per pair on the f16 MFMA pipe, the accumulator multiplied by 1 / (s_a s_b) -- half the matrix work for the same
 * f32-level product accuracy (2^-22 per product; measured against f64 in tests/test_f16x3_gpu.py).  Every x3 / weight operand
 * then ends in a 16-B trailer {float inv_scale, float bound, 0, 0} behind its zero row.  Where the bound comes from:
 *   dass_split3_rows / _packed     max |x| by dass_absmax_rows (one more read of the tensor),
 *   dass_bn_apply_train (out3)     batch statistics: |gamma xhat + beta| <= max|gamma| sqrt(M - 1) + max|beta| (+ residual_bound),
 *   dass_bn_bwd_apply_sums (dx3)   max over channels of |gamma invstd| (max|dz| + |mean dz| + sqrt(M - 1) |mean(dz xhat)|),
 *   weights                        max |w| (dass_weight_split_batch_f16).
 *   y3 of dass_conv2d_x3*          dass_x3_prepare_out BEFORE the launch: max_k(|scale_k| L1_k) * max|x| + max_k |shift_k| + max|res|
 *                                  with L1_k = sum |w_k| (dass_weight_l1) and the TRUE max |x| (trailer[2]; the conv epilogues
 *                                  maintain it for their own outputs, so the looseness does not compound along a fused chain),
 * Not available in the two-part format (DASS_ERR_UNSUPPORTED): out3 of dass_scale_shift_act, dx3 of dass_bn_bwd_apply / _gate
 * -- their consumers convert with dass_split3_rows instead.
 * y_amax (dass_conv2d_x3 / _per_image, nullable): device uint32, zeroed by the caller, that receives the bit pattern of
 * max |output| (atomic max) when only f32 rows are written; a two-part y3 keeps it in its own trailer. */
int dass_weight_l1(const float *w, int K, int64_t row_len, float *l1, void *stream);
int dass_x3_prepare_out(void *y3, int64_t out_rows, int K, const float *l1, const float *scale, const float *shift, const void *x3_in,
                        int64_t in_rows, int in_C, const float *res_amax, float mask_max, int act, void *stream);
/* parts: 3 (three bf16 parts, "bf16x6"), 2 (two scaled f16 parts, "f16x3") or 1 -- ONE bf16 part per element, one product per pair:
 * the "bf16x1" PERF engine (round 4).  Same kernels, same f32 tensors everywhere else; 64-B row-slabs, no per-tensor scales.  What it
 * multiplies is what autocast-bf16 multiplies: not parity-grade (tests/test_bf16_gpu.py measures its deviation). */
int dass_set_x3_parts(int parts);
int dass_get_x3_parts(void);
/* bound[0] = max over rows of |x[m][c] * nc_scale[m / rows_per_image][c]| (nc_scale nullable), as an atomic max: zero it first */
int dass_absmax_rows(const float *x, int64_t ld, int64_t M, int C, const float *nc_scale, int64_t rows_per_image, float *bound,
                     void *stream);
/* bytes of dass_w3_pack_per_image's output: N copies of the [rows][ceil(C/32)] operand + the trailer */
int64_t dass_w3_pack_bytes(int64_t rows, int C, int N);
/* tuning / test knob: tile + 10 * mode + 100 * shape (shape 0 = default MFMA shape = 16x16x32, 1 = 32x32x16, 2 = 16x16x32; DASS_X3_MFMA=32 makes 32x32x16 the default).  tile 0 = the dispatcher's choice, 1..7 = force one tile variant of
 * dass_conv2d_x3 (csrc/conv_x3.hip); mode 0 = auto, 1 = one output tile per workgroup, 2 = stream-K (whole rounds of
 * tiles one per workgroup + the remainder as equal slab ranges), 3 = stream-K slab ranges over all tiles */
int dass_x3_force_tile(int tile);
/* Diagnostic: which kernel the last dass_conv2d_x3* launch of this process ran -- (BM << 16) | (BN << 4) | 2 (whole-tile
 * specialisation) | 1 (stream-K schedule + fix-up pass).  Host state only (no device work); bench.py's in-step roofline uses it
 * to attribute launch times to tile classes.  No reference counterpart (the reference's convs are ATen calls). */
int dass_x3_last_pick(void);
/* workgroups of a whole-tile kernel resident per CU (hipOccupancyMaxActiveBlocksPerMultiprocessor): which = 0: 64x64, 1: 128x64,
 * 2: 256x128 tiles of the two-part engine -- the bytes a CU keeps in flight are this times the kernel's LDS ring */
int dass_x3_resident_workgroups(int which);
/* Measurement infrastructure (csrc/prof.hip; bench.py's in-step `roofline`, SURVEY.md 8d "measured live inside bench.py with HIP
 * events ... on the stream the kernel is launched on").  Between dass_prof_begin() and dass_prof_end() EVERY kernel this library
 * launches carries a start / stop event pair bound to that one dispatch (hipExtLaunchKernelGGL), on the stream of the launch:
 * dass_prof_get(i) returns launch i's kernel name, its own begin -> end duration in ms (waiting for it to finish), its grid size in
 * workgroups and its stream -- what a rocprofv3 kernel trace would list.  dass_prof_count(): launches recorded so far (callers
 * bracket an entry-point call with it to learn which kernels that call enqueued).  Outside a profile the launch path is the
 * plain one.  No reference counterpart. */
int dass_prof_begin(void);
int dass_prof_end(void);
int dass_prof_count(void);
int dass_prof_get(int i, char *name, int name_bytes, float *ms, int64_t *grid, void **stream);
/* Host-only helper: the (multiplier, shift) pair with which dass_conv2d_x3's kernels divide a pixel index n (0 <= n < 2^31)
 * by d = OH*OW or OW:  n / d == (n * mul >> 32) >> shift,  shift < 0 meaning d == 1 (quotient n).  Exported for the CPU tests. */
int dass_x3_magic(int d, unsigned *mul, int *shift);
/* bytes of scratch dass_conv2d_x3 wants for its stream-K schedule (partial tiles of split output tiles) */
int64_t dass_conv2d_x3_workspace_bytes(void);
int dass_split3_rows(const float *x, int64_t ld, void *out, int64_t M, int C, const float *nc_scale,
                     int64_t rows_per_image, void *stream);
/* nn.Conv2d forward (F.conv2d call sites of resnet.py:38-41, aspp.py:26, decoder.py:41-47) and, over dy with the
 * mode-1 weight operand, its input gradient -- same arguments as dass_conv2d_igemm with both operands pre-split:
 * x3 from dass_split3_rows (or a producer's y3), w3 from dass_weight_transform(DASS_F32X6) / dass_weight_split_batch.
 * Output: y (f32 rows, ldy; nullable) and/or y3 (x3 rows of the result, K % 4 == 0).  stat_partial/stat_rows
 * (nullable): per-M-tile BatchNorm partial sums as dass_conv2d_igemm_stats.  ustride > 1: phase-decomposed dgrad of a
 * strided conv (plain f32 output only).  Buffers beyond 4 GiB are DASS_ERR_UNSUPPORTED (32-bit buffer offsets).
 * workspace (nullable, 16-B aligned, dass_conv2d_x3_workspace_bytes()): with it the launch may balance the reduction slabs
 * over the CUs ("stream-K": output tiles split between workgroups are summed by a second, deterministic launch);
 * without it every output tile is one workgroup. */
int dass_conv2d_x3(const void *x3, const void *w3, void *y, int64_t ldy, void *y3, const float *scale,
                   const float *shift, const void *residual, int64_t ldr, int N, int H, int W, int C, int OH, int OW,
                   int K, int R, int S, int stride, int pad, int dil, int ustride, int act, float *stat_partial,
                   int *stat_rows, void *workspace, int64_t workspace_bytes, void *y_amax, void *stream);
/* Dropout2d-sparse form of the same conv for the MC-dropout tail (active_selection/mc_dropout.py:38-51: T stochastic
 * passes through models/decoder.py:23-36 with the ASPP Dropout2d(0.5) of models/aspp.py:89 active).  A dropped input
 * channel contributes exact zeros, so instead of multiplying zeros the surviving channels of every image are packed to the
 * front and the reduction stops after them:
 *   dass_dropout_compact     mask [N][C] (0 = dropped, else the multiplier 1/(1-p)) -> order [N][ceil(C/32)*32] (surviving
 *                            channel indices ascending, then -1) and cc_limit[N] (32-channel slabs that hold a survivor);
 *   dass_split3_rows_packed  f32 rows -> x3 rows whose channel j of image n is x[.., order[n][j]] * mask[n][order[n][j]];
 *   dass_w3_pack_per_image   pre-split weights [rows = K*R*S][ceil(C/32)][192 B] -> N copies [N][rows][..] in image n's order;
 *   dass_conv2d_x3_per_image image n multiplies with ITS weight copy over its first cc_limit[n] slabs; output tiles never
 *                            straddle two images.  Everything stays on the device (no host read of the mask).
 * The packed result differs from the masked dense one only in the order the surviving products are accumulated. */
int dass_dropout_compact(const float *mask, int N, int C, int *order, int *cc_limit, void *stream);
int dass_split3_rows_packed(const float *x, int64_t ld, void *out, int64_t M, int C, const float *mask, const int *order,
                            const int *cc_limit, int64_t rows_per_image, void *stream);
/* All T stochastic passes of a scoring batch as ONE launch each (active_selection/mc_dropout.py:37-49 runs T full forwards one after
 * the other; models/decoder.py:23-36 is the tail each of them re-runs).  dass_split3_rows_packed_rep: like
 * dass_split3_rows_packed_bound for M = (T x images) x rows_per_image OUTPUT rows over an x that holds only src_images images --
 * output image v packs source image v % src_images with its own mask / order / limit row.  dass_conv2d_x3_per_image_rep: like
 * dass_conv2d_x3_per_image with a residual of only res_images images (image g adds the rows of image g % res_images: the batch's
 * deterministic low-level share, the same in every pass).  N % res_images must be 0. */
int dass_split3_rows_packed_rep(const float *x, int64_t ld, void *out, int64_t M, int C, const float *mask, const int *order,
                                const int *cc_limit, int64_t rows_per_image, int src_images, const float *bound, float bound_mul,
                                void *stream);
int dass_conv2d_x3_per_image_rep(const void *x3, const void *w3, const int *cc_limit, void *y, int64_t ldy, void *y3,
                                 const float *scale, const float *shift, const void *residual, int64_t ldr, int res_images, int N,
                                 int H, int W, int C, int OH, int OW, int K, int R, int S, int stride, int pad, int dil, int act,
                                 void *workspace, int64_t workspace_bytes, void *y_amax, void *stream);
/* the same with the bound of the two-part format supplied by the caller: *bound * bound_mul >= max |x * mask| (no pass over x) */
int dass_split3_rows_packed_bound(const float *x, int64_t ld, void *out, int64_t M, int C, const float *mask, const int *order,
                                  const int *cc_limit, int64_t rows_per_image, const float *bound, float bound_mul, void *stream);
int dass_w3_pack_per_image(const void *w3, void *out, int64_t rows, int C, int N, const int *order, const int *cc_limit,
                           void *stream);
int dass_conv2d_x3_per_image(const void *x3, const void *w3, const int *cc_limit, void *y, int64_t ldy, void *y3,
                             const float *scale, const float *shift, const void *residual, int64_t ldr, int N, int H,
                             int W, int C, int OH, int OW, int K, int R, int S, int stride, int pad, int dil, int act,
                             void *workspace, int64_t workspace_bytes, void *y_amax, void *stream);
/* Weight gradients are accumulated over several pixel splits per tile with f32 atomics (fast, but the last bits depend on
 * arrival order).  dass_set_deterministic(1) makes every weight-gradient launch use ONE split per tile: each dW element is
 * summed by one workgroup in a fixed order -> bit-reproducible (slower on layers with few output tiles). */
int dass_set_deterministic(int on);
int dass_get_deterministic(void);
/* weight gradient of the same conv from pre-split operands: dw[K][R][S][C] (f32) (+)= sum over output pixels of
 * dy[pix][k] * x[pix @ tap][c]; x3 = the forward input, dy3 = the gradient of the conv output (both x3 rows with their zero
 * row; csrc/wgrad_x3.hip).  zero_first = 1 clears dw, 0 accumulates (pixel splits add with f32 atomics either way). */
int dass_conv2d_wgrad_x3(const void *x3, const void *dy3, float *dw, int N, int H, int W, int C, int OH, int OW, int K,
                         int R, int S, int stride, int pad, int dil, int zero_first, void *stream);
/* The weight gradients of MANY conv layers in one launch per tile class (csrc/wgrad_x3.hip "GROUPED form"): items = n x 16 int64
 * on the HOST {x3, dy3, dw, N, H, W, C, OH, OW, K, R, S, stride, pad, dil, 0}, arguments as dass_conv2d_wgrad_x3; every dw must
 * be zeroed by the caller (tiles are added).  Replaces the per-layer launches of a backward pass whose weight gradients have no
 * consumer before the optimizer step: with hundreds of output tiles in one grid no pixel reduction has to be cut into
 * atomically-added pieces just to fill the chip, and the fill / drain of ~100 launches is paid once.  scratch: device buffer of
 * dass_conv2d_wgrad_x3_group_scratch_bytes(n) bytes that receives the problem table (asynchronous copy on `stream`). */
int dass_conv2d_wgrad_x3_group(const int64_t *items, int n, void *scratch, int64_t scratch_bytes, void *stream);
int64_t dass_conv2d_wgrad_x3_group_scratch_bytes(int n);
/* Capturing a step that holds grouped weight-gradient launches into a hipGraph (dass_hip/graph.py; the reference loop it wraps:
 * active_train.py:103-107).  The copy of a launch's problem table becomes a graph node that re-reads its pinned source at every replay,
 * so those tables belong to the graph: dass_graph_capture_open(slots) -- BEFORE the capture starts, nothing may be allocated while a
 * stream captures -- makes sure `slots` free tables exist and returns the token that the launches captured until
 * dass_graph_capture_close() tag their tables with (-1: allocation failed); dass_graph_release(token) hands them back when the graph is
 * destroyed or its capture failed (-> number released).  A captured launch without an open token, or with no free table left, fails
 * with DASS_ERR_UNSUPPORTED and says why on stderr.  dass_graph_slots: tables owned by live graphs / free. */
int64_t dass_graph_capture_open(int slots);
int dass_graph_capture_close(void);
int dass_graph_release(int64_t token);
int dass_graph_slots(int *owned, int *free_slots);
/* ---------------------------------------------------------------- pool reader
 * dataloaders/dataset/paths_dataset.py:27-52: a record is uint8 [H][W][4] (RGB + label).  dass_resample_bilinear_u8 =
 * scipy.misc.imresize(image, (OH, OW)) of custom_transforms.py:153,228,291 = PIL's two-pass bilinear resampler, bit for bit:
 * src [H][W][src_ch] (first three channels) -> tmp [H][OW][3] -> dst [OH][OW][3]; x / y tables (first source index, tap
 * count, 22-bit fixed-point taps [out][ksize]) are built on the host (dataloaders/custom_transforms.py:resample_tables).
 * dass_pool_finalize = centre crop / centred paste + Normalize + ToTensor (+ the nearest-resized label plane read through
 * yidx / xidx source-index tables): output window [S][S] with the resized image at offset (oy0, ox0) (crop: -y1, -x1;
 * 512 canvas: +y0, +x0), image 0 / label 255 outside; divide255 / f64_chain select the arithmetic of the label chain
 * (custom Normalize via numpy float64) or the image-only chain (torchvision, float32).  out_lab nullable. */
int dass_resample_bilinear_u8(const void *src, int H, int W, int src_ch, void *tmp, void *dst, int OH, int OW,
                              const int *xmin, const int *xcnt, const int *xkk, int xksize,
                              const int *ymin, const int *ycnt, const int *ykk, int yksize, void *stream);
int dass_pool_finalize(const void *img, int OH, int OW, const void *rec, int W, int rec_ch, const int *yidx,
                       const int *xidx, int oy0, int ox0, int S, int divide255, int f64_chain, float *out_img,
                       float *out_lab, void *stream);

/* DIAGNOSTIC, not on the product path (tools/clock_probe.py): `blocks` workgroups run a bf16 MFMA loop (use_lds: with
 * LDS fragment reads) and report {shader cycles, 100 MHz ticks} per block into out[2 * blocks] (uint64): the clock the chip
 * sustains under that load = cycles / ticks * 100 MHz. */
int dass_clock_probe(void *out, int blocks, int iters, int use_lds, void *stream);

#ifdef __cplusplus
}
#endif
#endif
