/* libafd_hip -- C ABI of the MI355X (gfx950) hot path of audiofakedetect.
 *
 * The reference (gan-police/audiodeepfake-detection) is pure Python; the device work of
 * its hot path is launched implicitly by ptwt / torchaudio / torch.nn.  This header is the
 * boundary a maintainer binds (ctypes) to replace those launches.
 * Each entry point cites the reference call site it replaces.
 *
 * Conventions
 *   - every function returns 0 on success, a negative AFD_ERR_* code otherwise; the
 *     message is available from afd_last_error() (thread local);
 *   - pointers marked [dev] are device pointers owned by the caller (PyTorch); the library
 *     never allocates, frees or retains them; pointers marked [host] are host memory read
 *     during the call only;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it and the call
 *     returns without synchronising;
 *   - tensors are dense fp32 unless stated; layouts are given per function.
 */
#ifndef AFD_HIP_H
#define AFD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AFD_ERR_ARG (-1)
#define AFD_ERR_HIP (-2)
#define AFD_ERR_UNSUPPORTED (-3)
#define AFD_ERR_WORKSPACE (-4)

typedef void* afd_stream_t;

const char* afd_last_error(void);
int afd_version(void);

/* Optional measurement hook (bench.py): when enabled, every launch of the kernel classes
 * below is bracketed by HIP events on its own stream; afd_timing_collect sums, over the launches
 * recorded since the last afd_timing_reset: the durations, the launch count, the algorithmic work
 * (bytes for AFD_K_WPT / AFD_K_STFT, direct-form flops for the conv classes), the flops really
 * issued on the matrix cores (tile padding included; Winograd: its 16 GEMMs; 0 for kernels without
 * MFMAs) and the algorithmic bytes (every input and output tensor of the launch once). */
#define AFD_K_WPT 0
#define AFD_K_CONV_IGEMM 1 /* forward and backward-data launches of the implicit-GEMM kernel */
#define AFD_K_CONV_WGRAD 2
#define AFD_K_STFT 3
#define AFD_K_CONV_DIRECT 4 /* small-channel direct convolutions (dilated stack at time_dim <= 4) */
#define AFD_K_CONV_WGRAD_1X1 6 /* backward-weight of the 1x1 layers (a plain GEMM over pixels, HBM-leaning) */
#define AFD_K_CONV_WINOGRAD 5 /* 3x3 forward / backward-data launches on the Winograd F(2x2,3x3) kernel:
                                  work = direct-form flops, of which the kernel issues 16/36 as MFMAs */
#define AFD_K_LCNN_BF16 7 /* bf16 matrix-core launches of the LCNN evaluation forward (convolutions, LSTM / Linear
                             projections): work = the layer's flops, issued = with row / k padding; peak = bf16 MFMA */
#define AFD_K_CONV_FIRST 8  /* block 1 (Cin = 1): convolution + PReLU + 2x2 max-pool, forward and backward, on the
                               vector ALU; work = bytes = every tensor of the launch once (HBM-bound class) */
#define AFD_K_BATCHNORM 9   /* BatchNorm statistics / apply / backward passes and the small sum / fold kernels */
#define AFD_K_ELEMENTWISE 10 /* pool, dropout, permute, Linear + mean, cross entropy, gather, Adam, normalise */
#define AFD_K_CONV1X1 11    /* 1x1 convolutions forward / backward-data: GEMMs at 16 flop per byte, HBM-leaning;
                               work = direct-form flops, issued = the same with tile padding, bytes = tensors once */
#define AFD_K_COUNT 12
int afd_timing_enable(int on);
int afd_timing_collect(int id, double* total_ms, long long* count, double* total_work,
                       double* total_issued_flops, double* total_algorithmic_bytes);
int afd_timing_reset(void);

/* ------------------------------------------------------------------------------------
 * Wavelet-packet front end.
 * Replaces: compute_pytorch_packet_representation + Packets.forward
 *           (reference src/audiofakedetect/wavelet_math.py:167-263), i.e. the
 *           ptwt.WaveletPacket tree (:182), get_level (:185), stack (:206),
 *           log(|x|^power + eps) (:208-209), sign channel (:211-213), the final permute
 *           (:263), and torchvision Normalize with scalar statistics (:380-382).
 * ---------------------------------------------------------------------------------- */
#define AFD_WPT_LOG 1u  /* log(|x|^power + eps)                                       */
#define AFD_WPT_SIGN 2u /* second channel = +1/-1 sign pattern (loss_less), needs LOG */
#define AFD_WPT_NORM 4u /* per-channel Normalize: (v - mean) / std on the coefficient
                         * channel, (s - sign_mean) / sign_std on the sign channel    */

/* Node length at `level` for frames of N samples and L-tap filters (reflect mode). */
int afd_wpt_out_len(int N, int L, int level);

/* Bytes of device workspace afd_wpt_forward needs (may be 0). */
size_t afd_wpt_workspace_bytes(int B, int N, int L, int level);

/* x   [dev]  [B][N] frames
 * dec_lo, dec_hi [host] L decomposition taps (pywt convention)
 * out [dev]  [B][C][T][P], P = 2^level packets in frequency (Gray-code) order fastest,
 *            T = afd_wpt_out_len, C = 2 with AFD_WPT_SIGN else 1.  This is the memory
 *            order of the reference's returned view (logical [B][C][P][T]).
 * mean, std / sign_mean, sign_std: the per-channel statistics torchvision Normalize applies
 *            (calc_normalization's Welford runs per channel, wavelet_math.py:441); the sign
 *            pair is read only with AFD_WPT_SIGN | AFD_WPT_NORM.
 */
int afd_wpt_forward(const float* x, int B, int N, const float* dec_lo, const float* dec_hi,
                    int L, int level, unsigned flags, float power, float eps, float mean,
                    float std, float sign_mean, float sign_std, float* out, void* ws, size_t ws_bytes,
                    afd_stream_t stream);
/* One two-channel analysis step of frames of ANY length straight from global memory: ca / cd [B][n1],
 * n1 = afd_wpt_out_len(N, L, 1) (reflect mode, the step of afd_wpt_forward).  afd_wpt_forward keeps a frame's packet
 * tree in LDS and returns AFD_ERR_UNSUPPORTED for frames beyond about 27 000 samples; the host then splits the top
 * level(s) off with this call and transforms the children (the packets of a detail child come out in reversed
 * frequency order: wavelet_math.wpt_forward). */
int afd_wpt_analysis_step(const float* x, int B, int N, const float* dec_lo, const float* dec_hi, int L,
                          float* ca, float* cd, afd_stream_t stream);

/* The orthogonal lattice the level-9..14 kernel runs a filter bank in (host only, no GPU; exported for the tests):
 * dec_lo, dec_hi [host] L taps -> alpha, beta [L/2] stage coefficients, scales[2] = output scales of (cA, cD),
 * fit_residual = largest |lattice tap - table tap|.  One analysis step on the polyphase pairs (e_j, o_j) =
 * (xe[2j], xe[2j+1]) of the reflect-extended node: (A, B)[j] = (e_j + alpha[0] o_j, e_j + beta[0] o_j), then for
 * s = 1 .. L/2-1: (A, B)[j] <- (A[j] + alpha[s] B[j-1], A[j] + beta[s] B[j-1]); cA[i] = scales[0] A[i],
 * cD[i] = scales[1] B[i].  AFD_ERR_UNSUPPORTED when the taps are not an orthogonal bank. */
int afd_wpt_lattice(const float* dec_lo, const float* dec_hi, int L, double* alpha, double* beta, double* scales,
                    double* fit_residual);

/* ------------------------------------------------------------------------------------
 * STFT power spectrogram front end.
 * Replaces: STFTLayer.forward = torchaudio Spectrogram(n_fft, hop_length, power) + log
 *           (reference src/audiofakedetect/wavelet_math.py:47,63-68) and Normalize (:380-382).
 *           Periodic Hann window of n_fft, centre, reflect pad n_fft/2, one-sided.
 * afd_stft_dims   : F = n_fft/2+1 bins, T = 1 + N/hop frames, shape of the basis matrix.
 * afd_stft_basis  : fills the [basis_rows][basis_cols] windowed DFT basis on the HOST; the
 *                   caller uploads it once and passes the device copy to afd_stft_forward.
 * afd_stft_forward: x [dev] [B][N] -> out [dev] [B][1][F][T] (T fastest).
 * ---------------------------------------------------------------------------------- */
#define AFD_STFT_LOG 1u  /* log(|X|^power + eps) */
#define AFD_STFT_NORM 4u /* (v - mean) / std     */
int afd_stft_dims(int N, int n_fft, int hop, int* F, int* T, int* basis_rows, int* basis_cols);
int afd_stft_basis(int n_fft, float* basis /* [host] */);
int afd_stft_forward(const float* x, int B, int N, int n_fft, int hop, const float* basis,
                     unsigned flags, float power, float eps, float mean, float std, float* out,
                     afd_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Conv2d, stride 1, square kernel K, zero padding `pad`, dilation `dil`, NCHW fp32.
 * Replaces: the cuDNN forward / backward-data / backward-weight launches behind every
 *           nn.Conv2d of DCNN / LCNN (reference src/audiofakedetect/models.py:255-291,
 *           :85-110).  Exact-f32 MFMA (v_mfma_f32_32x32x2_f32) implicit GEMM.
 * Geometry arguments always describe the FORWARD convolution:
 *   x [N][Cin][H][W], w [Cout][Cin][K][K], y [N][Cout][Hout][Wout],
 *   Hout = H + 2 pad - dil (K - 1).   Limits: Cout <= 128 (and Cin <= 128 for
 *   backward_data).  `ws` is device scratch of afd_conv2d_workspace_bytes() bytes.
 * ---------------------------------------------------------------------------------- */
size_t afd_conv2d_workspace_bytes(int N, int Cin, int H, int W, int Cout, int K, int pad, int dil);
int afd_conv2d_forward(const float* x, const float* w, const float* bias /* may be NULL */,
                       float* y, int N, int Cin, int H, int W, int Cout, int K, int pad, int dil,
                       void* ws, size_t ws_bytes, afd_stream_t stream);
int afd_conv2d_backward_data(const float* dy, const float* w, float* dx, int N, int Cin, int H,
                             int W, int Cout, int K, int pad, int dil, void* ws, size_t ws_bytes,
                             afd_stream_t stream);

/* Conv2d(k=3, padding=1) -> PReLU -> MaxPool2d(2, 2) in one launch (reference models.py:263-265 and
 * :276-278, DCNN blocks 3 and 6): on the Winograd F(2x2,3x3) kernel the 2x2 output tile of a lane is
 * the pooling window, so the pooled value u and the 3-bit code of afd_prelu_pool_forward (argmax |
 * "winner <= 0") are written instead of the 4x larger convolution output, which never exists.
 * u, idx [N][Cout][H/2][W/2].  afd_conv3x3_prelu_pool_applicable: 1 if the layer geometry is one the
 * kernel takes (otherwise run afd_conv2d_forward_cropped + afd_prelu_pool_forward).  Backward is
 * afd_prelu_pool_backward followed by the convolution's backward entry points. */
int afd_conv3x3_prelu_pool_applicable(int Cin, int H, int W, int Cout);
int afd_conv3x3_prelu_pool_forward(const float* x, const float* w, const float* bias /* may be NULL */,
                                   const float* slope, float* u, uint8_t* idx, int N, int Cin, int H, int W,
                                   int Cout, void* ws, size_t ws_bytes, afd_stream_t stream);

/* 1x1 convolution forward that also produces the batch statistics of the BatchNorm behind its activation:
 * y = w x + bias ([N][Cout][HW]) and sums[c] = sum PReLU(y[c]), sums[Cout + c] = sum PReLU(y[c])^2 over all
 * pixels (the layout afd_bn_finalize takes; overwritten, deterministic).  Saves the statistics pass of
 * Conv2d(k=1) -> PReLU -> BatchNorm (reference models.py:262-264). */
size_t afd_conv1x1_forward_stats_workspace_bytes(int Cout);
int afd_conv1x1_forward_stats(const float* x, const float* w, const float* bias /* may be NULL */,
                              const float* slope, float* y, double* sums, int N, int Cin, int Cout, long HW,
                              void* ws, size_t ws_bytes, afd_stream_t stream);
/* BatchNorm2d(affine=False) -> Conv2d(k=1, pad 0) pair (reference models.py:260-262, DCNN blocks 1-2).
 * Forward needs no kernel of its own: the normalisation is a per-input-channel scale and shift, so the
 * caller folds it into the weights (wf[co][ci] = w[co][ci] * invstd[ci], bf = b - wf . mean) and runs
 * afd_conv2d_forward on the un-normalised tensor u.  Backward through both layers in one GEMM:
 *   du[n][ci][p] = sum_co wf[co][ci] dz[n][co][p] + alpha[ci] * u[n][ci][p] + beta[ci]
 * with alpha = -invstd^2 * E[dxhat xhat], beta = invstd^2 * E[dxhat xhat] * mean - invstd * E[dxhat]
 * (the batch means follow from the 1x1 weight / bias gradients, no pass over the activations).
 * Replaces, for this pair, cudnn's batch-norm backward + the convolution's backward-data. */
int afd_conv1x1_bn_backward_data(const float* dz, const float* wf, const float* u, const float* alpha,
                                 const float* beta, float* du, int N, int Cin, int Cout, long HW,
                                 afd_stream_t stream);
/* Backward of  u -> [BatchNorm(affine=False) folded into wf] -> Conv2d(Cin, C, 1) -> z -> PReLU -> P ->
 * BatchNorm(affine=False) -> xhat  (reference src/audiofakedetect/models.py:260-264, DCNN block 2) in ONE pass
 * over the activations; replaces cudnn's batch-norm backward, the PReLU backward and the convolution's
 * backward-weight and backward-data for this chain.  g = dL/dxhat [N][C][HW], z [N][C][HW], u [N][Cin][HW],
 * wf [C][Cin] the folded weights, coef [C][4] = (A, B, K, 0) with dP = A g + B P + K
 * (A = invstd, B = -invstd^2 E[g xhat], K = invstd^2 E[g xhat] mean - invstd E[g] of the SECOND BatchNorm).
 * Outputs: t [N][Cin][HW] = wf^T dz (the first BatchNorm's affine backward term is added by the consumer,
 * afd_conv1_pool_backward_affine), G [C][Cin] = sum_p dz u (gradient against the un-normalised input),
 * db [C] = sum_p dz, dslope (+=).  Cin, C <= 64. */
int afd_conv1x1_prelu_bn_backward_applicable(int Cin, int C);
size_t afd_conv1x1_prelu_bn_backward_workspace_bytes(int Cin, int C);
int afd_conv1x1_prelu_bn_backward(const float* g, const float* z, const float* u, const float* wf,
                                  const float* coef, const float* slope, float* t, float* G, float* db,
                                  float* dslope /* += */, int N, int Cin, int C, long HW, void* ws,
                                  size_t ws_bytes, afd_stream_t stream);
/* Backward-data of Conv2d(Cin, Cout, 3, padding=1) whose input xhat [N][Cin][H][W] was the output of a
 * training-mode BatchNorm without affine parameters, with that BatchNorm's backward sums from the same launch:
 * dx = conv_transpose(dy, w) as afd_conv2d_backward_data, and sums[c] = sum_px dx[c], sums[Cin + c] =
 * sum_px dx[c] xhat[c] (double, the layout afd_bn_backward_means takes; deterministic).  Saves the statistics
 * pass of the batch-norm backward (reference models.py:264-268: SyncBatchNorm -> Conv2d k3).
 * `ws` as for afd_conv2d_backward_data.
 * Where afd_conv3x3_backward_data_bnstats_needs_input returns 0, xhat may be NULL: sums[Cin + c] then comes back 0 and
 * the caller overwrites it with afd_conv_weight_dot(w, dw) once the layer's weight gradient exists -- for any
 * convolution, sum_px dx[c] x[c] = sum_{co,k} w[co][c][k] dw[co][c][k] (out[c], double), so the launch does not read
 * the activations a second time. */
int afd_conv3x3_backward_data_bnstats_applicable(int Cin, int H, int W, int Cout);
int afd_conv3x3_backward_data_bnstats_needs_input(int Cin, int H, int W, int Cout);
int afd_conv_weight_dot(const float* w, const float* dw, int Cout, int Cin, int KK, double* out, afd_stream_t stream);
/* Backward of Conv2d(Cin, Cout, 3, padding=1) -> PReLU -> MaxPool2d(2, 2) (reference models.py:263-266, 279-281) WITHOUT
 * the dense gradient of the convolution output: afd_prelu_pool_backward_compact leaves gg [N*Cout][H/2][W/2], the value
 * the pool's backward routes to position (code & 3) of each window (BatchNorm-backward coefficients `coef` [C][4] and
 * the PReLU slope applied, dslope += as afd_prelu_pool_backward_affine); the two convolution launches expand gg and the
 * codes while they load.  gg and codes must be followed by >= 16 readable bytes (vector loads at the row ends are
 * masked, not shortened).  sums as afd_conv3x3_backward_data_bnstats with xhat = NULL.  _applicable: both launches on the
 * F(4x4) Winograd kernels with 64 / 32 input channels. */
int afd_conv3x3_pooled_backward_applicable(int Cin, int H, int W, int Cout);
int afd_prelu_pool_backward_compact(const float* u, const float* slope, const uint8_t* idx, const float* du,
                                    const float* coef, int C, float* gg, float* dslope, int NC, int Hp, int Wp,
                                    afd_stream_t stream);
int afd_conv3x3_backward_data_bnstats_pooled(const float* gg, const uint8_t* codes, const float* w, float* dx,
                                             double* sums, int N, int Cin, int H, int W, int Cout, void* ws,
                                             size_t ws_bytes, void* stat_ws, size_t stat_ws_bytes, afd_stream_t stream);
int afd_conv3x3_backward_weight_pooled(const float* x, const float* gg, const uint8_t* codes, float* dw, float* dbias,
                                       int N, int Cin, int H, int W, int Cout, void* ws, size_t ws_bytes,
                                       afd_stream_t stream);
size_t afd_conv3x3_backward_data_bnstats_workspace_bytes(int N, int Cin, int H, int W);
int afd_conv3x3_backward_data_bnstats(const float* dy, const float* w, float* dx, const float* xhat, double* sums,
                                      int N, int Cin, int H, int W, int Cout, void* ws, size_t ws_bytes,
                                      void* stat_ws, size_t stat_ws_bytes, afd_stream_t stream);

/* Forward 3x3 / pad 1 convolution whose result feeds a training-mode BatchNorm, with that BatchNorm's batch sums from
 * the kernel's epilogue: sums[c] = sum v, sums[Cout + c] = sum v^2 (the layout afd_bn_stats writes), v = the pooled
 * value when u / idx are given (Conv2d + PReLU + MaxPool2d(2, 2) as afd_conv3x3_prelu_pool_forward, y unused), else
 * PReLU(y) with `slope` (y itself for a NULL slope).  Only for the layers the F(4x4) Winograd kernel takes
 * (afd_conv3x3_forward_stats_applicable); reference: nn.Conv2d -> nn.PReLU [-> nn.MaxPool2d] -> nn.BatchNorm2d of
 * DCNN blocks 3 and 4 (src/audiofakedetect/models.py:263-270). */
int afd_conv3x3_forward_stats_applicable(int Cin, int H, int W, int Cout, int pooled);
size_t afd_conv3x3_forward_stats_workspace_bytes(int N, int H, int W, int Cout);
int afd_conv3x3_forward_stats(const float* x, const float* w, const float* bias, const float* slope, float* y,
                              float* u, uint8_t* idx, double* sums, int N, int Cin, int H, int W, int Cout,
                              void* ws, size_t ws_bytes, void* stat_ws, size_t stat_ws_bytes, afd_stream_t stream);

/* The training-mode BatchNorm(affine=False) in FRONT of a 3x3 / pad 1 convolution applied while the convolution loads
 * (nn.[PReLU ->] nn.SyncBatchNorm(affine=False) -> nn.Conv2d(k=3, padding=1) of DCNN blocks 2->3, 3->4, 4->5, 5->6,
 * src/audiofakedetect/models.py:261-276): x is the BatchNorm's INPUT, in_aff [Cin][2] = (mean, invstd) of the batch,
 * in_slope the PReLU slope in front of the BatchNorm or NULL; a patch becomes (PReLU(x) - mean) * invstd in registers,
 * with afd_bn_apply_forward's arithmetic, so the normalised tensor is never written.
 *   afd_conv3x3_forward_fold: afd_conv3x3_forward_stats (sums != NULL) or afd_conv3x3_prelu_pool_forward (sums == NULL,
 *     u / idx given) on that input;
 *   afd_conv3x3_backward_weight_fold: afd_conv2d_backward_weight_sums (codes == NULL: dy dense, read up to
 *     dy_rows x dy_cols) or afd_conv3x3_backward_weight_pooled (codes != NULL: dy the pooled gradient);
 *   the backward-data launch of the layer does not read its input and is unchanged.
 * Only for layers on the F(4x4) Winograd kernels (afd_conv3x3_input_fold_applicable); AFD_ERR_UNSUPPORTED otherwise. */
int afd_conv3x3_input_fold_applicable(int Cin, int H, int W, int Cout, int pooled, int want_stats);

/* The BACKWARD of that BatchNorm (and of the PReLU in front of it) inside the convolution's backward-data launch
 * (models.py:267-276, what autograd runs for nn.PReLU -> nn.SyncBatchNorm(affine=False) -> nn.Conv2d(k=3, padding=1)).
 * The BatchNorm backward needs the batch sums of g = dL/d(its output) and of g * xhat before it touches an element; both
 * follow from small tensors once the convolution's backward-weight launch has run:
 *   sum g * xhat = afd_conv_weight_dot(w, dw);  sum g = afd_conv3x3_input_grad_sums (weights x border-aware sums of dy).
 * afd_conv3x3_backward_data_bnapply then writes dz = PReLU'(z) * invstd * (g - mean(g) - xhat * mean(g * xhat)) instead of g:
 * z the BatchNorm's input, bn_tab [Cin][4] = (mean, invstd, mean(g), mean(g * xhat)), bn_slope the PReLU slope or NULL;
 * sums[0 .. Cin) = sum(dz) per channel (the bias gradient of the convolution that produced z), sums[Cin .. 2 Cin) =
 * per-channel partial gradients of the slope.  dy dense, or (codes != NULL) the pooled gradient with the pool's codes.
 * bn_codes != NULL: the BatchNorm sits right behind nn.PReLU -> nn.MaxPool2d(2, 2) (models.py:264-266): z is the pooled
 * tensor, bn_codes the pool's argmax codes, bn_slope that PReLU's slope, and dz is the pooled gradient of the convolution
 * in front of the pool (what afd_prelu_pool_backward_compact leaves).
 * afd_conv3x3_input_grad_sums: sums must hold Cin + 8 Cout doubles (the tail is scratch); dy_sums (double) or dbias
 * (float) = the per-channel sums of dy, whichever the caller has. */
int afd_conv3x3_backward_data_bnapply_applicable(int Cin, int H, int W, int Cout, int pooled);
int afd_conv3x3_input_grad_sums(const float* dy, const uint8_t* codes /* may be NULL */, const float* w,
                                const double* dy_sums /* may be NULL */, const float* dbias /* may be NULL */,
                                double* sums, int N, int Cin, int H, int W, int Cout, int dy_rows, int dy_cols,
                                afd_stream_t stream);
int afd_conv3x3_backward_data_bnapply(const float* dy, const uint8_t* codes /* may be NULL */, const float* w,
                                      const float* z, const float* bn_tab, const float* bn_slope /* may be NULL */,
                                      const uint8_t* bn_codes /* may be NULL */, float* dz, double* sums, int N, int Cin,
                                      int H, int W, int Cout, void* ws, size_t ws_bytes, void* stat_ws,
                                      size_t stat_ws_bytes, afd_stream_t stream);
int afd_conv3x3_forward_fold(const float* x, const float* in_aff, const float* in_slope /* may be NULL */,
                             const float* w, const float* bias, const float* slope, float* y, float* u, uint8_t* idx,
                             double* sums /* may be NULL */, int N, int Cin, int H, int W, int Cout, void* ws,
                             size_t ws_bytes, void* stat_ws, size_t stat_ws_bytes, afd_stream_t stream);
int afd_conv3x3_backward_weight_fold(const float* x, const float* in_aff, const float* in_slope /* may be NULL */,
                                     const float* dy, const uint8_t* codes /* may be NULL */, float* dw,
                                     float* dbias /* may be NULL */, const double* dy_sums /* may be NULL */, int N,
                                     int Cin, int H, int W, int Cout, int dy_rows, int dy_cols, void* ws,
                                     size_t ws_bytes, afd_stream_t stream);
int afd_conv2d_backward_weight(const float* x, const float* dy, float* dw,
                               float* dbias /* may be NULL */, int N, int Cin, int H, int W,
                               int Cout, int K, int pad, int dil, void* ws, size_t ws_bytes,
                               afd_stream_t stream);

/* The same two calls for a convolution whose result feeds nn.MaxPool2d(2, 2) (models.py:264,
 * floor mode): the pool never reads an odd last row / column of y, and its backward leaves them
 * zero in dy.  Only y[.., :out_rows, :out_cols] is computed (the rest of y is left untouched);
 * dy is taken to be zero outside [:dy_rows, :dy_cols].  Pass the full extents for the plain
 * behaviour. */
int afd_conv2d_forward_cropped(const float* x, const float* w, const float* bias, float* y, int N,
                               int Cin, int H, int W, int Cout, int K, int pad, int dil,
                               int out_rows, int out_cols, void* ws, size_t ws_bytes,
                               afd_stream_t stream);
int afd_conv2d_backward_weight_cropped(const float* x, const float* dy, float* dw, float* dbias,
                                       int N, int Cin, int H, int W, int Cout, int K, int pad,
                                       int dil, int dy_rows, int dy_cols, void* ws,
                                       size_t ws_bytes, afd_stream_t stream);
/* The same with the per-channel sums of dy (double[Cout], e.g. from afd_bn_backward_apply_sums) when the caller
 * has them: the bias gradient is then taken from there where the kernel would otherwise make a pass over dy for
 * it (dy_sums may be NULL: identical to the call above).  Reference: the bias gradient of nn.Conv2d's backward. */
int afd_conv2d_backward_weight_sums(const float* x, const float* dy, float* dw, float* dbias,
                                    const double* dy_sums, int N, int Cin, int H, int W, int Cout, int K,
                                    int pad, int dil, int dy_rows, int dy_cols, void* ws, size_t ws_bytes,
                                    afd_stream_t stream);

/* First block for single-channel inputs, fused: Conv2d(1 -> Cout, 3x3, pad) + PReLU +
 * MaxPool2d(2,2) (reference models.py:255-259 with args.input_dim[1] == 1).  Forward writes
 * only the pooled tensor u [N][Cout][Hp][Wp] and the 3-bit code idx (as afd_prelu_pool_*);
 * backward consumes du, idx, u and x and produces dw [Cout][1][3][3], dbias [Cout] (may be
 * NULL) and dslope (+=); the pre-pool activation and its gradient are never materialised.
 * Hp = (H + 2 pad - 2) / 2.  sums != NULL (round 5; Cout <= 128): the forward launch also leaves the batch sums of the
 * BatchNorm behind the pool (models.py:260) -- sums[c] = sum of u over batch and pixels, sums[Cout + c] = sum of u^2, in
 * double precision from one partial row per workgroup in stat_ws (afd_conv1_pool_stats_workspace_bytes): no
 * statistics pass over u. */
size_t afd_conv1_pool_workspace_bytes(int N, int H, int W, int Cout, int pad);
size_t afd_conv1_pool_stats_workspace_bytes(int N, int H, int W, int Cout, int pad);
/* 1 when taking the sums in the forward launch is faster than a statistics pass over u (rows of at least one
 * workgroup's 1024 pooled columns: the level-14 input); the kernel is correct at every width. */
int afd_conv1_pool_stats_applicable(int N, int H, int W, int Cout, int pad);
int afd_conv1_pool_forward(const float* x, const float* w, const float* bias, const float* slope,
                           float* u, uint8_t* idx, double* sums, void* stat_ws, size_t stat_ws_bytes, int N, int H,
                           int W, int Cout, int pad, afd_stream_t stream);
int afd_conv1_pool_backward(const float* x, const float* du, const uint8_t* idx, const float* u,
                            const float* slope, float* dw, float* dbias, float* dslope /* += */,
                            int N, int H, int W, int Cout, int pad, void* ws, size_t ws_bytes,
                            afd_stream_t stream);
/* The same with the gradient of the pooled tensor given as du + alpha[c] * u + beta[c] (alpha, beta [Cout]):
 * the affine part of a following BatchNorm's backward is applied where du and u are read anyway
 * (see afd_conv1x1_prelu_bn_backward). */
int afd_conv1_pool_backward_affine(const float* x, const float* du, const uint8_t* idx, const float* u,
                                   const float* slope, const float* alpha, const float* beta, float* dw,
                                   float* dbias, float* dslope /* += */, int N, int H, int W, int Cout,
                                   int pad, void* ws, size_t ws_bytes, afd_stream_t stream);

/* ------------------------------------------------------------------------------------
 * HBM-bound layers.  `slope` is the device address of the single shared PReLU parameter
 * (nn.PReLU(), models.py:258); where it is "may be NULL" the PReLU is skipped.
 * ---------------------------------------------------------------------------------- */
/* Scalar moments of the transformed training set: replaces the per-batch WelfordEstimator
 * update of calc_normalization (wavelet_math.py:387-452, data_loader.py:27-71).
 * acc[0] += n, acc[1] += sum x, acc[2] += sum x^2, accumulated in double precision
 * (acc is device memory, zeroed by the caller before the first batch). */
int afd_moments_accumulate(const float* x, size_t n, double* acc, afd_stream_t stream);

/* Per-packet statistics and "block norm" of the packet front end: replaces the per-node
 * WelfordEstimator update and `node_wp / torch.max(torch.abs(node_wp))` inside the node loop of
 * compute_pytorch_packet_representation (wavelet_math.py:194-203) and the stack / log / sign
 * steps after it (:206-218).
 * x      [dev] raw coefficients [rows = B*T][P], P fastest (afd_wpt_forward with flags == 0)
 * sums   [dev] [2][P] doubles, += sum v and += sum v^2 per packet (zeroed by the caller)
 * absmax [dev] [P] floats, = max(absmax, max |v|) per packet (zeroed by the caller)
 * afd_packet_block_norm: out [B][C][T][P] = epilogue(x / absmax[p]) with the AFD_WPT_* flags of
 * afd_wpt_forward; absmax == NULL applies the epilogue alone. */
int afd_packet_stats(const float* x, long rows, int P, double* sums, float* absmax,
                     afd_stream_t stream);
int afd_packet_block_norm(const float* x, int B, int T, int P, const float* absmax /* may be NULL */,
                          unsigned flags, float power, float eps, float mean, float std, float sign_mean,
                          float sign_std, float* out, afd_stream_t stream);

/* torchvision Normalize with scalar statistics (wavelet_math.py:380-382): y = (x-mean)/std */
int afd_normalize_forward(const float* x, float* y, size_t n, float mean, float std,
                          afd_stream_t stream);
/* The per-channel form of torchvision's Normalize for [B][C][plane] features (the loss-less packets' two channels,
 * wavelet_math.py:380-382 with per-channel statistics): y = (x - means[c]) / stds[c]; means / stds are HOST arrays of
 * C <= 8 floats; one launch. */
int afd_normalize_channels_forward(const float* x, float* y, int B, int C, size_t plane, const float* means,
                                   const float* stds, afd_stream_t stream);
/* y[p][c][r] = x[p][r][c]: the .permute(0,1,3,2) of models.py:304 made contiguous */
int afd_transpose_last2(const float* x, float* y, int planes, int R, int C, afd_stream_t stream);

/* PReLU followed by Dropout(p) (models.py:291-292); the mask is a counter-based function of
 * (seed, element index) and is regenerated in backward */
int afd_prelu_dropout_forward(const float* z, const float* slope, float* y, size_t n, float p,
                              uint64_t seed, afd_stream_t stream);
int afd_prelu_dropout_backward(const float* z, const float* slope, const float* dy, float* dz,
                               float* dslope /* += */, size_t n, float p, uint64_t seed,
                               afd_stream_t stream);

/* PReLU + MaxPool2d(2,2) (models.py:258-259): z [NC][H][W] -> u [NC][H/2][W/2]; idx bits 0-1 =
 * argmax position (dy*2+dx) of each window, bit 2 = the winning z was <= 0; slope may be NULL
 * (plain max pool).  Backward takes the pooled OUTPUT u (not z): dz [NC][H][W] is fully
 * written; with a slope of exactly 0 the dslope term of pooled positions is lost. */
int afd_prelu_pool_forward(const float* z, const float* slope, float* u, uint8_t* idx, int NC,
                           int H, int W, afd_stream_t stream);
int afd_prelu_pool_backward(const float* u, const float* slope, const uint8_t* idx,
                            const float* du, float* dz, float* dslope /* += */, int NC, int H,
                            int W, afd_stream_t stream);
/* The same with the gradient of the pooled tensor given as A[c] du + B[c] u + K[c], coef [C][4] = (A, B, K, 0)
 * as written by afd_bn_backward_coef: the backward of a BatchNorm(affine=False) that follows the pool is applied
 * where du and u are read anyway (no batch-norm backward pass over the pooled tensor).  NC = N * C planes. */
int afd_prelu_pool_backward_affine(const float* u, const float* slope, const uint8_t* idx, const float* du,
                                   const float* coef /* may be NULL */, int C, float* dz,
                                   float* dslope /* += */, int NC, int H, int W, afd_stream_t stream);

/* (Sync)BatchNorm over x [N][C][HW] (models.py:260-289), optional PReLU fused on the input.
 * stats: sums[0..C) = sum, sums[C..2C) = sum of squares (double; the caller all-reduces them
 * across ranks and derives mean / invstd).  backward_stats: sum(dy), sum(dy * xhat).
 * backward_apply: dx = gamma invstd (dy - mean_dy - xhat mean_dy_xhat), chained through the
 * fused PReLU (dslope accumulates). gamma / beta may be NULL (affine=False). */
int afd_bn_stats(const float* x, const float* slope, double* sums, int N, int C, int HW,
                 afd_stream_t stream);
int afd_bn_apply_forward(const float* x, const float* slope, const float* mean,
                         const float* invstd, const float* gamma, const float* beta, float* y,
                         int N, int C, int HW, afd_stream_t stream);
int afd_bn_backward_stats(const float* x, const float* slope, const float* dy, const float* mean,
                          const float* invstd, double* sums, int N, int C, int HW,
                          afd_stream_t stream);
int afd_bn_backward_apply(const float* x, const float* slope, const float* dy, const float* mean,
                          const float* invstd, const float* gamma, const float* mean_dy,
                          const float* mean_dy_xhat, float* dx, float* dslope /* += */, int N,
                          int C, int HW, afd_stream_t stream);
/* The same, also adding the per-channel sums of dx into dx_sums (double[C], += ; NULL: none): what the convolution
 * that produced x needs as its bias gradient (SyncBatchNorm / BatchNorm2d backward, models.py:260-289). */
int afd_bn_backward_apply_sums(const float* x, const float* slope, const float* dy, const float* mean,
                               const float* invstd, const float* gamma, const float* mean_dy,
                               const float* mean_dy_xhat, float* dx, float* dslope /* += */,
                               double* dx_sums /* += */, int N, int C, int HW, afd_stream_t stream);

/* The small per-channel steps between the BatchNorm passes, one launch each instead of a chain of
 * elementwise torch kernels (nn.BatchNorm2d / SyncBatchNorm semantics, models.py:260-289):
 * finalize: from the packed sums [sum(C) | sum of squares(C) | count] -> mean, invstd (biased
 *   variance), running statistics updated in place with `momentum` (unbiased variance),
 *   *num_batches_tracked += 1.  count = sums[2C] when count < 0 (set by the caller before the
 *   cross-rank all-reduce), else the host value.  running_* / nbt may be NULL.
 * backward means: mdy = sums[0:C] / count, mdyx = sums[C:2C] / count (count as above, read from
 *   count_dev when count < 0). */
/* The small matrices of a BatchNorm(affine=False) folded into the 1x1 convolution after it (ops._BNConv1x1*;
 * reference models.py:260-262), one launch each:
 * fold forward: wf[co][ci] = w[co][ci] invstd[ci], bf[co] = b[co] - sum_ci wf[co][ci] mean[ci] (b may be NULL);
 * fold backward weights: dw = (G - db (x) mean) invstd from the gradient G against the un-normalised input, and
 *   the BatchNorm's backward sums sums[ci] = sum_co w db, sums[Cin + ci] = sum_co w dw (double; all-reduced by the
 *   caller across ranks);
 * fold backward affine: alpha, beta of du = wf^T dz + alpha u + beta from those sums and the count (count_dev when
 *   count < 0);
 * backward coef: coef[c] = (invstd, -invstd^2 mdyx, invstd^2 mdyx mean - invstd mdy, 0) for
 *   afd_conv1x1_prelu_bn_backward. */
int afd_bn_fold_forward(const float* w, const float* b, const float* mean, const float* invstd, float* wf,
                        float* bf, int C, int Cin, afd_stream_t stream);
int afd_bn_fold_backward_weights(const float* G, const float* db, const float* w, const float* mean,
                                 const float* invstd, float* dw, double* sums, int C, int Cin,
                                 afd_stream_t stream);
int afd_bn_fold_backward_affine(const double* sums, double count, const double* count_dev, const float* mean,
                                const float* invstd, float* alpha, float* beta, int Cin, afd_stream_t stream);
int afd_bn_backward_coef(const float* mean, const float* invstd, const float* mdy, const float* mdyx,
                         float* coef, int C, afd_stream_t stream);
/* fold_tab (may be NULL): also writes the pairs [C][2] = (mean, invstd) -- the table of a BatchNorm whose normalisation the
 * next convolution applies while it loads (afd_conv3x3_forward_fold). */
int afd_bn_finalize(const double* sums, int C, double count, float eps, float momentum, float* mean,
                    float* invstd, float* running_mean, float* running_var, long long* nbt,
                    double* count_out, float* fold_tab, afd_stream_t stream);
/* tab4 (may be NULL; needs mean and invstd): also writes [C][4] = (mean, invstd, mdy, mdyx), the table
 * afd_conv3x3_backward_data_bnapply takes. */
int afd_bn_backward_means(const double* sums, int C, double count, const double* count_dev,
                          float* mdy, float* mdyx, const float* mean, const float* invstd, float* tab4,
                          afd_stream_t stream);

/* Dropout(p) + permute(0,2,1,3).contiguous() (models.py:277,307): x [B][C][H][W] ->
 * y [B][H][C][W]; inverse != 0 runs the backward (dy [B][H][C][W] -> dx [B][C][H][W]) */
int afd_dropout_permute(const float* x, float* y, int B, int C, int H, int W, float p,
                        uint64_t seed, int inverse, afd_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Input side: the windowed read + resample of CustomDataset.__getitem__ (reference
 * src/audiofakedetect/data_loader.py:323-353: torchaudio.load(path, frame_offset, num_frames) +
 * torchaudio.functional.resample) for a whole batch.
 * afd_wav_read_windows (HOST pointers; no stream): window i = frames [frame_offsets[i], + win) of the first
 *   channel of the 16-bit PCM WAV file paths[i] -> out[i * win ..] (int16; frames past the end are 0), its sample
 *   rate -> rates[i]; `threads` reader threads.  AFD_ERR_UNSUPPORTED for files that are not 16-bit PCM.
 * afd_pcm16_resample (DEVICE pointers): out[b][i] = pcm[b][i] / 32768 when orig == nnew (bank NULL), else the
 *   polyphase Hann-windowed sinc filter out[j * nnew + ph] = sum_t bank[ph][t] x[j * orig + t - width], zero padded,
 *   bank [nnew][2 width + orig] (orig, nnew reduced by their gcd; n_out = ceil(nnew * n_in / orig)). */
int afd_wav_read_windows(const char* const* paths, const long long* frame_offsets, int n, int win, int16_t* out,
                         int* rates, int threads);
int afd_pcm16_resample(const int16_t* pcm, int B, int n_in, int orig, int nnew, int width, const float* bank,
                       float* out, int n_out, afd_stream_t stream);

/* Flatten(2) + Linear(F, O) + mean(1) (models.py:295-298,311): x [B][TD][F] -> y [B][O] */
int afd_linear_mean_forward(const float* x, const float* w, const float* bias, float* y, int B,
                            int TD, int F, int O, afd_stream_t stream);
int afd_linear_mean_backward(const float* x, const float* w, const float* dy, float* dx,
                             float* dw, float* db, int B, int TD, int F, int O,
                             afd_stream_t stream);

/* LCNN layers (reference models.py:68-131, 161-237).
 * afd_mfm_*    : MaxFeatureMap2D, y[n][c] = max(x[n][c], x[n][c + C/2]) over x [N][C][HW];
 *                sel (may be NULL in forward) records which half won, for the backward.
 * afd_gemm_nt  : C[M][N] = A[M][K] . B[N][K]^T + bias[N] (+ C when accumulate != 0), row-major
 *                with leading dimensions, exact-f32 MFMA: the input and recurrent projections
 *                of nn.LSTM (and any nn.Linear).
 * afd_lstm_cell: gates [B][4H] in torch order (i|f|g|o) -> c (in place), h (hstate [B][H] and
 *                a strided copy hout, the layer output slice). */
int afd_mfm_forward(const float* x, float* y, uint8_t* sel, int N, int C, int HW, afd_stream_t stream);
int afd_mfm_backward(const float* dy, const uint8_t* sel, float* dx, int N, int C, int HW,
                     afd_stream_t stream);
int afd_gemm_nt(const float* A, const float* B, const float* bias, float* C, int M, int N, int K,
                int lda, int ldb, int ldc, int accumulate, afd_stream_t stream);
int afd_lstm_cell(const float* gates, float* c, float* hout, float* hstate, int B, int H, int ldh,
                  afd_stream_t stream);
/* bf16 matrix-core path of the LCNN evaluation forward (BASELINE.json configs[4]; the reference runs the same
 * layers of models.py:68-131 under autocast): operands rounded to bf16, fp32 accumulation on
 * v_mfma_f32_32x32x16_bf16, fp32 tensors in memory.
 * afd_conv2d_forward_bf16: y = conv2d(x, w) + bias, stride 1, dilation 1, K in {1, 3, 5}, Cout <= 128;
 *                          mfm != 0 fuses MaxFeatureMap2D (models.py:203-209): y [N][Cout/2][Ho][Wo] =
 *                          max over the two channel halves; ws (afd_conv2d_bf16_workspace_bytes)
 *                          receives the bf16 weight image.
 * afd_gemm_nt_bf16       : afd_gemm_nt with bf16 operands. */
size_t afd_conv2d_bf16_workspace_bytes(int Cin, int Cout, int K);
int afd_conv2d_forward_bf16(const float* x, const float* w, const float* bias /* may be NULL */, float* y,
                            int N, int Cin, int H, int W, int Cout, int K, int pad, int mfm, void* ws,
                            size_t ws_bytes, afd_stream_t stream);
int afd_gemm_nt_bf16(const float* A, const float* B, const float* bias, float* C, int M, int N, int K,
                     int lda, int ldb, int ldc, int accumulate, afd_stream_t stream);
/* The same evaluation forward with bf16 STORAGE (csrc/lcnn_nhwc.hip): activations are channels-last bf16 tensors
 * [N][H][W][C] between the layers, weights are converted once per model state.  Replaces, per layer of
 * models.py:85-110, Conv2d -> MaxFeatureMap2D (:161-209) [-> MaxPool2d(2, 2)] [-> BatchNorm2d(affine=False) in
 * evaluation mode: a positive per-channel scale and a shift, folded into the convolution in front of it].
 * afd_lcnn_prep_conv_bf16 : w [Cout][Cin][K][K], bias (may be NULL), bn_mean / bn_var [Cout/2] (both NULL or both
 *                           given) -> wb_bb (afd_lcnn_prep_bytes): bf16 weight rows in matrix-tile order with
 *                           k = (ky K + kx) Cin + ci, followed by the folded fp32 bias.
 * afd_lcnn_conv1_nhwc_bf16: first layer, x fp32 [N][H][W] (one channel) -> y bf16 [N][Ho][Wo][Cout/2].
 * afd_lcnn_conv_nhwc_bf16 : K in {1, 3}, Cin in {32, 48, 64}: x bf16 [N][H][W][Cin] -> y bf16 [N][Ho][Wo][Cout/2].
 *                           pool (both convolutions; round 5): 0 = as above; 1 = the MaxPool2d(2, 2) that follows the
 *                           layer (models.py:87-107) runs in the epilogue and y is bf16 [N][Ho/2][Wo/2][Cout/2]; 2
 *                           (afd_lcnn_conv_nhwc_bf16 only) = the same with y in fp32 (bf16-rounded values) -- the
 *                           pre-pool tensor is never written, results bit-identical to conv + afd_lcnn_pool_nhwc_bf16.
 * afd_lcnn_pool_nhwc_bf16 : MaxPool2d(2, 2) (floor mode) on bf16 [N][H][W][C], C % 8 == 0; out_f32 != 0 writes
 *                           fp32 (the tensor that feeds the BLSTM layers as [N][H/2][(W/2) C]). */
size_t afd_lcnn_prep_bytes(int Cin, int Cout, int K);
int afd_lcnn_prep_conv_bf16(const float* w, const float* bias, const float* bn_mean, const float* bn_var, float eps,
                            void* wb_bb, int Cin, int Cout, int K, afd_stream_t stream);
int afd_lcnn_conv1_nhwc_bf16(const float* x, const void* wb_bb, void* y, int N, int H, int W, int Cout, int K, int pad,
                             int pool, afd_stream_t stream);
int afd_lcnn_conv_nhwc_bf16(const void* x, const void* wb_bb, void* y, int N, int H, int W, int Cin, int Cout, int K,
                            int pad, int pool, afd_stream_t stream);
int afd_lcnn_pool_nhwc_bf16(const void* x, void* y, int N, int H, int W, int C, int out_f32, afd_stream_t stream);
/* One step of an LSTM direction in evaluation (nn.LSTM inside BLSTMLayer, models.py:212-237), recurrent projection and
 * cell in one launch: gates = pre [B][4H] (input projection + biases of the step, order i | f | g | o) + hprev [B][H] .
 * Wh^T with bf16 operands (wh_bf16 = weight_hh [4H][H] converted by afd_f32_to_bf16), fp32 accumulation and cell;
 * c [B][H] is updated in place, h goes to hnext [B][H] (a different buffer than hprev) and to hout (row stride ldh). */
int afd_f32_to_bf16(const float* x, void* y, size_t n, afd_stream_t stream);
int afd_lstm_step_bf16(const float* pre, const void* wh_bf16, const float* hprev, float* c, float* hout, int ldh,
                       float* hnext, int B, int H, afd_stream_t stream);
/* The same step for BOTH directions of a bidirectional layer in one launch: every argument is a HOST array of two
 * pointers (forward direction, reverse direction). */
int afd_lstm_step_bf16_pair(const float* const* pre, const void* const* wh_bf16, const float* const* hprev,
                            float* const* c, float* const* hout, int ldh, float* const* hnext, int B, int H,
                            afd_stream_t stream);
/* A whole bidirectional layer in ONE launch (round 5): pre_fwd / pre_rev [T][B][4H] fp32 (input projections + biases of
 * every step, time-major); out [T][B][2H] fp32 (forward half | reverse half of every step); h_0 = c_0 = 0.
 * wh_*_bf16: weight_hh [4H][H] as bf16 in MATRIX-FRAGMENT order [H/8 tiles][H/16 k-steps][64 lanes][8]: element e of lane
 * (hh = lane / 32, r = lane % 32) of (tile, k-step) is weight_hh[(r / 8) H + 8 tile + r % 8][16 k-step + 8 hh + e], i.e.
 * weight_hh.view(4, H/8, 8, H/16, 2, 8).permute(1, 3, 4, 0, 2, 5) -- a wave's operand load is 1 KB of consecutive bytes.
 * A workgroup owns 32 batch rows of one direction and walks the T steps alone (the recurrence couples only the units of
 * one batch row): h in LDS, c in registers.  H a multiple of 16 up to 256.  Same products and accumulation order as
 * afd_lstm_step_bf16; the gates use the hardware exponential and reciprocal (sigmoid = 1 / (1 + exp(-x)), tanh = 2 sigmoid(2x) - 1),
 * where the step kernel calls expf / tanhf: the two agree to 2e-3 of the largest output over a layer (tests/test_lcnn_gpu.py),
 * not bit for bit. */
int afd_blstm_layer_bf16(const float* pre_fwd, const float* pre_rev, const void* wh_fwd_bf16, const void* wh_rev_bf16,
                         float* out, int T, int B, int H, afd_stream_t stream);
/* One step of the LSTM backward pass (BPTT of nn.LSTM inside BLSTMLayer, models.py:212-237):
 * gates = saved pre-activation sums [B][4H] of the step, c / cprev = cell state after / before it
 * (cprev NULL = zero), dh = gradient reaching h_t (row stride lddh), dc = running cell-state
 * gradient [B][H] (updated in place for the previous step), dpre = gradient of the
 * pre-activations [B][4H]. */
int afd_lstm_cell_backward(const float* gates, const float* c, const float* cprev, const float* dh,
                           int lddh, float* dc, float* dpre, int B, int H, afd_stream_t stream);

/* CrossEntropyLoss (mean) + its gradient + number of correct argmax predictions
 * (train_classifier.py:970-979); dlogits / correct may be NULL */
int afd_cross_entropy(const float* logits, const int64_t* labels, float* loss, float* dlogits,
                      float* correct, int B, int O, afd_stream_t stream);

/* n tensors srcs[k] (counts[k] floats each; NULL: zeros) -> dst + offsets[k], one launch per 96 tensors: the
 * per-parameter gradients of a backward pass into the optimizer's flat gradient arena (ops.FusedAdam.gather_grads).
 * srcs / offsets / counts are HOST arrays. */
int afd_multi_gather(const float* const* srcs, const long* offsets, const long* counts, int n, float* dst,
                     afd_stream_t stream);
/* Adam with coupled L2 weight decay over one flat parameter arena
 * (train_classifier.py:986,1215-1219); grads are multiplied by grad_scale first */
int afd_adam_step(float* params, const float* grads, float* m, float* v, size_t n, float lr,
                  float beta1, float beta2, float eps, float weight_decay, int step,
                  float grad_scale, afd_stream_t stream);

/* ------------------------------------------------------------------------------------
 * The in-step collectives of data-parallel training straight on RCCL, on the caller's stream (opt-in).
 * Replaces: the NCCL all-reduces DistributedDataParallel and nn.SyncBatchNorm issue per step
 *           (reference src/audiofakedetect/train_classifier.py:319-323, models.py:260-289), which through
 *           torch.distributed pay a stream hand-off each.  librccl is taken from the process with dlopen.
 * afd_rccl_unique_id : rank 0 fills 128 bytes [host]; the host layer broadcasts them to every rank
 * afd_rccl_init      : every rank, on its own device: ncclCommInitRank(world, id, rank)
 * afd_rccl_all_reduce_sum : in-place sum over the ranks of count f32 (is_double = 0) / f64 (1) values [dev]
 * afd_rccl_world     : ranks of the communicator, 0 when none is up
 * ---------------------------------------------------------------------------------- */
int afd_rccl_unique_id(void* id128 /* [host] 128 bytes */);
int afd_rccl_init(const void* id128 /* [host] */, int rank, int world);
int afd_rccl_all_reduce_sum(void* buf, long count, int is_double, afd_stream_t stream);
int afd_rccl_world(void);
int afd_rccl_destroy(void);

#ifdef __cplusplus
}
#endif
#endif /* AFD_HIP_H */
