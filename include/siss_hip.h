/* siss_hip.h -- C ABI of libsiss_hip.so: the MI355X (gfx950) kernels of the SISS unlearning step.
 *
 * The reference (claserken/SISS) is pure Python and has no FFI of its own; its hot path reaches
 * device code only through torch / diffusers.  These entry points are what a Python-side binding
 * for that path binds instead (ctypes stub: siss_amd/lib.py).  Conventions:
 *   - plain C: raw DEVICE pointers + sizes, no torch types; bf16 tensors are `void*` (uint16 storage);
 *   - every launcher takes the HIP stream as its last argument (`void*` = hipStream_t) and only
 *     enqueues work: no allocation, no synchronisation, safe under hipGraph capture;
 *   - return value: 0 = ok, 1 = bad argument (shape / alignment the kernel does not cover),
 *     2 = launch error;  the *_words functions return buffer sizes and take no stream;
 *   - activations: NHWC bf16 with a one-pixel zero halo, flattened to rows (siss_amd/layout.py);
 *     "sets" = the two cotangent / gradient sets (g_x, g_a) of the dual backward.
 */
#ifndef SISS_HIP_H
#define SISS_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- siss_loss.hip ----
 * Fused SISS pre/post kernels.  Replaces losses/ddpm_deletion_loss.py:12-53 (mixture select, eps_x/eps_a,
 * dist_x/dist_a, ratios, importance weights, weighted losses) and delete_celeb.py:602-603 (add_noise x2 with
 * the same noise) / :686-687 (loss normalisation that seeds the two backward passes).
 */
/* Number of f64 words the caller must provide in `partials` for a given (B, chw). */
long siss_loss_partials_words(int B, long chw);
int siss_mixture_fwd(const void* x0, const void* a0, const void* noise, int in_bf16, const int64_t* t, const
    float* u, const float* alphas_cumprod, const float* gamma_tab, const float* sigma_tab, float lambd, int B,
    long chw, void* x_mix, float* gamma_t, float* sigma_t, float* dist_x, float* dist_a, float* iw_x, float* iw_a,
    double* partials, void* stream);
/* Same outputs as siss_mixture_fwd, but from caller-provided noisy batches (DDPMDeletionLoss surface). */
int siss_mixture_select(const void* noisy_keep, const void* noisy_forget, const void* x0, const void* a0, int
    in_bf16, const int64_t* t, const float* u, const float* gamma_tab, const float* sigma_tab, float lambd, int B,
    long chw, void* x_mix, float* gamma_t, float* sigma_t, float* dist_x, float* dist_a, float* iw_x, float* iw_a,
    double* partials, void* stream);
int siss_loss_bwd_seed(const float* pred, const void* x_mix, const void* x0, const void* a0, int in_bf16,
    const float* gamma_t, const float* sigma_t, const float* iw_x, const float* iw_a, float scale, int B, long
    chw, float* c_x, float* c_a, float* loss_x, float* loss_a, float* sum_loss_x, float* sum_loss_a, double*
    partials, void* stream);
/* Plain squared error against `target` (No-IS / NegGrad / naive): c = 2*scale*(pred-target). */
int siss_mse_bwd_seed(const float* pred, const void* target, int target_bf16, float scale, int B, long chw,
    float* c, float* loss, float* sum_loss, double* partials, void* stream);

/* ---- gemm_nt.hip ----
 * Panelled NT GEMM on bf16 MFMA: conv3x3 / conv1x1 / linear fprop and dgrad.  Replaces the cuDNN/cuBLAS
 * kernels the reference reaches through diffusers' ResnetBlock2D / Downsample2D / Upsample2D / Attention
 * (call site: losses/ddpm_deletion_loss.py:24 `unet(...)`, backward: delete_celeb.py:691,:702).
 */
/* Flat argument list (ctypes-friendly).  shifts/coffs are HOST arrays of npanels ints. Returns SISS_ERR_ARG for shapes the kernel does not cover (Kp % 64, alignment, panel count). */
int siss_gemm_nt(const void* A, long lda, const void* W, void* C, long ldc, const float* bias, const float*
    rowbias, const void* R, long ldr, int M, int N, int Kp, int npanels, const int* shifts, const int* coffs, int
    rows_per_image, int Hp, int Wp, float alpha, int batch, long strideA, long strideW, long strideC, void*
    stream);

/* ---- gemm_tn.hip ----
 * Panelled TN GEMM on bf16 MFMA: weight gradients for both cotangent sets (g_x, g_a) in one pass.
 * Replaces the wgrad half of the two `accelerator.backward` calls (delete_celeb.py:691,:702) and the
 * clone / subtract split of :694-711.
 */
/* dW must be zeroed (or hold the running sum for gradient accumulation) before the call. Rows [row_begin, row_end) of every set are reduced; shifts/coffs are HOST arrays. */
int siss_gemm_tn(const void* Y, long ldy, const void* X, long ldx, float* dW, long set_stride, int N, int C,
    int npanels, const int* shifts, const int* coffs, int nsets, int rows_per_set, long x_set_rows, int row_begin,
    int row_end, int nsplits, const void* zero_page, void* stream);

/* ---- groupnorm.hip ----
 * GroupNorm(+SiLU) forward / backward on padded NHWC.  Replaces torch.nn.GroupNorm + F.silu inside
 * diffusers' ResnetBlock2D.norm1/norm2, Attention.group_norm and UNet2DModel.conv_norm_out.
 */
/* floats needed in `partial` for n samples */
long siss_gn_partial_words(int n, int H, int W, int C, int G);
/* y = act(GroupNorm(x)); x padded NHWC; y padded or compact ([N][H*W][C]).  Writes mean/rstd [N][G]. */
int siss_groupnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float*
    rstd, float* partial, int N, int H, int W, int C, int G, float eps, int silu, int out_compact, void* stream);
/* dx (padded, n2 samples) from dy (n2 samples, padded or compact) and the saved x (nx samples, x index = n2 % nx).  dgamma/dbeta: [sets][...] accumulated atomically at set = n2 / set_images with `set_stride` floats between sets.  accum (optional, padded like dx) is added to dx; colsum (optional, [n2][C] f32, pre-zeroed) receives the per-sample channel sums of dx. */
int siss_groupnorm_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const float*
    mean, const float* rstd, void* dx, const void* accum, float* dgamma, float* dbeta, float* colsum, float*
    partial, int n2, int nx, int set_images, long set_stride, int H, int W, int C, int G, int silu, int
    dy_compact, void* stream);

/* ---- conv_small.hip ----
 * conv_out (Cout = image channels): direct kernels for UNet2DModel.conv_out and its backward.
 */
int siss_conv_out_fprop(const void* x, const float* w, const float* bias, float* pred, int B, int H, int W,
    int C, int CO, void* stream);
int siss_conv_out_dgrad(const float* c, const float* w, void* dx, int N2, int H, int W, int C, int CO, void*
    stream);
/* c: [nsets*set_images][CO][H][W] f32 cotangent; x: saved conv_out input (nx images, index n2 % nx). */
int siss_conv_out_wgrad(const float* c, const void* x, float* dW, float* dbias, int nsets, int set_images, int
    nx, long set_stride_w, long set_stride_b, int H, int W, int C, int CO, void* stream);

/* ---- attention.hip ----
 * Row softmax fwd/bwd of the single-head attention block (diffusers Attention, upcast_softmax=True).
 */
int siss_softmax_fwd(const void* s, void* p, long rows, int S, void* stream);
int siss_softmax_bwd(const void* p, const void* dp, void* ds, long rows, long p_rows, int S, float scale,
    void* stream);

/* ---- timeemb.hip ----
 * Sinusoidal timestep embedding (diffusers Timesteps/get_timestep_embedding), TimestepEmbedding MLP and
 * ResnetBlock2D.time_emb_proj linears (M = batch rows), forward and backward.
 */
int siss_timestep_sincos(const int64_t* t, float* out, int B, int dim, int flip_sin_to_cos, float freq_shift,
    void* stream);
int siss_linear_small_fwd(const float* x, const float* W, const float* b, float* y, int M, int N, int K, int
    act_in_silu, void* stream);
/* dx may be NULL (first layer).  dW/db are accumulated in place (+=): zero them at step start. db2 (optional) receives the same bias sums (conv1 bias shares its gradient with time_emb_proj bias). */
int siss_linear_small_bwd(const float* dy, const float* yact, const float* x, const float* W, float* dx, int
    accumulate_dx, float* dW, float* db, float* db2, int M2, int Mx, int set_rows, long set_stride_w, long
    set_stride_b, int N, int K, int act_in_silu, void* stream);

/* ---- elementwise.hip ----
 * Data movement on the padded-NHWC layout: F.interpolate(nearest 2x), torch.cat of skip connections,
 * Downsample2D's F.pad+stride-2 gather (space-to-depth), attention reshape/residual, bias-gradient column sums.
 */
int siss_upsample2x(const void* in, void* out, int N, int H, int W, int C, void* stream);
int siss_upsample2x_bwd(const void* dout, void* din, int N, int H, int W, int C, void* stream);
int siss_concat(const void* a, const void* b, void* out, int N, int H, int W, int Ca, int Cb, void* stream);
int siss_concat_bwd(const void* dcat, void* da, void* db, int accumulate_b, int N, int H, int W, int Ca, int
    Cb, void* stream);
int siss_add_inplace(void* a, const void* b, int N, int H, int W, int C, void* stream);
int siss_space_to_depth(const void* in, void* z, int N, int H, int W, int C, void* stream);
int siss_depth_to_space(const void* dz, void* din, int accumulate, int N, int H, int W, int C, void* stream);
int siss_pad_to_compact(const void* in, void* out, int N, int H, int W, int C, void* stream);
/* out_padded = compact (+ res_padded) */
int siss_compact_add_to_pad(const void* comp, const void* res, void* out, int N, int H, int W, int C, void*
    stream);
int siss_transpose_bf16(const void* in, void* out, int batch, int R, int C, void* stream);
/* out[set][0:C] += column sums of y over each set's rows.  y: [nsets*rows_per_set][C] bf16. out2 (optional) receives the same sums (two biases that feed the same pre-activation). */
int siss_colsum(const void* y, long rows_per_set, int C, int nsets, long out_set_stride, float* out, float*
    out2, void* stream);
int siss_im2col3x3(const void* img, int img_bf16, void* out, int N, int Cin, int H, int W, int K, void*
    stream);

/* ---- optimizer.hip ----
 * Flat-buffer norm-fix + recombine + clip + AdamW.  Replaces delete_celeb.py:714-753 (five 450-tensor
 * loops), :767 (clip_grad_norm_) and :769 (torch.optim.AdamW.step).
 */
long siss_opt_partials_words(void);
long siss_opt_scalars_words(void);
/* pass 1 + on-device scalars.  `scalars` (16 floats, zero-initialised once; holds the step count). */
int siss_grad_norms_scale(const float* gx, const float* ga, long n, int mode, float knob, float max_norm,
    float beta1, float beta2, double* partials, float* scalars, void* stream);
/* pass 2.  shadow (bf16 copy of the updated parameters) and g_out (final clipped gradient) are optional. */
int siss_recombine_clip_adamw(const float* gx, const float* ga, float* p, float* m, float* v, void* shadow,
    float* g_out, long n, float lr, float beta1, float beta2, float eps, float wd, const float* scalars, void*
    stream);
int siss_cast_f32_bf16(const float* src, void* dst, long n, void* stream);
int siss_conv_weight_dgrad_layout(const float* w, void* wt, int taps, int co, int ci, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SISS_HIP_H */
