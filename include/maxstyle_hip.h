/* maxstyle_hip.h - C ABI of libmaxstyle_hip.so (gfx950 / MI355X kernels for the MaxStyle inner loop).
 *
 * The reference (cherise215/MaxStyle) has no FFI layer: its hot path is eager PyTorch.  The drop-in
 * boundary is therefore the reference's three Python call signatures and THIS header is
 * the C ABI underneath them: every entry point names the reference lines whose arithmetic it replaces.
 *
 * Conventions
 *   - all tensors are fp32, NCHW, contiguous, device memory owned by the caller (PyTorch caching allocator);
 *     the library allocates nothing; scratch is a caller-provided workspace (size queries below)
 *   - kernels are enqueued asynchronously on `stream` (a hipStream_t passed as void*); no host sync, so every
 *     entry point may be captured into a hipGraph
 *   - return value: 0 = ok, <0 = MS_ERR_* (invalid argument / alignment / workspace), >0 = hipError_t;
 *     ms_last_error() returns a thread-local description; nothing throws across the ABI
 *   - re-entrant and thread-safe: no mutable global state
 */
#ifndef MAXSTYLE_HIP_H
#define MAXSTYLE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int ms_version(void);
const char* ms_last_error(void);

/* ---- MaxStyle layer: src/advanced/maxstyle.py:140-189 ------------------------------------------------ */

/* bytes of scratch needed by ms_style_moments / ms_style_fwd / ms_style_bwd for a [B,C,H*W] tensor */
size_t ms_style_ws_bytes(int B, int C, int HW);

/* mu = mean_HW(x), sig = sqrt(var_HW(x, unbiased) + eps) per plane.          maxstyle.py:157-159 */
int ms_style_moments(const float* x, float* mu, float* sig, int planes, int HW, float eps, void* ws, size_t ws_bytes, void* stream);

/* Per-plane affine coefficients  A = sig(1-l)+sig[perm]l + gamma_noise*gamma_std,  S = mu(1-l)+mu[perm]l + beta_noise*beta_std
 * with l = clamp(lmda,0,1).  compute_std != 0: gamma_std[c]=std_b(sig[:,c]), beta_std[c]=std_b(mu[:,c]) (unbiased) are
 * computed from mu/sig and stored (the reference caches them on the first forward), else they are read.
 * lmda == NULL: no style mixing (mix_style=False); gamma_noise == beta_noise == NULL: no_noise=True.  maxstyle.py:165-185 */
int ms_style_coeffs(float* mu, float* sig, float* gamma_std, float* beta_std, int compute_std, const float* lmda,
                    const float* gamma_noise, const float* beta_noise, const int64_t* perm, float* coefA, float* coefS,
                    int B, int C, void* stream);

/* y = A * ((x - mu) / sig) + S                                                 maxstyle.py:161,184-185 */
int ms_style_apply(const float* x, float* y, const float* mu, const float* sig, const float* coefA, const float* coefS,
                   int planes, int HW, void* stream);

/* The fused forward: moments + coeffs + apply (K1 of SURVEY.md 2.2).  Outputs y, and mu/sig/coefA/coefS [B*C]
 * (kept for the backward pass), gamma_std/beta_std [C] (written when compute_std != 0). */
int ms_style_fwd(const float* x, float* y, float* mu, float* sig, float* gamma_std, float* beta_std, int compute_std,
                 const float* lmda, const float* gamma_noise, const float* beta_noise, const int64_t* perm,
                 float* coefA, float* coefS, int B, int C, int HW, float eps, void* ws, size_t ws_bytes, void* stream);

/* Backward of the layer (autograd of maxstyle.py:161-185 with mu/sig detached):
 *   dx = dy*A/sig (skipped when dx == NULL);  d_gamma = gamma_std*sum(dy*xhat);  d_beta = beta_std*sum(dy);
 *   d_lmda[b] = 1[0<=lmda<=1] * sum_c (sig[perm b]-sig[b])*S2 + (mu[perm b]-mu[b])*S1.   Any of d_* may be NULL. */
int ms_style_bwd(const float* dy, const float* x, float* dx, const float* mu, const float* sig, const float* coefA,
                 const float* gamma_std, const float* beta_std, const float* lmda, const int64_t* perm,
                 float* d_gamma, float* d_beta, float* d_lmda, int B, int C, int HW, void* ws, size_t ws_bytes, void* stream);

/* torch.optim.Adam(lr, betas=(b1,b2), eps, weight_decay=0, amsgrad=False) single update on a flat buffer
 * (advanced_triplet_recon_segmentation_model.py:537,562).  step is 1-based; if step_dev != NULL the kernel uses
 * *step_dev + 1 instead (graph replay) - advance it with ms_counter_incr. */
int ms_adam_step(float* p, const float* g, float* m, float* v, int n, float lr, float b1, float b2, float eps, int step,
                 const int* step_dev, void* stream);
int ms_counter_incr(int* counter, void* stream);

#ifdef __cplusplus
}
#endif
#endif
