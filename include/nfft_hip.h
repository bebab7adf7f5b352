/* nfft_hip.h -- C ABI of the MI355X-native NFFT forward/adjoint hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch types.  Every
 * entry point names the interface of the reference (dominikbuenger/torch_nfft)
 * it replaces.  All pointers are DEVICE pointers unless stated otherwise; all
 * work is enqueued on `stream` (a hipStream_t passed as void*) and the calls
 * return without synchronising the device.  Temporaries come from a
 * caller-provided workspace (the host side hands in memory from its caching
 * allocator), so no entry point allocates or frees device memory.
 *
 * Data layouts (reference: docs/source/theory/dataformat.rst:19-63):
 *   pos    float32 [n, dim]            points on the torus, nominally in [-1/2, 1/2)
 *   batch  int64   [n] or NULL         sorted point-set index of every point
 *   x      spatial coefficients  [n, C]            float32 (real) or complex64
 *   xhat   spectral coefficients [B, N^dim, C]     float32 (real) or complex64;
 *          frequency k in [-N/2, N/2) is stored at index k + N/2 on every axis
 * Complex data is interleaved (re, im) float32 pairs, i.e. torch.complex64.
 *
 * Return value: 0 on success, otherwise one of the NFFT_HIP_E* codes;
 * nfft_hip_last_error() returns a human-readable message for the calling thread.
 */
#ifndef NFFT_HIP_H
#define NFFT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NFFT_HIP_OK 0
#define NFFT_HIP_EINVAL 1     /* "Input mismatch" (reference: CHECK_INPUT, csrc/cuda/cuda_utils.cu:3) */
#define NFFT_HIP_EWORKSPACE 2 /* workspace missing or too small */
#define NFFT_HIP_EFFT 3       /* rocFFT plan creation/execution failed (reference: "Failed to create CUFFT plan", core_cuda.cu:255-268) */
#define NFFT_HIP_EHIP 4       /* HIP runtime error (reference aborts the process, cuda_utils.cu:7-14; we report) */
#define NFFT_HIP_EKERNEL 5    /* a kernel of an earlier call on this device reported a fault (nfft_hip_check_status) */

#define NFFT_HIP_ABI_VERSION 4

int nfft_hip_abi_version(void);
const char *nfft_hip_last_error(void);

/* Faults that only a running kernel can detect -- a bounded wait inside the streamed interpolation kernel ran out,
 * a batch index outside [0, batch_size) in the middle of the batch vector, a cached plan whose points have changed
 * (nfft_hip_plan_verify) -- are raised in a host-mapped status block
 * of the device and turned into an error by the NEXT entry point called on that device: it returns NFFT_HIP_EKERNEL
 * (NFFT_HIP_EINVAL for the batch vector) without running, and clears the flag.  nfft_hip_check_status looks at the
 * block on demand; with synchronize != 0 it first waits for `stream`, so that a fault of the work just enqueued is
 * seen.  Replaces CHECK_ERRORS of the reference (csrc/cuda/cuda_utils.cu:5-16: cudaDeviceSynchronize, print,
 * exit()) for the failures a kernel can only report itself. */
int nfft_hip_check_status(void *stream, int synchronize);

/* Problem description shared by the entry points below.
 * Mirrors the arguments of the reference operators (csrc/core.cpp:43-105):
 *   dim           pos.size(1), 1..3                       (core_cuda.cu:50-51)
 *   num_points    pos.size(0)
 *   num_columns   x.numel() / num_points  resp.  x.numel() / (B * N^dim)   (core_cuda.cu:84, 108-113)
 *   batch_size    batch[-1] + 1, or 1 when batch is NULL  (core_cuda.cu:60-65) -- read back by the HOST side
 *   N             bandwidth (even, >= 2);  m  window cutoff, 1 <= m, 2m+2 <= 2N
 *   flags         hints about the point geometry (0 = none).  They never change results, only which kernels run; a
 *                 point plan must be used with the flags it was built with.
 */
#define NFFT_HIP_POINTS_IN_QUARTER_BALL 1 /* every point lies within radius 1/4 of the origin -- the fastsum geometry
                                           * (test/test_fastsum.py:17-18, torch_nfft/kernel.py:77): the points occupy at
                                           * most 1/8 of the grid, so their local density is 8x the average */
typedef struct nfft_hip_problem {
    int32_t dim;
    int32_t flags;
    int64_t num_points;
    int64_t num_columns;
    int64_t batch_size;
    int64_t N;
    int64_t m;
} nfft_hip_problem;

/* Bytes of workspace needed by nfft_hip_adjoint / nfft_hip_forward for this problem
 * (creates and caches the rocFFT plans on the current device; returns < 0 on error).
 * x_is_complex / real_output as in the calls below. */
int64_t nfft_hip_adjoint_workspace_bytes(const nfft_hip_problem *p, int x_is_complex, int real_output);
int64_t nfft_hip_forward_workspace_bytes(const nfft_hip_problem *p, int x_is_complex, int real_output);

/* Adjoint NFFT:  y[b, k+N/2, c] ~= sum_{i: batch[i]=b} x[i,c] exp(+2 pi i k.pos[i]).
 * Replaces nfft_adjoint_cuda (csrc/cuda/core_cuda.cu:144-336), i.e. the operator
 * torch_nfft::nfft_adjoint(pos, x, batch, N, m, real_output) (csrc/core.cpp:43-55, 177).
 *   x  [n, C] float32 (x_is_complex = 0) or complex64 (x_is_complex = 1)
 *   y  [B, N^dim, C] complex64, or float32 holding the real part when real_output != 0;
 *      every element is written (the caller need not zero it). */
int nfft_hip_adjoint(const nfft_hip_problem *p, const float *pos, const void *x, int x_is_complex,
                     const int64_t *batch, int real_output, void *y,
                     void *workspace, int64_t workspace_bytes, void *stream);

/* Forward NFFT:  y[i,c] ~= sum_k xhat[batch[i], k+N/2, c] exp(-2 pi i k.pos[i]).
 * Replaces nfft_forward_cuda (csrc/cuda/core_cuda.cu:340-531), i.e. the operator
 * torch_nfft::nfft_forward(pos, x, batch, m, real_output) (csrc/core.cpp:94-105, 178).
 *   xhat [B, N^dim, C] float32 or complex64;  y [n, C] complex64 or float32 (real part). */
int nfft_hip_forward(const nfft_hip_problem *p, const float *pos, const void *xhat, int x_is_complex,
                     const int64_t *batch, int real_output, void *y,
                     void *workspace, int64_t workspace_bytes, void *stream);

/* 0 when the library runs this problem WITHOUT a point plan: problems whose oversampled grid ((2N)^dim cells, 2N a power
 * of two, at most 4096 cells) fits one workgroup's LDS and whose point sets are small run nfft_hip_adjoint / nfft_hip_forward as one
 * kernel each -- workspace may be NULL for them, and building a plan for the *_planned entry points would only add five
 * launches.  1 otherwise (callers that transform the same points repeatedly then build the plan once).  No reference
 * counterpart: the reference recomputes shifts and window values in every call (core_cuda.cu:188-211). */
int nfft_hip_plan_needed(const nfft_hip_problem *p);

/* The same two transforms on an existing point plan (nfft_hip_plan_points below): the plan depends only on
 * (pos, batch, dim, N, m) -- with them num_points and batch_size, from which the tiling is chosen -- and on whether
 * num_columns is 1 or larger (sparse 3-D problems: the spreading kernel's tiling of the plan differs), and can be shared by
 * any number of adjoint / forward calls on the same points with column counts of the same class -- pass the same
 * nfft_hip_problem to the plan and to its users.  The
 * reference re-derives shifts and psi in every call but reuses them when sources.is_same(targets)
 * (core_cuda.cu:552-564).  The workspace sizes are those of the un-planned calls. */
int nfft_hip_adjoint_planned(const nfft_hip_problem *p, const void *plan, const void *x, int x_is_complex,
                             int real_output, void *y, void *workspace, int64_t workspace_bytes, void *stream);
int nfft_hip_forward_planned(const nfft_hip_problem *p, const void *plan, const void *xhat, int x_is_complex,
                             int real_output, void *y, void *workspace, int64_t workspace_bytes, void *stream);

/* ---- stage-level entry points (used by the parity tests and by bench.py to time
 * the spreading kernel on its own; the two calls above are built from them) ---- */

/* Size in bytes of a point plan (tile-sorted copy of the points) for this problem. */
int64_t nfft_hip_plan_bytes(const nfft_hip_problem *p);

/* Bin the points into grid tiles.  Replaces compute_shifts_kernel + compute_psi_kernel
 * (csrc/cuda/spatial_window_operations.cu:38-97, launched at core_cuda.cu:188-211): instead of
 * materialising shifts and 2m+2 window values per point and axis in HBM, the points are
 * counting-sorted by tile and the window is re-evaluated in registers by the consumers. */
int nfft_hip_plan_points(const nfft_hip_problem *p, const float *pos, const int64_t *batch,
                         void *plan, int64_t plan_bytes, void *stream);

/* Verification of a plan that is kept across calls.  nfft_hip_plan_points leaves a 64-bit checksum of pos (and batch) in
 * the plan -- its seal, formed by the pass that counts the points, at no extra cost; nfft_hip_plan_verify recomputes it
 * from the arrays as they are NOW (one streaming pass, ~25 us for 10^7 3-D points) and raises a device fault when it
 * differs: the next entry point on the device returns NFFT_HIP_EINVAL ("stale point plan"), nfft_hip_check_status sees
 * it on demand.  A cache of plans keyed on buffer identity (core.so's: tensor address + version counter) cannot see a
 * write that bypasses its key; the reference has no such state -- it recomputes shifts and psi in every call
 * (csrc/cuda/core_cuda.cu:188-211) -- so a drop-in must not return a transform of points that are no longer there
 * without saying so. */
int nfft_hip_plan_verify(const nfft_hip_problem *p, const float *pos, const int64_t *batch, void *plan, void *stream);

/* Spreading (adjoint gridding):  grid[(b*Cr + cr), u] += xr[i, cr] * prod_k psi_k(i, u_k)
 * over real columns cr (a complex x is viewed as 2C real columns).  Replaces
 * real_/complex_adjoint_window_convolution_kernel (spatial_window_operations.cu:103-211).
 *   grid  float32 [B*Cr, (2N)^dim] real planes; every cell is written by this call.
 *   scratch  nfft_hip_spread_scratch_bytes(p, Cr) bytes: the tile-ordered copy of xr (n * Cr floats, or one per plan
 *            entry when the problem is sparse enough for the owner-computes kernel, whose plan enters a point into
 *            every tile its window touches) and one word per plane (its largest |x|, the operand scale of the
 *            matrix-core kernel). */
int64_t nfft_hip_spread_scratch_bytes(const nfft_hip_problem *p, int64_t real_columns);
int nfft_hip_spread(const nfft_hip_problem *p, const void *plan, const float *xr, int64_t real_columns,
                    float *grid, float *scratch, void *stream);

/* Interpolation (forward gather):  yr[i, cr] = sum_u grid[(b*Cr + cr), u] * prod_k psi_k(i, u_k).
 * Replaces complex_/real_forward_window_convolution_kernel (spatial_window_operations.cu:214-332). */
int nfft_hip_interpolate(const nfft_hip_problem *p, const void *plan, const float *grid,
                         int64_t real_columns, float *yr, void *stream);

/* ---- fast summation and kernel coefficients ----
 * nfft_fastsum of the reference (csrc/cuda/core_cuda.cu:535-852) is adjoint(sources) -> spectral multiply ->
 * forward(targets); its spectral step (spectral_window_operations.cu:269-402: g_hat *= coeffs * phi_hat_inv^2 on
 * the band, 0 elsewhere) is, on the band spectrum that nfft_hip_adjoint returns and nfft_hip_forward consumes,
 * the plain product below.  yhat [B, N^dim, C] complex64 in place; coeffs [N^dim] float32 or complex64. */
int nfft_hip_spectral_multiply(void *yhat, const void *coeffs, int coeffs_are_complex, int64_t batch_size,
                               int64_t band_size, int64_t num_columns, void *stream);

/* One-call fast summation:  y[j, c] = Re?[ sum_k coeffs[k] (sum_i x[i, c] e^{+2 pi i k.s_i}) e^{-2 pi i k.t_j} ].
 * Replaces nfft_fastsum_cuda (csrc/cuda/core_cuda.cu:535-852), i.e. the operator
 * torch_nfft::nfft_fastsum(sources, targets, x, coeffs, source_batch, target_batch, m) (csrc/core.cpp:108-121, 179):
 * spreading of the sources -> inverse FFT -> product with coeffs * phi_hat_inv^2 on the band
 * (spectral_window_operations.cu:269-402; here folded into the last spectral pass of the adjoint half) -> FFT ->
 * interpolation at the targets.
 *   src / tgt   the two point sets: same dim, N, m, batch_size and num_columns, their own num_points
 *   x  [n_s, C] float32 or complex64;  coeffs [N^dim] float32 or complex64 (index l + N/2 on every axis)
 *   y  [n_t, C] float32 when x is real (the real part, core_cuda.cu:817-821), complex64 otherwise
 * When `targets == sources` (same pointer, same batch pointer, same count) one point plan serves both halves, as in
 * the reference (core_cuda.cu:552-564).  The _planned variant takes existing point plans (the same pointer twice
 * for shared points).  Workspace: nfft_hip_fastsum_workspace_bytes(src, tgt, x_is_complex, shared_points, planned). */
int64_t nfft_hip_fastsum_workspace_bytes(const nfft_hip_problem *src, const nfft_hip_problem *tgt, int x_is_complex,
                                         int shared_points, int planned);
int nfft_hip_fastsum(const nfft_hip_problem *src, const float *sources, const int64_t *source_batch,
                     const nfft_hip_problem *tgt, const float *targets, const int64_t *target_batch, const void *x,
                     int x_is_complex, const void *coeffs, int coeffs_are_complex, void *y, void *workspace,
                     int64_t workspace_bytes, void *stream);
/* (The fastsum drivers treat both point sets as NFFT_HIP_POINTS_IN_QUARTER_BALL whatever `flags` says: plans handed
 * to the _planned variant must have been built with that flag set.) */
int nfft_hip_fastsum_planned(const nfft_hip_problem *src, const void *source_plan, const nfft_hip_problem *tgt,
                             const void *target_plan, const void *x, int x_is_complex, const void *coeffs,
                             int coeffs_are_complex, void *y, void *workspace, int64_t workspace_bytes, void *stream);

/* Coefficient set-up (csrc/cuda/kernel_coeffs.cu, drivers core_cuda.cu:855-1064).  Outputs are [N]^dim
 * arrays, index l + N/2 on every axis.
 *   gaussian_analytic_coeffs      float32:  prod_d sqrt(pi) sigma exp(-sigma^2 pi^2 l_d^2)      (kernel_coeffs.cu:6-30)
 *   interpolation_grid            float32 [N^dim, dim] (radial = 0) or [N^dim] norms (radial = 1)  (:76-123)
 *   gaussian_interpolated_coeffs  complex64: fftshift(FFT(ifftshift(K(k/N - 1/2)))) / N^dim, K Gaussian (p < 0) or
 *                                 Gaussian clipped outside radius 1/2 (p == 0); only p <= 0, eps == 0  (:33-73, :179-202)
 *   interpolated_kernel_coeffs    complex64: the same recipe on user samples (float32 or complex64)  (:126-202) */
int nfft_hip_gaussian_analytic_coeffs(double sigma, int64_t N, int32_t dim, float *coeffs, void *stream);
int nfft_hip_interpolation_grid(int64_t N, int32_t dim, int radial, float *grid, void *stream);
int64_t nfft_hip_coeffs_workspace_bytes(int64_t N, int32_t dim);
int nfft_hip_gaussian_interpolated_coeffs(double sigma, int64_t N, int32_t dim, int64_t p, double eps, void *coeffs,
                                          void *workspace, int64_t workspace_bytes, void *stream);
int nfft_hip_interpolated_kernel_coeffs(const void *grid_values, int values_are_complex, int64_t N, int32_t dim,
                                        void *coeffs, void *workspace, int64_t workspace_bytes, void *stream);

/* ---- measurement hooks (bench.py) ----
 * When enabled, nfft_hip_adjoint / nfft_hip_forward bracket each stage with HIP events recorded on the
 * caller's stream.  nfft_hip_profile_collect waits for the recorded events and returns, per stage, the
 * summed GPU time in milliseconds and the number of launches since the last collect.  Stage order:
 * 0 point plan (binning), 1 coefficient gather, 2 grid zero-fill, 3 spreading, 4 FFT, 5 roll-off, 6 interpolation.
 * The reference has no counterpart (it has no timers at all).
 * Two event records per stage cost 3-6 us of stream time each: ~70 us per adjoint + forward pair, which is 1-2 % of a
 * 10^7-point step but more than half of a 10^3-point one.  nfft_hip_profile_stages restricts the timers to the stages
 * whose bit is set (bit s = stage s; default: all), e.g. 1u << 3 times the spreading kernel alone. */
#define NFFT_HIP_NUM_STAGES 7
void nfft_hip_profile_enable(int enable);
void nfft_hip_profile_stages(unsigned stage_mask);
int nfft_hip_profile_collect(double *ms_per_stage, int64_t *launches_per_stage, int num_stages);

#ifdef __cplusplus
}
#endif
#endif /* NFFT_HIP_H */
