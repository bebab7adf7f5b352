// Trigonometric-kernel coefficients and the fastsum spectral multiply.
//
// Reference kernels restated (behaviour, not code): csrc/cuda/kernel_coeffs.cu
//   gaussian_analytic_coeffs        :6-30    b_l = prod_d sqrt(pi) sigma exp(-sigma^2 pi^2 l_d^2),  l in [-N/2, N/2)^d
//   gaussian_interpolated_coeffs    :33-73 + FFT + :179-202   b = fftshift(FFT(ifftshift(K(k/N - 1/2)))) / N^d with
//                                   K(r) = exp(-r^2/sigma^2) (p < 0), or the same clipped to exp(-1/(4 sigma^2)) for
//                                   r > 1/2 (p == 0, eps == 0; other p / eps are rejected, core_cuda.cu:890-891)
//   interpolation_grid              :76-97   grid[k, c] = k_c / N - 1/2
//   radial_interpolation_grid       :99-123  |grid[k]|
//   interpolated_kernel_coeffs      :126-202 the same FFT recipe applied to user samples (real or complex)
// and the spectral step of nfft_fastsum (csrc/cuda/spectral_window_operations.cu:269-402):
//   g_hat *= coeffs * phi_hat_inv^2 on the N^d band, 0 elsewhere.  Here the band spectrum is what the adjoint
//   transform returns (already carrying one phi_hat_inv) and what the forward transform consumes (it applies
//   the second), so the step is the plain product yhat[b, k, c] *= coeffs[k].
#include "../../include/nfft_hip.h"

#include "common.h"
#include "kernels.h"

namespace nfft {
namespace {

__device__ __forceinline__ void unravel(int64_t idx, int N, int dim, int k[3])
{
    k[0] = k[1] = k[2] = 0;
    for (int a = dim - 1; a >= 0; --a) {
        k[a] = (int)(idx % N);
        idx /= N;
    }
}

__global__ void analytic_kernel(float *__restrict__ out, float sigma, int N, int dim, int64_t total)
{
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int k[3];
        unravel(idx, N, dim, k);
        float v = 1.0f;
        for (int a = 0; a < dim; ++a) {
            const float l = (float)(k[a] - N / 2);
            v *= 1.77245385090551602729f * sigma * expf(-sigma * sigma * 9.86960440108935861883f * l * l);
        }
        out[idx] = v;
    }
}

__global__ void grid_kernel(float *__restrict__ out, int N, int dim, int radial, int64_t total)
{
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int k[3];
        unravel(idx, N, dim, k);
        float r2 = 0.0f;
        for (int a = 0; a < dim; ++a) {
            const float c = (float)k[a] / (float)N - 0.5f;
            r2 += c * c;
            if (!radial) out[idx * dim + a] = c;
        }
        if (radial) out[idx] = sqrtf(r2);
    }
}

// b[ifftshift(k)] = value(k): Gaussian samples (mode 0: p < 0, mode 1: p == 0) or user samples (mode 2/3: real/complex)
__global__ void fill_shifted_kernel(float2 *__restrict__ b, const void *__restrict__ values, int mode, float sigma2,
                                    int N, int dim, int64_t total)
{
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int k[3];
        unravel(idx, N, dim, k);
        int64_t bidx = 0;
        float r2 = 0.0f;
        for (int a = 0; a < dim; ++a) {
            bidx = bidx * N + (k[a] + N / 2) % N;
            const float c = (float)k[a] / (float)N - 0.5f;
            r2 += c * c;
        }
        float2 v;
        if (mode == 0) v = make_float2(expf(-r2 / sigma2), 0.f);
        else if (mode == 1) v = make_float2(r2 <= 0.25f ? expf(-r2 / sigma2) : expf(-0.25f / sigma2), 0.f);
        else if (mode == 2) v = make_float2(((const float *)values)[idx], 0.f);
        else v = ((const float2 *)values)[idx];
        b[bidx] = v;
    }
}

// coeffs[k] = b[ifftshift(k)] / N^d
__global__ void unshift_scale_kernel(float2 *__restrict__ out, const float2 *__restrict__ b, int N, int dim, int64_t total)
{
    const float inv = 1.0f / (float)total;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int k[3];
        unravel(idx, N, dim, k);
        int64_t bidx = 0;
        for (int a = 0; a < dim; ++a) bidx = bidx * N + (k[a] + N / 2) % N;
        out[idx] = make_float2(b[bidx].x * inv, b[bidx].y * inv);
    }
}

template <bool CCOMPLEX>
__global__ void spectral_multiply_kernel(float2 *__restrict__ yhat, const void *__restrict__ coeffs, int64_t band,
                                         int64_t C, int64_t total)
{
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t f = (e / C) % band;
        const float2 v = yhat[e];
        if (CCOMPLEX) {
            const float2 c = ((const float2 *)coeffs)[f];
            yhat[e] = make_float2(v.x * c.x - v.y * c.y, v.x * c.y + v.y * c.x);
        } else {
            const float c = ((const float *)coeffs)[f];
            yhat[e] = make_float2(v.x * c, v.y * c);
        }
    }
}

int blocks_for(int64_t n)
{
    int64_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

int check_nd(int64_t N, int dim)
{
    if (dim < 1 || dim > 3 || N < 2 || N > (1 << 20)) { set_error("Input mismatch"); return NFFT_HIP_EINVAL; }
    return 0;
}

int64_t ipow(int64_t N, int dim) { int64_t r = 1; for (int a = 0; a < dim; ++a) r *= N; return r; }

int fft_coeffs(const void *values, int mode, float sigma2, int64_t N, int dim, float2 *out, void *workspace,
               int64_t workspace_bytes, hipStream_t s)
{
    const int64_t total = ipow(N, dim);
    const int64_t wb = fft_work_bytes(kC2CForward, dim, (int)N, 1);
    if (wb < 0) return NFFT_HIP_EFFT;
    const int64_t need = align_up(total * 8, 256) + wb;
    if (!workspace || workspace_bytes < need) { set_error("workspace too small"); return NFFT_HIP_EWORKSPACE; }
    float2 *b = (float2 *)workspace;
    void *work = (char *)workspace + align_up(total * 8, 256);
    hipLaunchKernelGGL(fill_shifted_kernel, dim3(blocks_for(total)), dim3(256), 0, s, b, values, mode, sigma2, (int)N, dim, total);
    if (int rc = fft_execute(kC2CForward, dim, (int)N, 1, b, b, work, wb, s)) return rc;
    hipLaunchKernelGGL(unshift_scale_kernel, dim3(blocks_for(total)), dim3(256), 0, s, out, b, (int)N, dim, total);
    NFFT_HIP_CHECK(hipGetLastError());
    return 0;
}

} // namespace
} // namespace nfft

using namespace nfft;

extern "C" {

int nfft_hip_gaussian_analytic_coeffs(double sigma, int64_t N, int32_t dim, float *coeffs, void *stream)
{
    if (int rc = check_nd(N, dim)) return rc;
    const int64_t total = ipow(N, dim);
    hipLaunchKernelGGL(analytic_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, coeffs, (float)sigma, (int)N, dim, total);
    NFFT_HIP_CHECK(hipGetLastError());
    return 0;
}

int nfft_hip_interpolation_grid(int64_t N, int32_t dim, int radial, float *grid, void *stream)
{
    if (int rc = check_nd(N, dim)) return rc;
    const int64_t total = ipow(N, dim);
    hipLaunchKernelGGL(grid_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, grid, (int)N, dim, radial, total);
    NFFT_HIP_CHECK(hipGetLastError());
    return 0;
}

int64_t nfft_hip_coeffs_workspace_bytes(int64_t N, int32_t dim)
{
    if (check_nd(N, dim)) return -1;
    const int64_t wb = fft_work_bytes(kC2CForward, dim, (int)N, 1);
    if (wb < 0) return -1;
    return align_up(ipow(N, dim) * 8, 256) + wb + 256;
}

int nfft_hip_gaussian_interpolated_coeffs(double sigma, int64_t N, int32_t dim, int64_t p, double eps, void *coeffs,
                                          void *workspace, int64_t workspace_bytes, void *stream)
{
    if (int rc = check_nd(N, dim)) return rc;
    if (p > 0) { set_error("Gaussian interpolated coeffs are currently only implemented for p<=0"); return NFFT_HIP_EINVAL; }
    if (eps != 0.0) { set_error("Gaussian interpolated coeffs are currently only implemented for eps=0"); return NFFT_HIP_EINVAL; }
    return fft_coeffs(nullptr, p < 0 ? 0 : 1, (float)(sigma * sigma), N, dim, (float2 *)coeffs, workspace, workspace_bytes,
                      (hipStream_t)stream);
}

int nfft_hip_interpolated_kernel_coeffs(const void *grid_values, int values_are_complex, int64_t N, int32_t dim,
                                        void *coeffs, void *workspace, int64_t workspace_bytes, void *stream)
{
    if (int rc = check_nd(N, dim)) return rc;
    if (!grid_values) { set_error("Input mismatch: null input"); return NFFT_HIP_EINVAL; }
    return fft_coeffs(grid_values, values_are_complex ? 3 : 2, 1.0f, N, dim, (float2 *)coeffs, workspace, workspace_bytes,
                      (hipStream_t)stream);
}

int nfft_hip_spectral_multiply(void *yhat, const void *coeffs, int coeffs_are_complex, int64_t batch_size,
                               int64_t band_size, int64_t num_columns, void *stream)
{
    const int64_t total = batch_size * band_size * num_columns;
    if (total <= 0) return 0;
    if (!yhat || !coeffs) { set_error("Input mismatch: null input"); return NFFT_HIP_EINVAL; }
    if (coeffs_are_complex)
        hipLaunchKernelGGL((spectral_multiply_kernel<true>), dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, (float2 *)yhat, coeffs, band_size, num_columns, total);
    else
        hipLaunchKernelGGL((spectral_multiply_kernel<false>), dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, (float2 *)yhat, coeffs, band_size, num_columns, total);
    NFFT_HIP_CHECK(hipGetLastError());
    return 0;
}

} // extern "C"
