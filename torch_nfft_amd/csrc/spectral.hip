// Roll-off correction (deconvolution by the window's Fourier coefficients), fused with the
// band extraction / zero padding and with the Hermitian bookkeeping of the real FFTs.
//
// Reference: csrc/cuda/spectral_window_operations.cu
//   phi_hat_inv[k] = exp(k^2 * pi*m/(3 N^2))                                   (:2-3, :14-43)
//   adjoint: y[b, i, c] = g_hat[(b,c), (i - N/2) mod M] * prod_k phi_hat_inv[|i_k - N/2|]   (:51-153)
//   forward: g_hat[(b,c), (i - N/2) mod M] = x[b, i, c] * prod_k phi_hat_inv[...], 0 elsewhere (:158-265)
// The reference runs a C2C FFT on a complex grid (core_cuda.cu:254-272, 432-450).  Here the grid is a
// set of real planes and the FFTs are R2C / C2R, so:
//   adjoint: g_hat_p[kappa] (e^{+} convention) = conj(F_p[kappa]), F_p = R2C(plane p) stored for
//            kappa_last <= M/2; the other half follows from F_p[kappa] = conj(F_p[-kappa]).
//   forward: Re g and Im g are C2R transforms of the Hermitian / anti-Hermitian parts of g_hat.
#include "common.h"
#include "kernels.h"

namespace nfft {

struct SpecGeom {
    int dim, N, M, Mh;  // Mh = M/2 + 1
    int Na[3];          // band extent per internal axis (1 when degenerate)
    int Ma[3];
    float param;        // pi/3 * m / N^2
    int64_t band;       // N^dim
    int64_t half_cells; // M^(dim-1) * Mh
};

static SpecGeom make_spec_geom(const Geom &g)
{
    SpecGeom s;
    s.dim = g.dim;
    s.N = g.N;
    s.M = g.M;
    s.Mh = g.M / 2 + 1;
    s.band = 1;
    s.half_cells = s.Mh;
    for (int a = 0; a < 3; ++a) {
        s.Ma[a] = g.Ma[a];
        s.Na[a] = g.Ma[a] > 1 ? g.N : 1;
        s.band *= s.Na[a];
        if (a < 2) s.half_cells *= g.Ma[a];
    }
    s.param = 1.047197551196597746f * (float)g.m / ((float)g.N * (float)g.N);
    return s;
}

// (k * k in float: the integer product overflows from |k| = 46341, i.e. for bandwidths of 2^17 and up)
__device__ __forceinline__ float phi_hat_inv(int k, float param) { return expf((float)k * (float)k * param); }

// ---------------------------------------------------------------------------------------------
// adjoint: one thread per (column, band frequency); i2 fastest so that spectrum reads are contiguous.
template <bool XCOMPLEX, bool REAL_OUT>
__global__ void __launch_bounds__(256) deconv_adjoint_kernel(SpecGeom s, const float2 *__restrict__ spec, int64_t C,
                                                            int64_t col0, int64_t ncols, void *__restrict__ yv,
                                                            const void *__restrict__ mult, int mult_kind)
{
    const int64_t total = ncols * s.band;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t col_local = e / s.band;
        int64_t f = e - col_local * s.band;
        const int i2 = (int)(f % s.Na[2]); f /= s.Na[2];
        const int i1 = (int)(f % s.Na[1]); f /= s.Na[1];
        const int i0 = (int)f;
        const int h = s.N / 2;
        float fac = phi_hat_inv(abs(i2 - h), s.param);
        int k2 = i2 >= h ? i2 - h : s.M + i2 - h;
        int k1 = 0, k0 = 0;
        if (s.Ma[1] > 1) { fac *= phi_hat_inv(abs(i1 - h), s.param); k1 = i1 >= h ? i1 - h : s.M + i1 - h; }
        if (s.Ma[0] > 1) { fac *= phi_hat_inv(abs(i0 - h), s.param); k0 = i0 >= h ? i0 - h : s.M + i0 - h; }
        bool conj = true;
        if (k2 > s.M / 2) {  // stored half holds -kappa: g_hat[kappa] = conj(F[kappa]) = F[-kappa]
            k2 = s.M - k2;
            if (s.Ma[1] > 1) k1 = k1 ? s.M - k1 : 0;
            if (s.Ma[0] > 1) k0 = k0 ? s.M - k0 : 0;
            conj = false;
        }
        const int64_t sidx = ((int64_t)k0 * s.Ma[1] + k1) * s.Mh + k2;
        const int64_t colg = col0 + col_local;
        const int64_t b = colg / C, c = colg - b * C;
        const int64_t fidx = ((int64_t)i0 * s.Na[1] + i1) * s.Na[2] + i2;
        const int64_t oidx = (b * s.band + fidx) * C + c;
        float re, im;
        if (XCOMPLEX) {
            const float2 fr = spec[(col_local * 2) * s.half_cells + sidx];
            const float2 fi = spec[(col_local * 2 + 1) * s.half_cells + sidx];
            const float sgn = conj ? -1.0f : 1.0f;
            // (fr.x + i sgn fr.y) + i (fi.x + i sgn fi.y)
            re = fr.x - sgn * fi.y;
            im = sgn * fr.y + fi.x;
        } else {
            const float2 fr = spec[col_local * s.half_cells + sidx];
            re = fr.x;
            im = conj ? -fr.y : fr.y;
        }
        re *= fac;
        im *= fac;
        // fastsum: the kernel's Fourier coefficient of this frequency rides along (mult_kind 1 real, 2 complex)
        if (mult_kind == 1) {
            const float w = ((const float *)mult)[fidx];
            re *= w;
            im *= w;
        } else if (mult_kind == 2) {
            const float2 w = ((const float2 *)mult)[fidx];
            const float r2 = re * w.x - im * w.y;
            im = re * w.y + im * w.x;
            re = r2;
        }
        if (REAL_OUT) {
            ((float *)yv)[oidx] = re;
        } else {
            ((float2 *)yv)[oidx] = make_float2(re, im);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// forward: one thread per element of the Hermitian half-spectrum of every real plane.
//   a[kappa] = xhat[b, kappa + N/2, c] * fac   inside the band, 0 outside
//   Re g = C2R( (a[-kappa] + conj(a[kappa])) / 2 ),   Im g = C2R( (a[-kappa] - conj(a[kappa])) / (2i) )
template <bool XCOMPLEX>
__device__ __forceinline__ float2 load_band(const SpecGeom &s, const void *__restrict__ xhat, int64_t b, int64_t c,
                                            int64_t C, int s0, int s1, int s2)
{
    const int h = s.N / 2;
    // signed frequencies -> band membership
    if (s2 < -h || s2 > h - 1) return make_float2(0.f, 0.f);
    if (s.Ma[1] > 1 && (s1 < -h || s1 > h - 1)) return make_float2(0.f, 0.f);
    if (s.Ma[0] > 1 && (s0 < -h || s0 > h - 1)) return make_float2(0.f, 0.f);
    const int i2 = s2 + h;
    const int i1 = s.Ma[1] > 1 ? s1 + h : 0;
    const int i0 = s.Ma[0] > 1 ? s0 + h : 0;
    const int64_t idx = ((b * s.Na[0] + i0) * s.Na[1] + i1) * s.Na[2] + i2;
    if (XCOMPLEX) return ((const float2 *)xhat)[idx * C + c];
    return make_float2(((const float *)xhat)[idx * C + c], 0.f);
}

template <bool XCOMPLEX>
__global__ void __launch_bounds__(256) deconv_forward_kernel(SpecGeom s, const void *__restrict__ xhat, int64_t C,
                                                            int planes_per_col, int64_t plane0, int64_t nplanes,
                                                            float2 *__restrict__ spec)
{
    const int64_t total = nplanes * s.half_cells;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pl = e / s.half_cells;
        int64_t r = e - pl * s.half_cells;
        const int k2 = (int)(r % s.Mh); r /= s.Mh;
        const int k1 = (int)(r % s.Ma[1]); r /= s.Ma[1];
        const int k0 = (int)r;
        const int64_t plane = plane0 + pl;
        const int64_t colg = plane / planes_per_col;
        const int part = (int)(plane - colg * planes_per_col);
        const int64_t b = colg / C, c = colg - b * C;
        const int half = s.M / 2;
        const int s2 = k2 < half ? k2 : k2 - s.M;
        const int s1 = k1 < half ? k1 : k1 - s.M;
        const int s0 = k0 < half ? k0 : k0 - s.M;
        float fac = phi_hat_inv(abs(s2), s.param);
        if (s.Ma[1] > 1) fac *= phi_hat_inv(abs(s1), s.param);
        if (s.Ma[0] > 1) fac *= phi_hat_inv(abs(s0), s.param);
        const float2 ap = load_band<XCOMPLEX>(s, xhat, b, c, C, s0, s1, s2);     // a[kappa]
        const float2 am = load_band<XCOMPLEX>(s, xhat, b, c, C, -s0, -s1, -s2);  // a[-kappa]
        float2 out;
        if (part == 0) {
            out.x = 0.5f * (am.x + ap.x);
            out.y = 0.5f * (am.y - ap.y);
        } else {
            // (am - conj(ap)) / (2i) = -i/2 * ((am.x - ap.x) + i (am.y + ap.y))
            out.x = 0.5f * (am.y + ap.y);
            out.y = -0.5f * (am.x - ap.x);
        }
        out.x *= fac;
        out.y *= fac;
        spec[e] = out;
    }
}

static inline int grid_for(int64_t work, int block)
{
    int64_t g = (work + block - 1) / block;
    if (g < 1) g = 1;
    if (g > 256 * 64) g = 256 * 64;
    return (int)g;
}

int launch_deconv_adjoint(const Geom &g, const float2 *spec, int64_t C, int x_is_complex, int real_output,
                          int64_t plane0, int64_t nplanes, void *y, const void *mult, int mult_kind, hipStream_t stream)
{
    const SpecGeom s = make_spec_geom(g);
    const int ppc = x_is_complex ? 2 : 1;
    const int64_t col0 = plane0 / ppc, ncols = nplanes / ppc;
    if (ncols <= 0) return 0;
    const dim3 grid(grid_for(ncols * s.band, 256)), block(256);
    if (x_is_complex) {
        if (real_output) hipLaunchKernelGGL((deconv_adjoint_kernel<true, true>), grid, block, 0, stream, s, spec, C, col0, ncols, y, mult, mult_kind);
        else hipLaunchKernelGGL((deconv_adjoint_kernel<true, false>), grid, block, 0, stream, s, spec, C, col0, ncols, y, mult, mult_kind);
    } else {
        if (real_output) hipLaunchKernelGGL((deconv_adjoint_kernel<false, true>), grid, block, 0, stream, s, spec, C, col0, ncols, y, mult, mult_kind);
        else hipLaunchKernelGGL((deconv_adjoint_kernel<false, false>), grid, block, 0, stream, s, spec, C, col0, ncols, y, mult, mult_kind);
    }
    NFFT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_deconv_forward(const Geom &g, const void *xhat, int64_t C, int x_is_complex, int real_output,
                          int64_t plane0, int64_t nplanes, float2 *spec, hipStream_t stream)
{
    const SpecGeom s = make_spec_geom(g);
    const int ppc = real_output ? 1 : 2;
    if (nplanes <= 0) return 0;
    const dim3 grid(grid_for(nplanes * s.half_cells, 256)), block(256);
    if (x_is_complex)
        hipLaunchKernelGGL((deconv_forward_kernel<true>), grid, block, 0, stream, s, xhat, C, ppc, plane0, nplanes, spec);
    else
        hipLaunchKernelGGL((deconv_forward_kernel<false>), grid, block, 0, stream, s, xhat, C, ppc, plane0, nplanes, spec);
    NFFT_HIP_CHECK(hipGetLastError());
    return 0;
}

} // namespace nfft
