/* Exact NDFT in plain C (OpenMP) -- TEST INFRASTRUCTURE ONLY (oracle / CPU baseline).
 *
 * Restates the equations of the reference's ground truth
 * /root/reference/torch_nfft/ndft.py:
 *   ndft_adjoint (ndft.py:5-23):  y[k+N/2, c] = sum_i x[i,c] exp(+2 pi i k.pos[i])
 *   ndft_forward (ndft.py:26-44): y[i, c]     = sum_k x[k+N/2, c] exp(-2 pi i k.pos[i])
 * for ONE point set (the Python wrapper oracle/ndft_cpu.py loops over batches,
 * as ndft.py:22-23, 43-44 does).  k runs over [-N/2, N/2)^d, row-major
 * ("ij" meshgrid, ndft.py:10-11).  All arithmetic in double.
 *
 * Unlike ndft.py (which materialises the N^d x n matrix of exponentials,
 * ndft.py:14,36) the phases are formed per axis, exp(2 pi i k_a pos_a), and
 * multiplied; the sums are the same.
 *
 * Build: oracle/Makefile -> oracle/_build/libndft_oracle.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define CHUNK 256

typedef struct { double re, im; } cplx;

static void fill_phases(const float *pos, int64_t i0, int64_t np, int d, int64_t N, double sign, cplx *E)
{
    /* E[(a*np + p)*N + j] = exp(sign * 2 pi i * (j - N/2) * pos[(i0+p)*d + a]) */
    for (int a = 0; a < d; ++a)
        for (int64_t p = 0; p < np; ++p) {
            const double x = (double)pos[(i0 + p) * d + a];
            cplx *e = E + ((int64_t)a * np + p) * N;
            for (int64_t j = 0; j < N; ++j) {
                const double ang = sign * 2.0 * M_PI * (double)(j - N / 2) * x;
                e[j].re = cos(ang);
                e[j].im = sin(ang);
            }
        }
}

/* x: [n, C] complex interleaved (double pairs); y: [N^d, C] complex interleaved, overwritten. */
int ndft_oracle_adjoint(const float *pos, const double *x, int64_t n, int d, int64_t C, int64_t N,
                        double *y, int nthreads)
{
    if (d < 1 || d > 3 || N < 1 || C < 1) return 1;
    const int64_t N0 = N, N1 = d > 1 ? N : 1, N2 = d > 2 ? N : 1;
    const int64_t total = N0 * N1 * N2;
    memset(y, 0, sizeof(double) * 2 * total * C);
    cplx *E = (cplx *)malloc(sizeof(cplx) * 3 * CHUNK * N);
    if (!E) return 2;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    for (int64_t i0 = 0; i0 < n; i0 += CHUNK) {
        const int64_t np = (n - i0 < CHUNK) ? (n - i0) : CHUNK;
        fill_phases(pos, i0, np, d, N, +1.0, E);
        const cplx *E0 = E, *E1 = E + np * N, *E2 = E + 2 * np * N;
        /* the last axis of the point set is the innermost frequency axis */
        const cplx *Ea = E0, *Eb = (d > 1) ? E1 : NULL, *Ec = (d > 2) ? E2 : NULL;
#pragma omp parallel for collapse(2) schedule(static)
        for (int64_t k0 = 0; k0 < N0; ++k0)
            for (int64_t k1 = 0; k1 < N1; ++k1)
                for (int64_t p = 0; p < np; ++p) {
                    cplx w = Ea[p * N + k0];
                    if (Eb) {
                        const cplx b = Eb[p * N + k1];
                        const cplx t = { w.re * b.re - w.im * b.im, w.re * b.im + w.im * b.re };
                        w = t;
                    }
                    for (int64_t c = 0; c < C; ++c) {
                        const double xr = x[((i0 + p) * C + c) * 2], xi = x[((i0 + p) * C + c) * 2 + 1];
                        const cplx a = { xr * w.re - xi * w.im, xr * w.im + xi * w.re };
                        double *yo = y + ((k0 * N1 + k1) * N2) * C * 2 + c * 2;
                        if (Ec) {
                            const cplx *e2 = Ec + p * N;
                            for (int64_t k2 = 0; k2 < N2; ++k2) {
                                yo[k2 * C * 2] += a.re * e2[k2].re - a.im * e2[k2].im;
                                yo[k2 * C * 2 + 1] += a.re * e2[k2].im + a.im * e2[k2].re;
                            }
                        } else {
                            yo[0] += a.re;
                            yo[1] += a.im;
                        }
                    }
                }
    }
    free(E);
    return 0;
}

/* xhat: [N^d, C] complex interleaved; y: [n, C] complex interleaved, overwritten. */
int ndft_oracle_forward(const float *pos, const double *xhat, int64_t n, int d, int64_t C, int64_t N,
                        double *y, int nthreads)
{
    if (d < 1 || d > 3 || N < 1 || C < 1) return 1;
    const int64_t N0 = N, N1 = d > 1 ? N : 1, N2 = d > 2 ? N : 1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    int fail = 0;
#pragma omp parallel
    {
        cplx *E = (cplx *)malloc(sizeof(cplx) * 3 * N);
        if (!E) {
#pragma omp atomic write
            fail = 1;
        }
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            if (!E) continue;
            fill_phases(pos, i, 1, d, N, -1.0, E);
            const cplx *E0 = E, *E1 = E + N, *E2 = E + 2 * N;
            for (int64_t c = 0; c < C; ++c) {
                double sr = 0.0, si = 0.0;
                for (int64_t k0 = 0; k0 < N0; ++k0) {
                    double s1r = 0.0, s1i = 0.0;
                    for (int64_t k1 = 0; k1 < N1; ++k1) {
                        double s2r = 0.0, s2i = 0.0;
                        const double *xp = xhat + ((k0 * N1 + k1) * N2) * C * 2 + c * 2;
                        if (d > 2) {
                            for (int64_t k2 = 0; k2 < N2; ++k2) {
                                const double xr = xp[k2 * C * 2], xi = xp[k2 * C * 2 + 1];
                                s2r += xr * E2[k2].re - xi * E2[k2].im;
                                s2i += xr * E2[k2].im + xi * E2[k2].re;
                            }
                        } else {
                            s2r = xp[0];
                            s2i = xp[1];
                        }
                        if (d > 1) {
                            s1r += s2r * E1[k1].re - s2i * E1[k1].im;
                            s1i += s2r * E1[k1].im + s2i * E1[k1].re;
                        } else {
                            s1r = s2r;
                            s1i = s2i;
                        }
                    }
                    sr += s1r * E0[k0].re - s1i * E0[k0].im;
                    si += s1r * E0[k0].im + s1i * E0[k0].re;
                }
                y[(i * C + c) * 2] = sr;
                y[(i * C + c) * 2 + 1] = si;
            }
        }
        free(E);
    }
    return fail ? 2 : 0;
}

int ndft_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
