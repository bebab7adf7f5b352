// 1-D transforms whose oversampled grid fits one workgroup's LDS: ONE kernel per direction, no point plan.
//
// The general path runs a 1-D adjoint as point plan (5 launches) -> zero-fill -> spreading -> rocFFT -> roll-off and
// the forward transform as roll-off -> rocFFT -> gather: 13 launches per adjoint + forward pair at config C1 (N = 64,
// 10^3 points), ~5 us of host time each and 2-5 us of GPU time each with nothing to do -- 100 us per pair for 6 000 window
// taps.  Here workgroup (point set b, column c) keeps the whole grid of M = 2N cells in LDS:
//   adjoint: zero -> every point of the set adds its 2m+2 window taps (ds_add_f64: the 32-bit float LDS atomic is
//            serialised on gfx950, spread.hip) -> radix-2 Stockham FFT in LDS -> roll-off -> y[b, :, c];
//   forward: roll-off of x[b, :, c] into the zero-padded spectrum -> FFT -> every point gathers its taps -> y[i, c].
// The points are read where the caller has them (batch is sorted: a point set is a contiguous range found by bisection),
// complex coefficients are transformed as complex numbers (one complex FFT instead of two real planes).
// Same arithmetic as the general path: window as in spatial_window_operations.cu:1-28, 38-97 (common.h), fp64 sums of
// the taps, roll-off as in spectral_window_operations.cu:2-3, 51-153, 158-265 (spectral.hip), fp32 FFT.
#include "common.h"
#include "kernels.h"

namespace nfft {

constexpr int kS1Threads = 256;
constexpr int kS1MaxM = 4096;            // grid cells: 16 B of fp64 sums + 8 B of FFT buffer + 4 B of twiddles per cell
constexpr int64_t kS1MaxSetPoints = 32768;  // average points per point set (a set is one workgroup's serial loop)

bool small1d_supported(const nfft_hip_problem *p)
{
    static const bool off = [] {
        const char *env = std::getenv("NFFT_HIP_SMALL1D");
        return env && env[0] == '0';
    }();
    if (off || !p || p->dim != 1) return false;
    const int64_t M = 2 * p->N;
    if (M < 4 || M > kS1MaxM || (M & (M - 1)) != 0) return false;
    if (p->m < 1 || p->m > 8) return false;
    if (p->batch_size > 65535) return false;  // (point sets are the y dimension of the launch)
    const int64_t sets = p->batch_size < 1 ? 1 : (p->batch_size > 8 ? 8 : p->batch_size);
    return p->num_points <= kS1MaxSetPoints * sets;
}

namespace {

// LDS (dynamic, M * 28 + 16 bytes): [ double2 acc[M] | float2 buf[M] | float2 twiddle[M / 2] | 2 x int64 ]; the FFT
// ping-pongs between `buf` and the (by then free) accumulator area.

// rows [lo, hi) of point set b in the sorted batch vector (nullptr: one set)
__device__ __forceinline__ void set_range(const int64_t *__restrict__ batch, int64_t n, int64_t b, int64_t &lo, int64_t &hi)
{
    if (!batch) { lo = 0; hi = n; return; }
    auto lower = [&](int64_t key) {  // first row with batch[row] >= key
        int64_t a = 0, z = n;
        while (a < z) {
            const int64_t mid = (a + z) >> 1;
            if (batch[mid] >= key) z = mid; else a = mid + 1;
        }
        return a;
    };
    lo = lower(b);
    hi = lower(b + 1);
}

// In-LDS FFT of length M (power of two), radix-2 Stockham: out[kappa] = sum_j in[j] exp(sign 2 pi i j kappa / M).
// Returns the buffer that holds the result.  `tw[k] = exp(-2 pi i k / M)`, k < M / 2.
__device__ __forceinline__ float2 *lds_fft(float2 *a, float2 *b, const float2 *__restrict__ tw, const int M, const float sign)
{
    const int half = M >> 1;
    float2 *in = a, *out = b;
    for (int ns = 1, shift = 0; ns < M; ns <<= 1, ++shift) {
        const int tstep = half >> shift;  // M / (2 ns)
        for (int j = threadIdx.x; j < half; j += kS1Threads) {
            const int k = j & (ns - 1);
            const float2 w = tw[k * tstep];
            const float wy = -sign * w.y;  // table holds exp(-i ..): sign = +1 conjugates it
            const float2 u = in[j], v0 = in[j + half];
            const float2 v = make_float2(v0.x * w.x - v0.y * wy, v0.x * wy + v0.y * w.x);
            const int o = ((j - k) << 1) + k;
            out[o] = make_float2(u.x + v.x, u.y + v.y);
            out[o + ns] = make_float2(u.x - v.x, u.y - v.y);
        }
        __syncthreads();
        float2 *t = in; in = out; out = t;
    }
    return in;
}

__device__ __forceinline__ void fill_twiddles(float2 *tw, const int M)
{
    for (int k = threadIdx.x; k < M / 2; k += kS1Threads) {
        float s, c;
        sincospif(-2.0f * (float)k / (float)M, &s, &c);
        tw[k] = make_float2(c, s);
    }
}

__device__ __forceinline__ float rolloff(int k, float param) { return expf((float)k * (float)k * param); }

// x: [n, C] real or complex; y: [B, N, C] complex (or real: real_output); grid = (C, B)
__global__ void __launch_bounds__(kS1Threads)
small1d_adjoint_kernel(const int N, const int m, const float *__restrict__ pos, const int64_t *__restrict__ batch,
                       const int64_t n, const int64_t B, const int64_t C, const void *__restrict__ xv, const int x_is_complex,
                       const int real_output, void *__restrict__ yv, const void *__restrict__ mult, const int mult_kind,
                       int *__restrict__ status)
{
    extern __shared__ __align__(16) unsigned char s1_lds[];
    const int M = 2 * N;
    double *acc = (double *)s1_lds;                       // [M][2]
    float2 *buf = (float2 *)(s1_lds + (size_t)M * 16);    // [M]
    float2 *tw = buf + M;                                 // [M / 2]
    int64_t *range = (int64_t *)(tw + M / 2);             // {lo, hi}
    const int64_t c = blockIdx.x, b = blockIdx.y;
    if (threadIdx.x == 0) {
        int64_t lo, hi;
        set_range(batch, n, b, lo, hi);
        range[0] = lo;
        range[1] = hi;
        // (sorted: an index outside [0, B) shows at the ends; the general path reports the same fault from its sort)
        if (batch && n > 0 && b == 0 && c == 0 && (batch[0] < 0 || batch[n - 1] >= B)) report_fault(status, kFaultBatchIndex);
    }
    for (int i = threadIdx.x; i < 2 * M; i += kS1Threads) acc[i] = 0.0;
    fill_twiddles(tw, M);
    __syncthreads();
    const int64_t lo = range[0], hi = range[1];
    const float sc = win_exp_scale(m), norm = win_norm(m);
    const int W = 2 * m + 2;
    for (int64_t i = lo + threadIdx.x; i < hi; i += kS1Threads) {
        int cell;
        float frac;
        split_cell(pos[i], M, cell, frac);
        float xr, xi = 0.0f;
        if (x_is_complex) {
            const float2 v = ((const float2 *)xv)[i * C + c];
            xr = v.x * norm;
            xi = v.y * norm;
        } else {
            xr = ((const float *)xv)[i * C + c] * norm;
        }
        for (int l = 0; l < W; ++l) {
            const float t = frac + (float)(m - l);
            const float w = __builtin_amdgcn_exp2f(sc * t * t);
            const int col = (cell - m + l) & (M - 1);
            atomicAdd(&acc[2 * col], (double)(w * xr));
            if (x_is_complex) atomicAdd(&acc[2 * col + 1], (double)(w * xi));
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < M; j += kS1Threads) buf[j] = make_float2((float)acc[2 * j], (float)acc[2 * j + 1]);
    __syncthreads();
    const float2 *res = lds_fft(buf, (float2 *)acc, tw, M, +1.0f);
    const float param = 1.047197551196597746f * (float)m / ((float)N * (float)N);
    const int h = N / 2;
    for (int i = threadIdx.x; i < N; i += kS1Threads) {
        const int kappa = i - h;
        float2 v = res[kappa & (M - 1)];
        const float fac = rolloff(abs(kappa), param);
        v.x *= fac;
        v.y *= fac;
        if (mult_kind == 1) {
            const float w = ((const float *)mult)[i];
            v.x *= w;
            v.y *= w;
        } else if (mult_kind == 2) {
            const float2 w = ((const float2 *)mult)[i];
            v = make_float2(v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x);
        }
        const int64_t o = (b * N + i) * C + c;
        if (real_output) ((float *)yv)[o] = v.x;
        else ((float2 *)yv)[o] = v;
    }
}

// x: [B, N, C] real or complex spectrum; y: [n, C] complex (or real: real_output); grid = (C, B)
__global__ void __launch_bounds__(kS1Threads)
small1d_forward_kernel(const int N, const int m, const float *__restrict__ pos, const int64_t *__restrict__ batch,
                       const int64_t n, const int64_t B, const int64_t C, const void *__restrict__ xv, const int x_is_complex,
                       const int real_output, void *__restrict__ yv, int *__restrict__ status)
{
    extern __shared__ __align__(16) unsigned char s1_lds[];
    const int M = 2 * N;
    float2 *alt = (float2 *)s1_lds;                       // second FFT buffer (the adjoint's accumulator area)
    float2 *buf = (float2 *)(s1_lds + (size_t)M * 16);
    float2 *tw = buf + M;
    int64_t *range = (int64_t *)(tw + M / 2);
    const int64_t c = blockIdx.x, b = blockIdx.y;
    if (threadIdx.x == 0) {
        int64_t lo, hi;
        set_range(batch, n, b, lo, hi);
        range[0] = lo;
        range[1] = hi;
        if (batch && n > 0 && b == 0 && c == 0 && (batch[0] < 0 || batch[n - 1] >= B)) report_fault(status, kFaultBatchIndex);
    }
    fill_twiddles(tw, M);
    const float param = 1.047197551196597746f * (float)m / ((float)N * (float)N);
    const int h = N / 2;
    // a[kappa] = x[b, kappa + N/2, c] * roll-off inside the band, 0 outside (spectral.hip)
    for (int j = threadIdx.x; j < M; j += kS1Threads) {
        const int kappa = j < N ? j : j - M;  // signed frequency of slot j
        float2 v = make_float2(0.f, 0.f);
        if (kappa >= -h && kappa <= h - 1) {
            const int64_t idx = (b * N + (kappa + h)) * C + c;
            if (x_is_complex) v = ((const float2 *)xv)[idx];
            else v.x = ((const float *)xv)[idx];
            const float fac = rolloff(abs(kappa), param);
            v.x *= fac;
            v.y *= fac;
        }
        buf[j] = v;
    }
    __syncthreads();
    const int64_t lo = range[0], hi = range[1];
    const float2 *g = lds_fft(buf, alt, tw, M, -1.0f);  // g[j] = sum_kappa a[kappa] exp(-2 pi i j kappa / M)
    const float sc = win_exp_scale(m), norm = win_norm(m);
    const int W = 2 * m + 2;
    for (int64_t i = lo + threadIdx.x; i < hi; i += kS1Threads) {
        int cell;
        float frac;
        split_cell(pos[i], M, cell, frac);
        float re = 0.0f, im = 0.0f;
        for (int l = 0; l < W; ++l) {
            const float t = frac + (float)(m - l);
            const float w = __builtin_amdgcn_exp2f(sc * t * t);
            const float2 v = g[(cell - m + l) & (M - 1)];
            re = fmaf(w, v.x, re);
            im = fmaf(w, v.y, im);
        }
        if (real_output) ((float *)yv)[i * C + c] = re * norm;
        else ((float2 *)yv)[i * C + c] = make_float2(re * norm, im * norm);
    }
}

int prepare(const void *kernel)
{
    static DeviceOnce attr_done[2];
    DeviceOnce &once = attr_done[kernel == (const void *)small1d_adjoint_kernel ? 0 : 1];
    if (once.first_use()) {
        NFFT_HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)((size_t)kS1MaxM * 28 + 16)));
        once.mark();
    }
    return 0;
}

} // namespace

int launch_small1d_adjoint(const nfft_hip_problem *p, const float *pos, const int64_t *batch, const void *x, int x_is_complex,
                           int real_output, void *y, const void *mult, int mult_kind, hipStream_t stream)
{
    const int64_t B = p->batch_size, C = p->num_columns;
    if (B * C <= 0) return 0;
    if (int rc = prepare((const void *)small1d_adjoint_kernel)) return rc;
    const int M = (int)(2 * p->N);
    hipLaunchKernelGGL(small1d_adjoint_kernel, dim3((unsigned)C, (unsigned)B), dim3(kS1Threads), (size_t)M * 28 + 16, stream,
                       (int)p->N, (int)p->m, pos, batch, p->num_points, B, C, x, x_is_complex, real_output, y, mult, mult_kind,
                       device_status_block());
    NFFT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_small1d_forward(const nfft_hip_problem *p, const float *pos, const int64_t *batch, const void *xhat, int x_is_complex,
                           int real_output, void *y, hipStream_t stream)
{
    const int64_t B = p->batch_size, C = p->num_columns;
    if (B * C <= 0 || p->num_points <= 0) return 0;
    if (int rc = prepare((const void *)small1d_forward_kernel)) return rc;
    const int M = (int)(2 * p->N);
    hipLaunchKernelGGL(small1d_forward_kernel, dim3((unsigned)C, (unsigned)B), dim3(kS1Threads), (size_t)M * 28 + 16, stream,
                       (int)p->N, (int)p->m, pos, batch, p->num_points, B, C, xhat, x_is_complex, real_output, y,
                       device_status_block());
    NFFT_HIP_CHECK(hipGetLastError());
    return 0;
}

} // namespace nfft
