// Transforms whose oversampled grid fits one workgroup's LDS (at most 4096 cells: 1-D N <= 2048, 2-D N <= 32, 3-D N <= 8
// -- the reference's own target regime is N in {16, 32, 64}, torch_nfft/nfft.py:150-156): ONE kernel per direction, no
// point plan.
//
// The general path runs such an adjoint as point plan (5 launches) -> zero-fill -> spreading -> rocFFT -> roll-off and
// the forward transform as roll-off -> rocFFT -> gather: 13 launches per adjoint + forward pair at config C1 (1-D, N = 64,
// 10^3 points), ~5 us of host time each and 2-5 us of GPU time each with nothing to do -- 100 us per pair for 6 000 window
// taps.  Here workgroup (point set b, column c) keeps the whole grid of M^d cells (M = 2N) in LDS:
//   adjoint: zero -> every point of the set adds its (2m+2)^d window taps (ds_add_f64: the 32-bit float LDS atomic is
//            serialised on gfx950, spread.hip) -> radix-2 Stockham FFT in LDS, axis after axis -> roll-off -> y[b, .., c];
//   forward: roll-off of x[b, .., c] into the zero-padded spectrum -> FFT -> every point gathers its taps -> y[i, c].
// The points are read where the caller has them (batch is sorted: a point set is a contiguous range found by bisection),
// complex coefficients are transformed as complex numbers (one complex FFT instead of two real planes).
// Same arithmetic as the general path: window as in spatial_window_operations.cu:1-28, 38-97 (common.h), fp64 sums of
// the taps, roll-off as in spectral_window_operations.cu:2-3, 51-153, 158-265 (spectral.hip), fp32 FFT.
#include "common.h"
#include "kernels.h"

namespace nfft {

constexpr int kSgThreads = 1024;          // (16 waves: the LDS atomics and reads of the tap loops are latency-bound with fewer)
constexpr int kSgMaxCells = 4096;         // grid cells: 16 B of fp64 sums + 8 B of FFT buffer per cell (+ twiddles)
// average window taps per point set: a set is ONE workgroup's loop, ~0.35 ns per tap and direction (LDS atomics of a single
// CU) on top of ~25 us per adjoint + forward pair, against 80-145 us for the general path on these sizes -- measured
// break-even ~10^5 taps (profiles/r03_experiments.md).  8e4 since round 4: the reference's own test shape (test/test_adjoint.py:
// 2-D N = 16, m = 3, 1 000 points per set = 64 000 taps) sat just above the first limit of 6e4 and took the general path
// (0.122 ms against 0.07 here, profiles/r04_experiments.md)
constexpr int64_t kSgMaxSetTaps = 80000;

bool small_grid_supported(const nfft_hip_problem *p)
{
    static const bool off = [] {
        const char *env = std::getenv("NFFT_HIP_SMALL_GRID");
        return env && env[0] == '0';
    }();
    if (off || !p || p->dim < 1 || p->dim > 3) return false;
    const int64_t M = 2 * p->N;
    if (M < 4 || (M & (M - 1)) != 0) return false;
    int64_t cells = 1, taps = 1;
    for (int k = 0; k < p->dim; ++k) { cells *= M; taps *= 2 * p->m + 2; }
    if (cells > kSgMaxCells) return false;
    if (p->m < 1 || p->m > 8 || 2 * p->m + 2 > M) return false;
    if (p->batch_size > 65535) return false;  // (point sets are the y dimension of the launch)
    const int64_t sets = p->batch_size < 1 ? 1 : (p->batch_size > 8 ? 8 : p->batch_size);
    return p->num_points * taps <= kSgMaxSetTaps * sets;
}

namespace {

// LDS (dynamic, cells * 24 + M * 4 + 16 bytes): [ double2 acc[cells] | float2 buf[cells] | float2 twiddle[M / 2] |
// 2 x int64 ]; the FFT ping-pongs between `buf` and the (by then free) accumulator area.
inline size_t sg_lds_bytes(int64_t cells, int64_t M) { return (size_t)(cells * 24 + M * 4 + 16); }

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

// In-LDS FFT along one axis of the cube (length M, a power of two; element stride `stride` = 1, M or M^2; `cells` / M
// lines), radix-2 Stockham: out[.. kappa ..] = sum_j in[.. j ..] exp(sign 2 pi i j kappa / M).  Returns the buffer that holds
// the result.  `tw[k] = exp(-2 pi i k / M)`, k < M / 2.
__device__ __forceinline__ float2 *lds_fft_axis(float2 *a, float2 *b, const float2 *__restrict__ tw, const int M, const int logM,
                                                const int stride, const int cells, const float sign)
{
    const int half = M >> 1;
    float2 *in = a, *out = b;
    for (int ns = 1, shift = 0; ns < M; ns <<= 1, ++shift) {
        const int tstep = half >> shift;  // M / (2 ns)
        // butterfly e = (line, j): cells / 2 of them per pass
        for (int e = threadIdx.x; e < (cells >> 1); e += kSgThreads) {
            const int j = e & (half - 1), line = e >> (logM - 1);
            const int base = (line / stride) * (stride * M) + (line % stride);  // first element of the line
            const int k = j & (ns - 1);
            const float2 w = tw[k * tstep];
            const float wy = -sign * w.y;  // table holds exp(-i ..): sign = +1 conjugates it
            const float2 u = in[base + j * stride], v0 = in[base + (j + half) * stride];
            const float2 v = make_float2(v0.x * w.x - v0.y * wy, v0.x * wy + v0.y * w.x);
            const int o = ((j - k) << 1) + k;
            out[base + o * stride] = make_float2(u.x + v.x, u.y + v.y);
            out[base + (o + ns) * stride] = make_float2(u.x - v.x, u.y - v.y);
        }
        __syncthreads();
        float2 *t = in; in = out; out = t;
    }
    return in;
}

template <int DIM>
__device__ __forceinline__ float2 *lds_fft(float2 *a, float2 *b, const float2 *__restrict__ tw, const int M, const int logM,
                                           const int cells, const float sign)
{
    float2 *cur = a, *other = b;
    int stride = 1;
#pragma unroll
    for (int axis = 0; axis < DIM; ++axis) {
        float2 *res = lds_fft_axis(cur, other, tw, M, logM, stride, cells, sign);
        if (res != cur) { other = cur; cur = res; }
        stride *= M;
    }
    return cur;
}

__device__ __forceinline__ void fill_twiddles(float2 *tw, const int M)
{
    for (int k = threadIdx.x; k < M / 2; k += kSgThreads) {
        float s, c;
        sincospif(-2.0f * (float)k / (float)M, &s, &c);
        tw[k] = make_float2(c, s);
    }
}

__device__ __forceinline__ float rolloff(int k, float param) { return expf((float)k * (float)k * param); }

struct SgLds {
    double *acc;     // [cells][2]
    float2 *buf;     // [cells]
    float2 *tw;      // [M / 2]
    int64_t *range;  // {lo, hi}
};
__device__ __forceinline__ SgLds sg_carve(unsigned char *lds, const int cells, const int M)
{
    SgLds L;
    L.acc = (double *)lds;
    L.buf = (float2 *)(lds + (size_t)cells * 16);
    L.tw = L.buf + cells;
    L.range = (int64_t *)(L.tw + M / 2);
    return L;
}

// thread 0: the set's row range; a batch index outside [0, B) shows at the ends of the sorted vector (the general path
// reports the same fault from its sort)
__device__ __forceinline__ void sg_set_range(const SgLds &L, const int64_t *__restrict__ batch, int64_t n, int64_t B, int64_t b,
                                             int64_t c, int *__restrict__ status)
{
    if (threadIdx.x != 0) return;
    int64_t lo, hi;
    set_range(batch, n, b, lo, hi);
    L.range[0] = lo;
    L.range[1] = hi;
    if (batch && n > 0 && b == 0 && c == 0 && (batch[0] < 0 || batch[n - 1] >= B)) report_fault(status, kFaultBatchIndex);
    // The ranges the bisection finds must tile [0, n): set 0 starts at row 0, set B - 1 ends at row n (a row that no range
    // covers would never be written by the forward kernel).  Together with the per-point check in the kernels (every row
    // of a range carries the range's index) this reports what the general path's sort reports: an index outside [0, B)
    // or out of order ANYWHERE in the vector, not only at its ends.
    if (batch && n > 0 && c == 0 && ((b == 0 && lo != 0) || (b == B - 1 && hi != n))) report_fault(status, kFaultBatchIndex);
}
// row i lies in the range found for set b: its index must be b (c == 0 looks: one load per point and point set)
__device__ __forceinline__ void sg_check_row(const int64_t *__restrict__ batch, int64_t i, int64_t b, int64_t c, int *__restrict__ status)
{
    if (batch && c == 0 && batch[i] != b) report_fault(status, kFaultBatchOrder);
}

// band index i (0 .. N-1 per axis, row-major over the DIM axes) -> signed frequencies, cube slot and roll-off factor
template <int DIM>
__device__ __forceinline__ void band_slot(int f, const int N, const int M, const float param, int &slot, float &fac)
{
    const int h = N / 2;
    slot = 0;
    fac = 1.0f;
    int mul = 1;
#pragma unroll
    for (int axis = DIM - 1; axis >= 0; --axis) {  // last axis fastest
        const int i = f % N;
        f /= N;
        const int kappa = i - h;
        slot += (kappa & (M - 1)) * mul;
        fac *= rolloff(abs(kappa), param);
        mul *= M;
    }
}

// x: [n, C] real or complex; y: [B, N^DIM, C] complex (or real: real_output); grid = (C, B)
template <int DIM>
__global__ void __launch_bounds__(kSgThreads)
small_adjoint_kernel(const int N, const int m, const float *__restrict__ pos, const int64_t *__restrict__ batch,
                     const int64_t n, const int64_t B, const int64_t C, const void *__restrict__ xv, const int x_is_complex,
                     const int real_output, void *__restrict__ yv, const void *__restrict__ mult, const int mult_kind,
                     int *__restrict__ status)
{
    extern __shared__ __align__(16) unsigned char sg_lds[];
    const int M = 2 * N;
    const int logM = 31 - __builtin_clz(M);
    const int cells = DIM == 1 ? M : (DIM == 2 ? M * M : M * M * M);
    const SgLds L = sg_carve(sg_lds, cells, M);
    const int64_t c = blockIdx.x, b = blockIdx.y;
    sg_set_range(L, batch, n, B, b, c, status);
    for (int i = threadIdx.x; i < 2 * cells; i += kSgThreads) L.acc[i] = 0.0;
    fill_twiddles(L.tw, M);
    __syncthreads();
    const int64_t lo = L.range[0], hi = L.range[1];
    const float sc = win_exp_scale(m);
    float norm = win_norm(m);
    norm = DIM == 3 ? norm * norm * norm : (DIM == 2 ? norm * norm : norm);
    const int W = 2 * m + 2;
    // work item = (point, row of its window): the W taps along the last axis; W^(DIM-1) rows per point
    const int rows = DIM == 1 ? 1 : (DIM == 2 ? W : W * W);
    const int64_t items = (hi - lo) * rows;
    for (int64_t e = threadIdx.x; e < items; e += kSgThreads) {
        const int64_t i = lo + e / rows;
        const int row = (int)(e % rows);
        if (row == 0) sg_check_row(batch, i, b, c, status);
        int cell[DIM];
        float frac[DIM];
#pragma unroll
        for (int k = 0; k < DIM; ++k) split_cell(pos[i * DIM + k], M, cell[k], frac[k]);
        float wrow = norm;
        int base = 0;
        if constexpr (DIM >= 2) {
            const int l1 = row % W;  // tap along axis DIM - 2
            const float t1 = frac[DIM - 2] + (float)(m - l1);
            wrow *= __builtin_amdgcn_exp2f(sc * t1 * t1);
            base = ((cell[DIM - 2] - m + l1) & (M - 1)) * M;
        }
        if constexpr (DIM == 3) {
            const int l0 = row / W;
            const float t0 = frac[0] + (float)(m - l0);
            wrow *= __builtin_amdgcn_exp2f(sc * t0 * t0);
            base += ((cell[0] - m + l0) & (M - 1)) * M * M;
        }
        float xr, xi = 0.0f;
        if (x_is_complex) {
            const float2 v = ((const float2 *)xv)[i * C + c];
            xr = v.x * wrow;
            xi = v.y * wrow;
        } else {
            xr = ((const float *)xv)[i * C + c] * wrow;
        }
        for (int l = 0; l < W; ++l) {
            const float t = frac[DIM - 1] + (float)(m - l);
            const float w = __builtin_amdgcn_exp2f(sc * t * t);
            const int slot = base + ((cell[DIM - 1] - m + l) & (M - 1));
            atomicAdd(&L.acc[2 * slot], (double)(w * xr));
            if (x_is_complex) atomicAdd(&L.acc[2 * slot + 1], (double)(w * xi));
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < cells; j += kSgThreads) L.buf[j] = make_float2((float)L.acc[2 * j], (float)L.acc[2 * j + 1]);
    __syncthreads();
    const float2 *res = lds_fft<DIM>(L.buf, (float2 *)L.acc, L.tw, M, logM, cells, +1.0f);
    const float param = 1.047197551196597746f * (float)m / ((float)N * (float)N);
    const int band = DIM == 1 ? N : (DIM == 2 ? N * N : N * N * N);
    for (int f = threadIdx.x; f < band; f += kSgThreads) {
        int slot;
        float fac;
        band_slot<DIM>(f, N, M, param, slot, fac);
        float2 v = res[slot];
        v.x *= fac;
        v.y *= fac;
        if (mult_kind == 1) {
            const float w = ((const float *)mult)[f];
            v.x *= w;
            v.y *= w;
        } else if (mult_kind == 2) {
            const float2 w = ((const float2 *)mult)[f];
            v = make_float2(v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x);
        }
        const int64_t o = (b * band + f) * C + c;
        if (real_output) ((float *)yv)[o] = v.x;
        else ((float2 *)yv)[o] = v;
    }
}

// x: [B, N^DIM, C] real or complex spectrum; y: [n, C] complex (or real: real_output); grid = (C, B)
template <int DIM>
__global__ void __launch_bounds__(kSgThreads)
small_forward_kernel(const int N, const int m, const float *__restrict__ pos, const int64_t *__restrict__ batch,
                     const int64_t n, const int64_t B, const int64_t C, const void *__restrict__ xv, const int x_is_complex,
                     const int real_output, void *__restrict__ yv, int *__restrict__ status)
{
    extern __shared__ __align__(16) unsigned char sg_lds[];
    const int M = 2 * N;
    const int logM = 31 - __builtin_clz(M);
    const int cells = DIM == 1 ? M : (DIM == 2 ? M * M : M * M * M);
    const SgLds L = sg_carve(sg_lds, cells, M);
    float2 *const alt = (float2 *)L.acc;  // second FFT buffer (the adjoint's accumulator area)
    const int64_t c = blockIdx.x, b = blockIdx.y;
    sg_set_range(L, batch, n, B, b, c, status);
    fill_twiddles(L.tw, M);
    const float param = 1.047197551196597746f * (float)m / ((float)N * (float)N);
    const int band = DIM == 1 ? N : (DIM == 2 ? N * N : N * N * N);
    // a[kappa] = x[b, kappa + N/2, c] * roll-off inside the band, 0 outside (spectral.hip)
    for (int j = threadIdx.x; j < cells; j += kSgThreads) L.buf[j] = make_float2(0.f, 0.f);
    __syncthreads();
    for (int f = threadIdx.x; f < band; f += kSgThreads) {
        int slot;
        float fac;
        band_slot<DIM>(f, N, M, param, slot, fac);
        const int64_t idx = (b * band + f) * C + c;
        float2 v = make_float2(0.f, 0.f);
        if (x_is_complex) v = ((const float2 *)xv)[idx];
        else v.x = ((const float *)xv)[idx];
        L.buf[slot] = make_float2(v.x * fac, v.y * fac);
    }
    __syncthreads();
    const int64_t lo = L.range[0], hi = L.range[1];
    const float2 *g = lds_fft<DIM>(L.buf, alt, L.tw, M, logM, cells, -1.0f);  // g[j] = sum_kappa a[kappa] exp(-2 pi i j.kappa / M)
    const float sc = win_exp_scale(m);
    float norm = win_norm(m);
    norm = DIM == 3 ? norm * norm * norm : (DIM == 2 ? norm * norm : norm);
    const int W = 2 * m + 2;
    for (int64_t i = lo + threadIdx.x; i < hi; i += kSgThreads) {
        sg_check_row(batch, i, b, c, status);
        int cell[DIM];
        float frac[DIM];
#pragma unroll
        for (int k = 0; k < DIM; ++k) split_cell(pos[i * DIM + k], M, cell[k], frac[k]);
        float re = 0.0f, im = 0.0f;
        const int n0 = DIM == 3 ? W : 1, n1 = DIM >= 2 ? W : 1;
        for (int l0 = 0; l0 < n0; ++l0) {
            float w0 = 1.0f;
            int base0 = 0;
            if constexpr (DIM == 3) {
                const float t0 = frac[0] + (float)(m - l0);
                w0 = __builtin_amdgcn_exp2f(sc * t0 * t0);
                base0 = ((cell[0] - m + l0) & (M - 1)) * M * M;
            }
            for (int l1 = 0; l1 < n1; ++l1) {
                float w1 = w0;
                int base = base0;
                if constexpr (DIM >= 2) {
                    const float t1 = frac[DIM - 2] + (float)(m - l1);
                    w1 *= __builtin_amdgcn_exp2f(sc * t1 * t1);
                    base += ((cell[DIM - 2] - m + l1) & (M - 1)) * M;
                }
                float rr = 0.0f, ri = 0.0f;
                for (int l = 0; l < W; ++l) {
                    const float t = frac[DIM - 1] + (float)(m - l);
                    const float w = __builtin_amdgcn_exp2f(sc * t * t);
                    const float2 v = g[base + ((cell[DIM - 1] - m + l) & (M - 1))];
                    rr = fmaf(w, v.x, rr);
                    ri = fmaf(w, v.y, ri);
                }
                re = fmaf(w1, rr, re);
                im = fmaf(w1, ri, im);
            }
        }
        if (real_output) ((float *)yv)[i * C + c] = re * norm;
        else ((float2 *)yv)[i * C + c] = make_float2(re * norm, im * norm);
    }
}

template <typename K>
int prepare(K kernel, int slot)
{
    static DeviceOnce attr_done[6];
    DeviceOnce &once = attr_done[slot];
    if (once.first_use()) {
        NFFT_HIP_CHECK(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)sg_lds_bytes(kSgMaxCells, kSgMaxCells)));
        once.mark();
    }
    return 0;
}

} // namespace

int launch_small_grid_adjoint(const nfft_hip_problem *p, const float *pos, const int64_t *batch, const void *x, int x_is_complex,
                              int real_output, void *y, const void *mult, int mult_kind, hipStream_t stream)
{
    const int64_t B = p->batch_size, C = p->num_columns;
    if (B * C <= 0) return 0;
    const int64_t M = 2 * p->N;
    int64_t cells = 1;
    for (int k = 0; k < p->dim; ++k) cells *= M;
    const dim3 grid((unsigned)C, (unsigned)B);
    const size_t lds = sg_lds_bytes(cells, M);
    auto go = [&](auto kernel, int slot) -> int {
        if (int rc = prepare(kernel, slot)) return rc;
        hipLaunchKernelGGL(kernel, grid, dim3(kSgThreads), lds, stream, (int)p->N, (int)p->m, pos, batch, p->num_points, B, C,
                           x, x_is_complex, real_output, y, mult, mult_kind, device_status_block());
        return 0;
    };
    int rc = 1;
    switch (p->dim) {
    case 1: rc = go(small_adjoint_kernel<1>, 0); break;
    case 2: rc = go(small_adjoint_kernel<2>, 1); break;
    case 3: rc = go(small_adjoint_kernel<3>, 2); break;
    }
    if (rc) return rc;
    NFFT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_small_grid_forward(const nfft_hip_problem *p, const float *pos, const int64_t *batch, const void *xhat,
                              int x_is_complex, int real_output, void *y, hipStream_t stream)
{
    const int64_t B = p->batch_size, C = p->num_columns;
    if (B * C <= 0 || p->num_points <= 0) return 0;
    const int64_t M = 2 * p->N;
    int64_t cells = 1;
    for (int k = 0; k < p->dim; ++k) cells *= M;
    const dim3 grid((unsigned)C, (unsigned)B);
    const size_t lds = sg_lds_bytes(cells, M);
    auto go = [&](auto kernel, int slot) -> int {
        if (int rc = prepare(kernel, slot)) return rc;
        hipLaunchKernelGGL(kernel, grid, dim3(kSgThreads), lds, stream, (int)p->N, (int)p->m, pos, batch, p->num_points, B, C,
                           xhat, x_is_complex, real_output, y, device_status_block());
        return 0;
    };
    int rc = 1;
    switch (p->dim) {
    case 1: rc = go(small_forward_kernel<1>, 3); break;
    case 2: rc = go(small_forward_kernel<2>, 4); break;
    case 3: rc = go(small_forward_kernel<3>, 5); break;
    }
    if (rc) return rc;
    NFFT_HIP_CHECK(hipGetLastError());
    return 0;
}

} // namespace nfft
