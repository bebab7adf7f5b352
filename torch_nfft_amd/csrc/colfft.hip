// Pruned column FFT passes for the 3-D transforms (power-of-two M), fused with the roll-off.
//
// The reference transforms the whole (2N)^3 grid with cuFFT (csrc/cuda/core_cuda.cu:254-272, 432-450) and
// then picks the N^3 band / zero-pads it (spectral_window_operations.cu:51-265).  Only N of the 2N
// frequencies per axis are ever used, so after the contiguous axis-2 pass (rocFFT, real <-> half-complex
// rows) the two strided passes run here on the band only:
//
//   adjoint:  S[u0][u1][k2]  --axis 1-->  T[u0][k1 in band+][k2 <= N/2]  --axis 0 + roll-off-->  y
//   forward:  xhat --roll-off + Hermitian split + axis 0-->  T  --axis 1-->  S (zero for k2 > N/2)
//
// which cuts their HBM traffic to a quarter / an eighth of the full passes and removes the separate
// roll-off kernels (adjoint) and the zero-fill of the padded spectrum (forward).  One workgroup transforms
// a tile of NC adjacent k2-columns entirely in LDS (radix-8/4 DIF, digit-reversed output addressing), so
// every global access is a run of NC * 8 contiguous bytes.
//
// "band+" is k in [-N/2, N/2]: the extra +N/2 row feeds the Hermitian mirror g_hat[-k] = conj(g_hat[k])
// that reconstructs the k2 < 0 half of the spectrum from the stored k2 >= 0 half.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace nfft {

namespace {

constexpr int kFftThreads = 256;

struct ColGeom {
    int M, logM, N, H;  // H = N/2
    int Mh;             // M/2 + 1
    int SR;             // row stride of the axis-2 half spectrum S: Mh (rocFFT rows) or KC (own pruned row passes)
    int KC;             // kept k2 columns: 0..H
    int KS;             // row stride of T and of the compact S: KC rounded up to whole tiles of NC columns, so that the
                        // NC * 8-byte runs the passes read and write start on their own 64 / 128-byte lines (rows of
                        // 129 complex numbers put every run across two lines: the passes fetched ~2x their input)
    int NB;             // band+ rows: N + 1
    int NBm;            // rows of the MIDDLE axis in T: NB, or 1 for 2-D problems (no middle axis: T is the compact S itself, the
                        // "axis 0" pass is the only column pass and carries the roll-off)
    int two_d;
    int NC, logNC;      // columns per tile
    int TG;             // tiles that share a 128-byte line (16 / NC; 1: the XCD-aware tile mapping below is off)
    float param;        // pi/3 * m / N^2  (phi_hat_inv exponent scale)
};

__device__ __forceinline__ float phi_hat_inv_f(int k, float param) { return expf((float)k * (float)k * param); }

__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// Column passes, workgroup -> (column tile, row, plane).  A tile of NC < 16 columns is an NC * 8-byte run of every row: TG =
// 16 / NC neighbouring tiles share each 128-byte line, and consecutive workgroup ids go to different XCDs (round robin),
// each with its own L2 -- so every line was fetched from HBM once per tile (FETCH_SIZE 2.05x the input at C3).  Here the
// TG tiles of a line get ids that are 8 apart: same XCD, dispatched together, ONE fetch.  Ids are permuted inside blocks
// of 8 * TG consecutive ones (the launch rounds the tile count up to a multiple of TG; the spare tiles return at once).
__device__ __forceinline__ void column_tile_of_block(const int tg, int &tile, int &row, int64_t &plane)
{
    const unsigned gx = gridDim.x, gy = gridDim.y;
    unsigned long long id = blockIdx.x + (unsigned long long)gx * (blockIdx.y + (unsigned long long)gy * blockIdx.z);
    if (tg > 1) {
        const unsigned long long total = (unsigned long long)gx * gy * gridDim.z;
        const unsigned span = 8u * (unsigned)tg;
        const unsigned long long base = id - id % span;
        if (base + span <= total) {
            const unsigned r = (unsigned)(id - base);
            id = base + (unsigned long long)(r & 7u) * (unsigned)tg + (r >> 3);
        }
    }
    tile = (int)(id % gx);
    const unsigned long long q = id / gx;
    row = (int)(q % gy);
    plane = (int64_t)(q / gy);
}

// tw[j] = exp(-2 pi i j / M), j < M/2: compile-time tables in the code object (one per grid size the passes support), so
// that no transform has to launch a set-up kernel or keep a table in the caller's workspace.  Evaluated in double by
// Taylor series (|x| < pi: 14 terms leave < 1e-15) and rounded once to float.
constexpr double kPiD = 3.14159265358979323846264338327950288;
constexpr double taylor_sin(double x)
{
    double term = x, sum = x;
    for (int k = 1; k <= 16; ++k) {
        term *= -x * x / (double)((2 * k) * (2 * k + 1));
        sum += term;
    }
    return sum;
}
constexpr double taylor_cos(double x)
{
    double term = 1.0, sum = 1.0;
    for (int k = 1; k <= 16; ++k) {
        term *= -x * x / (double)((2 * k - 1) * (2 * k));
        sum += term;
    }
    return sum;
}
struct TwPair { float x, y; };
template <int M>
struct TwTable { TwPair v[M / 2]; };
template <int M>
constexpr TwTable<M> make_tw_table()
{
    TwTable<M> t{};
    for (int j = 0; j < M / 2; ++j) {
        // reduce to [0, pi/2]: cos(pi - a) = -cos a, sin(pi - a) = sin a  (j < M/2  =>  angle in [0, pi))
        const bool upper = 4 * j > M;
        const double a = 2.0 * kPiD * (double)(upper ? M / 2 - j : j) / (double)M;
        const double c = taylor_cos(a), sn = taylor_sin(a);
        t.v[j] = TwPair{(float)(upper ? -c : c), (float)-sn};
    }
    return t;
}
template <int M>
__device__ const TwTable<M> kTwTable = make_tw_table<M>();

template <int M>
const float2 *tw_symbol()
{
    void *ptr = nullptr;
    if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(kTwTable<M>)) != hipSuccess) return nullptr;
    return (const float2 *)ptr;
}
// device address of the table for grid size M on the current device (looked up once per device)
const float2 *twiddle_table(int M)
{
    static std::atomic<const float2 *> cache[kMaxDevices][7];
    const int dev = current_device();
    int slot = 0;
    while ((16 << slot) < M) ++slot;
    if (slot > 6 || (16 << slot) != M) return nullptr;
    if (dev < kMaxDevices) {
        const float2 *p = cache[dev][slot].load(std::memory_order_acquire);
        if (p) return p;
    }
    const float2 *p = nullptr;
    switch (M) {
    case 16: p = tw_symbol<16>(); break;
    case 32: p = tw_symbol<32>(); break;
    case 64: p = tw_symbol<64>(); break;
    case 128: p = tw_symbol<128>(); break;
    case 256: p = tw_symbol<256>(); break;
    case 512: p = tw_symbol<512>(); break;
    case 1024: p = tw_symbol<1024>(); break;
    }
    if (p && dev < kMaxDevices) cache[dev][slot].store(p, std::memory_order_release);
    return p;
}

// In-place decimation-in-frequency FFT along the row index of buf[M][NC], radix 8 / 4 stages with the butterflies in
// registers: a radix-R step on blocks of length L reads x_q = buf[base + j + q L/R], forms y_p = (sum_q x_q W_R^{pq})
// W_L^{jp} and stores y_p back to buf[base + p L/R + j]; sub-block p then holds the transform of the outputs
// k = p (mod R).  X[k] ends up in row out_row(k) (mixed-radix digit reversal).  M = 512 takes 3 LDS round trips
// (8 x 8 x 8) instead of the 9 of a radix-2 pass.  INV selects exp(+2 pi i ...).  ltw = exp(-2 pi i j / M), j < M/2.
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiply by -i (forward) / +i (inverse)
template <bool INV>
__device__ __forceinline__ float2 rot90(float2 a)
{
    return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}

template <bool INV>
__device__ __forceinline__ void dft4(float2 &x0, float2 &x1, float2 &x2, float2 &x3)
{
    const float2 a0 = cadd(x0, x2), a1 = csub(x0, x2), a2 = cadd(x1, x3), a3 = rot90<INV>(csub(x1, x3));
    x0 = cadd(a0, a2);
    x1 = cadd(a1, a3);
    x2 = csub(a0, a2);
    x3 = csub(a1, a3);
}

template <bool INV>
__device__ __forceinline__ void dft8(float2 (&x)[8])
{
    // even / odd radix-4 transforms, then y[p] = E[p] + W8^p O[p], y[p + 4] = E[p] - W8^p O[p]
    dft4<INV>(x[0], x[2], x[4], x[6]);
    dft4<INV>(x[1], x[3], x[5], x[7]);
    constexpr float r = 0.70710678118654752440f;
    const float2 o0 = x[1];
    const float2 o1 = INV ? make_float2((x[3].x - x[3].y) * r, (x[3].x + x[3].y) * r)
                          : make_float2((x[3].x + x[3].y) * r, (x[3].y - x[3].x) * r);
    const float2 o2 = rot90<INV>(x[5]);
    const float2 o3 = INV ? make_float2((-x[7].x - x[7].y) * r, (x[7].x - x[7].y) * r)
                          : make_float2((x[7].y - x[7].x) * r, (-x[7].x - x[7].y) * r);
    const float2 e0 = x[0], e1 = x[2], e2 = x[4], e3 = x[6];
    x[0] = cadd(e0, o0); x[4] = csub(e0, o0);
    x[1] = cadd(e1, o1); x[5] = csub(e1, o1);
    x[2] = cadd(e2, o2); x[6] = csub(e2, o2);
    x[3] = cadd(e3, o3); x[7] = csub(e3, o3);
}

template <bool INV>
__device__ __forceinline__ float2 twiddle_at(const float2 *ltw, int e, int M)
{
    e &= M - 1;
    float2 w = ltw[e & (M / 2 - 1)];
    if (e >= M / 2) w = make_float2(-w.x, -w.y);
    if (INV) w.y = -w.y;
    return w;
}

template <bool INV, int R>
__device__ __forceinline__ void fft_stage(float2 *buf, const float2 *ltw, const ColGeom &cg, int tid, int logL)
{
    constexpr int logR = R == 8 ? 3 : 2;
    const int logLs = logL - logR, Ls = 1 << logLs;
    const int tstep = cg.M >> logL;
    const int total = (cg.M >> logR) << cg.logNC;
    __syncthreads();
    for (int idx = tid; idx < total; idx += kFftThreads) {
        const int col = idx & (cg.NC - 1);
        const int t = idx >> cg.logNC;
        const int j = t & (Ls - 1);
        const int row0 = ((t >> logLs) << logL) + j;
        float2 *p0 = buf + (row0 << cg.logNC) + col;
        const int stride = Ls << cg.logNC;
        float2 x[R];
#pragma unroll
        for (int q = 0; q < R; ++q) x[q] = p0[q * stride];
        if (R == 8) {
            dft8<INV>(reinterpret_cast<float2(&)[8]>(x));
        } else {
            dft4<INV>(x[0], x[1], x[2], x[3]);
        }
        p0[0] = x[0];
        if (logLs == 0) {
#pragma unroll
            for (int q = 1; q < R; ++q) p0[q * stride] = x[q];
        } else {
#pragma unroll
            for (int q = 1; q < R; ++q) p0[q * stride] = cmul(x[q], twiddle_at<INV>(ltw, j * q * tstep, cg.M));
        }
    }
}

template <bool INV>
__device__ __forceinline__ void lds_fft(float2 *buf, const float2 *ltw, const ColGeom &cg, int tid)
{
    switch (cg.logM) {
    case 4: fft_stage<INV, 4>(buf, ltw, cg, tid, 4); fft_stage<INV, 4>(buf, ltw, cg, tid, 2); break;
    case 5: fft_stage<INV, 8>(buf, ltw, cg, tid, 5); fft_stage<INV, 4>(buf, ltw, cg, tid, 2); break;
    case 6: fft_stage<INV, 8>(buf, ltw, cg, tid, 6); fft_stage<INV, 8>(buf, ltw, cg, tid, 3); break;
    case 7: fft_stage<INV, 8>(buf, ltw, cg, tid, 7); fft_stage<INV, 4>(buf, ltw, cg, tid, 4);
            fft_stage<INV, 4>(buf, ltw, cg, tid, 2); break;
    case 8: fft_stage<INV, 8>(buf, ltw, cg, tid, 8); fft_stage<INV, 8>(buf, ltw, cg, tid, 5);
            fft_stage<INV, 4>(buf, ltw, cg, tid, 2); break;
    case 9: fft_stage<INV, 8>(buf, ltw, cg, tid, 9); fft_stage<INV, 8>(buf, ltw, cg, tid, 6);
            fft_stage<INV, 8>(buf, ltw, cg, tid, 3); break;
    default: fft_stage<INV, 8>(buf, ltw, cg, tid, 10); fft_stage<INV, 8>(buf, ltw, cg, tid, 7);
             fft_stage<INV, 4>(buf, ltw, cg, tid, 4); fft_stage<INV, 4>(buf, ltw, cg, tid, 2); break;
    }
    __syncthreads();
}

// Row that holds X[k] after lds_fft: the digits of k in the radix sequence above, most significant first.
__device__ __forceinline__ int brev_row(int k, int logM)
{
    int pos = 0, lg = logM;
    auto digit = [&](int lr) { lg -= lr; pos += (k & ((1 << lr) - 1)) << lg; k >>= lr; };
    switch (logM) {
    case 4: digit(2); digit(2); break;
    case 5: digit(3); digit(2); break;
    case 6: digit(3); digit(3); break;
    case 7: digit(3); digit(2); digit(2); break;
    case 8: digit(3); digit(3); digit(2); break;
    case 9: digit(3); digit(3); digit(3); break;
    default: digit(3); digit(3); digit(2); digit(2); break;
    }
    return pos;
}

// Fills LDS from global memory with UN loads per thread in flight before the first one is consumed: a tile has to
// keep ~64 KB outstanding per CU to cover the HBM latency (two 256-thread workgroups share a CU).
template <int UN, typename LoadF, typename StoreF>
__device__ __forceinline__ void batched_fill(int total, int tid, LoadF load, StoreF store)
{
    for (int base = tid; base < total; base += kFftThreads * UN) {
        float2 v[UN];
#pragma unroll
        for (int r = 0; r < UN; ++r) {
            const int idx = base + r * kFftThreads;
            v[r] = idx < total ? load(idx) : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int r = 0; r < UN; ++r) {
            const int idx = base + r * kFftThreads;
            if (idx < total) store(idx, v[r]);
        }
    }
}

__device__ __forceinline__ void stage_twiddles(float2 *ltw, const float2 *__restrict__ tw, int M, int tid)
{
    for (int j = tid; j < M / 2; j += kFftThreads) ltw[j] = tw[j];
}

// ---- adjoint, axis 1:  S[plane][u0][u1][Mh] -> T[plane][u0][NB][KC] --------------------------------------
// LOGM / LOGNC > 0: grid size and tile width known at compile time (the stage sequence, the digit reversal and the
// twiddle masks fold to constants); 0: run-time values of the geometry
template <int LOGM, int LOGNC>
__global__ void __launch_bounds__(kFftThreads)
adj_axis1_kernel(ColGeom cg_in, const float2 *__restrict__ tw, const float2 *__restrict__ S, float2 *__restrict__ T)
{
    ColGeom cg = cg_in;
    if constexpr (LOGM > 0) {
        cg.M = 1 << LOGM;
        cg.logM = LOGM;
        cg.NC = 1 << LOGNC;
        cg.logNC = LOGNC;
    }
    extern __shared__ float2 smem[];
    float2 *ltw = smem;
    float2 *buf = smem + cg.M / 2;
    const int tid = threadIdx.x;
    int tile_x, u0;
    int64_t plane;
    column_tile_of_block(cg.TG, tile_x, u0, plane);
    const int c0 = tile_x << cg.logNC;
    if (c0 >= cg.KC) return;  // (a spare tile of the rounded-up launch)
    stage_twiddles(ltw, tw, cg.M, tid);
    const float2 *src = S + ((plane * cg.M + u0) * cg.M) * cg.SR;
    batched_fill<16>(cg.M << cg.logNC, tid,
                     [&](int idx) {
                         const int col = idx & (cg.NC - 1), u1 = idx >> cg.logNC;
                         const int k2 = c0 + col;
                         return k2 < cg.KC ? src[(int64_t)u1 * cg.SR + k2] : make_float2(0.f, 0.f);
                     },
                     [&](int idx, float2 v) { buf[idx] = v; });
    lds_fft<false>(buf, ltw, cg, tid);
    float2 *dst = T + ((plane * cg.M + u0) * cg.NB) * cg.KS;
    for (int idx = tid; idx < (cg.NB << cg.logNC); idx += kFftThreads) {
        const int col = idx & (cg.NC - 1), j1 = idx >> cg.logNC;
        const int k2 = c0 + col;
        if (k2 < cg.KC) {
            const int k1 = (j1 - cg.H) & (cg.M - 1);
            dst[(int64_t)j1 * cg.KS + k2] = buf[(brev_row(k1, cg.logM) << cg.logNC) + col];
        }
    }
}

// fastsum: the kernel's Fourier coefficient of band frequency f rides along with the roll-off (1 real, 2 complex)
__device__ __forceinline__ void apply_mult(const void *__restrict__ mult, int kind, int64_t f, float &re, float &im)
{
    if (kind == 1) {
        const float w = ((const float *)mult)[f];
        re *= w;
        im *= w;
    } else if (kind == 2) {
        const float2 w = ((const float2 *)mult)[f];
        const float r2 = re * w.x - im * w.y;
        im = re * w.y + im * w.x;
        re = r2;
    }
}

// ---- adjoint, axis 0 + roll-off:  T -> y[b][N][N][N][C] ---------------------------------------------------
// F = forward DFT of the real plane at (k0, k1, k2 >= 0).  g_hat (e^{+} convention) = conj(F); the k2 < 0 half
// follows from g_hat[-k] = conj(g_hat[k]) = F[k].
template <int LOGM, int LOGNC, bool XCOMPLEX, bool REAL_OUT>
__global__ void __launch_bounds__(kFftThreads)
adj_axis0_kernel(ColGeom cg_in, const float2 *__restrict__ tw, const float2 *__restrict__ T, int64_t C, int64_t col0,
                 void *__restrict__ yv, const void *__restrict__ mult, int mult_kind)
{
    ColGeom cg = cg_in;
    if constexpr (LOGM > 0) {
        cg.M = 1 << LOGM;
        cg.logM = LOGM;
        cg.NC = 1 << LOGNC;
        cg.logNC = LOGNC;
    }
    extern __shared__ float2 smem[];
    float2 *ltw = smem;
    float2 *buf = smem + cg.M / 2;
    float2 *buf2 = buf + (cg.M << cg.logNC);
    const int tid = threadIdx.x;
    int tile_x, j1;
    int64_t col_local;  // (batch, column) pair inside this chunk of planes
    column_tile_of_block(cg.TG, tile_x, j1, col_local);
    const int c0 = tile_x << cg.logNC;
    if (c0 >= cg.KC) return;  // (a spare tile of the rounded-up launch)
    stage_twiddles(ltw, tw, cg.M, tid);
    const int64_t plane = XCOMPLEX ? col_local * 2 : col_local;
    const int64_t pstride = (int64_t)cg.M * cg.NBm * cg.KS;
    const float2 *src = T + plane * pstride + (int64_t)j1 * cg.KS;
    for (int part = 0; part < (XCOMPLEX ? 2 : 1); ++part) {
        float2 *const dstbuf = part ? buf2 : buf;
        const float2 *const srcp = src + part * pstride;
        batched_fill<16>(cg.M << cg.logNC, tid,
                         [&](int idx) {
                             const int col = idx & (cg.NC - 1), u0 = idx >> cg.logNC;
                             const int k2 = c0 + col;
                             return k2 < cg.KC ? srcp[(int64_t)u0 * cg.NBm * cg.KS + k2] : make_float2(0.f, 0.f);
                         },
                         [&](int idx, float2 v) { dstbuf[idx] = v; });
    }
    lds_fft<false>(buf, ltw, cg, tid);
    if (XCOMPLEX) lds_fft<false>(buf2, ltw, cg, tid);

    const int64_t colg = col0 + col_local;
    const int64_t b = colg / C, c = colg - b * C;
    const int H = cg.H, N = cg.N;
    const int k1 = cg.two_d ? 0 : j1 - H;  // (2-D: no middle axis -- frequency 0 of an axis of one cell, factor 1)
    const float f1 = cg.two_d ? 1.0f : phi_hat_inv_f(abs(k1), cg.param);
    const int64_t NN = cg.two_d ? (int64_t)N * N : (int64_t)N * N * N;
    for (int idx = tid; idx < (cg.NB << cg.logNC); idx += kFftThreads) {
        const int col = idx & (cg.NC - 1), j0 = idx >> cg.logNC;
        const int k2 = c0 + col;
        if (k2 >= cg.KC) continue;
        const int k0 = j0 - H;
        const int row = (brev_row(k0 & (cg.M - 1), cg.logM) << cg.logNC) + col;
        const float2 fr = buf[row];
        float2 fi = make_float2(0.f, 0.f);
        if (XCOMPLEX) fi = buf2[row];
        const float fac = phi_hat_inv_f(abs(k0), cg.param) * f1 * phi_hat_inv_f(k2, cg.param);
        if (k0 < H && k1 < H && k2 < H) {
            // direct: g_hat[k] = conj(F_re) + i conj(F_im)
            float re = (fr.x + fi.y) * fac, im = (-fr.y + fi.x) * fac;
            const int64_t f = cg.two_d ? (int64_t)(k0 + H) * N + (k2 + H) : ((int64_t)(k0 + H) * N + (k1 + H)) * N + (k2 + H);
            apply_mult(mult, mult_kind, f, re, im);
            const int64_t o = (b * NN + f) * C + c;
            if (REAL_OUT) ((float *)yv)[o] = re;
            else ((float2 *)yv)[o] = make_float2(re, im);
        }
        if (k2 >= 1 && k0 > -H && k1 > -H) {
            // mirror: g_hat[-k] = F_re + i F_im
            float re = (fr.x - fi.y) * fac, im = (fr.y + fi.x) * fac;
            const int64_t f = cg.two_d ? (int64_t)(H - k0) * N + (H - k2) : ((int64_t)(H - k0) * N + (H - k1)) * N + (H - k2);
            apply_mult(mult, mult_kind, f, re, im);
            const int64_t o = (b * NN + f) * C + c;
            if (REAL_OUT) ((float *)yv)[o] = re;
            else ((float2 *)yv)[o] = make_float2(re, im);
        }
    }
}

// ---- forward, roll-off + Hermitian split + axis 0:  xhat -> T[plane][u0][NB][KC] ------------------------
//   a[k] = xhat[b, k + N/2, c] * fac inside the band, 0 outside
//   Re g = C2R( (a[-k] + conj(a[k])) / 2 ),  Im g = C2R( (a[-k] - conj(a[k])) / (2i) )     (e^{+} transforms)
template <bool XCOMPLEX>
__device__ __forceinline__ float2 band_value(const void *__restrict__ xhat, int64_t b, int64_t c, int64_t C, int N,
                                             int H, int k0, int k1, int k2, const bool two_d = false)
{
    if (k0 < -H || k0 >= H || (!two_d && (k1 < -H || k1 >= H)) || k2 < -H || k2 >= H) return make_float2(0.f, 0.f);
    const int64_t idx = two_d ? ((b * N + (k0 + H)) * N + (k2 + H)) * C + c
                              : (((b * N + (k0 + H)) * N + (k1 + H)) * N + (k2 + H)) * C + c;
    if (XCOMPLEX) return ((const float2 *)xhat)[idx];
    return make_float2(((const float *)xhat)[idx], 0.f);
}

template <int LOGM, int LOGNC, bool XCOMPLEX>
__global__ void __launch_bounds__(kFftThreads)
fwd_axis0_kernel(ColGeom cg_in, const float2 *__restrict__ tw, const void *__restrict__ xhat, int64_t C, int ppc,
                 int64_t plane0, float2 *__restrict__ T)
{
    ColGeom cg = cg_in;
    if constexpr (LOGM > 0) {
        cg.M = 1 << LOGM;
        cg.logM = LOGM;
        cg.NC = 1 << LOGNC;
        cg.logNC = LOGNC;
    }
    extern __shared__ float2 smem[];
    float2 *ltw = smem;
    float2 *buf = smem + cg.M / 2;
    const int tid = threadIdx.x;
    int tile_x, j1;
    int64_t pl;
    column_tile_of_block(cg.TG, tile_x, j1, pl);
    const int c0 = tile_x << cg.logNC;
    if (c0 >= cg.KC) return;  // (a spare tile of the rounded-up launch)
    stage_twiddles(ltw, tw, cg.M, tid);
    for (int idx = tid; idx < (cg.M << cg.logNC); idx += kFftThreads) buf[idx] = make_float2(0.f, 0.f);
    __syncthreads();
    const int64_t plane = plane0 + pl;
    const int64_t colg = plane / ppc;
    const int part = (int)(plane - colg * ppc);
    const int64_t b = colg / C, c = colg - b * C;
    const int H = cg.H;
    const int k1 = cg.two_d ? 0 : j1 - H;
    const float f1 = cg.two_d ? 1.0f : phi_hat_inv_f(abs(k1), cg.param);
    batched_fill<8>(cg.NB << cg.logNC, tid,
                    [&](int idx) {
                        const int col = idx & (cg.NC - 1), j0 = idx >> cg.logNC;
                        const int k2 = c0 + col;
                        if (k2 >= cg.KC) return make_float2(0.f, 0.f);
                        const int k0 = j0 - H;
                        const float2 ap = band_value<XCOMPLEX>(xhat, b, c, C, cg.N, H, k0, k1, k2, cg.two_d != 0);
                        const float2 am = band_value<XCOMPLEX>(xhat, b, c, C, cg.N, H, -k0, -k1, -k2, cg.two_d != 0);
                        const float fac = 0.5f * phi_hat_inv_f(abs(k0), cg.param) * f1 * phi_hat_inv_f(k2, cg.param);
                        if (part == 0) return make_float2((am.x + ap.x) * fac, (am.y - ap.y) * fac);
                        return make_float2((am.y + ap.y) * fac, -(am.x - ap.x) * fac);
                    },
                    [&](int idx, float2 v) {
                        const int col = idx & (cg.NC - 1), j0 = idx >> cg.logNC;
                        if (c0 + col < cg.KC) buf[(((j0 - H) & (cg.M - 1)) << cg.logNC) + col] = v;
                    });
    lds_fft<true>(buf, ltw, cg, tid);
    float2 *dst = T + pl * ((int64_t)cg.M * cg.NBm * cg.KS) + (int64_t)j1 * cg.KS;
    for (int idx = tid; idx < (cg.M << cg.logNC); idx += kFftThreads) {
        const int col = idx & (cg.NC - 1), u0 = idx >> cg.logNC;
        const int k2 = c0 + col;
        if (k2 < cg.KC) dst[(int64_t)u0 * cg.NBm * cg.KS + k2] = buf[(brev_row(u0, cg.logM) << cg.logNC) + col];
    }
}

// ---- forward, axis 1:  T[plane][u0][NB][KC] -> S[plane][u0][u1][Mh]  (columns k2 >= KC are zero) ----------
template <int LOGM, int LOGNC>
__global__ void __launch_bounds__(kFftThreads)
fwd_axis1_kernel(ColGeom cg_in, const float2 *__restrict__ tw, const float2 *__restrict__ T, float2 *__restrict__ S)
{
    ColGeom cg = cg_in;
    if constexpr (LOGM > 0) {
        cg.M = 1 << LOGM;
        cg.logM = LOGM;
        cg.NC = 1 << LOGNC;
        cg.logNC = LOGNC;
    }
    extern __shared__ float2 smem[];
    float2 *ltw = smem;
    float2 *buf = smem + cg.M / 2;
    const int tid = threadIdx.x;
    int tile_x, u0;
    int64_t plane;
    column_tile_of_block(cg.TG, tile_x, u0, plane);
    const int c0 = tile_x << cg.logNC;
    if (c0 >= (cg.SR == cg.KS ? cg.KC : cg.Mh)) return;  // (a spare tile of the rounded-up launch)
    float2 *dst = S + ((plane * cg.M + u0) * cg.M) * cg.SR;
    const int width = cg.SR == cg.KS ? cg.KC : cg.Mh;  // columns of S that exist (the compact layout has no zero tail)
    if (c0 >= cg.KC) {  // zero tail of the padded half spectrum
        for (int idx = tid; idx < (cg.M << cg.logNC); idx += kFftThreads) {
            const int col = idx & (cg.NC - 1), u1 = idx >> cg.logNC;
            if (c0 + col < width) dst[(int64_t)u1 * cg.SR + c0 + col] = make_float2(0.f, 0.f);
        }
        return;
    }
    stage_twiddles(ltw, tw, cg.M, tid);
    for (int idx = tid; idx < (cg.M << cg.logNC); idx += kFftThreads) buf[idx] = make_float2(0.f, 0.f);
    __syncthreads();
    const float2 *src = T + ((plane * cg.M + u0) * cg.NB) * cg.KS;
    batched_fill<16>(cg.NB << cg.logNC, tid,
                     [&](int idx) {
                         const int col = idx & (cg.NC - 1), j1 = idx >> cg.logNC;
                         const int k2 = c0 + col;
                         return k2 < cg.KC ? src[(int64_t)j1 * cg.KS + k2] : make_float2(0.f, 0.f);
                     },
                     [&](int idx, float2 v) {
                         const int col = idx & (cg.NC - 1), j1 = idx >> cg.logNC;
                         if (c0 + col < cg.KC) buf[(((j1 - cg.H) & (cg.M - 1)) << cg.logNC) + col] = v;
                     });
    lds_fft<true>(buf, ltw, cg, tid);
    for (int idx = tid; idx < (cg.M << cg.logNC); idx += kFftThreads) {
        const int col = idx & (cg.NC - 1), u1 = idx >> cg.logNC;
        const int k2 = c0 + col;
        if (k2 < width) {
            dst[(int64_t)u1 * cg.SR + k2] =
                k2 < cg.KC ? buf[(brev_row(u1, cg.logM) << cg.logNC) + col] : make_float2(0.f, 0.f);
        }
    }
}

// ---- axis 2 (contiguous rows): pruned real <-> half-complex transforms, one wave per row ---------------------
// The M real samples of a row are packed as L = M/2 complex numbers z[n] = x[2n] + i x[2n+1]; one L-point complex FFT
// in the wave's private LDS (same radix-8/4 butterflies as the column passes, no workgroup barrier) plus the split
//     X[k] = (Z[k] + conj Z[L-k]) / 2 - i W_M^k (Z[k] - conj Z[L-k]) / 2,      k = 0 .. N/2   (the band; W_M = e^{-2 pi i / M})
// gives the kept half-spectrum columns directly: one pass over the grid (rocFFT needs two and writes / reads the
// M/2+1 - (N/2+1) columns nobody uses).  The inverse direction builds Z from the band (zero beyond it) and ends
// with x[2n] = Re z[n], x[2n+1] = Im z[n]; both are unnormalised like rocFFT's.
// A wave transforms RP = 512 / L consecutive rows at once (its LDS region always holds 512 complex numbers): the
// radix-8 stages then have 64 butterflies, one per lane, and RP rows' worth of loads are in flight.
constexpr int kWaveCplx = 512;
constexpr int kRowWaves = 4;       // waves per workgroup
constexpr int kR2cRounds = 1;  // the R2C pass reads 2 KB per row: one round in flight is enough (2 rounds 143 us, 1: 140, 3: 146)
constexpr int kC2rRounds = 2;  // rounds per wave of the C2R pass (all their loads go out first)
// (rounds processed one after the other, each loading its own rows, lost: 1 round 0.194 ms at C3 / 12.0 ms at C4-share,
// 2 rounds 0.212 / 13.7, 4 and 8 worse; 2 or 8 waves per workgroup lose as well)

// LDS index swizzle of the per-wave buffer: the butterflies of the later stages touch elements 4, 8 or 32 apart,
// which without it land in the same banks (up to 16-way conflicts).  Bits 4..2 are XORed with bits 7..5 and bits
// 1..0 with bits 3..2: every stage's 64 lanes then spread over all 32 eight-byte bank slots.  The map is linear over
// GF(2), so zsw(a | b) = zsw(a) ^ zsw(b) for disjoint a, b: one XOR per butterfly leg.
__device__ __host__ constexpr int zsw(int i) { return i ^ ((i >> 3) & 0x1c) ^ ((i >> 2) & 3); }

// one radix-R step on blocks of 2^LOGLB elements over the wave's 512-element buffer; ltw = exp(-2 pi i e / M), e < M
template <bool INV, int R, int LOGLB, int LOGM>
__device__ __forceinline__ void wave_fft_stage(float2 *z, const float2 *ltw, int lane)
{
    constexpr int logR = R == 8 ? 3 : 2;
    constexpr int logLs = LOGLB - logR, Ls = 1 << logLs;
    constexpr int logT = LOGM - LOGLB;  // W_Lb^{jq} = W_M^{jq 2^logT}
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t0 = 0; t0 < (kWaveCplx >> logR); t0 += 64) {
        const int t = t0 + lane;
        const int j = t & (Ls - 1);
        const int p0 = zsw(((t >> logLs) << LOGLB) + j);
        float2 x[R];
#pragma unroll
        for (int q = 0; q < R; ++q) x[q] = z[p0 ^ zsw(q * Ls)];
        if (R == 8) {
            dft8<INV>(reinterpret_cast<float2(&)[8]>(x));
        } else {
            dft4<INV>(x[0], x[1], x[2], x[3]);
        }
        z[p0] = x[0];
        if (logLs == 0) {
#pragma unroll
            for (int q = 1; q < R; ++q) z[p0 ^ zsw(q * Ls)] = x[q];
        } else {
            const int e1 = j << logT;
#pragma unroll
            for (int q = 1; q < R; ++q) {
                float2 w = ltw[(q * e1) & ((1 << LOGM) - 1)];
                if (INV) w.y = -w.y;
                z[p0 ^ zsw(q * Ls)] = cmul(x[q], w);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// L-point FFTs (L = 2^LOGL = M/2) of the RP rows in the wave's buffer; X[k] ends up at rev[k] inside its row
template <bool INV, int LOGL>
__device__ __forceinline__ void wave_fft(float2 *z, const float2 *ltw, int lane)
{
    constexpr int LOGM = LOGL + 1;
    if constexpr (LOGL == 6) {
        wave_fft_stage<INV, 8, 6, LOGM>(z, ltw, lane); wave_fft_stage<INV, 8, 3, LOGM>(z, ltw, lane);
    } else if constexpr (LOGL == 7) {
        wave_fft_stage<INV, 8, 7, LOGM>(z, ltw, lane); wave_fft_stage<INV, 4, 4, LOGM>(z, ltw, lane);
        wave_fft_stage<INV, 4, 2, LOGM>(z, ltw, lane);
    } else if constexpr (LOGL == 8) {
        wave_fft_stage<INV, 8, 8, LOGM>(z, ltw, lane); wave_fft_stage<INV, 8, 5, LOGM>(z, ltw, lane);
        wave_fft_stage<INV, 4, 2, LOGM>(z, ltw, lane);
    } else {
        wave_fft_stage<INV, 8, 9, LOGM>(z, ltw, lane); wave_fft_stage<INV, 8, 6, LOGM>(z, ltw, lane);
        wave_fft_stage<INV, 8, 3, LOGM>(z, ltw, lane);
    }
}

// LDS of the row kernels: full twiddle table, digit-reversal table, the waves' buffers
template <int LOGL>
struct RowLds {
    float2 tw[2 << LOGL];                 // exp(-2 pi i e / M), e < M
    float2 z[kRowWaves][kWaveCplx];
    unsigned short rev[1 << LOGL];        // row-local position of X[k] after wave_fft
};

template <int LOGL>
__device__ __forceinline__ void row_tables(RowLds<LOGL> &S, const float2 *__restrict__ tw, int tid)
{
    constexpr int L = 1 << LOGL, M = 2 * L;
    for (int e = tid; e < M; e += kRowWaves * 64) {
        float2 w = tw[e & (L - 1)];  // the global table covers e < M/2; the other half is its negative
        if (e >= L) w = make_float2(-w.x, -w.y);
        S.tw[e] = w;
    }
    for (int k = tid; k < L; k += kRowWaves * 64) S.rev[k] = (unsigned short)brev_row(k, LOGL);
    __syncthreads();
}

// grid rows [nrows][M] real  ->  S[nrows][KS] complex (KC of them kept)
template <int LOGL>
__global__ void __launch_bounds__(kRowWaves * 64)
row_r2c_kernel(int KC, int KS, int64_t nrows, const float2 *__restrict__ tw, const float *__restrict__ grid, float2 *__restrict__ out)
{
    constexpr int L = 1 << LOGL, M = 2 * L, RP = kWaveCplx / L;
    __shared__ RowLds<LOGL> S;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float2 *z = S.z[wave];
    row_tables<LOGL>(S, tw, tid);
    const int64_t row0 = ((int64_t)blockIdx.x * kRowWaves + wave) * kR2cRounds * RP;
    // 16 bytes per lane: the rows arrive in half as many load instructions (C3: 0.198 -> 0.173 ms per pass); the rows
    // of all rounds of the wave are requested before the first transform
    float4 v4[kR2cRounds][kWaveCplx / 128];
#pragma unroll
    for (int r = 0; r < kR2cRounds; ++r) {
        const int64_t row = row0 + (int64_t)r * RP;
        const int nr = (int)max((int64_t)0, min((int64_t)RP, nrows - row));
        const float4 *src4 = (const float4 *)(grid + row * M);
#pragma unroll
        for (int q = 0; q < kWaveCplx / 128; ++q) {
            const int n2 = q * 64 + lane;  // pair index: elements 2 n2, 2 n2 + 1
            v4[r][q] = 2 * n2 < nr * L ? src4[n2] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
#pragma unroll
    for (int r = 0; r < kR2cRounds; ++r) {
        const int64_t row = row0 + (int64_t)r * RP;
        if (row >= nrows) break;
        const int nr = (int)min((int64_t)RP, nrows - row);  // rows of this round (all RP except at the very end)
#pragma unroll
        for (int q = 0; q < kWaveCplx / 128; ++q) {
            const int n2 = q * 64 + lane;
            z[zsw(2 * n2)] = make_float2(v4[r][q].x, v4[r][q].y);
            z[zsw(2 * n2 + 1)] = make_float2(v4[r][q].z, v4[r][q].w);
        }
        wave_fft<false, LOGL>(z, S.tw, lane);
        for (int rr = 0; rr < nr; ++rr) {
            float2 *dst = out + (row + rr) * KS;
            for (int k = lane; k < KC; k += 64) {
                const float2 a = z[zsw(rr * L + S.rev[k])];
                float2 b = z[zsw(rr * L + S.rev[(L - k) & (L - 1)])];
                b.y = -b.y;  // conj Z[L - k]
                const float2 e = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y + b.y));
                const float2 o = make_float2(0.5f * (a.x - b.x), 0.5f * (a.y - b.y));
                const float2 wo = cmul(S.tw[k], o);
                dst[k] = make_float2(e.x + wo.y, e.y - wo.x);  // e - i w o
            }
        }
    }
}

// S[nrows][KS] complex (KC of them kept, zero beyond the band)  ->  grid rows [nrows][M] real
template <int LOGL>
__global__ void __launch_bounds__(kRowWaves * 64)
row_c2r_kernel(int KC, int KS, int64_t nrows, const float2 *__restrict__ tw, const float2 *__restrict__ in, float *__restrict__ grid)
{
    constexpr int L = 1 << LOGL, M = 2 * L, RP = kWaveCplx / L;
    __shared__ RowLds<LOGL> S;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float2 *z = S.z[wave];
    row_tables<LOGL>(S, tw, tid);
    // The half rows of ALL rounds of the wave are requested before the first transform: a half row is 1 KB, and with
    // one round in flight per wave the pass was bound by bytes in flight (4.0 TB/s against 5.9 for the R2C pass, which
    // reads 2 KB per row).  One load per element: beyond the band X is zero, so Z[k] needs X[k] for k < KC and
    // X[L - k] for L - k < KC, and with KC = L/2 + 1 (the band of an oversampled grid) exactly one of them is kept --
    // both only at k = L/2, where they are the same element.
    const int64_t row0 = ((int64_t)blockIdx.x * kRowWaves + wave) * kC2rRounds * RP;
    constexpr int NIT = kWaveCplx / 64;
    float2 va[kC2rRounds][NIT];  // (KC == L / 2 + 1 by construction: the launcher passes N / 2 + 1 and L = M / 2 = N)
#pragma unroll
    for (int r = 0; r < kC2rRounds; ++r) {
        const int64_t row = row0 + (int64_t)r * RP;
        const int nr = (int)max((int64_t)0, min((int64_t)RP, nrows - row));
        const float2 *src = in + row * KS;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int n = it * 64 + lane;
            const int rr = n >> LOGL, k = n & (L - 1);
            const bool live = rr < nr;
            const int kk = k <= L / 2 ? k : L - k;
            va[r][it] = live ? src[rr * KS + kk] : make_float2(0.f, 0.f);
        }
    }
#pragma unroll
    for (int r = 0; r < kC2rRounds; ++r) {
        const int64_t row = row0 + (int64_t)r * RP;
        if (row >= nrows) break;
        const int nr = (int)min((int64_t)RP, nrows - row);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            // Z[k] = (X[k] + conj X[L-k]) + i conj(W_M^k) (X[k] - conj X[L-k]),  X = 0 beyond the band
            const int n = it * 64 + lane;
            const int k = n & (L - 1);
            const float2 v = va[r][it], zero = make_float2(0.f, 0.f);
            const float2 a = k <= L / 2 ? v : zero;
            float2 b = k >= L / 2 ? v : zero;
            b.y = -b.y;
            const float2 sum = make_float2(a.x + b.x, a.y + b.y), dif = make_float2(a.x - b.x, a.y - b.y);
            float2 w = S.tw[k];
            w.y = -w.y;
            const float2 wd = cmul(w, dif);
            z[zsw(n)] = make_float2(sum.x - wd.y, sum.y + wd.x);  // sum + i w dif
        }
        wave_fft<true, LOGL>(z, S.tw, lane);
        float4 *dst4 = (float4 *)(grid + row * M);
#pragma unroll
        for (int q = 0; q < kWaveCplx / 128; ++q) {
            const int n2 = q * 64 + lane;
            const int na = 2 * n2, nb = 2 * n2 + 1;
            const int rr = na >> LOGL;
            if (rr < nr) {
                const float2 a = z[zsw(rr * L + S.rev[na & (L - 1)])], b = z[zsw(rr * L + S.rev[nb & (L - 1)])];
                dst4[n2] = make_float4(a.x, a.y, b.x, b.y);
            }
        }
    }
}

// =====================================================================================================================
// Column-innermost pipeline for several coefficient columns (C > 1).
//
// The reference stores spectra as [B, N^d, C] (columns innermost, docs/source/theory/dataformat.rst:43-63): the last
// adjoint pass / first forward pass of a plane-by-plane pipeline would touch elements 8 C bytes apart, and rounds 1-2
// therefore worked on planar copies and moved between the layouts with tiled transposes (column_layout_kernel: 17 GB of
// the C4-share step's ~110 GB).  Here the planes of a chunk are taken in groups of kCiQ = 16 and the group index is the
// INNERMOST index of both intermediate arrays,
//     S'[group][u0][u1][k2][q]      (axis-2 half spectrum, kept columns only)
//     T'[group][u0][j1][k2][q]      (after the axis-1 pass: band+ rows)
// so that the tile of a column pass is  [M rows][16 planes]  for ONE k2: every global access is still a run of
// 16 * 8 = 128 contiguous bytes, and the last pass writes y[b, k, c .. c + 15] -- 16 consecutive columns of one
// frequency -- in place.  The regrouping costs no HBM traffic: the row passes, which have every plane's row in LDS
// anyway, exchange 16 planes' rows through LDS (one wave per plane) before they write / after they read.
constexpr int kCiQ = 16;       // planes per group (the tile width of the column passes)
constexpr int kCiLogQ = 4;

// grid rows of 16 planes -> S'.  One workgroup = 16 waves = the RP rows [row0, row0 + RP) of the 16 planes of a group.
template <int LOGL>
struct RowCiLds {
    float2 tw[2 << LOGL];
    float2 z[kCiQ][kWaveCplx];            // the waves' transform buffers; ALSO the exchange buffer
                                          // xch[row of the round][k2][plane ^ (k2 & 15)] (RP * KC * 16 <= 16 * 512 entries),
                                          // used while no transform is in progress: 67 KB, two workgroups per CU
    unsigned short rev[1 << LOGL];
};
// slot of (row rk = rr * KC + k2, plane q) in the exchange buffer: a wave writes / reads one plane for 64 consecutive k2,
// 128 bytes apart -- without the XOR all 64 lanes hit one pair of LDS banks
__device__ __forceinline__ int xch_slot(int rk, int k2, int q) { return rk * kCiQ + (q ^ (k2 & (kCiQ - 1))); }

template <int LOGL>
__device__ __forceinline__ void row_ci_tables(RowCiLds<LOGL> &S, const float2 *__restrict__ tw, int tid)
{
    constexpr int L = 1 << LOGL, M = 2 * L;
    for (int e = tid; e < M; e += kCiQ * 64) {
        float2 w = tw[e & (L - 1)];
        if (e >= L) w = make_float2(-w.x, -w.y);
        S.tw[e] = w;
    }
    for (int k = tid; k < L; k += kCiQ * 64) S.rev[k] = (unsigned short)brev_row(k, LOGL);
    __syncthreads();
}

template <int LOGL>
__global__ void __launch_bounds__(kCiQ * 64)
row_r2c_ci_kernel(int64_t rows_per_plane, int64_t nplanes, const float2 *__restrict__ tw, const float *__restrict__ grid,
                  float2 *__restrict__ out)
{
    constexpr int L = 1 << LOGL, M = 2 * L, RP = kWaveCplx / L, KC = L / 2 + 1;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    RowCiLds<LOGL> &S = *reinterpret_cast<RowCiLds<LOGL> *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float2 *z = S.z[wave];
    row_ci_tables<LOGL>(S, tw, tid);
    const int64_t row0 = (int64_t)blockIdx.x * RP;          // RP divides M: the rows of a round lie in one plane
    const int64_t group = blockIdx.y;
    const int64_t plane = group * kCiQ + wave;
    const bool live = plane < nplanes;
    float4 v4[kWaveCplx / 128];
    const float4 *src4 = (const float4 *)(grid + (plane * rows_per_plane + row0) * M);
#pragma unroll
    for (int q = 0; q < kWaveCplx / 128; ++q) v4[q] = live ? src4[q * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int q = 0; q < kWaveCplx / 128; ++q) {
        const int n2 = q * 64 + lane;
        z[zsw(2 * n2)] = make_float2(v4[q].x, v4[q].y);
        z[zsw(2 * n2 + 1)] = make_float2(v4[q].z, v4[q].w);
    }
    wave_fft<false, LOGL>(z, S.tw, lane);
    // the kept half-spectrum values of the wave's RP rows, first into registers: the exchange buffer lies over the
    // transform buffers, which every wave must have finished reading
    constexpr int NV = (RP * KC + 63) / 64;
    float2 vals[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int t = i * 64 + lane;
        vals[i] = make_float2(0.f, 0.f);
        if (t < RP * KC) {
            const int rr = t / KC, k = t - rr * KC;
            const float2 a = z[zsw(rr * L + S.rev[k])];
            float2 b = z[zsw(rr * L + S.rev[(L - k) & (L - 1)])];
            b.y = -b.y;  // conj Z[L - k]
            const float2 e = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y + b.y));
            const float2 o = make_float2(0.5f * (a.x - b.x), 0.5f * (a.y - b.y));
            const float2 wo = cmul(S.tw[k], o);
            vals[i] = make_float2(e.x + wo.y, e.y - wo.x);  // e - i w o
        }
    }
    __syncthreads();
    float2 *const xch = &S.z[0][0];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int t = i * 64 + lane;
        if (t < RP * KC) xch[xch_slot(t, t % KC, wave)] = vals[i];
    }
    __syncthreads();
    // rows row0 .. row0 + RP - 1 of S' are contiguous: RP * KC * 16 complex numbers
    float2 *dst = out + ((group * rows_per_plane + row0) * KC) * kCiQ;
    for (int e = tid; e < RP * KC * kCiQ; e += kCiQ * 64) {
        const int rk = e >> kCiLogQ;
        dst[e] = xch[xch_slot(rk, rk % KC, e & (kCiQ - 1))];
    }
}

// S' -> grid rows of 16 planes
template <int LOGL>
__global__ void __launch_bounds__(kCiQ * 64)
row_c2r_ci_kernel(int64_t rows_per_plane, int64_t nplanes, const float2 *__restrict__ tw, const float2 *__restrict__ in,
                  float *__restrict__ grid)
{
    constexpr int L = 1 << LOGL, M = 2 * L, RP = kWaveCplx / L, KC = L / 2 + 1;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    RowCiLds<LOGL> &S = *reinterpret_cast<RowCiLds<LOGL> *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float2 *z = S.z[wave];
    const int64_t row0 = (int64_t)blockIdx.x * RP;
    const int64_t group = blockIdx.y;
    const int64_t plane = group * kCiQ + wave;
    const float2 *src = in + ((group * rows_per_plane + row0) * KC) * kCiQ;
    float2 *const xch = &S.z[0][0];  // (lies over the transform buffers: read out into registers before they are used)
    for (int e = tid; e < RP * KC * kCiQ; e += kCiQ * 64) {
        const int rk = e >> kCiLogQ;
        xch[xch_slot(rk, rk % KC, e & (kCiQ - 1))] = src[e];
    }
    row_ci_tables<LOGL>(S, tw, tid);  // (ends with the barrier that also publishes xch)
    constexpr int NIT = kWaveCplx / 64;
    float2 va[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int n = it * 64 + lane;
        const int rr = n >> LOGL, k = n & (L - 1);
        const int kk = k <= L / 2 ? k : L - k;
        va[it] = xch[xch_slot(rr * KC + kk, kk, wave)];
    }
    __syncthreads();
    if (plane >= nplanes) return;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        // Z[k] = (X[k] + conj X[L-k]) + i conj(W_M^k) (X[k] - conj X[L-k]),  X = 0 beyond the band (KC = L/2 + 1: exactly
        // one of X[k], X[L-k] is kept, both only at k = L/2 where they coincide)
        const int n = it * 64 + lane;
        const int k = n & (L - 1);
        const float2 v = va[it], zero = make_float2(0.f, 0.f);
        const float2 a = k <= L / 2 ? v : zero;
        float2 b = k >= L / 2 ? v : zero;
        b.y = -b.y;
        const float2 sum = make_float2(a.x + b.x, a.y + b.y), dif = make_float2(a.x - b.x, a.y - b.y);
        float2 w = S.tw[k];
        w.y = -w.y;
        const float2 wd = cmul(w, dif);
        z[zsw(n)] = make_float2(sum.x - wd.y, sum.y + wd.x);  // sum + i w dif
    }
    wave_fft<true, LOGL>(z, S.tw, lane);
    float4 *dst4 = (float4 *)(grid + (plane * rows_per_plane + row0) * M);
#pragma unroll
    for (int q = 0; q < kWaveCplx / 128; ++q) {
        const int n2 = q * 64 + lane;
        const int na = 2 * n2, nb = 2 * n2 + 1;
        const int rr = na >> LOGL;
        const float2 a = z[zsw(rr * L + S.rev[na & (L - 1)])], b = z[zsw(rr * L + S.rev[nb & (L - 1)])];
        dst4[n2] = make_float4(a.x, a.y, b.x, b.y);
    }
}

// the column passes on [M][16 planes] tiles of one k2: blockIdx = (k2, u0 or j1, group); cg.NC = 16
__global__ void __launch_bounds__(kFftThreads)
adj_axis1_ci_kernel(ColGeom cg, const float2 *__restrict__ tw, const float2 *__restrict__ S, float2 *__restrict__ T)
{
    extern __shared__ float2 smem[];
    float2 *ltw = smem;
    float2 *buf = smem + cg.M / 2;
    const int tid = threadIdx.x;
    const int k2 = blockIdx.x, u0 = blockIdx.y;
    const int64_t group = blockIdx.z;
    stage_twiddles(ltw, tw, cg.M, tid);
    const int64_t rs = (int64_t)cg.KC * kCiQ;  // row stride of S' and T'
    const float2 *src = S + ((group * cg.M + u0) * cg.M) * rs + (int64_t)k2 * kCiQ;
    batched_fill<16>(cg.M << kCiLogQ, tid,
                     [&](int idx) { return src[(int64_t)(idx >> kCiLogQ) * rs + (idx & (kCiQ - 1))]; },
                     [&](int idx, float2 v) { buf[idx] = v; });
    lds_fft<false>(buf, ltw, cg, tid);
    float2 *dst = T + ((group * cg.M + u0) * cg.NB) * rs + (int64_t)k2 * kCiQ;
    for (int idx = tid; idx < (cg.NB << kCiLogQ); idx += kFftThreads) {
        const int q = idx & (kCiQ - 1), j1 = idx >> kCiLogQ;
        const int k1 = (j1 - cg.H) & (cg.M - 1);
        dst[(int64_t)j1 * rs + q] = buf[(brev_row(k1, cg.logM) << kCiLogQ) + q];
    }
}

template <bool XCOMPLEX, bool REAL_OUT>
__global__ void __launch_bounds__(kFftThreads)
adj_axis0_ci_kernel(ColGeom cg, const float2 *__restrict__ tw, const float2 *__restrict__ T, int64_t C, int64_t plane0,
                    int64_t nplanes, void *__restrict__ yv, const void *__restrict__ mult, int mult_kind)
{
    extern __shared__ float2 smem[];
    float2 *ltw = smem;
    float2 *buf = smem + cg.M / 2;
    const int tid = threadIdx.x;
    const int k2 = blockIdx.x, j1 = blockIdx.y;
    const int64_t group = blockIdx.z;
    stage_twiddles(ltw, tw, cg.M, tid);
    const int64_t rs = (int64_t)cg.KC * kCiQ;
    const float2 *src = T + (group * cg.M * cg.NB + j1) * rs + (int64_t)k2 * kCiQ;
    batched_fill<16>(cg.M << kCiLogQ, tid,
                     [&](int idx) { return src[(int64_t)(idx >> kCiLogQ) * cg.NB * rs + (idx & (kCiQ - 1))]; },
                     [&](int idx, float2 v) { buf[idx] = v; });
    lds_fft<false>(buf, ltw, cg, tid);

    const int H = cg.H, N = cg.N;
    const int k1 = j1 - H;
    const float f12 = phi_hat_inv_f(abs(k1), cg.param) * phi_hat_inv_f(k2, cg.param);
    constexpr int PPC = XCOMPLEX ? 2 : 1;
    for (int idx = tid; idx < (cg.NB << kCiLogQ); idx += kFftThreads) {
        const int q = idx & (kCiQ - 1), j0 = idx >> kCiLogQ;
        if (XCOMPLEX && (q & 1)) continue;  // (the odd tile column is the imaginary-part plane of the column to its left)
        const int64_t pl = group * kCiQ + q;  // plane inside the chunk
        if (pl >= nplanes) continue;
        const int64_t colg = (plane0 + pl) / PPC;
        const int64_t b = colg / C, c = colg - b * C;
        const int k0 = j0 - H;
        const int row = (brev_row(k0 & (cg.M - 1), cg.logM) << kCiLogQ) + q;
        const float2 fr = buf[row];
        float2 fi = make_float2(0.f, 0.f);
        if (XCOMPLEX) fi = buf[row + 1];
        const float fac = phi_hat_inv_f(abs(k0), cg.param) * f12;
        if (k0 < H && k1 < H && k2 < H) {
            float re = (fr.x + fi.y) * fac, im = (-fr.y + fi.x) * fac;
            const int64_t f = ((int64_t)(k0 + H) * N + (k1 + H)) * N + (k2 + H);
            apply_mult(mult, mult_kind, f, re, im);
            const int64_t o = (b * N * N * N + f) * C + c;
            if (REAL_OUT) ((float *)yv)[o] = re;
            else ((float2 *)yv)[o] = make_float2(re, im);
        }
        if (k2 >= 1 && k0 > -H && k1 > -H) {
            float re = (fr.x - fi.y) * fac, im = (fr.y + fi.x) * fac;
            const int64_t f = ((int64_t)(H - k0) * N + (H - k1)) * N + (H - k2);
            apply_mult(mult, mult_kind, f, re, im);
            const int64_t o = (b * N * N * N + f) * C + c;
            if (REAL_OUT) ((float *)yv)[o] = re;
            else ((float2 *)yv)[o] = make_float2(re, im);
        }
    }
}

template <bool XCOMPLEX>
__global__ void __launch_bounds__(kFftThreads)
fwd_axis0_ci_kernel(ColGeom cg, const float2 *__restrict__ tw, const void *__restrict__ xhat, int64_t C, int ppc,
                    int64_t plane0, int64_t nplanes, float2 *__restrict__ T)
{
    extern __shared__ float2 smem[];
    float2 *ltw = smem;
    float2 *buf = smem + cg.M / 2;
    const int tid = threadIdx.x;
    const int k2 = blockIdx.x, j1 = blockIdx.y;
    const int64_t group = blockIdx.z;
    stage_twiddles(ltw, tw, cg.M, tid);
    for (int idx = tid; idx < (cg.M << kCiLogQ); idx += kFftThreads) buf[idx] = make_float2(0.f, 0.f);
    __syncthreads();
    const int H = cg.H;
    const int k1 = j1 - H;
    const float f12 = 0.5f * phi_hat_inv_f(abs(k1), cg.param) * phi_hat_inv_f(k2, cg.param);
    batched_fill<8>(cg.NB << kCiLogQ, tid,
                    [&](int idx) {
                        const int q = idx & (kCiQ - 1), j0 = idx >> kCiLogQ;
                        const int64_t pl = group * kCiQ + q;
                        if (pl >= nplanes) return make_float2(0.f, 0.f);
                        const int64_t plane = plane0 + pl;
                        const int64_t colg = plane / ppc;
                        const int part = (int)(plane - colg * ppc);
                        const int64_t b = colg / C, c = colg - b * C;
                        const int k0 = j0 - H;
                        const float2 ap = band_value<XCOMPLEX>(xhat, b, c, C, cg.N, H, k0, k1, k2);
                        const float2 am = band_value<XCOMPLEX>(xhat, b, c, C, cg.N, H, -k0, -k1, -k2);
                        const float fac = phi_hat_inv_f(abs(k0), cg.param) * f12;
                        if (part == 0) return make_float2((am.x + ap.x) * fac, (am.y - ap.y) * fac);
                        return make_float2((am.y + ap.y) * fac, -(am.x - ap.x) * fac);
                    },
                    [&](int idx, float2 v) {
                        const int q = idx & (kCiQ - 1), j0 = idx >> kCiLogQ;
                        buf[(((j0 - H) & (cg.M - 1)) << kCiLogQ) + q] = v;
                    });
    lds_fft<true>(buf, ltw, cg, tid);
    const int64_t rs = (int64_t)cg.KC * kCiQ;
    float2 *dst = T + (group * cg.M * cg.NB + j1) * rs + (int64_t)k2 * kCiQ;
    for (int idx = tid; idx < (cg.M << kCiLogQ); idx += kFftThreads) {
        const int q = idx & (kCiQ - 1), u0 = idx >> kCiLogQ;
        dst[(int64_t)u0 * cg.NB * rs + q] = buf[(brev_row(u0, cg.logM) << kCiLogQ) + q];
    }
}

__global__ void __launch_bounds__(kFftThreads)
fwd_axis1_ci_kernel(ColGeom cg, const float2 *__restrict__ tw, const float2 *__restrict__ T, float2 *__restrict__ S)
{
    extern __shared__ float2 smem[];
    float2 *ltw = smem;
    float2 *buf = smem + cg.M / 2;
    const int tid = threadIdx.x;
    const int k2 = blockIdx.x, u0 = blockIdx.y;
    const int64_t group = blockIdx.z;
    stage_twiddles(ltw, tw, cg.M, tid);
    for (int idx = tid; idx < (cg.M << kCiLogQ); idx += kFftThreads) buf[idx] = make_float2(0.f, 0.f);
    __syncthreads();
    const int64_t rs = (int64_t)cg.KC * kCiQ;
    const float2 *src = T + ((group * cg.M + u0) * cg.NB) * rs + (int64_t)k2 * kCiQ;
    batched_fill<16>(cg.NB << kCiLogQ, tid,
                     [&](int idx) { return src[(int64_t)(idx >> kCiLogQ) * rs + (idx & (kCiQ - 1))]; },
                     [&](int idx, float2 v) {
                         const int q = idx & (kCiQ - 1), j1 = idx >> kCiLogQ;
                         buf[(((j1 - cg.H) & (cg.M - 1)) << kCiLogQ) + q] = v;
                     });
    lds_fft<true>(buf, ltw, cg, tid);
    float2 *dst = S + ((group * cg.M + u0) * cg.M) * rs + (int64_t)k2 * kCiQ;
    for (int idx = tid; idx < (cg.M << kCiLogQ); idx += kFftThreads) {
        const int q = idx & (kCiQ - 1), u1 = idx >> kCiLogQ;
        dst[(int64_t)u1 * rs + q] = buf[(brev_row(u1, cg.logM) << kCiLogQ) + q];
    }
}

// ---- column-interleaved <-> planar copies for several coefficient columns -------------------------------
// The reference stores spectra as [B, N^d, C] (columns innermost).  A column pass that handles one (point set,
// column) plane would touch elements 8 C bytes apart (measured at C = 64: the last pass twice as slow).  With C > 1
// the passes therefore work on a planar copy [column][N^d] (it lives in the grid buffer, which is free at that point
// of either pipeline) and these tiled transposes move between the two layouts: 64 frequencies x 32 columns per
// workgroup through LDS, contiguous runs on both sides.
template <typename T, bool TO_INTERLEAVED>
__global__ void __launch_bounds__(256)
column_layout_kernel(const T *__restrict__ src, T *__restrict__ dst, int64_t K /* N^d */, int64_t C, int64_t col0,
                     int64_t ncols)
{
    __shared__ T tile[32][65];
    const int64_t k0 = (int64_t)blockIdx.x * 64;
    const int64_t cl0 = (int64_t)blockIdx.y * 32;
    const int tid = threadIdx.x;
    if (TO_INTERLEAVED) {
        // read planar rows (contiguous in k), write interleaved (contiguous in the column index)
        for (int e = tid; e < 32 * 64; e += 256) {
            const int cl = e >> 6, k = e & 63;
            if (cl0 + cl < ncols && k0 + k < K) tile[cl][k] = src[(cl0 + cl) * K + k0 + k];
        }
        __syncthreads();
        for (int e = tid; e < 32 * 64; e += 256) {
            const int k = e >> 5, cl = e & 31;
            if (cl0 + cl < ncols && k0 + k < K) {
                const int64_t colg = col0 + cl0 + cl, b = colg / C, c = colg - b * C;
                dst[(b * K + k0 + k) * C + c] = tile[cl][k];
            }
        }
    } else {
        for (int e = tid; e < 32 * 64; e += 256) {
            const int k = e >> 5, cl = e & 31;
            if (cl0 + cl < ncols && k0 + k < K) {
                const int64_t colg = col0 + cl0 + cl, b = colg / C, c = colg - b * C;
                tile[cl][k] = src[(b * K + k0 + k) * C + c];
            }
        }
        __syncthreads();
        for (int e = tid; e < 32 * 64; e += 256) {
            const int cl = e >> 6, k = e & 63;
            if (cl0 + cl < ncols && k0 + k < K) dst[(cl0 + cl) * K + k0 + k] = tile[cl][k];
        }
    }
}

// row stride (complex numbers) of the compact half spectrum and of T: the N/2+1 kept columns rounded up so that the
// runs of a tile start on 32- / 64-byte boundaries
int compact_stride(const Geom &g)
{
    static const int forced = [] {
        const char *env = std::getenv("NFFT_HIP_KS_ALIGN");
        const int v = env ? std::atoi(env) : 0;
        return (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) ? v : 0;
    }();
    // measured (profiles/r02_experiments.md): 8 at M = 512 (129 -> 136 columns), 4 at M = 256 (65 -> 68: padding to
    // whole 16-column tiles, 80, costs more bytes than the alignment wins)
    const int a = forced ? forced : (g.M >= 512 ? 8 : 4);
    return (g.N / 2 + 1 + a - 1) / a * a;
}

ColGeom make_col_geom(const Geom &g, bool two_buffers, bool compact, int64_t planes = 0)
{
    ColGeom cg;
    cg.M = g.M;
    cg.logM = 0;
    while ((1 << cg.logM) < g.M) ++cg.logM;
    cg.N = g.N;
    cg.H = g.N / 2;
    cg.Mh = g.M / 2 + 1;
    cg.KC = cg.H + 1;
    cg.KS = compact_stride(g);
    cg.SR = compact ? cg.KS : cg.Mh;
    cg.NB = g.N + 1;
    cg.two_d = g.dim == 2;
    cg.NBm = cg.two_d ? 1 : cg.NB;
    // tile of NC columns: M * NC * 8 bytes per buffer.  32 KB tiles (NC = 8 at M = 512) let four workgroups share a
    // CU and overlap their load / transform / store phases: 6 % faster than 64 KB tiles, 16 KB tiles (64-byte row
    // segments) lose 30 %; NFFT_HIP_COL_NC overrides.
    int nc = 16;
    const int64_t cap = g.M >= 1024 ? 65536 : 32768;
    while (nc > 1 && (int64_t)g.M * nc * 8 * (two_buffers ? 2 : 1) > cap) nc >>= 1;
    // 2-D: a launch is (column tiles) x (planes) workgroups -- a single 256^2 grid is 5 tiles of 16 columns, a latency chain
    // (config C2: 95 us per adjoint + forward with 16-column tiles, 78 with 4-column ones, 89 through rocFFT): narrower tiles
    // until the launch has ~64 workgroups
    if (g.dim == 2 && planes > 0)
        while (nc > 4 && ((g.N / 2 + 1 + nc - 1) / nc) * planes < 64) nc >>= 1;
    if (const char *env = std::getenv("NFFT_HIP_COL_NC")) {
        const int v = std::atoi(env);
        if ((v == 4 || v == 8 || v == 16) && (int64_t)g.M * v * 8 * (two_buffers ? 2 : 1) <= 131072) nc = v;
    }
    cg.NC = nc;
    cg.logNC = 0;
    while ((1 << cg.logNC) < nc) ++cg.logNC;
    cg.param = 1.047197551196597746f * (float)g.m / ((float)g.N * (float)g.N);
    static const bool pairs_off = [] {
        const char *env = std::getenv("NFFT_HIP_COL_XCD");
        return env && env[0] == '0';
    }();
    cg.TG = pairs_off || nc >= 16 ? 1 : 16 / nc;
    return cg;
}

// column tiles of a launch: whole groups of TG (column_tile_of_block)
static inline unsigned col_tiles(const ColGeom &cg, int columns = 0)
{
    const int t = ((columns > 0 ? columns : cg.KC) + cg.NC - 1) / cg.NC;
    return (unsigned)((t + cg.TG - 1) / cg.TG * cg.TG);
}

size_t col_lds_bytes(const ColGeom &cg, bool two_buffers)
{
    return (size_t)(cg.M / 2 + (int64_t)cg.M * cg.NC * (two_buffers ? 2 : 1)) * sizeof(float2);
}

// dynamic LDS above 64 KB has to be opted into per kernel
template <typename K>
void allow_lds(K kernel, size_t bytes)
{
    if (bytes > 48 * 1024) (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

} // namespace

bool colfft_supported(const Geom &g)
{
    // 3-D, and (round 4) 2-D grids the own row passes take: row pass + ONE column pass with the roll-off instead of rocFFT's
    // two kernels + the roll-off kernel (three launches per direction -> two; NFFT_HIP_COLFFT_2D=0: rocFFT as before)
    static const bool no2d = [] {
        const char *env = std::getenv("NFFT_HIP_COLFFT_2D");
        return env && env[0] == '0';
    }();
    if (g.dim == 2) return !no2d && g.M >= 128 && g.M <= 1024 && (g.M & (g.M - 1)) == 0;
    return g.dim == 3 && g.M >= 16 && g.M <= 1024 && (g.M & (g.M - 1)) == 0;
}

int64_t colfft_scratch_bytes(const Geom &g, int64_t nplanes)
{
    // T[plane][M][N+1][KS] complex (+ room that used to hold a twiddle table; the workspace contract keeps its size)
    // (2-D: no T -- the column pass reads / writes the compact half spectrum itself)
    return align_up(nplanes * (int64_t)g.M * (g.dim == 2 ? 1 : g.N + 1) * compact_stride(g) * 8, 256) + align_up((int64_t)g.M * 4, 256);
}

// Calls f(log2 M, log2 NC) as integral constants for the sizes the column kernels are specialised for -- 512^3 grids
// with their default tile widths (one and two tile buffers) -- and f(0, 0), the run-time version, otherwise
// (NFFT_HIP_COL_GENERIC=1: always).  Compile-time sizes take 6-14 % off a pass at M = 512 (C3: 110 / 90 / 110 / 120 ->
// 103 / 84 / 98 / 103 us); the same specialisation for 256^3 grids measured 6 % SLOWER (C4-share column passes 16.55 ->
// 17.5 ms) and is not instantiated.
template <typename F>
static void col_dispatch(const ColGeom &cg, F &&f)
{
    static const bool generic = [] {
        const char *env = std::getenv("NFFT_HIP_COL_GENERIC");
        return env && env[0] == '1';
    }();
    using std::integral_constant;
    if (!generic && cg.logM == 9 && cg.logNC == 3) return f(integral_constant<int, 9>{}, integral_constant<int, 3>{});
    if (!generic && cg.logM == 9 && cg.logNC == 2) return f(integral_constant<int, 9>{}, integral_constant<int, 2>{});
    return f(integral_constant<int, 0>{}, integral_constant<int, 0>{});
}

int launch_colfft_adjoint(const Geom &g, const float2 *spec, bool compact, void *scratch, int64_t scratch_planes,
                          int64_t C, int x_is_complex, int real_output, int64_t plane0, int64_t nplanes, void *y,
                          const void *mult, int mult_kind, hipStream_t stream)
{
    if (nplanes <= 0) return 0;
    const float2 *T = (const float2 *)scratch;
    const float2 *tw = twiddle_table(g.M);
    if (!tw) { set_error("no twiddle table for this grid size"); return 4; }
    if (g.dim == 2) {
        if (!compact) { set_error("2-D column passes need the compact half spectrum of the own row passes"); return 4; }
        T = spec;  // S[plane][u1][KS]: the only column pass transforms it in place of T
    } else {
        const ColGeom cg = make_col_geom(g, false, compact);
        const dim3 grid(col_tiles(cg), g.M, (unsigned)nplanes);
        const size_t lds = col_lds_bytes(cg, false);
        col_dispatch(cg, [&](auto lm, auto ln) {
            constexpr int LM = decltype(lm)::value, LN = decltype(ln)::value;
            allow_lds(adj_axis1_kernel<LM, LN>, lds);
            hipLaunchKernelGGL((adj_axis1_kernel<LM, LN>), grid, dim3(kFftThreads), lds, stream, cg, tw, spec, (float2 *)scratch);
        });
    }
    {
        const bool two = x_is_complex != 0;
        const int ppc = two ? 2 : 1;
        const int64_t col0 = plane0 / ppc, ncols = nplanes / ppc;
        const ColGeom cg = make_col_geom(g, two, compact, ncols);
        const dim3 grid(col_tiles(cg), cg.NBm, (unsigned)ncols);
        const size_t lds = col_lds_bytes(cg, two);
        col_dispatch(cg, [&](auto lm, auto ln) {
            constexpr int LM = decltype(lm)::value, LN = decltype(ln)::value;
            auto go = [&](auto kernel) {
                allow_lds(kernel, lds);
                hipLaunchKernelGGL(kernel, grid, dim3(kFftThreads), lds, stream, cg, tw, T, C, col0, y, mult, mult_kind);
            };
            if (two) {
                if (real_output) go(adj_axis0_kernel<LM, LN, true, true>);
                else go(adj_axis0_kernel<LM, LN, true, false>);
            } else {
                if (real_output) go(adj_axis0_kernel<LM, LN, false, true>);
                else go(adj_axis0_kernel<LM, LN, false, false>);
            }
        });
    }
    NFFT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_colfft_forward(const Geom &g, const void *xhat, void *scratch, int64_t scratch_planes, int64_t C,
                          int x_is_complex, int real_output, int64_t plane0, int64_t nplanes, float2 *spec, bool compact,
                          hipStream_t stream)
{
    if (nplanes <= 0) return 0;
    float2 *T = (float2 *)scratch;
    const float2 *tw = twiddle_table(g.M);
    if (!tw) { set_error("no twiddle table for this grid size"); return 4; }
    if (g.dim == 2) {
        if (!compact) { set_error("2-D column passes need the compact half spectrum of the own row passes"); return 4; }
        T = spec;  // the only column pass writes the compact half spectrum the row pass reads
    }
    const ColGeom cg = make_col_geom(g, false, compact, nplanes);
    const size_t lds = col_lds_bytes(cg, false);
    const int ppc = real_output ? 1 : 2;
    col_dispatch(cg, [&](auto lm, auto ln) {
        constexpr int LM = decltype(lm)::value, LN = decltype(ln)::value;
        {
            const dim3 grid(col_tiles(cg), cg.NBm, (unsigned)nplanes);
            if (x_is_complex) {
                allow_lds(fwd_axis0_kernel<LM, LN, true>, lds);
                hipLaunchKernelGGL((fwd_axis0_kernel<LM, LN, true>), grid, dim3(kFftThreads), lds, stream, cg, tw, xhat, C, ppc, plane0, T);
            } else {
                allow_lds(fwd_axis0_kernel<LM, LN, false>, lds);
                hipLaunchKernelGGL((fwd_axis0_kernel<LM, LN, false>), grid, dim3(kFftThreads), lds, stream, cg, tw, xhat, C, ppc, plane0, T);
            }
        }
        if (g.dim != 2) {
            const dim3 grid(col_tiles(cg, compact ? cg.KC : cg.Mh), g.M, (unsigned)nplanes);
            allow_lds(fwd_axis1_kernel<LM, LN>, lds);
            hipLaunchKernelGGL((fwd_axis1_kernel<LM, LN>), grid, dim3(kFftThreads), lds, stream, cg, tw, T, spec);
        }
    });
    NFFT_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- column-innermost pipeline (several coefficient columns): launchers ---------------------------------------------
bool colfft_ci_supported(const Geom &g) { return g.dim == 3 && colfft_supported(g) && g.M >= 128 && g.M <= 1024; }

// planes the buffers of a chunk of `nplanes` planes must hold in column-innermost mode (whole groups of 16)
int64_t colfft_ci_planes(int64_t nplanes) { return (nplanes + kCiQ - 1) / kCiQ * kCiQ; }

static ColGeom make_ci_geom(const Geom &g)
{
    ColGeom cg = make_col_geom(g, false, true);
    cg.NC = kCiQ;
    cg.logNC = kCiLogQ;
    return cg;
}

template <int LOGL>
static int launch_rows_ci_t(bool c2r, const Geom &g, int64_t nplanes, const float2 *tw, const void *in, void *out,
                            hipStream_t stream)
{
    constexpr int RP = kWaveCplx >> LOGL;
    const int64_t rows = (int64_t)g.M * g.M;
    const dim3 blocks((unsigned)(rows / RP), (unsigned)((nplanes + kCiQ - 1) / kCiQ));
    const size_t lds = sizeof(RowCiLds<LOGL>);
    static DeviceOnce attr_done;
    if (attr_done.first_use()) {
        NFFT_HIP_CHECK(hipFuncSetAttribute((const void *)row_r2c_ci_kernel<LOGL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        NFFT_HIP_CHECK(hipFuncSetAttribute((const void *)row_c2r_ci_kernel<LOGL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done.mark();
    }
    if (c2r)
        hipLaunchKernelGGL((row_c2r_ci_kernel<LOGL>), blocks, dim3(kCiQ * 64), lds, stream, rows, nplanes, tw, (const float2 *)in,
                           (float *)out);
    else
        hipLaunchKernelGGL((row_r2c_ci_kernel<LOGL>), blocks, dim3(kCiQ * 64), lds, stream, rows, nplanes, tw, (const float *)in,
                           (float2 *)out);
    NFFT_HIP_CHECK(hipGetLastError());
    return 0;
}

static int launch_rows_ci(bool c2r, const Geom &g, int64_t nplanes, const void *in, void *out, hipStream_t stream)
{
    if (nplanes <= 0) return 0;
    const float2 *tw = twiddle_table(g.M);
    if (!tw) { set_error("no twiddle table for this grid size"); return 4; }
    switch (g.M) {
    case 128: return launch_rows_ci_t<6>(c2r, g, nplanes, tw, in, out, stream);
    case 256: return launch_rows_ci_t<7>(c2r, g, nplanes, tw, in, out, stream);
    case 512: return launch_rows_ci_t<8>(c2r, g, nplanes, tw, in, out, stream);
    case 1024: return launch_rows_ci_t<9>(c2r, g, nplanes, tw, in, out, stream);
    }
    set_error("row passes support M = 128 .. 1024");
    return 1;
}

int launch_row_r2c_ci(const Geom &g, const float *grid, int64_t nplanes, float2 *spec, hipStream_t stream)
{
    return launch_rows_ci(false, g, nplanes, grid, spec, stream);
}
int launch_row_c2r_ci(const Geom &g, const float2 *spec, int64_t nplanes, float *grid, hipStream_t stream)
{
    return launch_rows_ci(true, g, nplanes, spec, grid, stream);
}

// S' -> T' -> y [B, N^3, C] (the planes plane0 .. plane0 + nplanes of the call), all in column-innermost order
int launch_colfft_adjoint_ci(const Geom &g, const float2 *spec, void *scratch, int64_t C, int x_is_complex,
                             int real_output, int64_t plane0, int64_t nplanes, void *y, const void *mult, int mult_kind,
                             hipStream_t stream)
{
    if (nplanes <= 0) return 0;
    float2 *T = (float2 *)scratch;
    const float2 *tw = twiddle_table(g.M);
    if (!tw) { set_error("no twiddle table for this grid size"); return 4; }
    const ColGeom cg = make_ci_geom(g);
    const size_t lds = col_lds_bytes(cg, false);
    const unsigned groups = (unsigned)((nplanes + kCiQ - 1) / kCiQ);
    allow_lds(adj_axis1_ci_kernel, lds);
    hipLaunchKernelGGL(adj_axis1_ci_kernel, dim3(cg.KC, g.M, groups), dim3(kFftThreads), lds, stream, cg, tw, spec, T);
    auto go = [&](auto kernel) {
        allow_lds(kernel, lds);
        hipLaunchKernelGGL(kernel, dim3(cg.KC, cg.NB, groups), dim3(kFftThreads), lds, stream, cg, tw, T, C, plane0, nplanes, y,
                           mult, mult_kind);
    };
    if (x_is_complex) {
        if (real_output) go(adj_axis0_ci_kernel<true, true>);
        else go(adj_axis0_ci_kernel<true, false>);
    } else {
        if (real_output) go(adj_axis0_ci_kernel<false, true>);
        else go(adj_axis0_ci_kernel<false, false>);
    }
    NFFT_HIP_CHECK(hipGetLastError());
    return 0;
}

// xhat [B, N^3, C] -> T' -> S'
int launch_colfft_forward_ci(const Geom &g, const void *xhat, float2 *spec, void *scratch, int64_t C, int x_is_complex,
                             int real_output, int64_t plane0, int64_t nplanes, hipStream_t stream)
{
    if (nplanes <= 0) return 0;
    float2 *T = (float2 *)scratch;
    const float2 *tw = twiddle_table(g.M);
    if (!tw) { set_error("no twiddle table for this grid size"); return 4; }
    const ColGeom cg = make_ci_geom(g);
    const size_t lds = col_lds_bytes(cg, false);
    const unsigned groups = (unsigned)((nplanes + kCiQ - 1) / kCiQ);
    const int ppc = real_output ? 1 : 2;
    if (x_is_complex) {
        allow_lds(fwd_axis0_ci_kernel<true>, lds);
        hipLaunchKernelGGL((fwd_axis0_ci_kernel<true>), dim3(cg.KC, cg.NB, groups), dim3(kFftThreads), lds, stream, cg, tw, xhat, C,
                           ppc, plane0, nplanes, T);
    } else {
        allow_lds(fwd_axis0_ci_kernel<false>, lds);
        hipLaunchKernelGGL((fwd_axis0_ci_kernel<false>), dim3(cg.KC, cg.NB, groups), dim3(kFftThreads), lds, stream, cg, tw, xhat, C,
                           ppc, plane0, nplanes, T);
    }
    allow_lds(fwd_axis1_ci_kernel, lds);
    hipLaunchKernelGGL(fwd_axis1_ci_kernel, dim3(cg.KC, g.M, groups), dim3(kFftThreads), lds, stream, cg, tw, T, spec);
    NFFT_HIP_CHECK(hipGetLastError());
    return 0;
}

bool rowfft_supported(const Geom &g)
{
    // one wave per row: M/2 in {64, 128, 256, 512} complex points; part of the pruned column-pass pipeline
    return colfft_supported(g) && g.M >= 128 && g.M <= 1024;
}

template <int LOGL>
static void launch_rows_t(bool c2r, const Geom &g, int64_t nrows, const float2 *tw, const void *in, void *out,
                          hipStream_t stream)
{
    const int64_t per_wg = (int64_t)kRowWaves * (c2r ? kC2rRounds : kR2cRounds) * (kWaveCplx >> LOGL);
    const dim3 blocks((unsigned)((nrows + per_wg - 1) / per_wg));
    if (c2r)
        hipLaunchKernelGGL((row_c2r_kernel<LOGL>), blocks, dim3(kRowWaves * 64), 0, stream, g.N / 2 + 1, compact_stride(g), nrows, tw,
                           (const float2 *)in, (float *)out);
    else
        hipLaunchKernelGGL((row_r2c_kernel<LOGL>), blocks, dim3(kRowWaves * 64), 0, stream, g.N / 2 + 1, compact_stride(g), nrows, tw,
                           (const float *)in, (float2 *)out);
}

static int launch_rows(bool c2r, const Geom &g, int64_t nplanes, const float2 *tw, const void *in, void *out,
                       hipStream_t stream)
{
    const int64_t nrows = nplanes * g.Ma[0] * g.Ma[1];  // (2-D: M rows per plane)
    // (the kept band of a row is N/2 + 1 of its M/2 = N complex points: what the C2R pass's single load per element assumes)
    switch (g.M) {
    case 128: launch_rows_t<6>(c2r, g, nrows, tw, in, out, stream); break;
    case 256: launch_rows_t<7>(c2r, g, nrows, tw, in, out, stream); break;
    case 512: launch_rows_t<8>(c2r, g, nrows, tw, in, out, stream); break;
    case 1024: launch_rows_t<9>(c2r, g, nrows, tw, in, out, stream); break;
    default: set_error("row passes support M = 128 .. 1024"); return 1;
    }
    NFFT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_row_r2c(const Geom &g, const float *grid, void *scratch, int64_t scratch_planes, int64_t nplanes,
                   float2 *spec, hipStream_t stream)
{
    if (nplanes <= 0) return 0;
    const float2 *tw = twiddle_table(g.M);
    if (!tw) { set_error("no twiddle table for this grid size"); return 4; }
    return launch_rows(false, g, nplanes, tw, grid, spec, stream);
}

int launch_row_c2r(const Geom &g, const float2 *spec, void *scratch, int64_t scratch_planes, int64_t nplanes,
                   float *grid, hipStream_t stream)
{
    if (nplanes <= 0) return 0;
    const float2 *tw = twiddle_table(g.M);
    if (!tw) { set_error("no twiddle table for this grid size"); return 4; }
    return launch_rows(true, g, nplanes, tw, spec, grid, stream);
}

// planar [ncols][N^d] <-> interleaved [B, N^d, C] for the columns col0 .. col0 + ncols (global column = b * C + c)
int launch_column_layout(bool to_interleaved, const void *src, void *dst, int64_t K, int64_t C, int64_t col0,
                         int64_t ncols, int elem_bytes, hipStream_t stream)
{
    if (ncols <= 0 || K <= 0) return 0;
    const dim3 blocks((unsigned)((K + 63) / 64), (unsigned)((ncols + 31) / 32));
    if (elem_bytes == 8) {
        if (to_interleaved)
            hipLaunchKernelGGL((column_layout_kernel<float2, true>), blocks, dim3(256), 0, stream, (const float2 *)src,
                               (float2 *)dst, K, C, col0, ncols);
        else
            hipLaunchKernelGGL((column_layout_kernel<float2, false>), blocks, dim3(256), 0, stream, (const float2 *)src,
                               (float2 *)dst, K, C, col0, ncols);
    } else {
        if (to_interleaved)
            hipLaunchKernelGGL((column_layout_kernel<float, true>), blocks, dim3(256), 0, stream, (const float *)src,
                               (float *)dst, K, C, col0, ncols);
        else
            hipLaunchKernelGGL((column_layout_kernel<float, false>), blocks, dim3(256), 0, stream, (const float *)src,
                               (float *)dst, K, C, col0, ncols);
    }
    NFFT_HIP_CHECK(hipGetLastError());
    return 0;
}

} // namespace nfft
