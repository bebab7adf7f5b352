// Register-tile ("output-stationary") spreading kernel for 3-D grids -- no atomics anywhere.
//
// Same result as spread.hip / the reference's adjoint_window_convolution kernels
// (csrc/cuda/spatial_window_operations.cu:103-211), different ownership: ONE WAVE owns a kSub x kSub column of
// grid cells (lane = cell in (axis 1, axis 2)) and sweeps it along axis 0 with NP = TC + 2m+1 fp32 accumulators per
// lane (one per resident plane) held in REGISTERS.  For every point whose window reaches the column (its own
// sub-block and the 8 neighbours in the point plan) the wave evaluates the in-plane weight of each lane once
// (one v_exp_f32; lanes outside the (2m+2)^2 window get 0) and issues 2m+2 FMAs against wave-uniform axis-0
// weights.  A plane that can receive no further taps is stored with plain, row-contiguous stores; the
// accumulators slide down by register moves.
//
// Why: on gfx950 the LDS float atomic is unusable (ds_add_f32 ~193 cycles / wave instruction) and the native
// 64-bit one costs 8.2 cycles of the CU-wide LDS pipe per 64 taps, which bounds spread.hip at ~3 ms for
// 1e10 taps; the VALU has 16x that rate.  Here every cell is written exactly once by its owner, so the grid
// needs no zero-fill, no global atomics, and the result is bitwise reproducible run to run.
// Cost: every point is visited by each of the ~4.5 columns its window overlaps.
#include <climits>

#include "common.h"
#include "kernels.h"
#include "window.h"

namespace nfft {

namespace {

template <int W>
struct RegCfg {
    static constexpr TileCfg tc = tile_cfg(3, W);
    static constexpr int TC = tc.TC;
    static constexpr int NP = TC + W - 1;
    static constexpr int MC = W / 2 - 1;  // cutoff m
    static_assert(MC + 1 <= kSub, "window must not reach beyond the neighbouring sub-blocks");
};

// The accumulators are a clang vector value (not an array): every element access has a compile-time index, so
// they live in VGPRs; an array here ends up in scratch memory.
template <int N>
using vecf = float __attribute__((ext_vector_type(N)));

template <int W>
__global__ void __launch_bounds__(64)
spread_reg_kernel(const Geom g, const int *__restrict__ tile_offsets, const float *__restrict__ spos,
                  const float *__restrict__ xs, const int64_t n, const int Cr, const int plane0,
                  float *__restrict__ grid, const int nsegr, const int seg_chunks)
{
    using C = RegCfg<W>;
    constexpr int TC = C::TC, NP = C::NP, m = C::MC;
    const int lane = threadIdx.x;
    const int row = lane >> 3, col = lane & 7;
    const float rowf = (float)row, colf = (float)col;

    // block -> (sub-block column (gs1, gs2), segment along axis 0); plane -> (batch, real column)
    const int nsb1 = g.M / kSub, nsb2 = g.M / kSub;
    int bid = blockIdx.x;
    const int seg = bid % nsegr; bid /= nsegr;
    const int gs2 = bid % nsb2;
    const int gs1 = bid / nsb2;
    const int plane_local = blockIdx.y;
    const int plane = plane0 + plane_local;
    const int b = plane / Cr;
    const int cr = plane - b * Cr;

    const int nt0 = g.nta[0];
    const int kb = seg * seg_chunks;
    const int ke = min(nt0, kb + seg_chunks);
    if (kb >= ke) return;
    // planes owned by this wave: [z_lo, z_hi)
    const int z_lo = kb * TC;
    const int z_hi = min(ke * TC, g.M);

    const float sc = win_exp_scale(m);
    float norm = win_norm(m);
    norm = norm * norm * norm;
    const float *const xcol = xs + (int64_t)cr * n;
    float *const gcol = grid + (int64_t)plane_local * g.cells + (int64_t)(gs1 * kSub + row) * g.M + (gs2 * kSub + col);

    vecf<NP> acc = 0.0f;

    // Sweep every chunk holding a cell in [z_lo - (m+1), z_hi + m): exactly the points that can reach an owned
    // plane (plus chunk-mates that cannot; their taps only touch accumulators that are never stored).  The sweep
    // runs on an unwrapped plane axis so that the periodic neighbours of the first / last chunk line up.
    const int u_end = z_hi + m;
    int u = z_lo - (m + 1);
    int pb = INT_MIN;  // acc[i] holds the (unwrapped) plane pb + i; set by the first chunk
    while (u < u_end) {
        int wcell = u % g.M;
        if (wcell < 0) wcell += g.M;
        const int k = wcell / TC;
        const int cbase = k * TC;                       // first cell (wrapped) of the chunk
        const int clen = min(TC, g.M - cbase);          // the last chunk of the axis may be short
        const int ucs = u - (wcell - cbase);            // unwrapped position of that first cell
        const int want_pb = ucs - m;                    // plane acc[0] must hold while the chunk is processed
        if (pb == INT_MIN) pb = want_pb;
        // retire planes below want_pb: store the owned ones, slide the accumulators down
        while (pb < want_pb) {
            if (pb >= z_lo && pb < z_hi) gcol[(int64_t)pb * g.M * g.M] = acc[0];
#pragma unroll
            for (int i = 0; i + 1 < NP; ++i) acc[i] = acc[i + 1];
            acc[NP - 1] = 0.0f;
            ++pb;
        }
        u = ucs + clen;
        // points of this chunk in the 3 x 3 sub-blocks around the owned column
        for (int d1 = -1; d1 <= 1; ++d1) {
            const int nb1 = gs1 + d1 < 0 ? gs1 + d1 + nsb1 : (gs1 + d1 >= nsb1 ? gs1 + d1 - nsb1 : gs1 + d1);
            const int J1 = nb1 / g.sb1, s1 = nb1 - J1 * g.sb1;
            for (int d2 = -1; d2 <= 1; ++d2) {
                const int nb2 = gs2 + d2 < 0 ? gs2 + d2 + nsb2 : (gs2 + d2 >= nsb2 ? gs2 + d2 - nsb2 : gs2 + d2);
                const int J2 = nb2 / g.sb2, s2 = nb2 - J2 * g.sb2;
                const int tile = b * g.tiles_per_batch + (J1 * g.nta[2] + J2) * g.np0 + k;
                const int fine = tile * g.SB + s1 * g.sb2 + s2;
                const int s = tile_offsets[fine], e = tile_offsets[fine + 1];
                // cell coordinates of that sub-block relative to the owned column
                const int o1 = nb1 * kSub - d1 * kSub, o2 = nb2 * kSub - d2 * kSub;
                for (int j0 = s; j0 < e; j0 += 64) {
                    const int j = j0 + lane;
                    int c0 = 0, c1 = 0, c2 = 0;
                    float f0 = 0.f, f1 = 0.f, f2 = 0.f, xv = 0.f;
                    bool hit = false;
                    if (j < e) {
                        const f32x4 rec = *(const f32x4 *)(spos + (int64_t)j * 4);  // plan record {p0, p1, p2, x}
                        split_cell(rec.x, g.M, c0, f0);
                        split_cell(rec.y, g.M, c1, f1);
                        split_cell(rec.z, g.M, c2, f2);
                        xv = xcol[j] * norm;
                        c1 -= o1;  // in [-kSub, 2 kSub): row of the point's cell relative to the owned rows
                        c2 -= o2;
                        hit = (c1 + m + 1 >= 0) && (c1 - m < kSub) && (c2 + m + 1 >= 0) && (c2 - m < kSub);
                    }
                    // distances of the lanes' cells from the point are (f + c) - row / col
                    const float g1 = f1 + (float)c1, g2 = f2 + (float)c2;
                    const int p0v = c0 - cbase;  // plane of tap 0 relative to acc[0]: (c0 - m) - (cbase - m)
                    unsigned long long todo = __ballot(hit);
                    while (todo) {
                        const int q = __builtin_ctzll(todo);
                        todo &= todo - 1;
                        // axis-0 weight of resident plane `lane`: tap index lane - p0, zero outside the window, so the
                        // accumulation below needs no branch on p0 (a switch over register offsets makes the compiler
                        // copy all accumulators on every path)
                        const int l0 = lane - readlane_i(p0v, q);
                        const float d0 = readlane_f(f0, q) + (float)(m - l0);
                        float psi0 = __builtin_amdgcn_exp2f(sc * d0 * d0);
                        psi0 = (unsigned)l0 < (unsigned)W ? psi0 : 0.0f;
                        const float e1 = readlane_f(g1, q) - rowf;
                        const float e2 = readlane_f(g2, q) - colf;
                        // lane is inside the window iff its tap index row - c1 + m lies in [0, 2m+1]
                        const int l1 = row - readlane_i(c1, q) + m;
                        const int l2 = col - readlane_i(c2, q) + m;
                        const bool in = (unsigned)l1 < (unsigned)W && (unsigned)l2 < (unsigned)W;
                        float w = __builtin_amdgcn_exp2f(sc * fmaf(e1, e1, e2 * e2)) * readlane_f(xv, q);
                        w = in ? w : 0.0f;
#pragma unroll
                        for (int i = 0; i < NP; ++i) acc[i] = fmaf(readlane_f(psi0, i), w, acc[i]);
                    }
                }
            }
        }
    }
    // drain: everything still resident that this wave owns
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int z = pb + i;
        if (z >= z_lo && z < z_hi) gcol[(int64_t)z * g.M * g.M] = acc[i];
    }
}

} // namespace

bool spread_reg_supported(const Geom &g)
{
    // sub-blocks must tile the grid and the plan's tiles; the 3 x 3 neighbourhood must cover the window reach
    return g.dim == 3 && g.SB > 1 && g.m + 1 <= kSub && g.M / kSub >= 3;
}

template <int W>
static int launch_reg_t(const Geom &g, const int *to, const float *spos, const float *xs, int64_t n, int64_t Cr,
                        int64_t plane0, int64_t nplanes, float *grid, hipStream_t stream)
{
    const int nsb = g.M / kSub;
    // segments: enough waves to fill the chip, few enough that the two extra halo chunks stay cheap
    int seg_chunks = 32;
    if (seg_chunks > g.nta[0]) seg_chunks = g.nta[0];
    const int nsegr = (g.nta[0] + seg_chunks - 1) / seg_chunks;
    const dim3 blocks((unsigned)(nsb * nsb * nsegr), (unsigned)nplanes);
    hipLaunchKernelGGL((spread_reg_kernel<W>), blocks, dim3(64), 0, stream, g, to, spos, xs, n, (int)Cr, (int)plane0,
                       grid, nsegr, seg_chunks);
    NFFT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_spread_reg(const Geom &g, const PlanLayout &L, const void *plan, const float *xs, int64_t n, int64_t Cr,
                      int64_t plane0, int64_t nplanes, float *grid, hipStream_t stream)
{
    const char *base = (const char *)plan;
    const int *to = (const int *)(base + L.off_offsets);
    const float *spos = (const float *)(base + L.off_spos);
    if (nplanes <= 0) return 0;
    switch (g.m) {
    case 1: return launch_reg_t<4>(g, to, spos, xs, n, Cr, plane0, nplanes, grid, stream);
    case 2: return launch_reg_t<6>(g, to, spos, xs, n, Cr, plane0, nplanes, grid, stream);
    case 3: return launch_reg_t<8>(g, to, spos, xs, n, Cr, plane0, nplanes, grid, stream);
    case 4: return launch_reg_t<10>(g, to, spos, xs, n, Cr, plane0, nplanes, grid, stream);
    case 5: return launch_reg_t<12>(g, to, spos, xs, n, Cr, plane0, nplanes, grid, stream);
    case 6: return launch_reg_t<14>(g, to, spos, xs, n, Cr, plane0, nplanes, grid, stream);
    case 7: return launch_reg_t<16>(g, to, spos, xs, n, Cr, plane0, nplanes, grid, stream);
    }
    set_error("register-tile spreading supports cutoff 1..7");
    return 1;
}

} // namespace nfft
