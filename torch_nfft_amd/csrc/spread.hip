// Spreading (adjoint gridding) for gfx950.
//
// Computes what the reference's real_/complex_adjoint_window_convolution_kernel
// (csrc/cuda/spatial_window_operations.cu:103-211) computes,
//     g[(b, c), (shift_i + l) mod M] += x[i, c] * prod_k psi_k(i, l_k),   l in [0, 2m+2)^d,
// but organised for CDNA4:
//   * points arrive counting-sorted by (pencil, chunk) tile (binning.hip);
//   * one workgroup sweeps a segment of a T1 x T2 pencil along axis 0.  NP = TC + 2m+1 padded planes are
//     resident in LDS; a chunk of TC planes worth of points is accumulated, the finished planes are written
//     out, the remaining 2m+1 planes slide down and the sweep continues -- the halo along axis 0 never
//     leaves LDS;
//   * one wave handles one point at a time: its lanes are the (l1, l2) taps of the point, the l0 taps are an
//     unrolled loop with compile-time LDS offsets; the window is evaluated in registers (one v_exp_f32 per
//     lane and pass plus one for axis 0) instead of being read back from HBM (reference: point_psi, 1.2 GB
//     at N=256, n=1e7);
//   * accumulation is ds_add_f64 on 8-byte cells.  Measured on MI355X (scripts/ubench/lds_ops.hip):
//     ds_add_f32 193 cycles per wave instruction (serialised), ds_add_f64 8.2, ds_add_u32 4.3 -- the 32-bit
//     float LDS atomic is unusable, the 64-bit one is native.  Side effect: tile sums are exact to fp64;
//   * finished planes leave LDS as contiguous row segments of global_atomic_add_f32 (one padded row =
//     one <=256-byte run of a wave instruction), so HBM sees line-sized updates, never scattered dwords.

#include "common.h"
#include "kernels.h"
#include "window.h"

namespace nfft {

template <int DIM>
constexpr int spread_threads() { return DIM == 3 ? 1024 : 256; }

template <int DIM, int W>
__global__ void __launch_bounds__((spread_threads<DIM>()))
spread_kernel(const Geom g, const int *__restrict__ tile_offsets, const int *__restrict__ perm,
              const float *__restrict__ spos, const float *__restrict__ xr, const float *__restrict__ xs, const int64_t n,
              const int Cr, const int plane0, float *__restrict__ grid)
{
    using C = TapCfg<DIM, W>;
    constexpr int NT = spread_threads<DIM>();
    constexpr int NWAVES = NT / 64;
    __shared__ double acc[C::CELLS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    // block -> (pencil (j1, j2), segment), plane -> (batch, real column)
    const int seg = blockIdx.x % g.nseg;
    const int pencil = blockIdx.x / g.nseg;
    const int j2 = pencil % g.nta[2];
    const int j1 = pencil / g.nta[2];
    const int plane_local = blockIdx.y;
    const int plane = plane0 + plane_local;
    const int b = plane / Cr;
    const int cr = plane - b * Cr;

    const int k_begin = seg * kSegChunks;
    const int k_end = min(g.nta[0], k_begin + kSegChunks);
    const int bin0 = b * g.tiles_per_batch + pencil * g.np0;  // first plan bin of this pencil
    {
        int s0, e0, s1, e1;
        chunk_range(g, tile_offsets, bin0, k_begin, s0, e0);
        chunk_range(g, tile_offsets, bin0, k_end - 1, s1, e1);
        if (s0 == e1) return;  // no points in this segment
    }  // no points in this segment

    for (int i = tid; i < C::CELLS; i += NT) acc[i] = 0.0;

    const int m = g.m;
    const int tb1 = j1 * g.Ta[1], tb2 = j2 * g.Ta[2];
    const float sc = win_exp_scale(m);
    float norm = win_norm(m);
    norm = DIM == 3 ? norm * norm * norm : (DIM == 2 ? norm * norm : norm);

    LaneTaps<DIM, W> taps;
    taps.init(lane, m);
    const float c0 = (float)(m - lane);  // axis-0 tap of this lane (lanes < W)

    float *const gplane = grid + (int64_t)plane_local * g.cells;
    // coefficients: the caller's row-major [point][Cr] array read through the plan's permutation (one or two real
    // columns: no gather pass), or the planar copy in plan order
    const float *const xcol = xs + (int64_t)cr * n;
    // A grid with few tiles (1-D, 2-D: 64 tiles at N = 128) would leave most CUs idle: blockIdx.z splits the points of
    // every chunk over gridDim.z workgroups, whose partial tiles meet in the flush atomics.
    const int nsplit = gridDim.z, split = blockIdx.z;

    // Resident plane p holds the (unwrapped) grid plane base_z + p.
    int base_z = 0;
    bool live = false;

    // Write out the lowest `shift` planes, slide the others down, clear the top.
    // Batch 0 visits the retiring planes row by row (one wave per padded row, lanes = columns, so a row
    // leaves as one contiguous run of global atomics) and refills them from `shift` planes above; the
    // remaining batches are a linear LDS move.  Batch b only reads what batch b+1 will overwrite.
    auto retire = [&](int shift) {
        __syncthreads();
        // Batch 0: all 64 lanes of every wave instruction carry a cell (a padded row is S2 = T2 + 2m+2 cells, so
        // consecutive lanes run across row ends): the memory side accepts roughly one float-atomic wave
        // instruction per 50 ns per CU whatever its lane count, so fewer, fuller instructions matter.
        for (int idx = tid; idx < shift * C::S0; idx += NT) {
            const int p = idx / C::S0;
            const int rem = idx - p * C::S0;
            const int r = rem / C::S2;
            const int c = rem - r * C::S2;
            const double v = acc[idx];
            const int src = idx + shift * C::S0;
            acc[idx] = src < C::CELLS ? acc[src] : 0.0;
            if (v != 0.0) {
                const int64_t gz = DIM == 3 ? wrap_near(base_z + p, g.Ma[0]) : 0;
                const int64_t g1 = DIM >= 2 ? wrap_near(tb1 - m + r, g.Ma[1]) : 0;
                atomicAdd(gplane + (gz * g.Ma[1] + g1) * g.Ma[2] + wrap_near(tb2 - m + c, g.Ma[2]), (float)v);
            }
        }
        for (int lo = shift * C::S0; lo < C::CELLS; lo += shift * C::S0) {
            __syncthreads();
            const int hi = min(lo + shift * C::S0, C::CELLS);
            for (int idx = lo + tid; idx < hi; idx += NT) {
                const int src = idx + shift * C::S0;
                acc[idx] = src < C::CELLS ? acc[src] : 0.0;
            }
        }
        __syncthreads();
    };

    __syncthreads();
    for (int k = k_begin; k < k_end; ++k) {
        int s, e;
        chunk_range(g, tile_offsets, bin0, k, s, e);
        if (nsplit > 1) {
            const int span = (e - s + nsplit - 1) / nsplit;
            s = min(e, s + split * span);
            e = min(e, s + span);
        }
        if (e == s) continue;
        const int want_z = k * C::TC - C::M0OFF;
        if (live && want_z != base_z) retire(min(want_z - base_z, C::NP));
        base_z = want_z;
        live = true;
        const int tb0 = k * C::TC;

        // static split of the chunk's points over the waves
        const int len = (e - s + NWAVES - 1) / NWAVES;
        const int a = s + wave * len;
        const int bnd = min(e, a + len);
        for (int j0 = a; j0 < bnd; j0 += 64) {
            const int cnt = min(64, bnd - j0);
            PointPrep<DIM, W> pp;
            float xv = 0.0f;
            if (lane < cnt) {
                pp.load(g, spos, (int64_t)j0 + lane, tb0, tb1, tb2);
                if (xr) {
                    const int64_t src = DIM == 3 ? __float_as_int(spos[((int64_t)j0 + lane) * 4 + 3]) : perm[(int64_t)j0 + lane];
                    xv = xr[src * Cr + cr] * norm;
                } else {
                    xv = xcol[(int64_t)j0 + lane] * norm;
                }
            } else {
                pp.clear();
            }
            for (int q = 0; q < cnt; ++q) {
                const float f1 = readlane_f(pp.f1, q), f2 = readlane_f(pp.f2, q);
                const float xq = readlane_f(xv, q);
                double *const origin = acc + readlane_i(pp.base, q);
                float ps0[C::W0];
                if (DIM == 3) {
                    const float d0 = readlane_f(pp.f0, q) + c0;
                    const float psi0 = __builtin_amdgcn_exp2f(sc * d0 * d0);
#pragma unroll
                    for (int l0 = 0; l0 < C::W0; ++l0) ps0[l0] = readlane_f(psi0, l0);
                } else {
                    ps0[0] = 1.0f;
                }
#pragma unroll
                for (int p = 0; p < C::PASSES; ++p) {
                    if (taps.valid[p]) {
                        const float d1 = f1 + taps.c1[p], d2 = f2 + taps.c2[p];
                        const float r2 = DIM >= 2 ? fmaf(d1, d1, d2 * d2) : d2 * d2;
                        const float w12 = __builtin_amdgcn_exp2f(sc * r2) * xq;
                        double *dst = origin + taps.off[p];
#pragma unroll
                        for (int l0 = 0; l0 < C::W0; ++l0) atomicAdd(dst + l0 * C::S0, (double)(w12 * ps0[l0]));
                    }
                }
            }
        }
    }
    if (live) retire(C::NP);
}

template <int DIM, int W>
static int launch_spread_t(const Geom &g, const int *tile_offsets, const int *perm, const float *spos, const float *xr,
                           const float *xs, int64_t n, int64_t Cr, int64_t plane0, int64_t nplanes, int splits, float *grid,
                           hipStream_t stream)
{
    const dim3 blocks((unsigned)(g.nta[1] * g.nta[2] * g.nseg), (unsigned)nplanes, (unsigned)splits);
    hipLaunchKernelGGL((spread_kernel<DIM, W>), blocks, dim3(spread_threads<DIM>()), 0, stream, g, tile_offsets, perm, spos,
                       xr, xs, n, (int)Cr, (int)plane0, grid);
    NFFT_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int DIM>
static int launch_spread_d(const Geom &g, const int *to, const int *perm, const float *spos, const float *xr, const float *xs,
                           int64_t n, int64_t Cr, int64_t plane0, int64_t nplanes, int splits, float *grid, hipStream_t stream)
{
    switch (g.m) {
    case 1: return launch_spread_t<DIM, 4>(g, to, perm, spos, xr, xs, n, Cr, plane0, nplanes, splits, grid, stream);
    case 2: return launch_spread_t<DIM, 6>(g, to, perm, spos, xr, xs, n, Cr, plane0, nplanes, splits, grid, stream);
    case 3: return launch_spread_t<DIM, 8>(g, to, perm, spos, xr, xs, n, Cr, plane0, nplanes, splits, grid, stream);
    case 4: return launch_spread_t<DIM, 10>(g, to, perm, spos, xr, xs, n, Cr, plane0, nplanes, splits, grid, stream);
    case 5: return launch_spread_t<DIM, 12>(g, to, perm, spos, xr, xs, n, Cr, plane0, nplanes, splits, grid, stream);
    case 6: return launch_spread_t<DIM, 14>(g, to, perm, spos, xr, xs, n, Cr, plane0, nplanes, splits, grid, stream);
    case 7: return launch_spread_t<DIM, 16>(g, to, perm, spos, xr, xs, n, Cr, plane0, nplanes, splits, grid, stream);
    case 8: return launch_spread_t<DIM, 18>(g, to, perm, spos, xr, xs, n, Cr, plane0, nplanes, splits, grid, stream);
    }
    set_error("cutoff m must be in 1..8");
    return 1;
}

// xr != nullptr: the caller's row-major [point][Cr] coefficients (read through the plan's permutation); else xs, the
// planar copy in plan order (gather_rows)
int launch_spread(const Geom &g, const PlanLayout &L, const void *plan, const float *xr, const float *xs, int64_t n,
                  int64_t Cr, int64_t plane0, int64_t nplanes, float *grid, hipStream_t stream)
{
    const char *base = (const char *)plan;
    const int *to = (const int *)(base + L.off_offsets);
    const int *perm = (const int *)(base + L.off_perm);
    const float *spos = (const float *)(base + L.off_spos);
    if (nplanes <= 0 || n <= 0) return 0;
    const int splits = point_splits(g, L, n, nplanes);
    switch (g.dim) {
    case 1: return launch_spread_d<1>(g, to, perm, spos, xr, xs, n, Cr, plane0, nplanes, splits, grid, stream);
    case 2: return launch_spread_d<2>(g, to, perm, spos, xr, xs, n, Cr, plane0, nplanes, splits, grid, stream);
    case 3: return launch_spread_d<3>(g, to, perm, spos, xr, xs, n, Cr, plane0, nplanes, splits, grid, stream);
    }
    set_error("dim must be 1, 2 or 3");
    return 1;
}

} // namespace nfft
