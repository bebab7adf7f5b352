// Spreading (adjoint gridding) for gfx950.
//
// Computes what the reference's real_/complex_adjoint_window_convolution_kernel
// (csrc/cuda/spatial_window_operations.cu:103-211) computes,
//     g[(b, c), (shift_i + l) mod M] += x[i, c] * prod_k psi_k(i, l_k),   l in [0, 2m+2)^d,
// but organised for CDNA4:
//   * points arrive counting-sorted by (pencil, chunk) tile (binning.hip);
//   * one workgroup sweeps a segment of a T1 x T2 pencil along axis 0, accumulating into a ring of
//     padded planes in LDS with ds_add_f32 -- one wave handles one point at a time, its lanes are the
//     (l1, l2) taps, the l0 taps are an unrolled loop over ring planes;
//   * the window is evaluated in registers (one v_exp_f32 per lane and pass plus one for axis 0) instead
//     of being read back from HBM (reference: point_psi, 1.2 GB at N=256, n=1e7);
//   * completed planes leave LDS as contiguous row segments of global_atomic_add_f32 (a row of a padded
//     plane is one <=256-byte wave instruction), so HBM sees whole-line updates, never scattered dwords.
#include <cstdlib>

#include "common.h"
#include "kernels.h"
#include "window.h"

namespace nfft {

template <int DIM, int W>
__global__ void __launch_bounds__((TapCfg<DIM, W>::NT))
spread_kernel(const Geom g, const int *__restrict__ tile_offsets, const float *__restrict__ spos,
              const float *__restrict__ xs, const int64_t n, const int Cr, const int plane0, float *__restrict__ grid,
              const int dbg)
{
    using C = TapCfg<DIM, W>;
    __shared__ float ring[C::LDS_FLOATS];

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
    const int tile0 = b * g.tiles_per_batch + pencil * g.nta[0];
    if (tile_offsets[tile0 + k_begin] == tile_offsets[tile0 + k_end]) return;  // no points in this segment

    for (int i = tid; i < C::LDS_FLOATS; i += C::NT) ring[i] = 0.0f;

    const int m = g.m;
    const int tb1 = j1 * g.Ta[1], tb2 = j2 * g.Ta[2];
    const float sc = win_exp_scale(m);
    float norm = win_norm(m);
    norm = DIM == 3 ? norm * norm * norm : (DIM == 2 ? norm * norm : norm);

    LaneTaps<DIM, W> taps;
    taps.init(lane, m);
    const float c0 = (float)(m - lane);  // axis-0 tap of this lane (lanes < W)

    float *const gplane = grid + (int64_t)plane_local * g.cells;
    const float *const xcol = xs + (int64_t)cr * n;

    int flushed = k_begin * C::TC - C::M0OFF;  // first plane (unwrapped) not yet written out
    int dirty = flushed;                      // planes below this may hold data
    __syncthreads();

    for (int k = k_begin; k < k_end; ++k) {
        const int s = tile_offsets[tile0 + k], e = tile_offsets[tile0 + k + 1];
        if (e > s) {
            // static split of the chunk's points over the waves
            const int len = (e - s + C::NWAVES - 1) / C::NWAVES;
            const int a = s + wave * len;
            const int bnd = min(e, a + len);
            for (int j0 = a; j0 < bnd; j0 += 64) {
                const int cnt = min(64, bnd - j0);
                PointPrep<DIM, W> pp;
                float xv = 0.0f;
                if (lane < cnt) {
                    pp.load(g, spos, (int64_t)j0 + lane, tb1, tb2);
                    xv = xcol[(int64_t)j0 + lane] * norm;
                } else {
                    pp.f0 = pp.f1 = pp.f2 = 0.0f;
                    pp.base12 = 0;
                    pp.z0 = 0;
                }
                for (int q = 0; q < cnt; ++q) {
                    const float f1 = readlane_f(pp.f1, q), f2 = readlane_f(pp.f2, q);
                    const float xq = readlane_f(xv, q);
                    const int b12 = readlane_i(pp.base12, q);
                    float psi0 = 1.0f;
                    int z0 = 0;
                    if (DIM == 3) {
                        const float d0 = readlane_f(pp.f0, q) + c0;
                        psi0 = __builtin_amdgcn_exp2f(sc * d0 * d0);
                        z0 = readlane_i(pp.z0, q) + 4 * C::R;  // keep the ring index non-negative
                    }
#pragma unroll
                    for (int p = 0; p < C::PASSES; ++p) {
                        if (taps.valid[p]) {
                            const float d1 = f1 + taps.c1[p], d2 = f2 + taps.c2[p];
                            const float r2 = DIM >= 2 ? fmaf(d1, d1, d2 * d2) : d2 * d2;
                            const float w12 = __builtin_amdgcn_exp2f(sc * r2) * xq;
                            float *dst = ring + b12 + taps.off[p];
                            if (DIM == 3) {
#pragma unroll
                                for (int l0 = 0; l0 < C::W0; ++l0) {
                                    const int slot = (z0 + l0) & (C::R - 1);
                                    if (dbg & 2) { if (dbg & 4) dst[slot * C::S0] = w12 * readlane_f(psi0, l0); else asm volatile("" ::"v"(w12 * readlane_f(psi0, l0)), "v"(dst + slot * C::S0)); }
                                    else atomicAdd(dst + slot * C::S0, w12 * readlane_f(psi0, l0));
                                }
                            } else {
                                atomicAdd(dst, w12);
                            }
                        }
                    }
                }
            }
            dirty = (k + 1) * C::TC + (C::W0 - 1 - C::M0OFF);
        }
        // planes below `upto` receive no further taps from this segment
        int upto = (k + 1) * C::TC - C::M0OFF;
        if (k == k_end - 1) upto = min((k + 1) * C::TC, g.Ma[0]) + (C::W0 - 1 - C::M0OFF);
        if (dirty > flushed) {
            __syncthreads();
            const int hi = min(upto, dirty);
            const int total = (hi - flushed) * C::S0;
            for (int idx = tid; idx < total; idx += C::NT) {
                const int pz = idx / C::S0;
                const int rem = idx - pz * C::S0;
                const int r = rem / C::S2;
                const int c = rem - r * C::S2;
                const int z = flushed + pz;
                float *src = ring + ((z + 4 * C::R) & (C::R - 1)) * C::S0 + rem;
                const float v = *src;
                if (v != 0.0f) {
                    *src = 0.0f;
                    const int64_t gz = DIM == 3 ? wrap(z, g.Ma[0]) : 0;
                    const int64_t g1 = DIM >= 2 ? wrap(tb1 - m + r, g.Ma[1]) : 0;
                    const int64_t g2 = wrap(tb2 - m + c, g.Ma[2]);
                    if (dbg & 1) { if (dbg & 8) gplane[(gz * g.Ma[1] + g1) * g.Ma[2] + g2] = v; }
                    else atomicAdd(gplane + (gz * g.Ma[1] + g1) * g.Ma[2] + g2, v);
                }
            }
            __syncthreads();
        }
        flushed = upto;
    }
}

template <int DIM, int W>
static int launch_spread_t(const Geom &g, const int *tile_offsets, const float *spos, const float *xs, int64_t n,
                           int64_t Cr, int64_t plane0, int64_t nplanes, float *grid, hipStream_t stream)
{
    using C = TapCfg<DIM, W>;
    const dim3 blocks((unsigned)(g.nta[1] * g.nta[2] * g.nseg), (unsigned)nplanes);
    hipLaunchKernelGGL((spread_kernel<DIM, W>), blocks, dim3(C::NT), 0, stream, g, tile_offsets, spos, xs, n, (int)Cr,
                       (int)plane0, grid, getenv("NFFT_HIP_DBG") ? atoi(getenv("NFFT_HIP_DBG")) : 0);
    NFFT_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int DIM>
static int launch_spread_d(const Geom &g, const int *to, const float *spos, const float *xs, int64_t n, int64_t Cr,
                           int64_t plane0, int64_t nplanes, float *grid, hipStream_t stream)
{
    switch (g.m) {
    case 1: return launch_spread_t<DIM, 4>(g, to, spos, xs, n, Cr, plane0, nplanes, grid, stream);
    case 2: return launch_spread_t<DIM, 6>(g, to, spos, xs, n, Cr, plane0, nplanes, grid, stream);
    case 3: return launch_spread_t<DIM, 8>(g, to, spos, xs, n, Cr, plane0, nplanes, grid, stream);
    case 4: return launch_spread_t<DIM, 10>(g, to, spos, xs, n, Cr, plane0, nplanes, grid, stream);
    case 5: return launch_spread_t<DIM, 12>(g, to, spos, xs, n, Cr, plane0, nplanes, grid, stream);
    case 6: return launch_spread_t<DIM, 14>(g, to, spos, xs, n, Cr, plane0, nplanes, grid, stream);
    case 7: return launch_spread_t<DIM, 16>(g, to, spos, xs, n, Cr, plane0, nplanes, grid, stream);
    case 8: return launch_spread_t<DIM, 18>(g, to, spos, xs, n, Cr, plane0, nplanes, grid, stream);
    }
    set_error("cutoff m must be in 1..8");
    return 1;
}

int launch_spread(const Geom &g, const PlanLayout &L, const void *plan, const float *xs, int64_t n, int64_t Cr,
                  int64_t plane0, int64_t nplanes, float *grid, hipStream_t stream)
{
    const char *base = (const char *)plan;
    const int *to = (const int *)(base + L.off_offsets);
    const float *spos = (const float *)(base + L.off_spos);
    if (nplanes <= 0 || n <= 0) return 0;
    switch (g.dim) {
    case 1: return launch_spread_d<1>(g, to, spos, xs, n, Cr, plane0, nplanes, grid, stream);
    case 2: return launch_spread_d<2>(g, to, spos, xs, n, Cr, plane0, nplanes, grid, stream);
    case 3: return launch_spread_d<3>(g, to, spos, xs, n, Cr, plane0, nplanes, grid, stream);
    }
    set_error("dim must be 1, 2 or 3");
    return 1;
}

} // namespace nfft
