// Interpolation (forward gather) for gfx950.
//
// Computes what the reference's complex_/real_forward_window_convolution_kernel
// (csrc/cuda/spatial_window_operations.cu:214-332) computes,
//     y[i, c] = sum_{l in [0,2m+2)^d} prod_k psi_k(i, l_k) * g[(b, c), (shift_i + l) mod M],
// as a true gather (the reference reduces the (2m+2)^d products of one output with atomics on y):
//   * same tile-sorted point plan and pencil sweep as the spreading kernel;
//   * the NP = TC + 2m+1 planes a chunk needs are resident in LDS: between chunks the 2m+1 still-needed
//     planes slide down and only TC new ones are fetched (coalesced row loads), so every plane of a pencil
//     is read from HBM/L2 once per segment;
//   * one wave per point, lanes = (l1, l2) taps, axis-0 taps unrolled; the 64 partial sums are reduced
//     with DPP row shifts/broadcasts (no LDS traffic) and the wave writes 64 outputs at a time.
// g is held as real planes (re and im of a complex grid are separate planes = separate real columns).
#include "common.h"
#include "kernels.h"
#include "window.h"

namespace nfft {

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_zero(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}

// Sum over the 64 lanes; the total ends up in lane 63.
__device__ __forceinline__ float wave_sum_to_lane63(float v)
{
    v += dpp_zero<0x111, 0xf>(v);  // row_shr:1
    v += dpp_zero<0x112, 0xf>(v);  // row_shr:2
    v += dpp_zero<0x114, 0xf>(v);  // row_shr:4
    v += dpp_zero<0x118, 0xf>(v);  // row_shr:8  -> lane 15 of every row holds the row sum
    v += dpp_zero<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v += dpp_zero<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}

template <int DIM>
constexpr int interp_threads() { return DIM == 3 ? 512 : 256; }

template <int DIM, int W>
__global__ void __launch_bounds__((interp_threads<DIM>()))
interp_kernel(const Geom g, const int *__restrict__ tile_offsets, const int *__restrict__ perm,
              const float *__restrict__ spos, const float *__restrict__ grid, const int Cr, const int plane0,
              float *__restrict__ yr)
{
    using C = TapCfg<DIM, W>;
    constexpr int NT = interp_threads<DIM>();
    constexpr int NWAVES = NT / 64;
    __shared__ float planes[C::CELLS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

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
    if (tile_offsets[tile0 + k_begin] == tile_offsets[tile0 + k_end]) return;

    const int m = g.m;
    const int tb1 = j1 * g.Ta[1], tb2 = j2 * g.Ta[2];
    const float sc = win_exp_scale(m);
    float norm = win_norm(m);
    norm = DIM == 3 ? norm * norm * norm : (DIM == 2 ? norm * norm : norm);

    LaneTaps<DIM, W> taps;
    taps.init(lane, m);
    const float c0 = (float)(m - lane);

    const float *const gplane = grid + (int64_t)plane_local * g.cells;

    // Resident plane p holds the (unwrapped) grid plane base_z + p; planes [0, have) are valid.
    int base_z = 0, have = 0;

    for (int k = k_begin; k < k_end; ++k) {
        const int s = tile_offsets[tile0 + k], e = tile_offsets[tile0 + k + 1];
        if (e == s) continue;
        const int want_z = k * C::TC - C::M0OFF;
        // slide the planes that are still needed down (linear LDS move in batches of `shift` planes: batch b
        // only reads what batch b+1 overwrites), then fetch the missing planes row by row (one wave per padded
        // row, lanes = columns: coalesced)
        const int shift = have > 0 ? min(want_z - base_z, have) : 0;
        const int kept = have - shift;
        __syncthreads();  // every wave is done reading the planes about to move
        if (kept > 0) {
            for (int lo = 0; lo < kept * C::S0; lo += shift * C::S0) {
                const int hi = min(lo + shift * C::S0, kept * C::S0);
                for (int idx = lo + tid; idx < hi; idx += NT) planes[idx] = planes[idx + shift * C::S0];
                __syncthreads();
            }
        }
        for (int row = kept * C::P1 + wave; row < C::NP * C::P1; row += NWAVES) {
            const int p = row / C::P1;
            const int r = row - p * C::P1;
            const int64_t gz = DIM == 3 ? wrap(want_z + p, g.Ma[0]) : 0;
            const int64_t g1 = DIM >= 2 ? wrap(tb1 - m + r, g.Ma[1]) : 0;
            const float *const grow = gplane + (gz * g.Ma[1] + g1) * g.Ma[2];
            for (int c = lane; c < C::S2; c += 64)
                planes[row * C::S2 + c] = c < C::P2 ? grow[wrap_near(tb2 - m + c, g.Ma[2])] : 0.0f;
        }
        base_z = want_z;
        have = C::NP;
        __syncthreads();
        const int tb0 = k * C::TC;

        const int len = (e - s + NWAVES - 1) / NWAVES;
        const int a = s + wave * len;
        const int bnd = min(e, a + len);
        for (int j0 = a; j0 < bnd; j0 += 64) {
            const int cnt = min(64, bnd - j0);
            PointPrep<DIM, W> pp;
            if (lane < cnt) pp.load(g, spos, (int64_t)j0 + lane, tb0, tb1, tb2);
            else pp.clear();
            float result = 0.0f;
            for (int q = 0; q < cnt; ++q) {
                const float f1 = readlane_f(pp.f1, q), f2 = readlane_f(pp.f2, q);
                const float *const origin = planes + readlane_i(pp.base, q);
                float ps0[C::W0];
                if (DIM == 3) {
                    const float d0 = readlane_f(pp.f0, q) + c0;
                    const float psi0 = __builtin_amdgcn_exp2f(sc * d0 * d0);
#pragma unroll
                    for (int l0 = 0; l0 < C::W0; ++l0) ps0[l0] = readlane_f(psi0, l0);
                } else {
                    ps0[0] = 1.0f;
                }
                float acc = 0.0f;
#pragma unroll
                for (int p = 0; p < C::PASSES; ++p) {
                    if (taps.valid[p]) {
                        const float d1 = f1 + taps.c1[p], d2 = f2 + taps.c2[p];
                        const float r2 = DIM >= 2 ? fmaf(d1, d1, d2 * d2) : d2 * d2;
                        const float w12 = __builtin_amdgcn_exp2f(sc * r2);
                        const float *src = origin + taps.off[p];
                        float part = 0.0f;
#pragma unroll
                        for (int l0 = 0; l0 < C::W0; ++l0) part = fmaf(ps0[l0], src[l0 * C::S0], part);
                        acc = fmaf(w12, part, acc);
                    }
                }
                const float total = readlane_f(wave_sum_to_lane63(acc), 63);
                if (lane == q) result = total;
            }
            if (lane < cnt) yr[(int64_t)perm[j0 + lane] * Cr + cr] = result * norm;
        }
    }
}

template <int DIM, int W>
static int launch_interp_t(const Geom &g, const int *to, const int *perm, const float *spos, const float *grid,
                           int64_t Cr, int64_t plane0, int64_t nplanes, float *yr, hipStream_t stream)
{
    const dim3 blocks((unsigned)(g.nta[1] * g.nta[2] * g.nseg), (unsigned)nplanes);
    hipLaunchKernelGGL((interp_kernel<DIM, W>), blocks, dim3(interp_threads<DIM>()), 0, stream, g, to, perm, spos, grid, (int)Cr,
                       (int)plane0, yr);
    NFFT_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int DIM>
static int launch_interp_d(const Geom &g, const int *to, const int *perm, const float *spos, const float *grid,
                           int64_t Cr, int64_t plane0, int64_t nplanes, float *yr, hipStream_t stream)
{
    switch (g.m) {
    case 1: return launch_interp_t<DIM, 4>(g, to, perm, spos, grid, Cr, plane0, nplanes, yr, stream);
    case 2: return launch_interp_t<DIM, 6>(g, to, perm, spos, grid, Cr, plane0, nplanes, yr, stream);
    case 3: return launch_interp_t<DIM, 8>(g, to, perm, spos, grid, Cr, plane0, nplanes, yr, stream);
    case 4: return launch_interp_t<DIM, 10>(g, to, perm, spos, grid, Cr, plane0, nplanes, yr, stream);
    case 5: return launch_interp_t<DIM, 12>(g, to, perm, spos, grid, Cr, plane0, nplanes, yr, stream);
    case 6: return launch_interp_t<DIM, 14>(g, to, perm, spos, grid, Cr, plane0, nplanes, yr, stream);
    case 7: return launch_interp_t<DIM, 16>(g, to, perm, spos, grid, Cr, plane0, nplanes, yr, stream);
    case 8: return launch_interp_t<DIM, 18>(g, to, perm, spos, grid, Cr, plane0, nplanes, yr, stream);
    }
    set_error("cutoff m must be in 1..8");
    return 1;
}

int launch_interp(const Geom &g, const PlanLayout &L, const void *plan, const float *grid, int64_t n, int64_t Cr,
                  int64_t plane0, int64_t nplanes, float *yr, hipStream_t stream)
{
    const char *base = (const char *)plan;
    const int *to = (const int *)(base + L.off_offsets);
    const int *perm = (const int *)(base + L.off_perm);
    const float *spos = (const float *)(base + L.off_spos);
    if (nplanes <= 0 || n <= 0) return 0;
    switch (g.dim) {
    case 1: return launch_interp_d<1>(g, to, perm, spos, grid, Cr, plane0, nplanes, yr, stream);
    case 2: return launch_interp_d<2>(g, to, perm, spos, grid, Cr, plane0, nplanes, yr, stream);
    case 3: return launch_interp_d<3>(g, to, perm, spos, grid, Cr, plane0, nplanes, yr, stream);
    }
    set_error("dim must be 1, 2 or 3");
    return 1;
}

} // namespace nfft
