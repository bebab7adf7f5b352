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
//   * one LANE per point: a lane walks the (2m+2)^d window of its own point -- per row of the window
//     ceil((2m+5)/4) aligned ds_read_b128 (the axis-2 weights are evaluated on the aligned positions and are
//     zero outside the window) and 4 FMAs per read; the axis-0/1 weights are one v_exp_f32 per row/plane.  No
//     cross-lane reduction, no per-point scalar broadcasts: ~40 wave instructions per point instead of ~120
//     for the tap-per-lane layout this kernel started with.
// g is held as real planes (re and im of a complex grid are separate planes = separate real columns).
#include "common.h"
#include "kernels.h"
#include "window.h"

namespace nfft {

// Geometry of the gather kernel.  It shares the point plan (pencils, chunks of TC planes) with the spreading
// kernel but keeps 4-byte cells and rows padded to a multiple of 4 floats so that a lane can fetch its
// 2m+2 taps of a row with aligned ds_read_b128.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int DIM, int W, bool WIDE>
struct GatherCfg {
    static constexpr TileCfg tc = tile_cfg(DIM, W, WIDE);
    static constexpr int T1 = tc.T1, T2 = tc.T2, TC = tc.TC;
    static constexpr int W0 = DIM == 3 ? W : 1;
    static constexpr int W1 = DIM >= 2 ? W : 1;
    static constexpr int M0OFF = DIM == 3 ? (W / 2 - 1) : 0;
    static constexpr int NP = TC + W0 - 1;
    static constexpr int P1 = T1 + W1 - 1;
    static constexpr int P2 = T2 + W - 1;
    static constexpr int NR = (W + 3 + 3) / 4;            // aligned 16-byte reads covering any 2m+2 window
    static constexpr int S2 = (P2 + 3 + 3) / 4 * 4;       // row stride (floats): room for the aligned over-read
    static constexpr int S0 = P1 * S2;
    static constexpr int CELLS = NP * S0;
    static constexpr int NT = DIM == 3 ? (WIDE ? 1024 : 512) : 256;  // the wide tiling fills the LDS with one workgroup
    static constexpr int NWAVES = NT / 64;
    static_assert(CELLS * 4 <= 160 * 1024, "LDS budget");
};

template <int DIM, int W, bool WIDE>
__global__ void __launch_bounds__((GatherCfg<DIM, W, WIDE>::NT))
interp_kernel(const Geom g, const int *__restrict__ tile_offsets, const int *__restrict__ perm,
              const float *__restrict__ spos, const float *__restrict__ grid, const int Cr, const int plane0,
              float *__restrict__ yr)
{
    using C = GatherCfg<DIM, W, WIDE>;
    constexpr int NT = C::NT;
    constexpr int NWAVES = C::NWAVES;
    __shared__ float4 planes4[C::CELLS / 4];
    float *const planes = (float *)planes4;

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
    const int bin0 = b * g.tiles_per_batch + pencil * g.np0;  // first plan bin of this pencil
    {
        int s0, e0, s1, e1;
        chunk_range(g, tile_offsets, bin0, k_begin, s0, e0);
        chunk_range(g, tile_offsets, bin0, k_end - 1, s1, e1);
        if (s0 == e1) return;  // no points in this segment
    }

    const int m = g.m;
    const int tb1 = j1 * g.Ta[1], tb2 = j2 * g.Ta[2];
    const float sc = win_exp_scale(m);
    float norm = win_norm(m);
    norm = DIM == 3 ? norm * norm * norm : (DIM == 2 ? norm * norm : norm);
    const float *const gplane = grid + (int64_t)plane_local * g.cells;

    // Resident plane p holds the (unwrapped) grid plane base_z + p; planes [0, have) are valid.
    int base_z = 0, have = 0;

    // (few tiles: blockIdx.z splits the points of every chunk over gridDim.z workgroups, as in spread.hip)
    const int nsplit = gridDim.z, split = blockIdx.z;
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

        // one lane per point: each lane walks the (2m+2)^d window of its own point
        // full 64-point batches are dealt round-robin to the waves (only the chunk's last batch is partial)
        for (int j0 = s + wave * 64; j0 < e; j0 += NWAVES * 64) {
            const int j = j0 + lane;
            if (j >= e) continue;
            int c0 = 0, c1 = 0, c2 = 0;
            float f0 = 0.f, f1 = 0.f, f2 = 0.f;
            if (DIM == 3) {
                const f32x4 rec = *(const f32x4 *)(spos + (int64_t)j * 4);  // plan record {p0, p1, p2, x}
                split_cell(rec.x, g.M, c0, f0);
                split_cell(rec.y, g.M, c1, f1);
                split_cell(rec.z, g.M, c2, f2);
            } else if (DIM == 2) {
                split_cell(spos[(int64_t)j * 2 + 0], g.M, c1, f1);
                split_cell(spos[(int64_t)j * 2 + 1], g.M, c2, f2);
            } else {
                split_cell(spos[j], g.M, c2, f2);
            }
            const int col = c2 - tb2;       // window origin column inside the padded row
            const int sh = col & 3;         // its offset from the 16-byte boundary below it
            // axis-2 weights on the aligned positions; zero outside the window so the over-read is harmless.
            // Held as float pairs: the inner loop is packed math (v_pk_fma_f32, two FMAs per lane and instruction
            // -- a wave64 VALU instruction costs ~4 cycles per SIMD on gfx950, so packing halves the VALU time).
            f32x2 w2[2 * C::NR];
#pragma unroll
            for (int q = 0; q < 2 * C::NR; ++q) {
                float pair[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int l2 = 2 * q + h - sh;
                    const float t = f2 + (float)(m - l2);
                    const float v = __builtin_amdgcn_exp2f(sc * t * t);
                    pair[h] = (l2 >= 0 && l2 < W) ? v : 0.0f;
                }
                w2[q] = f32x2{pair[0], pair[1]};
            }
            const f32x4 *row0 = (const f32x4 *)(planes + (c0 - tb0) * C::S0 + (c1 - tb1) * C::S2 + (col - sh));
            // axis-1 weights (one v_exp_f32 each, reused by every plane)
            float w1[C::W1];
#pragma unroll
            for (int l1 = 0; l1 < C::W1; ++l1) {
                const float t1 = f1 + (float)(m - l1);
                w1[l1] = DIM >= 2 ? __builtin_amdgcn_exp2f(sc * t1 * t1) : 1.0f;
            }
            // Rows are processed RB at a time with all their reads in flight: the loop is LDS-latency bound
            // otherwise (one wave-wide ds_read_b128 -> wait -> dependent FMAs).
            constexpr int RB = C::W1 % 2 == 0 ? 2 : 1;  // more rows in flight cost occupancy (152 VGPRs at 5 rows)
            float acc0 = 0.0f;
            for (int l0 = 0; l0 < C::W0; ++l0) {
                const f32x4 *rowp = row0 + l0 * (C::S0 / 4);
                f32x2 plane_acc = {0.0f, 0.0f};
#pragma unroll
                for (int l1 = 0; l1 < C::W1; l1 += RB) {
                    f32x4 v[RB][C::NR];
#pragma unroll
                    for (int r = 0; r < RB; ++r)
#pragma unroll
                        for (int q = 0; q < C::NR; ++q) v[r][q] = rowp[(l1 + r) * (C::S2 / 4) + q];
#pragma unroll
                    for (int r = 0; r < RB; ++r) {
                        f32x2 ra = w2[0] * v[r][0].xy;
                        f32x2 rb = w2[1] * v[r][0].zw;
#pragma unroll
                        for (int q = 1; q < C::NR; ++q) {
                            ra = __builtin_elementwise_fma(w2[2 * q], v[r][q].xy, ra);
                            rb = __builtin_elementwise_fma(w2[2 * q + 1], v[r][q].zw, rb);
                        }
                        const f32x2 wr = {w1[l1 + r], w1[l1 + r]};
                        plane_acc = __builtin_elementwise_fma(wr, ra + rb, plane_acc);
                    }
                }
                float p0 = 1.0f;
                if (DIM == 3) {
                    const float t0 = f0 + (float)(m - l0);
                    p0 = __builtin_amdgcn_exp2f(sc * t0 * t0);
                }
                acc0 = fmaf(p0, plane_acc.x + plane_acc.y, acc0);
            }
            yr[(int64_t)(DIM == 3 ? __float_as_int(spos[(int64_t)j * 4 + 3]) : perm[j]) * Cr + cr] = acc0 * norm;  // (3-D: the index sits in the record)
        }
    }
}

template <int DIM, int W>
static int launch_interp_t(const Geom &g, const int *to, const int *perm, const float *spos, const float *grid,
                           int64_t Cr, int64_t plane0, int64_t nplanes, int splits, float *yr, hipStream_t stream)
{
    const dim3 blocks((unsigned)(g.nta[1] * g.nta[2] * g.nseg), (unsigned)nplanes, (unsigned)splits);
    if constexpr (DIM == 3) {
        if (g.wide) {
            hipLaunchKernelGGL((interp_kernel<DIM, W, true>), blocks, dim3(GatherCfg<DIM, W, true>::NT), 0, stream, g, to,
                               perm, spos, grid, (int)Cr, (int)plane0, yr);
            NFFT_HIP_CHECK(hipGetLastError());
            return 0;
        }
    }
    hipLaunchKernelGGL((interp_kernel<DIM, W, false>), blocks, dim3(GatherCfg<DIM, W, false>::NT), 0, stream, g, to, perm,
                       spos, grid, (int)Cr, (int)plane0, yr);
    NFFT_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int DIM>
static int launch_interp_d(const Geom &g, const int *to, const int *perm, const float *spos, const float *grid,
                           int64_t Cr, int64_t plane0, int64_t nplanes, int splits, float *yr, hipStream_t stream)
{
    switch (g.m) {
    case 1: return launch_interp_t<DIM, 4>(g, to, perm, spos, grid, Cr, plane0, nplanes, splits, yr, stream);
    case 2: return launch_interp_t<DIM, 6>(g, to, perm, spos, grid, Cr, plane0, nplanes, splits, yr, stream);
    case 3: return launch_interp_t<DIM, 8>(g, to, perm, spos, grid, Cr, plane0, nplanes, splits, yr, stream);
    case 4: return launch_interp_t<DIM, 10>(g, to, perm, spos, grid, Cr, plane0, nplanes, splits, yr, stream);
    case 5: return launch_interp_t<DIM, 12>(g, to, perm, spos, grid, Cr, plane0, nplanes, splits, yr, stream);
    case 6: return launch_interp_t<DIM, 14>(g, to, perm, spos, grid, Cr, plane0, nplanes, splits, yr, stream);
    case 7: return launch_interp_t<DIM, 16>(g, to, perm, spos, grid, Cr, plane0, nplanes, splits, yr, stream);
    case 8: return launch_interp_t<DIM, 18>(g, to, perm, spos, grid, Cr, plane0, nplanes, splits, yr, stream);
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
    const int splits = point_splits(g, L, n, nplanes);
    switch (g.dim) {
    case 1: return launch_interp_d<1>(g, to, perm, spos, grid, Cr, plane0, nplanes, splits, yr, stream);
    case 2: return launch_interp_d<2>(g, to, perm, spos, grid, Cr, plane0, nplanes, splits, yr, stream);
    case 3: return launch_interp_d<3>(g, to, perm, spos, grid, Cr, plane0, nplanes, splits, yr, stream);
    }
    set_error("dim must be 1, 2 or 3");
    return 1;
}

} // namespace nfft
