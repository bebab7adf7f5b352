// Interpolation (forward gather) for several coefficient columns: one wave per column ("wave = column").
//
// Same result as interp_mfma.hip / the reference's forward_window_convolution kernels
// (csrc/cuda/spatial_window_operations.cu:214-332); the reference loops over the trailing columns inside its
// kernel (":281-330") re-using shift and psi of a point for every column.  Here the sharing is:
//   * a workgroup of 8 waves takes one work item of the wide tiling (point set, pencil, range of slabs) for 8
//     consecutive grid planes of that point set -- 8 real coefficient columns; wave w owns column w;
//   * what depends only on the POINTS is built once per block of 32 points and shared through LDS by all 8
//     columns: the B fragments (psi2 on the 64 padded columns, f16-split, in MFMA register order), the 16 psi1
//     weights per lane, the cell / fraction along axis 0 -- in interp_mfma.hip every (set, column) plane has its
//     own workgroup and rebuilds them (64 x at config C4);
//   * what depends on the COLUMN never touches LDS: each wave streams the padded 32 x 64 tile of its own column's
//     plane straight from global memory into A-fragment registers (eight 16-byte loads per lane, the next plane
//     prefetched while the current one is used), scales it by the tile's power of two and splits it into f16
//     hi / lo in registers.  No staging phases, no barrier inside the plane sweep;
//   * per plane z and active block:  T_z = G_z Psi2 (12 MFMAs), t = sum_u1 psi1[u1] T_z[u1, i], y_i += psi0_i[z] t
//     exactly as in interp_mfma.hip; a wave keeps one y accumulator per block of the group in registers.
// Blocks are handled in groups of kIcBlocks (what fits the LDS); a group's plane sweep runs from the first
// point's window to the last one's, so a sparse item (config C4: 6.5 points per slab and pencil) sweeps its
// planes ~1.2 times, a dense one more often -- the kernel is chosen for multi-column problems, where the column
// sharing pays for that (api.hip gather_any).
// The results leave through LDS: 8 columns of a point are 32 contiguous bytes of y.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "kernels.h"
#include "mfma_split.h"

namespace nfft {

namespace {

constexpr int kIcWaves = 8;                 // columns per workgroup
constexpr int kIcThreads = kIcWaves * 64;
constexpr int kIcBlocks = 10;               // blocks of 32 points resident at a time

// Everything a block of 32 points shares between the columns, contiguous (13 KB): the sweep addresses it as
// (block base) + lane * 16 + instruction offset, one address register for all of it.
struct __align__(16) IcBlock {
    f16x8 bfrag[4][2][64];                  // [k-step][hi/lo][lane = 32 (column half) + point]
    f32x4 w1[5][64];                        // [quad][lane]: psi1 of the lane's point on its 16 rows (quads 0-3); quad 4 =
                                            // {fraction along axis 0, cell along axis 0 (int bits; far outside every
                                            // window for padding lanes), -, -}
};
struct __align__(16) IcLds {
    IcBlock blk[kIcBlocks];                 // 130 KB
    float yacc[kIcWaves][kIcBlocks][64];    // running sums of a wave's column, per block and lane   20 KB
    int zf[kIcBlocks], zl[kIcBlocks];       // first / last plane of the block's window
    int ticket;                             // work-list entry of the workgroup (persistent launch: next_work_item)
};
// after a group's sweep the block area is reused to transpose the results: [point of the group][column]
constexpr int kIcStageStride = kIcWaves + 1;
static_assert(kIcBlocks * 32 * kIcStageStride * 4 <= (int)sizeof(IcBlock) * kIcBlocks, "stage fits");
static_assert(sizeof(IcLds) <= 160 * 1024, "LDS budget");

template <int W, bool OVERFLOW>
__global__ void __launch_bounds__(kIcThreads) __attribute__((amdgpu_waves_per_eu(2, 2)))
interp_cols_kernel(const Geom g, const int *__restrict__ tile_offsets,
                   const float *__restrict__ spos, const float *__restrict__ grid, const int Cr, const int64_t plane0,
                   const int64_t nplanes, const int64_t group0, float *__restrict__ yr, const int seg_slabs,
                   const int nsegm, const int4 *__restrict__ work, const int4 *__restrict__ sorted, const WorkTickets tickets)
{
    constexpr int m = W / 2 - 1;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    IcLds &L = *reinterpret_cast<IcLds *>(smem_raw);
    float *const ystage = reinterpret_cast<float *>(&L.blk[0]);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, h = lane >> 5;

    // column group -> (point set, first column); wave -> plane
    const int G = (Cr + kIcWaves - 1) / kIcWaves;
    const int64_t gid = group0 + blockIdx.y;
    const int b = (int)(gid / G);
    const int cr0 = (int)(gid - (int64_t)b * G) * kIcWaves;
    const int cr = cr0 + wave;
    const int64_t plane = (int64_t)b * Cr + cr;
    const bool col_active = cr < Cr && plane >= plane0 && plane < plane0 + nplanes;  // wave-uniform
    const float *const gplane = grid + (plane - plane0) * g.cells;  // (only dereferenced when col_active)
    const int pencils = g.nta[1] * g.nta[2];
    const int M = g.M;

    // (work items as in spread_mfma.hip: one workgroup per range, or a persistent grid over the plan's work list)
    const int listed = work[0].z;
    if (OVERFLOW ? !listed : listed) return;
    // (a plane walks its own point set's part of the sorted list: set_hdr[b] = {entries, first entry})
    const int2 set_hdr = OVERFLOW ? ((const int2 *)(work + 1))[b] : make_int2(1, 0);
    const int n_items = set_hdr.x;
    const int4 *const entries = sorted + set_hdr.y;
    for (int item = OVERFLOW ? next_work_item(tickets, &L.ticket, -1, (int)blockIdx.y) : 0; item < n_items;
         item = OVERFLOW ? next_work_item(tickets, &L.ticket, item, (int)blockIdx.y) : 1) {
    int pencil, sb, se;
    if constexpr (OVERFLOW) {
        const int4 it = tickets.ring ? entries[item] : listed_item(entries, item, n_items);
        pencil = it.x - b * pencils;
        sb = it.y;
        se = it.z;
    } else {
        pencil = (int)blockIdx.x / nsegm;
        const int seg = (int)blockIdx.x - pencil * nsegm;
        sb = min(seg * seg_slabs, M);
        se = min(sb + seg_slabs, M);
    }
    if (se <= sb) continue;
    const int bin0 = b * g.tiles_per_batch + pencil * g.np0;  // one plan bin per slab
    const int p_begin = tile_offsets[bin0 + sb], p_end = tile_offsets[bin0 + se];
    if (p_begin == p_end) continue;
    const int j2 = pencil % g.nta[2], j1 = pencil / g.nta[2];
    const int tb1 = j1 * g.Ta[1], tb2 = j2 * g.Ta[2];
    const float sc = win_exp_scale(m);
    float norm = win_norm(m);
    norm = norm * norm * norm;

    // row of this lane inside the plane and its eight 8-column pieces (two 16-byte loads each)
    const int64_t grow_off = (int64_t)wrap_near(tb1 - m + r32, M) * M;
    const int ccol0 = tb2 - m + 8 * h;  // + 16 ks
    // Periodic wrap of the padded columns (edge pencils only): 4-float pieces are wrapped as a whole; if the boundary
    // would cut through a piece (start column not a multiple of 4) the tile is read float by float.
    const bool wraps = tb2 - m < 0 || tb2 - m + 64 > M;
    const bool straddle = wraps && ((tb2 - m) & 3) != 0;  // workgroup-uniform
    const int cbase = ccol0 < 0 ? ccol0 + M : ccol0;      // first column of this lane's pieces, in [0, M)

    auto load_tile = [&](const int z, f32x4 (&raw)[8]) {
        const float *const prow = gplane + (int64_t)wrap(z, M) * M * M + grow_off;
        if (!straddle) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = 16 * (e >> 1) + 4 * (e & 1);
                const int c = cbase + k;
                raw[e] = *(const f32x4_dw *)(prow + (c >= M ? c - M : c));
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = cbase + 16 * (e >> 1) + 4 * (e & 1);
                f32x4 v;
                v.x = prow[c >= M ? c - M : c];
                v.y = prow[c + 1 >= M ? c + 1 - M : c + 1];
                v.z = prow[c + 2 >= M ? c + 2 - M : c + 2];
                v.w = prow[c + 3 >= M ? c + 3 - M : c + 3];
                raw[e] = v;
            }
        }
    };

    for (int c_begin = p_begin; c_begin < p_end; c_begin += kIcBlocks * 32) {
        const int npts = min(kIcBlocks * 32, p_end - c_begin);
        const int nblk = (npts + 31) >> 5;
        __syncthreads();  // the previous group's results have left the staging area

        // ---- shared tables of the group's blocks: wave w builds blocks w, w + 8 ----------------------------------
        for (int j = wave; j < nblk; j += kIcWaves) {
            const int pt = c_begin + 32 * j + r32;
            const bool valid = pt < p_end;
            int c0 = 0, c1 = 0, c2 = 0;
            float f0 = 0.f, f1 = 0.f, f2 = 0.f;
            if (valid) {
                const f32x4 rec = *(const f32x4 *)(spos + (int64_t)pt * 4);  // plan record {p0, p1, p2, x}
                split_cell(rec.x, M, c0, f0);
                split_cell(rec.y, M, c1, f1);
                split_cell(rec.z, M, c2, f2);
            }
            const int nvalid = min(32, p_end - (c_begin + 32 * j));
            const int zf = __builtin_amdgcn_readlane(c0, 0) - m;
            const int zl = __builtin_amdgcn_readlane(c0, nvalid - 1) + m + 1;
            if (lane == 0) { L.zf[j] = zf; L.zl[j] = zl; }
            IcBlock &Bk = L.blk[j];
            Bk.w1[4][lane] = f32x4{f0, __int_as_float(valid ? c0 : -(1 << 28)), 0.0f, 0.0f};  // padding: outside every window
            // B fragments: psi2 of my point on the padded columns 16 ks + 8 h + jj (zero outside the window)
            const int o2 = c2 - tb2;  // padded column of tap 0
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                float w[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int l2 = 16 * ks + 8 * h + jj - o2;
                    const float d = f2 + (float)(m - l2);
                    const float ev = __builtin_amdgcn_exp2f(sc * d * d) * kOpScale;
                    w[jj] = (valid && (unsigned)l2 < (unsigned)W) ? ev : 0.0f;
                }
                unsigned h0, h1, h2, h3, q0, q1, q2, q3;
                split_pair(w[0], w[1], h0, q0);
                split_pair(w[2], w[3], h1, q1);
                split_pair(w[4], w[5], h2, q2);
                split_pair(w[6], w[7], h3, q3);
                Bk.bfrag[ks][0][lane] = __builtin_bit_cast(f16x8, u32x4{h0, h1, h2, h3});
                Bk.bfrag[ks][1][lane] = __builtin_bit_cast(f16x8, u32x4{q0, q1, q2, q3});
            }
            // psi1 of my point on the 16 rows this lane holds of every T_z (MFMA result layout)
            const int o1 = c1 - tb1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float wv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = r + 8 * q + 4 * h;
                    const int l1 = row - o1;
                    const float d = f1 + (float)(m - l1);
                    const float ev = __builtin_amdgcn_exp2f(sc * d * d);
                    wv[r] = (valid && (unsigned)l1 < (unsigned)W) ? ev : 0.0f;
                }
                Bk.w1[q][lane] = f32x4{wv[0], wv[1], wv[2], wv[3]};
            }
        }
        __syncthreads();

        // ---- plane sweep of this wave's column -------------------------------------------------------------------
        for (int j = 0; j < nblk; ++j) L.yacc[wave][j][lane] = 0.0f;  // (read and written by this lane only)
        if (col_active) {
            const int z_first = L.zf[0], z_last = L.zl[nblk - 1];
            // windows of the blocks, wave-uniform: lane j holds block j's.  They are sorted (the points are sorted by
            // slab), so the blocks whose window holds plane z are a range [jlo, jhi) that only moves forward.
            const int zf_l = lane < nblk ? L.zf[lane] : (1 << 28);
            const int zl_l = lane < nblk ? L.zl[lane] : (1 << 28);
            int jlo = 0, jhi = 0;
            // one plane of the sweep: `raw` holds its tile (requested a whole plane step earlier)
            auto do_plane = [&](const int z, f32x4 (&raw)[8]) {
                while (jhi < nblk && __builtin_amdgcn_readlane(zf_l, jhi) <= z) ++jhi;
                while (jlo < jhi && __builtin_amdgcn_readlane(zl_l, jlo) < z) ++jlo;
                // power-of-two scale: max |G| of the tile lands in [1024, 2048); odd planes enter negated (the MFMA
                // accumulation truncates with a small sign-independent bias that cancels over alternating planes)
                float mx = 0.0f;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    mx = fmaxf(fmaxf(fmaxf(mx, fabsf(raw[e].x)), fabsf(raw[e].y)), fmaxf(fabsf(raw[e].z), fabsf(raw[e].w)));
                for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
                if (mx > 0.0f && jlo < jhi) {  // (an all-zero tile adds nothing)
                    float scale = 1.0f, pinv = 1.0f / kOpScale;
                    if (mx > 1.0e-30f && mx < 3.0e38f) {
                        int ex;
                        frexpf(mx, &ex);
                        scale = ldexpf(1.0f, 11 - ex);
                        pinv = ldexpf(1.0f, ex - 11) * (1.0f / kOpScale);
                    }
                    if (z & 1) { scale = -scale; pinv = -pinv; }
                    u32x4 ah[4], al[4];
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const f32x4 a = raw[2 * ks], c = raw[2 * ks + 1];
                        unsigned h0, h1, h2, h3, q0, q1, q2, q3;
                        split_pair(a.x * scale, a.y * scale, h0, q0);
                        split_pair(a.z * scale, a.w * scale, h1, q1);
                        split_pair(c.x * scale, c.y * scale, h2, q2);
                        split_pair(c.z * scale, c.w * scale, h3, q3);
                        ah[ks] = u32x4{h0, h1, h2, h3};
                        al[ks] = u32x4{q0, q1, q2, q3};
                    }
                    for (int j = jlo; j < jhi; ++j) {
                        const IcBlock &Bk = L.blk[j];
                        f32x16 acc = 0.0f;
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) {
                            const f16x8 bh = Bk.bfrag[ks][0][lane], bl = Bk.bfrag[ks][1][lane];
                            const f16x8 ahk = __builtin_bit_cast(f16x8, ah[ks]), alk = __builtin_bit_cast(f16x8, al[ks]);
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahk, bh, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahk, bl, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(alk, bh, acc, 0, 0, 0);
                        }
                        float t = 0.0f;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 wq = Bk.w1[q][lane];
                            t = fmaf(wq.x, acc[4 * q + 0], t);
                            t = fmaf(wq.y, acc[4 * q + 1], t);
                            t = fmaf(wq.z, acc[4 * q + 2], t);
                            t = fmaf(wq.w, acc[4 * q + 3], t);
                        }
                        // axis-0 weight of plane z for my point (zero outside its window), times the tile's scale
                        const f32x4 aux = Bk.w1[4][lane];
                        const int l0 = z - (__float_as_int(aux.y) - m);
                        const float d0 = aux.x + (float)(m - l0);
                        float p0 = __builtin_amdgcn_exp2f(sc * d0 * d0) * pinv;
                        p0 = (unsigned)l0 < (unsigned)W ? p0 : 0.0f;
                        L.yacc[wave][j][lane] = fmaf(p0, t, L.yacc[wave][j][lane]);
                    }
                }
                // the registers of this tile take the tile two planes ahead (the next plane's is already in flight)
                if (z + 2 <= z_last) load_tile(z + 2, raw);
            };
            f32x4 ta[8], tb[8];
            load_tile(z_first, ta);
            if (z_first + 1 <= z_last) load_tile(z_first + 1, tb);
            for (int z = z_first; z <= z_last; z += 2) {
                do_plane(z, ta);
                if (z + 1 <= z_last) do_plane(z + 1, tb);
            }
        }
        // this lane's sums out of the LDS before the block area becomes the result stage
        float ysum[kIcBlocks];
#pragma unroll
        for (int j = 0; j < kIcBlocks; ++j) ysum[j] = j < nblk ? L.yacc[wave][j][lane] : 0.0f;
        __syncthreads();  // every wave is done with the fragments
#pragma unroll
        for (int j = 0; j < kIcBlocks; ++j) {
            float y = ysum[j];
            y += __shfl_xor(y, 32);  // the two row halves of the point
            if (j < nblk && h == 0) ystage[(j * 32 + r32) * kIcStageStride + wave] = y * norm;
        }
        __syncthreads();
        // thread -> (point of the group, column): the columns of a point are contiguous in y
        for (int e = tid; e < npts * kIcWaves; e += kIcThreads) {
            const int p = e / kIcWaves, w = e - p * kIcWaves;
            const int c = cr0 + w;
            const int64_t pl = (int64_t)b * Cr + c;
            if (c < Cr && pl >= plane0 && pl < plane0 + nplanes)
                yr[(int64_t)__float_as_int(spos[(int64_t)(c_begin + p) * 4 + 3]) * Cr + c] = ystage[p * kIcStageStride + w];
        }
    }
    }  // work items
}

} // namespace

// Worth it from 4 real columns up (half the waves busy); 3-D wide tiling only.
bool interp_cols_supported(const Geom &g, int64_t Cr)
{
    static const bool off = [] {
        const char *env = std::getenv("NFFT_HIP_GATHER");
        return env && (env[0] == 'l' || env[0] == 'm');  // lds: lane-per-point kernel, mfma: plane-ring kernel
    }();
    return !off && g.dim == 3 && g.wide && !g.owned && Cr >= 4;
}

template <int W>
static int launch_ic_t(const Geom &g, const PlanLayout &L, const void *plan, const float *grid, int64_t n, int64_t Cr,
                       int64_t plane0, int64_t nplanes, float *yr, hipStream_t stream)
{
    const char *base = (const char *)plan;
    const int *to = (const int *)(base + L.off_offsets);
    const float *spos = (const float *)(base + L.off_spos);
    const int4 *work = (const int4 *)(base + L.off_work), *sorted = work + L.work_head + L.work_cap;
    const int64_t pencils = (int64_t)g.nta[1] * g.nta[2];
    int64_t nsets = g.tiles_per_batch > 0 ? L.ntiles / g.tiles_per_batch : 1;
    if (nsets < 1) nsets = 1;
    const int nsegm = seg_base_runs(n, nsets, pencils, g.M, device_cu_count());
    const int seg_slabs = (g.M + nsegm - 1) / nsegm;
    // column groups that intersect the planes [plane0, plane0 + nplanes)
    const int64_t G = (Cr + kIcWaves - 1) / kIcWaves;
    auto group_of = [&](int64_t pl) { return (pl / Cr) * G + (pl % Cr) / kIcWaves; };
    const int64_t group0 = group_of(plane0), ngroups = group_of(plane0 + nplanes - 1) - group0 + 1;
    if (ngroups > 65535) { set_error("Input mismatch: too many planes for one interpolate call"); return 1; }
    static DeviceOnce attr_done;
    if (attr_done.first_use()) {
        NFFT_HIP_CHECK(hipFuncSetAttribute((const void *)interp_cols_kernel<W, false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(IcLds)));
        NFFT_HIP_CHECK(hipFuncSetAttribute((const void *)interp_cols_kernel<W, true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(IcLds)));
        attr_done.mark();
    }
    const dim3 blocks((unsigned)(pencils * nsegm), (unsigned)ngroups);
    hipLaunchKernelGGL((interp_cols_kernel<W, false>), blocks, dim3(kIcThreads), sizeof(IcLds), stream, g, to, spos,
                       grid, (int)Cr, plane0, nplanes, group0, yr, seg_slabs, nsegm, work, sorted, WorkTickets{nullptr, 0u});
    // the persistent launch over the work list (unbalanced plans; its workgroups leave at once otherwise); entries are
    // handed out by tickets when the launch's planes fit its share of the ticket ring, else round robin
    const WorkTickets tickets{ngroups <= kTicketPlanes ? device_ticket_ring() : nullptr, next_launch_number()};
    const dim3 oblocks(work_list_workgroups(n, nsets, pencils, nsegm, device_cu_count()), (unsigned)ngroups);
    hipLaunchKernelGGL((interp_cols_kernel<W, true>), oblocks, dim3(kIcThreads), sizeof(IcLds), stream, g, to,
                       spos, grid, (int)Cr, plane0, nplanes, group0, yr, seg_slabs, nsegm, work, sorted, tickets);
    NFFT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_interp_cols(const Geom &g, const PlanLayout &L, const void *plan, const float *grid, int64_t n, int64_t Cr,
                       int64_t plane0, int64_t nplanes, float *yr, hipStream_t stream)
{
    if (nplanes <= 0 || n <= 0) return 0;
    switch (g.m) {
    case 1: return launch_ic_t<4>(g, L, plan, grid, n, Cr, plane0, nplanes, yr, stream);
    case 2: return launch_ic_t<6>(g, L, plan, grid, n, Cr, plane0, nplanes, yr, stream);
    case 3: return launch_ic_t<8>(g, L, plan, grid, n, Cr, plane0, nplanes, yr, stream);
    case 4: return launch_ic_t<10>(g, L, plan, grid, n, Cr, plane0, nplanes, yr, stream);
    case 5: return launch_ic_t<12>(g, L, plan, grid, n, Cr, plane0, nplanes, yr, stream);
    case 6: return launch_ic_t<14>(g, L, plan, grid, n, Cr, plane0, nplanes, yr, stream);
    case 7: return launch_ic_t<16>(g, L, plan, grid, n, Cr, plane0, nplanes, yr, stream);
    }
    set_error("matrix-core interpolation supports cutoff 1..7");
    return 1;
}

} // namespace nfft
