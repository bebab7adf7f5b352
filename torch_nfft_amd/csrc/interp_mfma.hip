// Interpolation (forward gather) on the matrix cores, 3-D grids with the wide pencil tiling.
//
// Same result as interp.hip / the reference's forward_window_convolution kernels
// (csrc/cuda/spatial_window_operations.cu:214-332).  For one plane z of a pencil and a block of 32 points
//     T_z[u1, i] = sum_{u2} G_z[u1, u2] psi2_i[u2]           -- a GEMM: (32 rows x 64 columns) x (64 x 32 points)
//     y_i       += psi0_i[z] * sum_{u1} psi1_i[u1] T_z[u1, i]
// i.e. the innermost window axis is contracted by v_mfma_f32_32x32x16_f16 (two-way f16 split operands, fp32
// accumulation, see mfma_split.h), the other two by 17 FMAs per lane and plane:
//   * the 16 planes a chunk of 17 - (2m+2) slabs needs are resident in LDS, already split into f16 hi / lo and laid
//     out as MFMA A fragments (8 KB per plane; slot = plane mod 16, so advancing a chunk only overwrites the planes
//     that fell out of the window).  Every plane is scaled by its own power of two (max |G| of the tile -> [1024,
//     2048)), undone per plane when the row sums are accumulated;
//   * a wave takes 32 consecutive points of the slab-sorted plan: lane (i, h) builds the B fragments of point i
//     (psi2 on the 64 padded columns, zero outside the window) once, then for every plane of the block's window
//     issues 12 MFMAs (4 k-steps x {hi hi, hi lo, lo hi}) and folds the 16 rows it holds with its psi1 weights;
//   * no atomics, no cross-wave reduction: the two half-sums of a point meet in one DPP add at the end.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "kernels.h"
#include "mfma_split.h"

namespace nfft {

namespace {

constexpr int kGmThreads = 1024;
constexpr int kGmWaves = kGmThreads / 64;
constexpr int kRing = 16;  // resident planes: TC + 2m+1 = 16 for every cutoff of the wide tiling

struct __align__(16) GatherMfmaLds {
    f16x8 frag[kRing][4][2][64];  // [plane slot][k-step][hi/lo][lane = 32 (column half) + row]
    unsigned pmax[kRing];         // bit pattern of max |G| over the plane tile (staging)
    float pinv[kRing];            // what one unit of the scaled plane is worth, times the B operand scale
    int ticket;                   // work-list entry of the workgroup (persistent launch: next_work_item)
};

template <int W, bool OVERFLOW>
__global__ void __launch_bounds__(kGmThreads) __attribute__((amdgpu_waves_per_eu(4, 4)))
interp_mfma_kernel(const Geom g, const int *__restrict__ tile_offsets,
                   const float *__restrict__ spos, const float *__restrict__ grid, const int Cr, const int plane0,
                   float *__restrict__ yr, const int seg_slabs, const int nsegm, const int4 *__restrict__ work, const int4 *__restrict__ sorted, const WorkTickets tickets)
{
    constexpr int m = W / 2 - 1;
    constexpr int TC = 17 - W;
    static_assert(TC >= 1 && TC + W - 1 == kRing, "ring holds exactly one chunk's planes");
    extern __shared__ __align__(16) unsigned char smem_raw[];
    GatherMfmaLds &L = *reinterpret_cast<GatherMfmaLds *>(smem_raw);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform by construction: keep it in an SGPR
    const int r32 = lane & 31, h = lane >> 5;

    const int plane_local = blockIdx.y;
    const int plane = plane0 + plane_local;
    const int b = plane / Cr;
    const int cr = plane - b * Cr;
    const int pencils = g.nta[1] * g.nta[2];

    // ---- work items: the same ranges of slabs (or, for unbalanced plans, the entries of the plan's work list) as the
    // spreading kernel (spread_mfma.hip); an item owns the chunks whose first slab lies in its range.
    // (work items as in spread_mfma.hip: one workgroup per range, or a persistent grid over the plan's work list)
    const int listed = work[0].z;
    if (OVERFLOW ? !listed : listed) return;
    // (a plane walks its own point set's part of the sorted list: set_hdr[b] = {entries, first entry})
    const int2 set_hdr = OVERFLOW ? ((const int2 *)(work + 1))[b] : make_int2(1, 0);
    const int n_items = set_hdr.x;
    const int4 *const entries = sorted + set_hdr.y;
    for (int item = OVERFLOW ? next_work_item(tickets, &L.ticket, -1, plane_local) : 0; item < n_items;
         item = OVERFLOW ? next_work_item(tickets, &L.ticket, item, plane_local) : 1) {
    if (OVERFLOW && item != (int)blockIdx.x) __syncthreads();  // the previous item is done with the LDS
    int pencil, sb, se;
    if constexpr (OVERFLOW) {
        const int4 it = tickets.ring ? entries[item] : listed_item(entries, item, n_items);
        pencil = it.x - b * pencils;
        sb = it.y;
        se = it.z;
    } else {
        pencil = (int)blockIdx.x / nsegm;
        const int seg = (int)blockIdx.x - pencil * nsegm;
        sb = min(seg * seg_slabs, g.M);
        se = min(sb + seg_slabs, g.M);
    }
    const int j2 = pencil % g.nta[2];
    const int j1 = pencil / g.nta[2];
    const int k_begin = (sb + TC - 1) / TC;
    const int k_end = min(g.nta[0], (se + TC - 1) / TC);
    if (k_begin >= k_end) continue;
    const int bin0 = b * g.tiles_per_batch + pencil * g.np0;
    {
        int s0, e0, s1, e1;
        chunk_range(g, tile_offsets, bin0, k_begin, s0, e0);
        chunk_range(g, tile_offsets, bin0, k_end - 1, s1, e1);
        if (s0 == e1) continue;  // no points in these chunks
    }

    const int tb1 = j1 * g.Ta[1], tb2 = j2 * g.Ta[2];
    const float sc = win_exp_scale(m);
    float norm = win_norm(m);
    norm = norm * norm * norm;
    const float *const gplane = grid + (int64_t)plane_local * g.cells;
    const int M = g.M;

    // resident planes: unwrapped z in [res_lo, res_hi), plane z in slot z & 15
    int res_lo = 0, res_hi = 0;

    for (int k = k_begin; k < k_end; ++k) {
        int s, e;
        chunk_range(g, tile_offsets, bin0, k, s, e);
        if (e == s) continue;
        const int zl = k * TC - m, zh = zl + kRing;
        const int new_lo = (res_hi > zl && res_lo <= zl) ? res_hi : zl;  // planes [new_lo, zh) have to be fetched

        // ---- stage the new planes, 8 at a time: thread task -> (plane, row, group of 8 columns) ---------------
        __syncthreads();  // every wave is done with the planes about to be replaced
        if (tid < kRing) L.pmax[tid] = 0u;
        __syncthreads();
        for (int pz = new_lo; pz < zh; pz += 8) {
            const int ntask = min(8, zh - pz) * 256;
            float v[2][8];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int t = tid + r * kGmThreads;
                const bool live = t < ntask;
                // a wave's 64 tasks: 16 rows x 4 column groups (half rows of 128 contiguous bytes in HBM; 16
                // consecutive lanes write 16 consecutive 16-byte LDS slots: no bank conflicts)
                const int cg = ((t >> 4) & 3) + 4 * ((t >> 7) & 1), row = (t & 15) + 16 * ((t >> 6) & 1), z = pz + (t >> 8);
                const int64_t gz = wrap(z, M);
                const int64_t g1 = wrap_near(tb1 - m + row, M);
                const float *const grow = gplane + (gz * M + g1) * M;
                const int c0 = tb2 - m + 8 * cg;
                if (live && c0 >= 0 && c0 + 8 <= M) {
                    const f32x4 a = *(const f32x4_dw *)(grow + c0), bq = *(const f32x4_dw *)(grow + c0 + 4);
                    v[r][0] = a.x; v[r][1] = a.y; v[r][2] = a.z; v[r][3] = a.w;
                    v[r][4] = bq.x; v[r][5] = bq.y; v[r][6] = bq.z; v[r][7] = bq.w;
                } else {
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) v[r][jj] = live ? grow[wrap_near(c0 + jj, M)] : 0.0f;
                }
                // a wave's 64 tasks belong to one plane: wave-reduce, then one LDS max per wave
                float mx = 0.0f;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) mx = fmaxf(mx, fabsf(v[r][jj]));
                for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
                if (lane == 0 && live) atomicMax(&L.pmax[z & (kRing - 1)], __float_as_uint(mx));
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int t = tid + r * kGmThreads;
                if (t < ntask) {
                    const int cg = ((t >> 4) & 3) + 4 * ((t >> 7) & 1), row = (t & 15) + 16 * ((t >> 6) & 1);
                    const int slot = (pz + (t >> 8)) & (kRing - 1);
                    // power-of-two scale: max |G| of the plane tile lands in [1024, 2048).  Odd planes are stored
                    // negated (and un-negated through pinv): the MFMA accumulation truncates with a small
                    // sign-independent bias (scripts/ubench/mfma_bias.hip) that cancels over the alternating planes
                    // of a window instead of adding up over millions of outputs.
                    const float mx = __uint_as_float(L.pmax[slot]);
                    float scale = 1.0f, inv = 1.0f;
                    if (mx > 1.0e-30f && mx < 3.0e38f) {
                        int ex;
                        frexpf(mx, &ex);
                        scale = ldexpf(1.0f, 11 - ex);
                        inv = ldexpf(1.0f, ex - 11);
                    }
                    if (slot & 1) { scale = -scale; inv = -inv; }
                    unsigned h0, h1, h2, h3, q0, q1, q2, q3;
                    split_pair(v[r][0] * scale, v[r][1] * scale, h0, q0);
                    split_pair(v[r][2] * scale, v[r][3] * scale, h1, q1);
                    split_pair(v[r][4] * scale, v[r][5] * scale, h2, q2);
                    split_pair(v[r][6] * scale, v[r][7] * scale, h3, q3);
                    const u32x4 hi = {h0, h1, h2, h3}, lo = {q0, q1, q2, q3};
                    const int ln = 32 * (cg & 1) + row;
                    L.frag[slot][cg >> 1][0][ln] = __builtin_bit_cast(f16x8, hi);
                    L.frag[slot][cg >> 1][1][ln] = __builtin_bit_cast(f16x8, lo);
                    if (cg == 0 && row == 0) L.pinv[slot] = inv * (1.0f / kOpScale);
                }
            }
        }
        res_lo = zl;
        res_hi = zh;
        __syncthreads();

        // ---- blocks of 32 consecutive points, dealt round-robin to the waves ------------------------------
        for (int j0 = s + wave * 32; j0 < e; j0 += kGmWaves * 32) {
            const int j = j0 + r32;
            const bool valid = j < e;
            int c0 = 0, c1 = 0, c2 = 0;
            float f0 = 0.f, f1 = 0.f, f2 = 0.f;
            if (valid) {
                const f32x4 rec = *(const f32x4 *)(spos + (int64_t)j * 4);  // plan record {p0, p1, p2, x}
                split_cell(rec.x, M, c0, f0);
                split_cell(rec.y, M, c1, f1);
                split_cell(rec.z, M, c2, f2);
            }
            // the plan is sorted by slab: the block's planes run from the first point's window to the last one's
            const int nvalid = min(32, e - j0);
            const int z_first = __builtin_amdgcn_readlane(c0, 0) - m;
            const int z_last = __builtin_amdgcn_readlane(c0, nvalid - 1) + m + 1;

            // B fragments: psi2 of my point on the padded columns 16 ks + 8 h + jj (zero outside the window)
            u32x4 bh[4], bl[4];
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
                bh[ks] = u32x4{h0, h1, h2, h3};
                bl[ks] = u32x4{q0, q1, q2, q3};
            }
            // psi1 of my point on the 16 rows this lane holds of every T_z (MFMA result layout)
            float w1[16];
            const int o1 = c1 - tb1;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
                const int l1 = row - o1;
                const float d = f1 + (float)(m - l1);
                const float ev = __builtin_amdgcn_exp2f(sc * d * d);
                w1[reg] = (unsigned)l1 < (unsigned)W ? ev : 0.0f;
            }

            float y = 0.0f;
            for (int z = z_first; z <= z_last; ++z) {
                const int slot = z & (kRing - 1);
                f32x16 acc = 0.0f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const f16x8 ah = L.frag[slot][ks][0][lane], al = L.frag[slot][ks][1][lane];
                    const f16x8 bhk = __builtin_bit_cast(f16x8, bh[ks]), blk = __builtin_bit_cast(f16x8, bl[ks]);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bhk, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, blk, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bhk, acc, 0, 0, 0);
                }
                float t = 0.0f;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) t = fmaf(w1[reg], acc[reg], t);
                // axis-0 weight of plane z for my point (zero outside its window), times the plane's scale
                const int l0 = z - (c0 - m);
                const float d0 = f0 + (float)(m - l0);
                float p0 = __builtin_amdgcn_exp2f(sc * d0 * d0) * L.pinv[slot];
                p0 = (unsigned)l0 < (unsigned)W ? p0 : 0.0f;
                y = fmaf(p0, t, y);
            }
            y += __shfl_xor(y, 32);  // the two row halves of the point
            if (valid && h == 0) yr[(int64_t)__float_as_int(spos[(int64_t)j * 4 + 3]) * Cr + cr] = y * norm;  // (index: fourth word of the record)
        }
    }
    }  // work items
}

} // namespace

bool interp_mfma_supported(const Geom &g) { return g.dim == 3 && g.wide; }

template <int W>
static int launch_gm_t(const Geom &g, const PlanLayout &L, const void *plan, const int *to,
                       const float *spos, const float *grid, int64_t n, int64_t Cr, int64_t plane0, int64_t nplanes,
                       float *yr, hipStream_t stream)
{
    // the work decomposition of the spreading kernel (ranges of M / runs slabs per pencil + the plan's work list
    // for dense ranges); every item starts by staging all 16 planes of its first chunk
    const int64_t pencils = (int64_t)g.nta[1] * g.nta[2];
    int64_t nsets = g.tiles_per_batch > 0 ? L.ntiles / g.tiles_per_batch : 1;
    if (nsets < 1) nsets = 1;
    const int nsegm = seg_base_runs(n, nsets, pencils, g.M, device_cu_count());
    const int seg_slabs = (g.M + nsegm - 1) / nsegm;
    const dim3 blocks((unsigned)(pencils * nsegm), (unsigned)nplanes);
    static DeviceOnce attr_done;
    if (attr_done.first_use()) {
        NFFT_HIP_CHECK(hipFuncSetAttribute((const void *)interp_mfma_kernel<W, false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(GatherMfmaLds)));
        NFFT_HIP_CHECK(hipFuncSetAttribute((const void *)interp_mfma_kernel<W, true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(GatherMfmaLds)));
        attr_done.mark();
    }
    const char *base = (const char *)plan;
    const int4 *work = (const int4 *)(base + L.off_work), *sorted = work + L.work_head + L.work_cap;
    hipLaunchKernelGGL((interp_mfma_kernel<W, false>), blocks, dim3(kGmThreads), sizeof(GatherMfmaLds), stream, g, to,
                       spos, grid, (int)Cr, (int)plane0, yr, seg_slabs, nsegm, work, sorted, WorkTickets{nullptr, 0u});
    // the persistent launch over the work list (unbalanced plans; its workgroups leave at once otherwise); entries are
    // handed out by tickets when the launch's planes fit its share of the ticket ring, else round robin
    const WorkTickets tickets{nplanes <= kTicketPlanes ? device_ticket_ring() : nullptr, next_launch_number()};
    const dim3 oblocks(work_list_workgroups(n, nsets, pencils, nsegm, device_cu_count()), (unsigned)nplanes);
    hipLaunchKernelGGL((interp_mfma_kernel<W, true>), oblocks, dim3(kGmThreads), sizeof(GatherMfmaLds), stream, g,
                       to, spos, grid, (int)Cr, (int)plane0, yr, seg_slabs, nsegm, work, sorted, tickets);
    NFFT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_interp_mfma(const Geom &g, const PlanLayout &L, const void *plan, const float *grid, int64_t n, int64_t Cr,
                       int64_t plane0, int64_t nplanes, float *yr, hipStream_t stream)
{
    const char *base = (const char *)plan;
    const int *to = (const int *)(base + L.off_offsets);
    const float *spos = (const float *)(base + L.off_spos);
    if (nplanes <= 0 || n <= 0) return 0;
    switch (g.m) {
    case 1: return launch_gm_t<4>(g, L, plan, to, spos, grid, n, Cr, plane0, nplanes, yr, stream);
    case 2: return launch_gm_t<6>(g, L, plan, to, spos, grid, n, Cr, plane0, nplanes, yr, stream);
    case 3: return launch_gm_t<8>(g, L, plan, to, spos, grid, n, Cr, plane0, nplanes, yr, stream);
    case 4: return launch_gm_t<10>(g, L, plan, to, spos, grid, n, Cr, plane0, nplanes, yr, stream);
    case 5: return launch_gm_t<12>(g, L, plan, to, spos, grid, n, Cr, plane0, nplanes, yr, stream);
    case 6: return launch_gm_t<14>(g, L, plan, to, spos, grid, n, Cr, plane0, nplanes, yr, stream);
    case 7: return launch_gm_t<16>(g, L, plan, to, spos, grid, n, Cr, plane0, nplanes, yr, stream);
    }
    set_error("matrix-core interpolation supports cutoff 1..7");
    return 1;
}

} // namespace nfft
