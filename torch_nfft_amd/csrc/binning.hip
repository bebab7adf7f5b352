// Point plan: counting sort of the points by (batch, pencil, chunk) tile -- (batch, pencil, slab) for the wide tiling
// of the matrix-core kernels, plus their load-balance tables (segment_split_kernel).
//
// Replaces the reference's per-point HBM temporaries (shifts: spatial_window_operations.cu:38-61,
// psi: :68-97; allocated at core_cuda.cu:188-211 and re-read 2m+2 times per axis by the spreading
// kernel).  Here the points themselves are reordered once (three streaming passes over n points) and
// every consumer re-derives cell index and window weights in registers.
#include <algorithm>
#include <hipcub/hipcub.hpp>

#include "common.h"
#include "kernels.h"

namespace nfft {

constexpr int kSortBlockPoints = 8192;  // most points per workgroup in the first-level passes: runs of ~6 16-byte records
                                        // per (workgroup, bin) at C3.  Measured at the end of round 3 (lean kernels, plan stage
                                        // at C3): 4 096: 0.368, 8 192: 0.334, 16 384: 0.367, 32 768: 0.386 ms
constexpr int kSortThreads = 512;
constexpr int kMaxPencilsLds = 8192;    // first-level bins that fit an LDS histogram
constexpr int kSort2Parts = 8;          // workgroups that share one first-level bin in the second level (a dense bin of a
                                        // clustered input is otherwise one workgroup's serial loop: 1.4 ms at C3-clustered)

PlanLayout plan_layout(const Geom &g, int64_t n, int64_t B)
{
    PlanLayout L;
    L.cap = g.owned ? 4 * n : n;
    L.ntiles = (int64_t)g.tiles_per_batch * B * g.SB;  // (tile, sub-block) bins
    L.npencils = (int64_t)g.nta[1] * g.nta[2] * B * g.l1seg;  // first-level bins: (batch, pencil, segment)
    // first-level slices: 16 384 points for big inputs, smaller ones (>= 1 024) as long as that gives fewer than ~512
    // workgroups -- at n = 10^5 seven workgroups of 16 384 took 22 + 14 us for the two passes
    L.block_points = 1024;
    while (L.block_points < kSortBlockPoints &&
           (n > L.block_points * 512 || L.npencils * ((n + L.block_points - 1) / L.block_points) > (int64_t(1) << 20)))
        L.block_points *= 2;  // (and a table of per-(bin, slice) counts that stays small next to the points)
    L.nblocks = (n + L.block_points - 1) / L.block_points;
    L.two_level = L.npencils <= kMaxPencilsLds && L.npencils * L.nblocks < (int64_t(1) << 28) && n > 0;
    const int64_t scan_items = L.two_level ? L.npencils * L.nblocks + 1 : L.ntiles + 1;
    // hipcub's temporary storage for an int scan grows with the item count; bound it without touching the device
    L.scan_bytes = align_up(scan_items / 64 + 65536, 256);
    int64_t o = 0;
    L.off_offsets = o; o = align_up(o + (L.ntiles + 1) * 4, 256);
    L.off_cursor = o;  o = align_up(o + (L.ntiles + 1) * 4, 256);
    L.off_seal = o;    o += kSealBytes;  // the plan's seal (kernels.h), right behind the cursors: one memset zeroes both
    L.off_perm = o;    o = align_up(o + (g.dim == 3 ? 0 : L.cap * 4), 256);  // (3-D: the index sits in the record)
    L.off_spos = o;    o = align_up(o + L.cap * g.pstride * 4, 256);
    L.off_scan = o;    o = align_up(o + L.scan_bytes, 256);
    // scratch of the two-level sort: per-(pencil, block) counts + their scan, and the pencil-ordered records
    L.off_hist = o;    o = align_up(o + (L.two_level ? scan_items * 4 : 0), 256);
    L.off_hscan = o;   o = align_up(o + (L.two_level ? scan_items * 4 : 0), 256);
    L.off_tmp = o;     o = align_up(o + (L.two_level ? L.cap * 16 : 0), 256);
    // second level: per (first-level bin, part) counts of the fine keys
    L.off_hist2 = o;   o = align_up(o + (L.two_level ? L.npencils * kSort2Parts * (int64_t)g.l1bins * g.SB * g.CG * 4 : 0), 256);
    L.grouped = L.two_level && g.CG == 3;
    L.off_groups = o;  o = align_up(o + (L.grouped ? L.ntiles * 2 * 4 : 0), 256);
    // work list of the matrix-core kernels (wide tiling; segment_split_kernel / work_order_kernel below): a header,
    // the list and its copy in launch order.  ranges + entries / 2048 bounds the list (pieces hold >= 2048 points)
    L.work_cap = g.wide ? (int64_t)g.nta[1] * g.nta[2] * B * kSegMax + L.cap / 2048 + 16 : 0;
    L.work_head = 1 + (B * 8 + 15) / 16;
    L.off_work = o;    o = align_up(o + (g.wide ? (L.work_head + 2 * L.work_cap) * 16 : 0), 256);
    // the count passes leave the keys they computed for the scatter passes (two bytes per point / record instead of
    // two or three more split_cell + tile-index evaluations)
    L.off_sealpart = o; o = align_up(o + (L.two_level ? L.nblocks * 8 : 0), 256);  // per-slice shares of the seal (count pass)
    L.off_key1 = o;    o = align_up(o + (L.two_level && !g.owned ? n * 2 : 0), 256);
    L.off_key2 = o;    o = align_up(o + (L.two_level ? L.cap * 2 : 0), 256);
    L.total = o;
    return L;
}

// Plan bins of point i: one, or up to four for the owned tiling (one per tile its window touches).
__device__ __forceinline__ int point_tiles(const Geom &g, const float *__restrict__ pos, const int64_t *__restrict__ batch,
                                           int64_t i, int64_t B, int tiles[4])
{
    int cell[3] = {0, 0, 0};
    for (int u = 0; u < g.dim; ++u) {
        float fr;
        split_cell(pos[i * g.dim + u], g.M, cell[u + 3 - g.dim], fr);
    }
    int64_t b = batch ? batch[i] : 0;
    b = b < 0 ? 0 : (b >= B ? B - 1 : b);  // (the host layer rejects a batch vector that is not in [0, B))
    if (g.owned) {
        int pen[4];
        const int k = owned_pencils(g, cell[1], cell[2], pen);
        for (int q = 0; q < k; ++q) tiles[q] = (int)b * g.tiles_per_batch + pen[q] * g.np0 + cell[0];
        return k;
    }
    tiles[0] = ((int)b * g.tiles_per_batch + tile_of_cells(g, cell)) * g.SB + sub_of_cells(g, cell);
    return 1;
}

// ---- seal of a plan: checksum of the data it was built from -------------------------------------------------------------
// A plan cached across calls is keyed on tensor identity + version counter; a write that bypasses the counter (pos.data,
// a foreign kernel, a DLPack alias) would make the cached plan silently wrong -- the reference recomputes shifts and psi
// in every call (core_cuda.cu:188-211) and has no such state.  Every point enters an order-independent 64-bit sum with a
// 32-bit hash of its words and its index (bijective in every word: a changed word always changes the point's hash, two
// changes cancel with probability 2^-32).  The count passes of the sort form it on the side -- ~14 integer operations per
// point; a first version with one avalanche per WORD made the instruction-bound count pass 18 % slower -- and the
// verification of a cached plan recomputes it in one streaming pass (~25 us for 10^7 3-D points).
constexpr int kSealThreads = 1024;  // few, big workgroups: every one ends with two atomics on the SAME words, which serialise at
                                    // their L2 channel (2 441 workgroups of 256 threads: 120 us for 10^7 points; 512 of 1 024: see
                                    // profiles/r04_experiments.md -- the same effect as in plane_absmax_kernel)
__device__ __forceinline__ unsigned long long seal_point(int dim, int64_t i, float c0, float c1, float c2, bool has_batch, int64_t b)
{
    unsigned h = __float_as_uint(c0) * 0x9E3779B1u + (unsigned)i * 0x85EBCA77u + (unsigned)((unsigned long long)i >> 32) * 0x27D4EB2Fu;
    if (dim > 1) h += __float_as_uint(c1) * 0xC2B2AE3Du;
    if (dim > 2) h += __float_as_uint(c2) * 0x165667B1u;
    if (has_batch) h += (unsigned)b * 0x7FEB352Du + (unsigned)((unsigned long long)b >> 32) * 0x846CA68Bu;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return (unsigned long long)h;
}
// sum over the workgroup (256 .. 1024 threads), valid in thread 0
__device__ __forceinline__ unsigned long long seal_block_sum(unsigned long long sum, unsigned long long *part /* LDS, 16 words */)
{
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sum;
    __syncthreads();
    unsigned long long s = 0;
    if (threadIdx.x == 0)
        for (unsigned i = 0; i < (blockDim.x + 63) / 64; ++i) s += part[i];
    return s;
}

__global__ void __launch_bounds__(256) bin_count_kernel(Geom g, const float *__restrict__ pos,
                                                       const int64_t *__restrict__ batch, int64_t n, int64_t B,
                                                       int *__restrict__ count, unsigned long long *__restrict__ seal,
                                                       int *__restrict__ status)
{
    __shared__ unsigned long long part[16];
    unsigned long long sum = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (batch && (batch[i] < 0 || batch[i] >= B)) report_fault(status, kFaultBatchIndex);  // (binned clamped; the next call fails)
        int tiles[4];
        const int k = point_tiles(g, pos, batch, i, B, tiles);
        for (int q = 0; q < k; ++q) atomicAdd(&count[tiles[q]], 1);
        // the plan's seal (checksum of the points it is built from): this pass reads every word of pos / batch anyway
        sum += seal_point(g.dim, i, pos[i * g.dim], g.dim > 1 ? pos[i * g.dim + 1] : 0.f, g.dim > 2 ? pos[i * g.dim + 2] : 0.f,
                          batch != nullptr, batch ? batch[i] : 0);
    }
    sum = seal_block_sum(sum, part);
    if (threadIdx.x == 0 && sum) atomicAdd(seal, sum);
}

__global__ void __launch_bounds__(256) bin_fill_kernel(Geom g, const float *__restrict__ pos,
                                                      const int64_t *__restrict__ batch, int64_t n, int64_t B,
                                                      const int *__restrict__ offsets, int *__restrict__ cursor,
                                                      int *__restrict__ perm, float *__restrict__ spos)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int tiles[4];
        const int k = point_tiles(g, pos, batch, i, B, tiles);
        for (int q = 0; q < k; ++q) {
            const int slot = offsets[tiles[q]] + atomicAdd(&cursor[tiles[q]], 1);
            for (int u = 0; u < g.dim; ++u) spos[(int64_t)slot * g.pstride + u] = pos[i * g.dim + u];
            if (g.dim == 3) spos[(int64_t)slot * 4 + 3] = __int_as_float((int)i);
            else perm[slot] = (int)i;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Two-level counting sort without global atomics.
//   level 1: (batch, pencil) bins.  Each workgroup owns a slice of kSortBlockPoints points, histograms it in
//            LDS (ds_add_u32 runs at the ds_write_b32 rate), the per-(pencil, block) counts are scanned on
//            the device, and the slice is scattered with LDS cursors -> records grouped by pencil.
//   level 2: one workgroup per pencil counting-sorts its records by chunk in LDS and writes the final
//            tile offsets, the tile-ordered positions and the permutation.
// first-level bin: (batch, pencil, segment of l1bins plan bins along axis 0)
__device__ __forceinline__ int pencil_of(const Geom &g, const int cell[3], int64_t b)
{
    const int pencil = ((int)b * g.nta[1] + div_small(cell[1], g.Ta[1])) * g.nta[2] + div_small(cell[2], g.Ta[2]);
    if (g.l1seg == 1) return pencil;
    const int k0 = g.bin0 == 1 ? cell[0] : div_small(cell[0], g.bin0);
    return pencil * g.l1seg + div_small(k0, g.l1bins);
}

// first-level bins of a point given by its (up to three) coordinates, internal axis order: one bin, or up to four
// for the owned tiling (an entry per touched tile).  Returns the count.
__device__ __forceinline__ int l1_bins_of(const Geom &g, float c0, float c1, float c2, int64_t b, int bins[4])
{
    int cell[3] = {0, 0, 0};
    float fr;
    if (g.dim == 3) {
        split_cell(c0, g.M, cell[0], fr);
        split_cell(c1, g.M, cell[1], fr);
        split_cell(c2, g.M, cell[2], fr);
    } else if (g.dim == 2) {
        split_cell(c0, g.M, cell[1], fr);
        split_cell(c1, g.M, cell[2], fr);
    } else {
        split_cell(c0, g.M, cell[2], fr);
    }
    if (g.owned) {
        int pen[4];
        const int k = owned_pencils(g, cell[1], cell[2], pen);
        const int seg = div_small(g.bin0 == 1 ? cell[0] : div_small(cell[0], g.bin0), g.l1bins);
        for (int q = 0; q < k; ++q) bins[q] = ((int)b * g.nta[1] * g.nta[2] + pen[q]) * g.l1seg + seg;
        return k;
    }
    bins[0] = pencil_of(g, cell, b);
    return 1;
}

// coordinates of point i (zeros beyond the dimension); 3-D: one 12-byte load instead of three 4-byte ones
__device__ __forceinline__ void load_point(const float *__restrict__ pos, int dim, int64_t i, bool live, float &a0,
                                           float &a1, float &a2)
{
    a0 = a1 = a2 = 0.f;
    if (!live) return;
    if (dim == 3) {
        typedef float f32x3 __attribute__((ext_vector_type(3)));
        f32x3 v;
        __builtin_memcpy(&v, pos + i * 3, 12);
        a0 = v.x; a1 = v.y; a2 = v.z;
    } else {
        a0 = pos[i * dim];
        if (dim > 1) a1 = pos[i * dim + 1];
    }
}

// The first-level passes read kSortUnroll points per thread before touching the LDS counters: a wave then keeps
// several global loads in flight instead of one load -> atomic (-> store) chain per point.
constexpr int kSortUnroll = 8;

template <bool COMMON>
__global__ void __launch_bounds__(kSortThreads)
sort1_count_kernel(Geom g_in, const float *__restrict__ pos, const int64_t *__restrict__ batch, int64_t n, int64_t B,
                   int npencils, int nblocks, int block_points, int *__restrict__ hist /* [pencil][block] */,
                   unsigned short *__restrict__ key1 /* first-level bin of every point (not for the owned tiling) */,
                   unsigned long long *__restrict__ seal_part /* [block]: the slice's share of the plan's seal, or null */,
                   int *__restrict__ status)
{
    // COMMON: a 3-D problem on the scatter tiling (every benchmark configuration): with the two flags constant the
    // per-point paths lose their run-time branches on dimension and tiling (8-fold unrolled: 43 KB of code per kernel
    // before), the instruction-bound count passes most of all
    Geom g = g_in;
    if constexpr (COMMON) { g.dim = 3; g.owned = 0; }
    extern __shared__ int lds_hist[];
    for (int i = threadIdx.x; i < npencils; i += kSortThreads) lds_hist[i] = 0;
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * block_points;
    const int64_t hi = min(n, lo + block_points);
    unsigned long long seal = 0;  // the plan's seal (a checksum of pos / batch, kernels.h): this pass reads every word anyway
    for (int64_t i0 = lo + threadIdx.x; i0 < hi; i0 += (int64_t)kSortThreads * kSortUnroll) {
        float c0[kSortUnroll], c1[kSortUnroll], c2[kSortUnroll];
        int64_t bb[kSortUnroll];
#pragma unroll
        for (int q = 0; q < kSortUnroll; ++q) {
            const int64_t i = i0 + (int64_t)q * kSortThreads;
            const bool live = i < hi;
            load_point(pos, g.dim, i, live, c0[q], c1[q], c2[q]);
            bb[q] = live && batch ? batch[i] : 0;
        }
#pragma unroll
        for (int q = 0; q < kSortUnroll; ++q) {
            if (i0 + (int64_t)q * kSortThreads >= hi) continue;
            if (seal_part) seal += seal_point(g.dim, i0 + (int64_t)q * kSortThreads, c0[q], c1[q], c2[q], batch != nullptr, bb[q]);
            // (the host layer reads the first and the last entry only: an index outside [0, B) in between is binned
            // clamped, keeps every access in range, and is reported -- the next entry point fails with "Input mismatch")
            if (bb[q] < 0 || bb[q] >= B) report_fault(status, kFaultBatchIndex);
            const int64_t b = bb[q] < 0 ? 0 : (bb[q] >= B ? B - 1 : bb[q]);
            int bins[4];
            const int k = l1_bins_of(g, c0[q], c1[q], c2[q], b, bins);
            for (int r = 0; r < k; ++r) atomicAdd(&lds_hist[bins[r]], 1);
            if (!g.owned) key1[i0 + (int64_t)q * kSortThreads] = (unsigned short)bins[0];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < npencils; i += kSortThreads) hist[(int64_t)i * nblocks + blockIdx.x] = lds_hist[i];
    if (seal_part) {
        __shared__ unsigned long long part[16];
        seal = seal_block_sum(seal, part);
        if (threadIdx.x == 0) seal_part[blockIdx.x] = seal;
    }
}

template <bool COMMON>
__global__ void __launch_bounds__(kSortThreads)
sort1_scatter_kernel(Geom g_in, const float *__restrict__ pos, const int64_t *__restrict__ batch, int64_t n, int64_t B,
                     int npencils, int nblocks, int block_points, const int *__restrict__ hscan, const unsigned short *__restrict__ key1,
                     float4 *__restrict__ tmp, const unsigned long long *__restrict__ seal_part, unsigned long long *__restrict__ seal,
                     int *__restrict__ final_offsets, int *__restrict__ final_perm, float *__restrict__ final_spos,
                     const int *__restrict__ hist_raw)
{
    // 1-D / 2-D plans have ONE plan bin per first-level bin (no axis-0 bins, no sub-blocks, no column groups): this pass
    // then writes the plan itself -- offsets, permutation, tile-ordered points -- and the second level (two launches that
    // only copied the records) does not run
    const bool final_pass = !COMMON && final_offsets != nullptr;
    // ... and for a small table of counts (hist_raw != nullptr: <= 16 384 of them) every workgroup scans it itself, so that
    // the scan launch between the two passes is gone as well: cursor of bin p in this slice = points of the bins before p
    // (all slices) + points of bin p in the slices before this one
    const bool local_scan = final_pass && hist_raw != nullptr;
    if (final_pass && !local_scan && blockIdx.x == 0) {
        for (int i = threadIdx.x; i <= npencils; i += kSortThreads) final_offsets[i] = hscan[(int64_t)i * nblocks];
    }
    if (seal && blockIdx.x == 0) {
        // the plan's seal: sum of the count pass's per-slice shares; the rest of the seal block (accumulators and arrival
        // counters of verifications) starts at zero
        __shared__ unsigned long long part[16];
        unsigned long long sum = 0;
        for (int i = threadIdx.x; i < nblocks; i += kSortThreads) sum += seal_part[i];
        sum = seal_block_sum(sum, part);
        if (threadIdx.x == 0) seal[0] = sum;
        if (threadIdx.x >= 1 && threadIdx.x < (int)(kSealBytes / 8)) seal[threadIdx.x] = 0ull;
        __syncthreads();
    }
    // COMMON: a 3-D problem on the scatter tiling (every benchmark configuration): with the two flags constant the
    // per-point paths lose their run-time branches on dimension and tiling (8-fold unrolled: 43 KB of code per kernel
    // before), the instruction-bound count passes most of all
    Geom g = g_in;
    if constexpr (COMMON) { g.dim = 3; g.owned = 0; }
    extern __shared__ int lds_cur[];
    if (local_scan) {
        int *const row = lds_cur + npencils;  // [npencils] points per bin over all slices (LDS: 2 * npencils ints)
        for (int i = threadIdx.x; i < npencils; i += kSortThreads) { lds_cur[i] = 0; row[i] = 0; }
        __syncthreads();
        // G threads per bin (a power of two, 512 / bins at most), each sums every G-th slice: G LDS atomics per bin
        int G = 1;
        while (2 * G * npencils <= kSortThreads && G < 64) G *= 2;
        const int sub = threadIdx.x & (G - 1);
        for (int p = threadIdx.x / G; p < npencils; p += kSortThreads / G) {
            int all = 0, before = 0;
            for (int bl = sub; bl < nblocks; bl += G) {
                const int c = hist_raw[p * nblocks + bl];
                all += c;
                before += bl < (int)blockIdx.x ? c : 0;
            }
            if (all) atomicAdd(&row[p], all);
            if (before) atomicAdd(&lds_cur[p], before);
        }
        __syncthreads();
        if (threadIdx.x < 64) {  // exclusive scan of the bin totals, one wave
            int carry = 0;
            for (int base = 0; base < npencils; base += 64) {
                const int idx = base + threadIdx.x;
                const int tot = idx < npencils ? row[idx] : 0;
                int incl = tot;
                for (int off = 1; off < 64; off <<= 1) {
                    const int t = __shfl_up(incl, off);
                    if ((int)threadIdx.x >= off) incl += t;
                }
                if (idx < npencils) {
                    const int excl = carry + incl - tot;
                    lds_cur[idx] += excl;
                    if (blockIdx.x == 0) final_offsets[idx] = excl;
                }
                carry += __shfl(incl, 63);
            }
            if (blockIdx.x == 0 && threadIdx.x == 0) final_offsets[npencils] = carry;
        }
    } else {
        for (int i = threadIdx.x; i < npencils; i += kSortThreads) lds_cur[i] = hscan[(int64_t)i * nblocks + blockIdx.x];
    }
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * block_points;
    const int64_t hi = min(n, lo + block_points);
    for (int64_t i0 = lo + threadIdx.x; i0 < hi; i0 += (int64_t)kSortThreads * kSortUnroll) {
        float c0[kSortUnroll], c1[kSortUnroll], c2[kSortUnroll];
        int64_t bb[kSortUnroll];
        int kk[kSortUnroll];
#pragma unroll
        for (int q = 0; q < kSortUnroll; ++q) {
            const int64_t i = i0 + (int64_t)q * kSortThreads;
            const bool live = i < hi;
            load_point(pos, g.dim, i, live, c0[q], c1[q], c2[q]);
            bb[q] = live && batch && g.owned ? batch[i] : 0;
            kk[q] = live && !g.owned ? key1[i] : 0;
        }
#pragma unroll
        for (int q = 0; q < kSortUnroll; ++q) {
            const int64_t i = i0 + (int64_t)q * kSortThreads;
            if (i >= hi) continue;
            int bins[4];
            int k = 1;
            if (!g.owned) {
                bins[0] = kk[q];  // (left by the count pass)
            } else {
                const int64_t b = bb[q] < 0 ? 0 : (bb[q] >= B ? B - 1 : bb[q]);
                k = l1_bins_of(g, c0[q], c1[q], c2[q], b, bins);
            }
            for (int r = 0; r < k; ++r) {
                const int slot = atomicAdd(&lds_cur[bins[r]], 1);
                if (final_pass) {
                    final_perm[slot] = (int)i;
                    final_spos[(int64_t)slot * g.dim] = c0[q];
                    if (g.dim > 1) final_spos[(int64_t)slot * g.dim + 1] = c1[q];
                    continue;
                }
                tmp[slot] = make_float4(c0[q], c1[q], c2[q], __int_as_float((int)i));
            }
        }
    }
}

// (axis-0 bin, sub-block) key of a level-1 record inside its first-level bin (whose first axis-0 bin is bin_lo)
__device__ __forceinline__ int fine_key(const Geom &g, const float4 rec, const int bin_lo, const int col0)
{
    if (g.dim != 3) return 0;
    int cell[3];
    float fr;
    split_cell(rec.x, g.M, cell[0], fr);
    int key = (g.bin0 == 1 ? cell[0] : div_small(cell[0], g.bin0)) - bin_lo;
    if (g.SB > 1) {
        split_cell(rec.y, g.M, cell[1], fr);
        split_cell(rec.z, g.M, cell[2], fr);
        key = key * g.SB + sub_of_cells(g, cell);
    }
    if (g.CG > 1) {
        // ordering inside the slab: column group of the point's window in the padded tile
        // (col0 = first column of the pencil the record was binned into)
        split_cell(rec.z, g.M, cell[2], fr);
        key = key * g.CG + column_group(cell[2] - col0, g.W);
    }
    return key;
}

// Second level: the records of first-level bin l1 are counting-sorted by fine key (slab, sub-block) by kSort2Parts
// workgroups, each taking an equal share of the bin's records: a count pass (per-part histograms to HBM) and a
// scatter pass whose prologue turns the bin's small table of counts into this part's cursors.
__device__ __forceinline__ void sort2_range(const int *__restrict__ hscan, int l1, int nblocks, int part, int &p0,
                                            int &r0, int &r1)
{
    // (the scan has one item more than there are (bin, block) counts: its last entry is the number of plan entries,
    // n for ordinary plans, larger for the owned tiling)
    p0 = hscan[(int64_t)l1 * nblocks];
    const int p1 = hscan[(int64_t)(l1 + 1) * nblocks];
    const int64_t len = p1 - p0;
    r0 = p0 + (int)(len * part / kSort2Parts);
    r1 = p0 + (int)(len * (part + 1) / kSort2Parts);
}

template <bool COMMON>
__global__ void __launch_bounds__(kSortThreads)
sort2_count_kernel(Geom g_in, int nblocks, const int *__restrict__ hscan, const float4 *__restrict__ tmp,
                   int *__restrict__ hist2 /* [l1][part][key] */, unsigned short *__restrict__ key2 /* fine key of every record */)
{
    // COMMON: a 3-D problem on the scatter tiling (every benchmark configuration): with the two flags constant the
    // per-point paths lose their run-time branches on dimension and tiling (8-fold unrolled: 43 KB of code per kernel
    // before), the instruction-bound count passes most of all
    Geom g = g_in;
    if constexpr (COMMON) { g.dim = 3; g.owned = 0; }
    extern __shared__ int lds2[];
    const int l1 = blockIdx.x, part = blockIdx.y;
    const int pencil = l1 / g.l1seg, sg = l1 - pencil * g.l1seg;
    const int bin_lo = sg * g.l1bins;
    const int col0 = (pencil % g.nta[2]) * g.Ta[2];
    const int nt0 = (min(g.np0, bin_lo + g.l1bins) - bin_lo) * g.SB * g.CG;  // fine keys of this first-level bin
    int p0, r0, r1;
    sort2_range(hscan, l1, nblocks, part, p0, r0, r1);
    for (int i = threadIdx.x; i < nt0; i += kSortThreads) lds2[i] = 0;
    __syncthreads();
    for (int j0 = r0 + threadIdx.x; j0 < r1; j0 += kSortThreads * 8) {
        float4 rec[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int j = j0 + q * kSortThreads;
            rec[q] = j < r1 ? tmp[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (j0 + q * kSortThreads < r1) {
                const int key = fine_key(g, rec[q], bin_lo, col0);
                atomicAdd(&lds2[key], 1);
                key2[j0 + q * kSortThreads] = (unsigned short)key;
            }
    }
    __syncthreads();
    int *out = hist2 + ((int64_t)l1 * kSort2Parts + part) * g.l1bins * g.SB * g.CG;
    for (int i = threadIdx.x; i < nt0; i += kSortThreads) out[i] = lds2[i];
}

__global__ void __launch_bounds__(kSortThreads)
sort2_scatter_kernel(Geom g, int npencils, int nblocks, const int *__restrict__ hscan, const float4 *__restrict__ tmp,
                     const int *__restrict__ hist2, const unsigned short *__restrict__ key2, int *__restrict__ offsets,
                     int *__restrict__ groups, int *__restrict__ perm, float *__restrict__ spos)
{
    extern __shared__ int lds2[];  // [fine keys] this part's cursors, [fine keys] totals over the parts
    const int l1 = blockIdx.x, part = blockIdx.y;
    const int pencil = l1 / g.l1seg, sg = l1 - pencil * g.l1seg;
    const int bin_lo = sg * g.l1bins;
    const int nt0 = (min(g.np0, bin_lo + g.l1bins) - bin_lo) * g.SB * g.CG;
    const int64_t obase = ((int64_t)pencil * g.np0 + bin_lo) * g.SB;   // first entry of these keys in the offsets table
                                                                        // (one entry per CG ordering keys)
    int p0, r0, r1;
    sort2_range(hscan, l1, nblocks, part, p0, r0, r1);
    const int *table = hist2 + (int64_t)l1 * kSort2Parts * g.l1bins * g.SB * g.CG;
    const int kstride = g.l1bins * g.SB * g.CG;
    // cursor of a key = p0 + keys before (totals over the parts) + the same key in the parts before this one.
    // All threads fetch the counts (one round of loads), one wave scans the totals out of LDS.
    int *const tot_lds = lds2 + kstride;
    for (int idx = threadIdx.x; idx < nt0; idx += kSortThreads) {
        int tot = 0, before = 0;
#pragma unroll
        for (int q = 0; q < kSort2Parts; ++q) {
            const int c = table[q * kstride + idx];
            tot += c;
            before += q < part ? c : 0;
        }
        tot_lds[idx] = tot;
        lds2[idx] = before;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        int carry = 0;
        for (int base = 0; base < nt0; base += 64) {
            const int idx = base + threadIdx.x;
            const int tot = idx < nt0 ? tot_lds[idx] : 0;
            int incl = tot;
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(incl, off);
                if ((int)threadIdx.x >= off) incl += t;
            }
            if (idx < nt0) {
                const int excl = carry + incl - tot;
                lds2[idx] += p0 + excl;
                if (part == 0) {
                    const int bin = idx / g.CG, grp = idx - bin * g.CG;
                    if (grp == 0) offsets[obase + bin] = p0 + excl;
                    else groups[(obase + bin) * 2 + grp - 1] = p0 + excl;
                }
            }
            carry += __shfl(incl, 63);
        }
        if (part == 0 && l1 == npencils - 1 && threadIdx.x == 0) offsets[obase + nt0 / g.CG] = p0 + carry;  // = entries
    }
    __syncthreads();
    for (int j0 = r0 + threadIdx.x; j0 < r1; j0 += kSortThreads * 8) {
        float4 recs[8];
        int keys[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int j = j0 + q * kSortThreads;
            recs[q] = j < r1 ? tmp[j] : make_float4(0.f, 0.f, 0.f, 0.f);
            keys[q] = j < r1 ? key2[j] : 0;  // (left by the count pass)
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (j0 + q * kSortThreads >= r1) continue;
            const float4 rec = recs[q];
            const int slot = atomicAdd(&lds2[keys[q]], 1);
            if (g.dim == 3) {
                // the 16-byte record {p0, p1, p2, index of the point} moves as it is: ONE scattered store per point (the
                // pass is bound by scattered store requests, not bytes; there is no separate permutation array in 3-D)
                ((float4 *)spos)[slot] = rec;
            } else {
                perm[slot] = __float_as_int(rec.w);
                spos[(int64_t)slot * g.dim] = rec.x;
                if (g.dim > 1) spos[(int64_t)slot * g.dim + 1] = rec.y;
            }
        }
    }
}

// Work list of the wide tiling (common.h): every range of slabs of every (point set, pencil), dense ranges cut
// into pieces of about `target` points.  One wave per (point set, pencil), lane r = range r of the pencil.
// work[0] = {entries, 1 if any range was cut, 0, 0}; entries {point set * pencils + pencil, first slab, end slab, points}.
__global__ void __launch_bounds__(64)
segment_split_kernel(Geom g, int npl /* point sets x pencils */, int pencils, int runs, int target,
                     const int *__restrict__ offsets, int4 *__restrict__ work, int4 *__restrict__ list, int capacity)
{
    const int pl = blockIdx.x;
    if (pl >= npl) return;
    const int r = threadIdx.x;
    if (r >= runs) return;
    const int *off = offsets + (int64_t)pl * g.np0;  // np0 == M bins per pencil, SB == 1
    const int seg_slabs = (g.M + runs - 1) / runs;
    const int sb = min(r * seg_slabs, g.M), se = min(sb + seg_slabs, g.M);
    const int o_sb = off[sb], pts = off[se] - o_sb;
    if (se <= sb) return;
    int pieces = (int)(((int64_t)pts + target / 2) / target);
    pieces = pieces < 1 ? 1 : (pieces > kSegPieces ? kSegPieces : pieces);
    // (pieces <= 1 + pts / target with target >= 2048: the list's capacity, ranges + entries / 2048, always suffices)
    // the plan runs from its list (work[0].z) when a range is cut or holds >= 1.5 x the mean over ALL ranges of its point
    // set, empty ones included (work_order_kernel)
    {
        const int64_t first = (int64_t)(pl / pencils) * pencils * g.np0;
        const int64_t set_pts = offsets[first + (int64_t)pencils * g.np0] - offsets[first];
        if (pieces > 1 || (pts > 0 && (double)pts * (double)(pencils * runs) >= 1.5 * (double)set_pts))
            atomicOr(&((int *)work)[2], 1);
    }
    int prev = sb;  // end of the previous piece
    for (int p = 1; p <= pieces; ++p) {
        int end = se;
        if (p < pieces) {
            const int t = o_sb + (int)((int64_t)pts * p / pieces);
            int lo = prev, hi = se;  // smallest slab in [prev, se] whose offset is >= t
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (off[mid] >= t) hi = mid; else lo = mid + 1;
            }
            end = lo;
        }
        const int piece_pts = off[end] - off[prev];
        // (pieces without points are listed too: the owned tiling writes every plane, and the gather kernels give an
        // item the chunks of slabs that START in it -- such a chunk may reach into the next item's slabs)
        if (end > prev) {
            const int slot = atomicAdd(&((int *)work)[0], 1);
            if (slot < capacity) {
                list[slot] = make_int4(pl, prev, end, piece_pts);
                atomicAdd(&((int2 *)(work + 1))[pl / pencils].x, 1);
            }
        }
        prev = end;
    }
}

// Puts the work list of an unbalanced plan in launch order (a balanced one -- segment_split_kernel saw no range that was
// cut or holds >= 1.5 x its set's mean: every uniform input -- returns at once: its kernels run one workgroup per range in
// grid order, straight-line code, neighbouring ranges side by side; measured 1 % faster at config C3 than any sorted
// order).  One workgroup per point set b: its entries are copied to sorted[start_b ...), biggest first (a counting sort
// into 16 size classes, sixteenths of the set's largest entry), and set_hdr[b] = {entries, start_b}.  ONE persistent
// launch per plane then walks the set's part of `sorted`, so that the tail of the launch is made of small items:
// clustered inputs leave most CUs idle behind their few heavy ranges in grid order (43 % CU utilisation at C3-clustered),
// and a separate launch for the cut-off pieces (rounds 2-3) started only when the last first piece was done.
constexpr int kOrderClasses = 16;
__global__ void __launch_bounds__(1024)
work_order_kernel(int4 *__restrict__ work, const int4 *__restrict__ list, int4 *__restrict__ sorted, int capacity,
                  int pencils, int forced)
{
    __shared__ int cnt[kOrderClasses][1024];  // [class][thread]: entries of the class in the thread's chunk -> their first slot
    __shared__ int total[kOrderClasses];
    __shared__ int maxpts, start;
    const int b = blockIdx.x;
    const int4 hdr = work[0];
    if (!hdr.z && !forced) return;  // balanced: nobody reads the list
    if (forced && b == 0 && threadIdx.x == 0) ((int *)work)[2] = 1;
    const int nitems = min(hdr.x, capacity);
    int2 *const set_hdr = (int2 *)(work + 1);
    if (threadIdx.x == 0) { maxpts = 1; start = 0; }
    __syncthreads();
    {   // entries of the sets in front of this one
        int before = 0;
        for (int q = threadIdx.x; q < b; q += 1024) before += set_hdr[q].x;
        if (before) atomicAdd(&start, before);
    }
    // thread t owns the contiguous chunk [t per, (t + 1) per) of the list; only this set's entries count
    const int per = (nitems + 1023) / 1024;
    const int lo = min(nitems, (int)threadIdx.x * per), hi = min(nitems, lo + per);
    const int pl0 = b * pencils, pl1 = pl0 + pencils;
    int mx = 0;
    for (int it = lo; it < hi; ++it) {
        const int4 e = list[it];
        if (e.x >= pl0 && e.x < pl1) mx = max(mx, e.w);
    }
    atomicMax(&maxpts, mx);
    __syncthreads();
    if (threadIdx.x == 0) set_hdr[b].y = start;
    const float scale = (float)kOrderClasses / (float)maxpts;
    auto size_class = [&](const int pts) { return kOrderClasses - 1 - min(kOrderClasses - 1, (int)((float)pts * scale)); };  // 0 = biggest
    int mine[kOrderClasses];
#pragma unroll
    for (int c = 0; c < kOrderClasses; ++c) mine[c] = 0;
    for (int it = lo; it < hi; ++it) {
        const int4 e = list[it];
        if (e.x < pl0 || e.x >= pl1) continue;
        const int c = size_class(e.w);
#pragma unroll
        for (int q = 0; q < kOrderClasses; ++q) mine[q] += q == c ? 1 : 0;
    }
#pragma unroll
    for (int c = 0; c < kOrderClasses; ++c) cnt[c][threadIdx.x] = mine[c];
    __syncthreads();
    // wave c scans class c over the 1024 threads (exclusive); total[c] = the class total
    {
        const int c = threadIdx.x >> 6, lane = threadIdx.x & 63;
        int carry = 0;
        for (int base = 0; base < 1024; base += 64) {
            const int v = cnt[c][base + lane];
            int incl = v;
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(incl, off);
                if (lane >= off) incl += t;
            }
            cnt[c][base + lane] = carry + incl - v;
            carry += __shfl(incl, 63);
        }
        if (lane == 0) total[c] = carry;
    }
    __syncthreads();
    int slot[kOrderClasses];
    {
        int before = start;
#pragma unroll
        for (int c = 0; c < kOrderClasses; ++c) {
            slot[c] = before + cnt[c][threadIdx.x];
            before += total[c];
        }
    }
    for (int it = lo; it < hi; ++it) {
        const int4 e = list[it];
        if (e.x < pl0 || e.x >= pl1) continue;
        const int c = size_class(e.w);
        int s_ = 0;
#pragma unroll
        for (int q = 0; q < kOrderClasses; ++q) {
            if (q == c) { s_ = slot[q]; slot[q] += 1; }
        }
        sorted[s_] = e;
    }
}

// xs[c, slot] = xr[perm[slot], c]  (tile-ordered, column-major copy of the real coefficient columns; column stride
// `stride` = plan capacity, `*total` = entries the plan holds)
__global__ void __launch_bounds__(256) gather_rows_kernel(const int *__restrict__ perm /* stride pstep ints */, const int pstep,
                                                         const float *__restrict__ xr,
                                                         float *__restrict__ xs, const int *__restrict__ total_entries,
                                                         int64_t stride, int64_t cols)
{
    const int64_t n = *total_entries;
    const int64_t total = n * cols;
    // eight elements per thread and step: first the eight permutation entries, then the eight (random) coefficient
    // reads, so that a wave has 8 x 64 independent loads in flight
    constexpr int UN = 8;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e0 < total; e0 += step * UN) {
        int64_t src[UN];
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            const int64_t e = e0 + q * step;
            if (e < total) {
                const int64_t slot = e / cols, c = e - slot * cols;
                src[q] = (int64_t)perm[slot * pstep] * cols + c;
            } else {
                src[q] = -1;
            }
        }
        float v[UN];
#pragma unroll
        for (int q = 0; q < UN; ++q) v[q] = src[q] >= 0 ? xr[src[q]] : 0.0f;
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            const int64_t e = e0 + q * step;
            if (e < total) {
                const int64_t slot = e / cols, c = e - slot * cols;
                xs[c * stride + slot] = v[q];
            }
        }
    }
}

static inline int grid_for(int64_t work, int block)
{
    int64_t g = (work + block - 1) / block;
    if (g < 1) g = 1;
    if (g > 256 * 32) g = 256 * 32;
    return (int)g;
}

static int launch_segment_split(const Geom &g, const PlanLayout &L, int64_t n, int64_t B, const int *offsets, int4 *work,
                                 hipStream_t stream)
{
    const int64_t pencils = (int64_t)g.nta[1] * g.nta[2], npl = pencils * B;
    const int ncu = device_cu_count();
    const int runs = seg_base_runs(n, B, pencils, g.M, ncu);
    const int target = (int)std::min<int64_t>(seg_target_points(n, B, ncu), int64_t(1) << 30);
    NFFT_HIP_CHECK(hipMemsetAsync(work, 0, (size_t)L.work_head * 16, stream));
    if (npl <= 0) return 0;
    int4 *const list = work + L.work_head;
    hipLaunchKernelGGL(segment_split_kernel, dim3((unsigned)npl), dim3(64), 0, stream, g, (int)npl, (int)pencils, runs, target,
                       offsets, work, list, (int)L.work_cap);
    hipLaunchKernelGGL(work_order_kernel, dim3((unsigned)B), dim3(1024), 0, stream, work, list, list + L.work_cap,
                       (int)L.work_cap, (int)pencils, work_list_forced() ? 1 : 0);
    return 0;
}

// Exclusive prefix sum of up to kSmallScanItems ints in ONE workgroup (small problems: hipCUB's device scan is two launches,
// ~5 us of host time each, for a few thousand counters).  out[i] = in[0] + ... + in[i - 1].
constexpr int kSmallScanItems = 1 << 16;
// IPT consecutive items per thread and round: 4 for a few thousand items (one round), 16 beyond (a round is three barriers and a
// serial sum over the waves, ~5 us: 23 500 items took 31 us in rounds of 4 096 -- more than hipCUB's two launches)
template <int IPT>
__global__ void __launch_bounds__(1024) small_scan_kernel(const int *__restrict__ in, int *__restrict__ out, int items)
{
    __shared__ int wave_sums[16];
    __shared__ int carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < items; base += 1024 * IPT) {
        const int i0 = base + IPT * (int)threadIdx.x;
        int v[IPT];
        int mine = 0;
        if (i0 + IPT <= items && ((uintptr_t)in & 15) == 0) {
            // (16-byte loads: as single words a thread's 16 items were 16 load instructions of 64-byte stride across the lanes --
            // 47 us for 23 500 items)
#pragma unroll
            for (int k = 0; k < IPT; k += 4) {
                const int4 q = *(const int4 *)(in + i0 + k);
                v[k] = q.x; v[k + 1] = q.y; v[k + 2] = q.z; v[k + 3] = q.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < IPT; ++k) v[k] = i0 + k < items ? in[i0 + k] : 0;
        }
#pragma unroll
        for (int k = 0; k < IPT; ++k) mine += v[k];
        int incl = mine;
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wave_sums[wave] = incl;
        __syncthreads();
        int before = carry_s;
        for (int w = 0; w < wave; ++w) before += wave_sums[w];
        int run = before + incl - mine;
        if (i0 + IPT <= items && ((uintptr_t)out & 15) == 0) {
#pragma unroll
            for (int k = 0; k < IPT; k += 4) {
                int4 q;
                q.x = run; run += v[k];
                q.y = run; run += v[k + 1];
                q.z = run; run += v[k + 2];
                q.w = run; run += v[k + 3];
                *(int4 *)(out + i0 + k) = q;
            }
        } else {
#pragma unroll
            for (int k = 0; k < IPT; ++k) {
                if (i0 + k < items) out[i0 + k] = run;
                run += v[k];
            }
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = run;  // (the last thread's running sum: everything up to this round's end)
        __syncthreads();
    }
}

static int exclusive_scan(const int *in, int *out, int64_t items, void *scratch, int64_t scratch_bytes, hipStream_t stream)
{
    if (items <= kSmallScanItems) {
        if (items <= 4096) hipLaunchKernelGGL(small_scan_kernel<4>, dim3(1), dim3(1024), 0, stream, in, out, (int)items);
        else hipLaunchKernelGGL(small_scan_kernel<16>, dim3(1), dim3(1024), 0, stream, in, out, (int)items);
        NFFT_HIP_CHECK(hipGetLastError());
        return 0;
    }
    size_t scan_bytes = 0;
    NFFT_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, in, out, (int)items, stream));
    if ((int64_t)scan_bytes > scratch_bytes) { set_error("scan scratch too small"); return 2; }
    NFFT_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(scratch, scan_bytes, in, out, (int)items, stream));
    return 0;
}

int launch_plan_points(const Geom &g, const PlanLayout &L, const float *pos, const int64_t *batch, int64_t n, int64_t B,
                       void *plan, hipStream_t stream)
{
    char *base = (char *)plan;
    int *offsets = (int *)(base + L.off_offsets);
    int *cursor = (int *)(base + L.off_cursor);
    int *perm = (int *)(base + L.off_perm);
    float *spos = (float *)(base + L.off_spos);
    // every plan is sealed by the pass that counts its points (kernels.h: kSealBytes)
    unsigned long long *const seal = (unsigned long long *)(base + L.off_seal);
    if (L.two_level) {
        int *hist = (int *)(base + L.off_hist);
        int *hscan = (int *)(base + L.off_hscan);
        float4 *tmp = (float4 *)(base + L.off_tmp);
        const int npencils = (int)L.npencils, nblocks = (int)L.nblocks;
        const int64_t items = L.npencils * L.nblocks + 1;
        const size_t lds1 = (size_t)npencils * 4;
        unsigned short *key1 = (unsigned short *)(base + L.off_key1), *key2 = (unsigned short *)(base + L.off_key2);
        const bool common = g.dim == 3 && !g.owned;
        hipLaunchKernelGGL(common ? sort1_count_kernel<true> : sort1_count_kernel<false>, dim3(nblocks), dim3(kSortThreads), lds1, stream, g, pos, batch, n, B,
                           npencils, nblocks, (int)L.block_points, hist, key1, seal ? (unsigned long long *)(base + L.off_sealpart) : nullptr,
                           device_status_block());
        // (1-D / 2-D: the first level is the whole sort; a small table of counts is scanned inside the scatter pass)
        const bool one_level = g.dim < 3 && g.l1seg == 1 && (int64_t)g.l1bins * g.SB * g.CG == 1 && L.ntiles == L.npencils;
        const bool local_scan = one_level && items <= 16384;
        if (!local_scan)
            if (int rc = exclusive_scan(hist, hscan, items, base + L.off_scan, L.scan_bytes, stream)) return rc;
        hipLaunchKernelGGL(common ? sort1_scatter_kernel<true> : sort1_scatter_kernel<false>, dim3(nblocks), dim3(kSortThreads),
                           local_scan ? 2 * lds1 : lds1, stream, g, pos, batch, n, B,
                           npencils, nblocks, (int)L.block_points, hscan, key1, tmp, (const unsigned long long *)(base + L.off_sealpart), seal,
                           one_level ? offsets : nullptr, perm, spos, local_scan ? hist : nullptr);
        if (one_level) {
            NFFT_HIP_CHECK(hipGetLastError());
            return 0;
        }
        int *hist2 = (int *)(base + L.off_hist2);
        const size_t lds2 = (size_t)g.l1bins * g.SB * g.CG * 4;
        hipLaunchKernelGGL(common ? sort2_count_kernel<true> : sort2_count_kernel<false>, dim3(npencils, kSort2Parts), dim3(kSortThreads), lds2, stream, g, nblocks,
                           hscan, tmp, hist2, key2);
        hipLaunchKernelGGL(sort2_scatter_kernel, dim3(npencils, kSort2Parts), dim3(kSortThreads), 2 * lds2, stream, g,
                           npencils, nblocks, hscan, tmp, hist2, key2, offsets, (int *)(base + L.off_groups), perm, spos);
        if (g.wide && launch_segment_split(g, L, n, B, offsets, (int4 *)(base + L.off_work), stream)) return 2;
        NFFT_HIP_CHECK(hipGetLastError());
        return 0;
    }
    // counts are accumulated in `cursor`, scanned into `offsets`, then `cursor` restarts at zero (the plan's seal block lies
    // right behind `cursor`: the same memset zeroes it, the count pass adds the checksum of the points to it)
    static_assert(kSealBytes == 256, "the seal block fills one alignment unit behind the cursors");
    NFFT_HIP_CHECK(hipMemsetAsync(cursor, 0, (size_t)(L.off_seal + kSealBytes - L.off_cursor), stream));
    if (n > 0) {
        hipLaunchKernelGGL(bin_count_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, g, pos, batch, n, B, cursor,
                           (unsigned long long *)(base + L.off_seal), device_status_block());
    }
    if (int rc = exclusive_scan(cursor, offsets, L.ntiles + 1, base + L.off_scan, L.scan_bytes, stream)) return rc;
    NFFT_HIP_CHECK(hipMemsetAsync(cursor, 0, (L.ntiles + 1) * 4, stream));
    if (n > 0) {
        hipLaunchKernelGGL(bin_fill_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, g, pos, batch, n, B, offsets,
                           cursor, perm, spos);
    }
    if (g.wide && launch_segment_split(g, L, n, B, offsets, (int4 *)(base + L.off_work), stream)) return 2;
    NFFT_HIP_CHECK(hipGetLastError());
    return 0;
}

__global__ void __launch_bounds__(kSealThreads)
points_seal_kernel(const float *__restrict__ pos, const int64_t *__restrict__ batch, int64_t n, int dim,
                   unsigned long long *__restrict__ acc, const unsigned long long *__restrict__ expect,
                   unsigned *__restrict__ arrivals, int *__restrict__ status)
{
    unsigned long long sum = 0;
    const int64_t stride = (int64_t)gridDim.x * kSealThreads;
    int64_t i = (int64_t)blockIdx.x * kSealThreads + threadIdx.x;
    if (dim == 3 && ((uintptr_t)pos & 15) == 0 && (!batch || ((uintptr_t)batch & 15) == 0)) {
        // 3-D (every large problem): four points = three aligned 16-byte loads (and two for their batch indices)
        const int64_t quads = n / 4;
        const uint4 *p4 = (const uint4 *)pos;
        const uint4 *b4 = (const uint4 *)batch;
        auto f = [](unsigned u) { return __uint_as_float(u); };
        auto b64 = [](unsigned lo, unsigned hi) { return (int64_t)((unsigned long long)lo | ((unsigned long long)hi << 32)); };
        auto four = [&](const int64_t q, const uint4 v0, const uint4 v1, const uint4 v2, const uint4 w0, const uint4 w1) {
            sum += seal_point(3, 4 * q + 0, f(v0.x), f(v0.y), f(v0.z), batch != nullptr, b64(w0.x, w0.y));
            sum += seal_point(3, 4 * q + 1, f(v0.w), f(v1.x), f(v1.y), batch != nullptr, b64(w0.z, w0.w));
            sum += seal_point(3, 4 * q + 2, f(v1.z), f(v1.w), f(v2.x), batch != nullptr, b64(w1.x, w1.y));
            sum += seal_point(3, 4 * q + 3, f(v2.y), f(v2.z), f(v2.w), batch != nullptr, b64(w1.z, w1.w));
        };
        const uint4 z4 = make_uint4(0, 0, 0, 0);
        int64_t q = i;
        for (; q + 2 * stride < quads; q += 3 * stride) {  // three groups of four points (nine 16-byte loads) in flight per thread
            uint4 v[3][3], w[3][2];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int64_t qq = q + u * stride;
                v[u][0] = p4[3 * qq]; v[u][1] = p4[3 * qq + 1]; v[u][2] = p4[3 * qq + 2];
                w[u][0] = batch ? b4[2 * qq] : z4;
                w[u][1] = batch ? b4[2 * qq + 1] : z4;
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) four(q + u * stride, v[u][0], v[u][1], v[u][2], w[u][0], w[u][1]);
        }
        for (; q < quads; q += stride)
            four(q, p4[3 * q], p4[3 * q + 1], p4[3 * q + 2], batch ? b4[2 * q] : z4, batch ? b4[2 * q + 1] : z4);
        i += 4 * quads;  // (the last n % 4 points: the loops below, which then run for the first threads only)
        if (i >= n) i = n;
    }
    for (; i + 3 * stride < n; i += 4 * stride) {  // four points in flight per thread
        float c[4][3];
        int64_t bb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            load_point(pos, dim, i + q * stride, true, c[q][0], c[q][1], c[q][2]);
            bb[q] = batch ? batch[i + q * stride] : 0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) sum += seal_point(dim, i + q * stride, c[q][0], c[q][1], c[q][2], batch != nullptr, bb[q]);
    }
    for (; i < n; i += stride) {
        float c0, c1, c2;
        load_point(pos, dim, i, true, c0, c1, c2);
        sum += seal_point(dim, i, c0, c1, c2, batch != nullptr, batch ? batch[i] : 0);
    }
    __shared__ unsigned long long part[16];
    sum = seal_block_sum(sum, part);
    if (threadIdx.x == 0) {
        if (sum) atomicAdd(acc, sum);
        if (expect) {
            // verification: the last workgroup to arrive compares the finished sum with the plan's seal
            __threadfence();
            if (atomicAdd(arrivals, 1u) == gridDim.x - 1) {
                __threadfence();
                if (__hip_atomic_load(acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != *expect)
                    report_fault(status, kFaultStalePlan);
                // leave the slot as it was found -- zero -- for the next verification that uses it: no memsets per call
                __hip_atomic_store(acc, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(arrivals, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

static int seal_launch(const float *pos, const int64_t *batch, int64_t n, int dim, unsigned long long *acc,
                       const unsigned long long *expect, unsigned *arrivals, hipStream_t stream)
{
    // >= 4 points per thread (one round of loads in flight: a small input is a latency chain, 9 us -> 5 for 10^5 points);
    // big inputs are capped at 512 workgroups, i.e. many points per thread
    int64_t blocks = (n + kSealThreads * 4 - 1) / (kSealThreads * 4);
    blocks = blocks < 1 ? 1 : (blocks > 512 ? 512 : blocks);
    hipLaunchKernelGGL(points_seal_kernel, dim3((unsigned)blocks), dim3(kSealThreads), 0, stream, pos, batch, n, dim, acc, expect,
                       arrivals, device_status_block());
    NFFT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_points_verify(const float *pos, const int64_t *batch, int64_t n, int dim, void *seal, int slot, hipStream_t stream)
{
    // (accumulator and arrival counter of the slot are zero: the plan's count pass left them so, and every verification
    // zeroes them again when it is through)
    unsigned long long *words = (unsigned long long *)seal;
    unsigned *arrivals = (unsigned *)(words + 9) + (slot & 7);
    return seal_launch(pos, batch, n, dim, words + 1 + (slot & 7), words, arrivals, stream);
}

int launch_gather_rows(const Geom &g, const PlanLayout &L, const void *plan, int64_t n, const float *xr, int64_t cols,
                       float *xs, hipStream_t stream)
{
    // 3-D: the original index of a point is the fourth word of its 16-byte plan record
    const int *perm = g.dim == 3 ? (const int *)((const char *)plan + L.off_spos) + 3 : (const int *)((const char *)plan + L.off_perm);
    const int *total = (const int *)((const char *)plan + L.off_offsets) + L.ntiles;  // offsets[ntiles] = entries
    if (n * cols > 0)
        hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(L.cap * cols, 256)), dim3(256), 0, stream, perm,
                           g.dim == 3 ? 4 : 1, xr, xs, total, L.cap, cols);
    NFFT_HIP_CHECK(hipGetLastError());
    return 0;
}

} // namespace nfft
