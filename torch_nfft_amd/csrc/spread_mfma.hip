// Spreading on the matrix cores: the default for 3-D grids of 64^3 and up with m <= 7 (wide pencil tiling).
//
// Same result as the reference's adjoint_window_convolution kernels (csrc/cuda/spatial_window_operations.cu:
// 103-211).  For one grid plane z of a pencil the window sum is an outer-product accumulation
//     G_z[u1, u2] = sum_i (x_i psi0_i[z] psi1_i[u1]) * psi2_i[u2]  =  (A_z B)[u1, u2],
// A_z = [32 rows x K points], B = [K points x 64 columns] -- a GEMM whose operands hold only 32 + 64 window
// values per point instead of the (2m+2)^2 products, and whose 32 x 64 result is exactly the padded pencil
// (T1 + 2m+1 = 32 rows, T2 + 2m+1 = 64 columns).  It runs on v_mfma_f32_32x32x16_f16 with two-way split
// operands (2^11 x = hi + lo in f16; hi*hi + hi*lo + lo*hi keeps ~22 bits), fp32 accumulators in registers:
//   * the plan is sorted by single planes ("slabs"); a K-block is 16 points of ONE slab, so the axis-0 weight of
//     a plane is a wave-uniform row of a small table.  Inside a slab the plan orders the points by the 32-column half of
//     the tile their window lies in (column groups, common.h): the operand build records which halves a K-block
//     touches, and the owners skip the 3 MFMAs of an untouched half (a third of all MFMAs at config C3);
//   * the coefficients are read in place: every 16-byte plan record carries the index of its point in the caller's
//     arrays, the staging pipeline fetches x[index] by LDS-DMA two steps after the record (one or two real columns; more
//     columns come from the plan-ordered copy of gather_rows).  The f16 operand scale is the plane's largest |x|, taken
//     by plane_absmax_kernel in a pass of its own (no per-item prologue);
//   * one workgroup (16 waves) sweeps a segment of a pencil.  Plane-owner wave w holds the plane z = w (mod NOWN) of
//     the sliding window in its accumulators: no LDS accumulator, no barrier to accumulate; it flushes its 32 x 64
//     tile with global atomics (each instruction = two 128-byte row segments) as soon as the sweep has passed it
//     and moves on to plane z + NOWN.  NOWN = 12 for 2m+2 <= 12 -- the other four waves, one per SIMD, only stage
//     points and build operands -- and 16 for wider windows;
//   * per batch of 8 K-blocks the operands are built once in LDS, one batch ahead of the MFMAs, in wave-sized tasks (one
//     per K-block) handed out through an LDS counter: f16-split B fragments in MFMA register order, and the f16 splits of
//     the psi1 table [row][point] and of the axis-0 table [tap][point] (x' psi0).  Every owner whose plane lies in a
//     K-block's window forms the split of the PRODUCT psi1 * (x' psi0) -- its A fragment -- from the two splits with 16
//     packed f16 instructions (mfma_split.h: split_product_f16x4; until round 4: 24 mixed fp32 / f16 instructions on fp32
//     tables) and issues 3 MFMAs per touched column tile.  One raw s_barrier per batch.
// Owner-computes variant (template flag OWNED, plans of the owned tiling: sparse inputs, common.h choose_owned): the
// 32 x 64 accumulator tile is the OWNED region of the grid -- a point has a plan entry in every tile its window
// touches (1.46x entries at m = 4) and taps that fall outside the tile are masked -- and a work item owns the planes
// [sb, se) of its tile: it sweeps the slabs [sb - m - 1, se + m) (cyclically) and writes every owned plane exactly
// once with plain 128-byte row stores.  No zero-fill, no atomics, bitwise reproducible.  At low density the atomic
// flush of the padded tiles is what the scatter variant spends its time on (1.3 TB/s chip-wide for float atomics
// against ~6 TB/s for stores); at the density of config C3 it is not (profiles/r02_flush_variants.txt).
// Why: ds_add_f32 is unusable on gfx950 and the f64 LDS atomic bounds spread.hip at ~3.7 ms for 1e10 taps
// (DESIGN.md section 4); here the taps are ~0.86 PFLOP of issued matrix work per launch and the kernel runs 1.34 ms at C3
// (round 4; 93 VGPRs, no scratch).  What bounds it is not one resource: timing-only builds without the MFMAs, without the
// packed arithmetic, without the operand loads, without the table builds or without the flush atomics each gain 3-7 %, and
// a flag-driven variant without the per-batch barrier is slower (profiles/r04_experiments.md).
#include <algorithm>
#include <climits>
#include <cstdlib>

#include "common.h"
#include "kernels.h"
#include "mfma_split.h"

namespace nfft {

#ifdef NFFT_HIP_TRACE
// Developer instrumentation (variant builds only, scripts/exp_build.sh -DNFFT_HIP_TRACE): eight 64-bit words per
// workgroup of the primary launch -- 100 MHz real-time stamps at entry / after the max-|x| pass / at the first batch /
// at the end, the hardware id of the CU, the item's K-block and point counts.
__device__ unsigned long long *g_spread_trace = nullptr;
#define NFFT_TRACE(slot, value)                                                                                   \
    do {                                                                                                          \
        if (!OVERFLOW && threadIdx.x == 0 && g_spread_trace)                                                      \
            g_spread_trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (slot)] = (value);                 \
    } while (0)
// Step-level stamps (shader clock, s_memtime) of the first kStepTraceWgs workgroups: per wave and pipeline step the time
// at the loop top, after the accumulation, after the operand build and after the barrier (scripts/spread_steps.py).
constexpr int kStepTraceWgs = 16, kStepTraceSteps = 64;
__device__ unsigned long long *g_step_trace = nullptr;
#define NFFT_STEP_V(i, point, extra)                                                                              \
    do {                                                                                                          \
        if (!OVERFLOW && g_step_trace && blockIdx.y == 0 && blockIdx.x < kStepTraceWgs && (i) >= 0 &&            \
            (i) < kStepTraceSteps && lane == 0)                                                                   \
            g_step_trace[((((size_t)blockIdx.x * 16 + wave) * kStepTraceSteps + (i)) * 4) + (point)] =            \
                __builtin_amdgcn_s_memtime() | (extra);                                                           \
    } while (0)
#define NFFT_STEP(i, point) NFFT_STEP_V(i, point, 0ull)
#else
#define NFFT_TRACE(slot, value) do { } while (0)
#define NFFT_STEP(i, point) do { } while (0)
#define NFFT_STEP_V(i, point, extra) do { } while (0)
#endif

namespace {

constexpr int kKB = 16;        // points per K-block (the MFMA K dimension; the table builds map lane = 4 * point + tap group: 4 kKB = 64)
constexpr int kNKB = 8;        // K-blocks per batch
constexpr int kSlots = kKB * kNKB;
constexpr int kMfmaThreads = 1024;
constexpr int kMaxSegSlabs = 128;   // planes of one work item (a range of M / runs slabs or a piece of it)
constexpr int kMaxSweep = kMaxSegSlabs + 2 * kMaxCutoff;  // slabs it sweeps: the owned variant adds 2m+1 halo slabs
constexpr float kPsiScale = 16.0f;  // psi1 <= 1 enters its f16 split times 16: psi1 * (x' psi0 * 2^11) <= 2^15 fits f16, and the
                                    // lo parts of the three central taps on either side are normal f16 numbers

// Operands of one batch of K-blocks, double-buffered: while the waves run the MFMAs of batch i they already build
// the operands of batch i + 1.
template <int W>
struct __align__(16) MfmaOps {
    f16x8 bfrag[kNKB][2][2][64];          // [K-block][column tile][hi/lo][lane]
    _Float16 p1[kNKB][2][2][32][8];       // [K-block][hi/lo][point >> 3][row][point & 7]   16 psi1, f16 split: lane (row, h) of an
                                          // owner wave reads its 8 points with one ds_read_b128 at 16 * lane (no bank conflict)
    _Float16 a0[kNKB][W][2][kKB];         // [K-block][axis-0 tap][hi/lo][point]   2^11 x' psi0, f16 split
    int slab[kNKB];
    int halves[kNKB];                     // bit t: some tap of the K-block lies in column tile t (else its MFMAs are skipped)
};

// Points of one batch (cell fractions, in-pencil cells, scaled value), double-buffered as well.
struct __align__(16) MfmaStage {
    float f0[kSlots], f1[kSlots], f2[kSlots], x[kSlots];
    float x1[kSlots];                     // (paired variant: the second column's scaled value)
    int c1[kSlots], c2[kSlots];
    int slab[kNKB];
};

// Staging rings: the 16-byte plan records of a batch are requested six steps ahead of its MFMAs, its coefficients four
// steps ahead (their address comes out of the landed record when the kernel gathers x through the plan), the batch is
// converted two steps ahead.
constexpr int kRecRing = 8;
constexpr int kXRing = 4;
template <int W>
struct __align__(16) MfmaLds {
    MfmaOps<W> ops[2];
    MfmaStage stag[2];
    f32x4 raw[kRecRing][kSlots];          // landing zones of the LDS-DMA (batch b uses b & 7): plan records {p0, p1, p2, index}
    float rawx[kXRing][kSlots];           // ... and of the coefficients (batch b uses b & 3)
    float rawx1[kXRing][kSlots];          // (paired variant: the second column's)
    int raw_idx[kRecRing][kSlots];        // plan entry of the slot (only used with the plan-ordered coefficient copy)
    signed char raw_have[kRecRing][kSlots];
    int raw_slab[kRecRing][kNKB];
    int task_counter[2];
    int ticket;                           // work-list entry of the workgroup (persistent launch: next_work_item)
    float inv_xscale;                     // (read by the staging threads once per step: a register would be spilled to scratch)
    float inv_xscale1;                    // (paired variant: the second column's plane)
    int2 sched[kMaxSweep + 8];            // per slab: {K-blocks before it, point offset}; padded with the totals
    int sched_end[kMaxSweep + 8];         // per slab: end of its point range
};
static_assert(sizeof(MfmaLds<16>) <= 160 * 1024, "LDS budget");

// Paired variant (template flag PAIR, owned plans of problems with two or more columns: 32 x 32 tiles): accumulator 0 holds
// the tile of real column 2q of the point set, accumulator 1 the SAME cells of column 2q + 1 -- the points are staged once,
// the B fragments (psi2) and the psi1 table are built once, only the axis-0 table (x' psi0) and the A fragment exist per
// column.  At C4's density (10^5 points per 256^3 grid) the kernel's time is its K-block count, not its stores (measured:
// profiles/r04_experiments.md), and a 32 x 32 tile's slab holds ~10 points -- one K-block -- where a 32 x 64 tile's held ~18,
// i.e. two half-empty ones per column.  The y grid of the launch enumerates pair slots: slot s = (point set, pair) covers
// planes set * Cr + 2 pair (+ 1); a chunk of planes that starts or ends inside a pair runs that slot with one column.
template <int W, bool OVERFLOW, bool OWNED, bool PAIR>
__global__ void __launch_bounds__(kMfmaThreads) __attribute__((amdgpu_waves_per_eu(4, 4)))
spread_mfma_kernel(const Geom g, const int *__restrict__ tile_offsets, const float *__restrict__ spos,
                   const float *__restrict__ xr, const float *__restrict__ xs, const int64_t xs_stride,
                   const unsigned *__restrict__ xmax, const int Cr,
                   const int plane0, const int nplanes, float *__restrict__ grid, const int seg_slabs, const int nsegm,
                   const int4 *__restrict__ work, const int4 *__restrict__ sorted, const WorkTickets tickets, int *__restrict__ status)
{
    static_assert(!PAIR || OWNED, "the paired variant is an owner-computes kernel");
    constexpr int m = W / 2 - 1;
    constexpr int TW = PAIR ? 32 : 64;  // columns of the accumulator tile
    extern __shared__ __align__(16) unsigned char smem_raw[];
    MfmaLds<W> &L = *reinterpret_cast<MfmaLds<W> *>(smem_raw);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform by construction: keep it in an SGPR
    const int r32 = lane & 31, h = lane >> 5;
    NFFT_TRACE(0, __builtin_amdgcn_s_memrealtime());
    NFFT_TRACE(4, (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32));

    // plane(s) of this workgroup: one, or the pair slot's two (the second one may be missing: odd column count, or a chunk
    // of planes that ends inside the pair; a chunk that STARTS inside a pair runs the pair's second column alone)
    int plane_local = blockIdx.y, plane_local1 = -1;
    if constexpr (PAIR) {
        const int P = (Cr + 1) >> 1;  // pair slots per point set
        const int b0 = plane0 / Cr, c0 = plane0 - b0 * Cr;
        const int slot = b0 * P + (c0 >> 1) + (int)blockIdx.y;
        const int sb_ = slot / P, q = slot - sb_ * P;
        const int pa = sb_ * Cr + 2 * q - plane0;  // local plane of the pair's first column
        const bool va = pa >= 0 && pa < nplanes, vb = 2 * q + 1 < Cr && pa + 1 >= 0 && pa + 1 < nplanes;
        if (!va && !vb) return;  // (cannot happen for the slots the launch enumerates)
        plane_local = va ? pa : pa + 1;
        plane_local1 = va && vb ? pa + 1 : -1;
    }
    const bool two = PAIR && plane_local1 >= 0;  // workgroup-uniform
    const int plane = plane0 + plane_local;
    const int b = plane / Cr;
    const int cr = plane - b * Cr;
    const int pencils = g.nta[1] * g.nta[2];

    // ---- work items (common.h).  Balanced plan: workgroup (pencil, range) sweeps its range of seg_slabs slabs,
    // straight-line code.  Otherwise a persistent grid walks the plan's work list, biggest items first.  Both launches
    // are enqueued; the one that is not the plan's leaves here.  (two instantiations: the item loop costs registers)
    const int listed = work[0].z;
    if (OVERFLOW ? !listed : listed) return;
    // (a plane walks its own point set's part of the sorted list: set_hdr[b] = {entries, first entry})
    const int2 set_hdr = OVERFLOW ? ((const int2 *)(work + 1))[b] : make_int2(1, 0);
    const int n_items = set_hdr.x;
    const int4 *const entries = sorted + set_hdr.y;
    for (int item = OVERFLOW ? next_work_item(tickets, &L.ticket, -1, (int)blockIdx.y) : 0; item < n_items;
         item = OVERFLOW ? next_work_item(tickets, &L.ticket, item, (int)blockIdx.y) : 1) {
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
    const int nplane = se - sb;  // planes (scatter variant: slabs) of this item
    const int bin0 = b * g.tiles_per_batch + pencil * g.np0;  // np0 == M: one plan bin per slab
    if (nplane <= 0) continue;
    if (!OWNED && tile_offsets[bin0 + sb] == tile_offsets[bin0 + se]) continue;
    // slabs swept: [s_lo, s_lo + nslab), unwrapped; the owned variant reads the 2m+1 slabs around its planes as well
    const int s_lo = OWNED ? sb - m - 1 : sb;
    const int nslab = OWNED ? nplane + W - 1 : nplane;

    const int tb1 = j1 * g.Ta[1], tb2 = j2 * g.Ta[2];
    const float sc = win_exp_scale(m);
    float norm = win_norm(m);
    norm = norm * norm * norm;
    // Coefficients: either the caller's row-major [point][Cr] array `xr`, read through the index that every plan record
    // carries (one or two real columns: no permutation pass anywhere), or `xs`, a copy in plan order, one column after
    // the other (gather_rows: more columns).  Both arrive through the staging pipeline below.
    const float *const xcol = xs + (int64_t)cr * xs_stride;
    [[maybe_unused]] const float *const xcol1 = xcol + (two ? xs_stride : 0);  // (paired variant: column cr + 1)
    // x is scaled into [-1, 1] by a power of two so that every operand fits f16; undone at the flush.  The scale is the
    // plane's own -- the largest |x| of its point set in its column, found by plane_absmax_kernel before the launch --
    // so columns and point sets of very different magnitude each keep their full ~22 bits (a single scale for the whole
    // call would flush a column 1e-10 below the largest one to zero; the reference spreads every column independently
    // in fp32).  (Until round 3 every work item scanned its own points for it: a latency-bound prologue, 7 % of the
    // kernel's time at config C3, profiles/r03_spread_trace.txt.)
    auto plane_scale = [&](const int pl) {
        float sc1 = 1.0f;
        const float mx = __uint_as_float(xmax[pl]);
        if (mx > 1.0e-30f && mx < 3.0e38f) {  // (tinier inputs: 1 / scale would overflow; their taps flush to zero anyway)
            int e;
            frexpf(mx, &e);
            sc1 = ldexpf(1.0f, e > 127 ? 127 : e);  // (|x| in [2^127, 3e38): 2^128 is not a float; 1 / it would be 0)
        }
        // (wave-uniform: keep the scales in scalar registers -- the kernel sits at its 128-VGPR limit)
        return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(sc1)));
    };
    const float xscale = plane_scale(plane);
    NFFT_TRACE(1, __builtin_amdgcn_s_memrealtime());
    if (tid == 0) L.inv_xscale = 1.0f / xscale;  // (visible behind the barriers of the schedule set-up below)
    const float unscale =
        __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(xscale * norm * (1.0f / (kOpScale * kOpScale * kPsiScale)))));
    float *const gplane = grid + (int64_t)plane_local * g.cells;
    [[maybe_unused]] float unscale1 = 0.0f;
    [[maybe_unused]] float *gplane1 = nullptr;
    if constexpr (PAIR) {
        if (two) {
            const float xscale1 = plane_scale(plane + 1);
            if (tid == 0) L.inv_xscale1 = 1.0f / xscale1;
            unscale1 = __int_as_float(
                __builtin_amdgcn_readfirstlane(__float_as_int(xscale1 * norm * (1.0f / (kOpScale * kOpScale * kPsiScale)))));
            gplane1 = grid + (int64_t)plane_local1 * g.cells;
        }
    }

    f32x16 acc0 = 0.0f, acc1 = 0.0f;
    bool dirty = false;
    // wave w owns the plane congruent to w (mod 16) inside the sliding window that starts at sb - m
    // (for 2m+2 <= 12 only 12 waves own planes: the other four -- one per SIMD -- stage points and build operands
    // full time instead of idling through the accumulation)
    constexpr int NOWN = W <= 12 ? 12 : 16;
    const bool owner = wave < NOWN;
    const int z_lo = OWNED ? sb : sb - m;  // first plane any owner holds
    int myz = z_lo + (((wave - z_lo) % NOWN) + NOWN) % NOWN;
    // staging threads: the first two non-owner waves if there are any, else waves 0 and 1
    constexpr int kStageWave0 = NOWN == 16 ? 0 : NOWN;
    const int st = tid - kStageWave0 * 64;  // slot of a staging thread, in [0, kSlots)
    const bool stager = (unsigned)st < (unsigned)kSlots;

    auto flush = [&]() __attribute__((always_inline)) {
        if constexpr (PAIR) {
            // accumulator 0 = the 32 x 32 tile of the first column's grid, accumulator 1 = the same cells of the second
            // column's: four 16-byte stores per lane and grid (transposed tile as below: lane = grid row)
            if (owner && myz >= sb && myz < se) {
                const int64_t off = ((int64_t)myz * g.M + tb1 + r32) * g.M + tb2 + 4 * h;
                const float zs0 = ((myz + m) & 1) ? -unscale : unscale;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v;
                    v.x = acc0[4 * q + 0] * zs0; v.y = acc0[4 * q + 1] * zs0; v.z = acc0[4 * q + 2] * zs0; v.w = acc0[4 * q + 3] * zs0;
                    *(f32x4 *)(gplane + off + 8 * q) = v;
                }
                if (two) {
                    const float zs1 = ((myz + m) & 1) ? -unscale1 : unscale1;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v;
                        v.x = acc1[4 * q + 0] * zs1; v.y = acc1[4 * q + 1] * zs1; v.z = acc1[4 * q + 2] * zs1; v.w = acc1[4 * q + 3] * zs1;
                        *(f32x4 *)(gplane1 + off + 8 * q) = v;
                    }
                }
            }
            acc0 = 0.0f;
            acc1 = 0.0f;
            dirty = false;
        } else if constexpr (OWNED) {
            // every owned plane is written exactly once, whether or not points reached it
            if (owner && myz >= sb && myz < se) {  // (the builder waves own nothing)
                // the owned variant accumulates the TRANSPOSED tile (operands swapped in the MFMA): lane = grid row,
                // registers 4q .. 4q+3 = four consecutive grid columns -> eight 16-byte stores per lane and plane
                // instead of 32 4-byte ones (the flush is bound by store instructions, not by bytes)
                const float zscale = ((myz + m) & 1) ? -unscale : unscale;
                float *const grow = gplane + ((int64_t)myz * g.M + tb1 + r32) * g.M + tb2 + 4 * h;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v;
                        v.x = (t == 0 ? acc0[4 * q + 0] : acc1[4 * q + 0]) * zscale;
                        v.y = (t == 0 ? acc0[4 * q + 1] : acc1[4 * q + 1]) * zscale;
                        v.z = (t == 0 ? acc0[4 * q + 2] : acc1[4 * q + 2]) * zscale;
                        v.w = (t == 0 ? acc0[4 * q + 3] : acc1[4 * q + 3]) * zscale;
                        *(f32x4 *)(grow + 32 * t + 8 * q) = v;
                    }
                }
            }
            acc0 = 0.0f;
            acc1 = 0.0f;
            dirty = false;
        } else if (dirty) {
            // (the tile origin and the lane's position pass through an empty asm: everything derived from them -- the 32
            // wrapped row offsets of the boundary pencils above all -- is then computed HERE, once per flush, instead of being
            // hoisted out of the K-block loop into ~35 registers the kernel does not have: it sat at the 128-VGPR limit with
            // 15 VGPRs and 62 SGPRs spilled to scratch until round 4)
            int o1 = tb1 - m, o2 = tb2 - m, M = g.M, hh = h, rr = r32;
            asm volatile("" : "+s"(o1), "+s"(o2), "+s"(M), "+v"(hh), "+v"(rr));
            const int gz = wrap(myz, M);
            const float zscale = ((myz + m) & 1) ? -unscale : unscale;  // plane s - m + l0: parity of s + l0 + m
            // (no test for zero: a plane that received points has few zero cells, and the test costs as much
            // instruction issue as the atomic)
            if (o1 >= 0 && o1 + 32 <= M) {
                // the tile's 32 rows do not cross the periodic boundary (all pencils but the first and the last of a
                // column of pencils): one 32-bit offset per lane and column tile, the rows are wave-uniform strides
                // from it -- 2 vector instructions per atomic instead of 14 (the flush was 13 % of the kernel's)
                float *const pbase = gplane + (int64_t)gz * M * M;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int gc = wrap_near(o2 + 32 * t + rr, M);
                    const unsigned off0 = (unsigned)((o1 + 4 * hh) * M + gc);
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const unsigned row_off = (unsigned)(((reg & 3) + 8 * (reg >> 2)) * M);  // wave-uniform
                        atomicAdd(pbase + (off0 + row_off), (t == 0 ? acc0[reg] : acc1[reg]) * zscale);
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int gc = wrap_near(o2 + 32 * t + rr, M);
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                        const float v = (t == 0 ? acc0[reg] : acc1[reg]) * zscale;
                        const int gr = wrap_near(o1 + row, M);
                        atomicAdd(gplane + ((int64_t)gz * M + gr) * M + gc, v);
                    }
                }
            }
            acc0 = 0.0f;
            acc1 = 0.0f;
            dirty = false;
        }
    };

    // ---- K-block schedule: slab s holds ceil(count / 16) K-blocks; sched[s] = {K-blocks before s, point offset} ---
    if (wave == 0) {
        // three slabs per lane (up to kMaxSweep <= 192 slabs); slab k of the sweep is plan bin wrap(s_lo + k)
        int ob[3], oe[3], nk[3];
        int sum = 0;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int k = 3 * lane + q;
            const int sw = wrap(s_lo + min(k, nslab), g.M);
            ob[q] = tile_offsets[bin0 + sw];
            oe[q] = k < nslab ? tile_offsets[bin0 + sw + 1] : ob[q];
            nk[q] = (oe[q] - ob[q] + kKB - 1) / kKB;  // 0 beyond the sweep
            sum += nk[q];
        }
        int incl = sum;
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        int run = incl - sum;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int k = 3 * lane + q;
            if (k < kMaxSweep + 8) {
                L.sched[k] = make_int2(run, ob[q]);
                L.sched_end[k] = oe[q];
            }
            run += nk[q];
        }
        // (entries at and beyond nslab hold the total: run == total there, and the probes below may read 4 past)
    }
    __syncthreads();
    const int total = L.sched[nslab].x;
    const int nbatch = (total + kNKB - 1) / kNKB;
    if (OWNED && total == 0) {
        // no points anywhere near these planes: they are still this item's to write
        constexpr int QW = TW / 4;  // 16-byte pieces per tile row
        for (int e = tid; e < nplane * 32 * QW; e += kMfmaThreads) {
            const int pz = e / (32 * QW), row = (e / QW) & 31, c4 = e & (QW - 1);
            const int64_t off = ((int64_t)(sb + pz) * g.M + tb1 + row) * g.M + tb2 + 4 * c4;
            *(f32x4 *)(gplane + off) = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (PAIR) {
                if (two) *(f32x4 *)(gplane1 + off) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        continue;
    }

    // ---- staging of a batch: thread -> (K-block, point).  While batch i is accumulated the records of batch i + 6 are
    // requested (LDS-DMA into L.raw[(i + 6) & 7]), the coefficients of batch i + 4 (their records have landed and say
    // where the point's coefficient lives), and batch i + 2 is converted to (cell, fraction, scaled value) form: every
    // load has two steps to arrive -- under the atomic traffic of the flushes one step is not enough.  Every step issues
    // exactly two DMA instructions per wave (lanes without a point read a dummy address), so that the consumer can
    // wait with a count.
    int cur = 0;  // slab of this thread's previous K-block (K-blocks only move forward)
    // this thread's slot of a batch: plan index of its point (any valid one for a padding slot), whether it has one,
    // the K-block's slab (unwrapped)
    auto locate = [&](const int batch, int &idx, int &have, int &slab) {
        const int j = st / kKB, i = st - j * kKB;
        const int q = batch * kNKB + j;
        have = 0;
        slab = INT_MAX;
        idx = 0;
        if (q < total) {
            // sched[lo].x <= q < sched[lo + 1].x; probe four slabs ahead per LDS round trip
            int lo = cur;
            int2 e0;
            while (true) {
                const int2 c0 = L.sched[lo], c1 = L.sched[lo + 1], c2 = L.sched[lo + 2], c3 = L.sched[lo + 3],
                           c4 = L.sched[lo + 4];
                if (q < c1.x) { e0 = c0; break; }
                if (q < c2.x) { e0 = c1; lo += 1; break; }
                if (q < c3.x) { e0 = c2; lo += 2; break; }
                if (q < c4.x) { e0 = c3; lo += 3; break; }
                lo += 4;
            }
            cur = lo;
            slab = s_lo + lo;  // unwrapped
            idx = e0.y + kKB * (q - e0.x) + i;
            have = idx < L.sched_end[lo];
            if (!have) idx = e0.y;  // any valid point: the value is not used
        }
    };
    // record request of a batch (B): locate this thread's slot, one 16-byte DMA, bookkeeping for the later steps
    auto request_records = [&](const int batch) {
        int idx, have, slab;
        locate(batch, idx, have, slab);
        const int j = st / kKB, i = st - j * kKB;
        const int buf = batch & (kRecRing - 1);
        lds_dma_dwordx4(spos + (int64_t)idx * 4, &L.raw[buf][(wave - kStageWave0) * 64]);
        L.raw_have[buf][st] = (signed char)have;
        if (i == 0) L.raw_slab[buf][j] = slab;
        if (!xr) L.raw_idx[buf][st] = idx;
    };
    // coefficient request of a batch (A), two steps behind its record request: the record has landed and holds the
    // index of the point in the caller's arrays
    auto request_coefficients = [&](const int batch) {
        const float *src;
        if (xr) {
            const int orig = __float_as_int(L.raw[batch & (kRecRing - 1)][st].w);  // (padding slots hold a valid record)
            src = xr + (int64_t)orig * Cr + cr;
        } else {
            src = xcol + L.raw_idx[batch & (kRecRing - 1)][st];
        }
        lds_dma_dword(src, &L.rawx[batch & (kXRing - 1)][(wave - kStageWave0) * 64]);
        if constexpr (PAIR) {
            // the second column's coefficient (the same address again in a slot of one column: the count of DMA
            // instructions per step stays uniform)
            const float *src1 = xr ? src + (two ? 1 : 0) : xcol1 + L.raw_idx[batch & (kRecRing - 1)][st];
            lds_dma_dword(src1, &L.rawx1[batch & (kXRing - 1)][(wave - kStageWave0) * 64]);
        }
    };
    auto stage_convert = [&](MfmaStage &S, const int batch, const bool newest_in_flight) {
        const int buf = batch & (kRecRing - 1);
        // the stager waves of the 12-owner layout issue no other vector-memory traffic: a counted wait leaves the two
        // requests of the previous step in flight (plane-owner waves also have flush atomics outstanding: wait for all)
        if (newest_in_flight && NOWN != 16) { if constexpr (PAIR) wait_lds_dma_but_newest3(); else wait_lds_dma_but_newest(); }
        else wait_lds_dma();
        float f0 = 0.f, f1 = 0.f, f2 = 0.f, xv = 0.f;
        [[maybe_unused]] float xv1 = 0.f;
        int c1 = -1000, c2 = -1000;  // padding slots: outside every window
        if (L.raw_have[buf][st]) {
            int c0;
            const f32x4 rec = L.raw[buf][st];
            split_cell(rec.x, g.M, c0, f0);
            split_cell(rec.y, g.M, c1, f1);
            split_cell(rec.z, g.M, c2, f2);
            if constexpr (OWNED) {
                // cell relative to the tile, in [-m - 1, T + m): tap l sits at row c1 - m + l1 of the (unpadded) tile
                c1 -= tb1;
                c2 -= tb2;
                // (periodic image whose window reaches the tile: offsets in [-m - 1, T + m); M >= 128 keeps it unique)
                c1 = c1 >= 32 + m ? c1 - g.M : (c1 < -(m + 1) ? c1 + g.M : c1);
                c2 = c2 >= TW + m ? c2 - g.M : (c2 < -(m + 1) ? c2 + g.M : c2);
            } else {
                c1 -= tb1 - m;  // row of the point's cell inside the padded pencil (tap l1 sits at row c1 - m + l1)
                c2 -= tb2 - m;
            }
            xv = L.rawx[batch & (kXRing - 1)][st] * L.inv_xscale;
            // |x| above the plane's maximum: the batch vector is not sorted (the maxima are taken over the row range
            // of every point set).  Reported; the value is clamped so that no operand leaves the f16 range.
            // (NaN / infinite coefficients are the caller's business: they pass through and poison their plane, as in fp32)
            if (fabsf(xv) > 1.0f && fabsf(xv) < __builtin_huge_valf()) {
                report_fault(status, kFaultBatchOrder);
                xv = fminf(fmaxf(xv, -1.0f), 1.0f);
            }
            if constexpr (PAIR) {
                if (two) {
                    xv1 = L.rawx1[batch & (kXRing - 1)][st] * L.inv_xscale1;
                    if (fabsf(xv1) > 1.0f && fabsf(xv1) < __builtin_huge_valf()) {
                        report_fault(status, kFaultBatchOrder);
                        xv1 = fminf(fmaxf(xv1, -1.0f), 1.0f);
                    }
                }
            }
        }
        if constexpr (PAIR) S.x1[st] = xv1;
        S.f0[st] = f0; S.f1[st] = f1; S.f2[st] = f2; S.x[st] = xv;
        S.c1[st] = c1; S.c2[st] = c2;
        if ((st & (kKB - 1)) == 0) S.slab[st / kKB] = L.raw_slab[buf][st / kKB];  // written by this same thread
    };

    // ---- operands of a batch: three wave-sized tasks per K-block, handed out through an LDS counter so that the
    // waves whose plane lies outside the batch's windows (6 of 16 for m = 4) build them while the others run MFMAs.
    // Only the 16 (2m+2) taps per axis are evaluated and scattered into zero-filled tables.
    [[maybe_unused]] int tasks_taken = 0;  // (trace builds only)
    auto build_tasks = [&](const MfmaStage &S, MfmaOps<W> &O, const int nkb, int *counter) {
        tasks_taken = 0;
        while (true) {
            // all 64 lanes add 1 (the compiler folds this into one ds_add of 64 per wave): the counter runs in units of
            // 64, lane 0 sees the wave's base value
            const int j = __builtin_amdgcn_readfirstlane(atomicAdd(counter, 1)) >> 6;
            if (j >= nkb) break;
            ++tasks_taken;
            // One task = the three operand tables of K-block j.  A task is a chain of LDS round trips (counter, inputs,
            // zero fill, scattered writes), not arithmetic: as three tasks of one table each the builds took 57 % of the
            // workgroup's wave time (profiles/r03_experiments.md), as two (B side / A side) they change nothing
            // (profiles/r04_experiments.md); in one task the chains overlap and the point's inputs are read once.
            // ---- zero fills: B fragments [column tile][hi/lo][lane = 32 (k / 8) + column] element k % 8, psi1 [row][point]
            {
                const f16x8 zero = (_Float16)0.0f;
                O.bfrag[j][0][0][lane] = zero;
                O.bfrag[j][0][1][lane] = zero;
                if constexpr (!PAIR) {  // (paired variant: one column tile; the second one's space holds the second axis-0 table)
                    O.bfrag[j][1][0][lane] = zero;
                    O.bfrag[j][1][1][lane] = zero;
                }
                f32x4 *pz = (f32x4 *)&O.p1[j][0][0][0][0];  // 2 x 2 x 32 x 8 halves = 128 x 16 bytes
                const f32x4 zero4 = 0.0f;
                pz[lane] = zero4;
                pz[lane + 64] = zero4;
            }
            asm volatile("" ::: "memory");  // the scattered writes below must stay behind the zero fills
            // lane = 4 k + g: point k of the K-block, taps g, g + 4, g + 8 (, g + 12) -- the point's cell and fraction
            // are read once, the tap loops have a compile-time trip count and no division
            const int k = lane >> 2, g4 = lane & 3;
            const int slot = j * kKB + k;
            const int c2v = S.c2[slot], c1v = S.c1[slot];
            const float f2v = S.f2[slot], f1v = S.f1[slot], f0v = S.f0[slot], xv = S.x[slot];
            [[maybe_unused]] const float xv1 = PAIR ? S.x1[slot] : 0.0f;
            // (paired variant) axis-0 table of the second column, [tap][hi/lo][point], in the unused second column tile
            [[maybe_unused]] _Float16 *const a1 = (_Float16 *)&O.bfrag[j][1][0][0];
            const int sl = S.slab[j];
            int touched = 0;
            _Float16 *const base_h = (_Float16 *)&O.bfrag[j][0][0][32 * (k >> 3)] + (k & 7);
#pragma unroll
            for (int t = 0; t < (W + 3) / 4; ++t) {
                const int l = g4 + 4 * t;
                // B fragments: f16 hi / lo of psi2 (times the operand scale)
                {
                    const int col = c2v - m + l;
                    const float d = f2v + (float)(m - l);
                    const float v = __builtin_amdgcn_exp2f(sc * d * d) * kOpScale;
                    unsigned hi, lo;
                    split_pair(v, 0.0f, hi, lo);
                    if (l < W && (unsigned)col < (unsigned)TW) {  // (padding slots fail the second test)
                        // element [col >> 5][hi/lo][32 (k >> 3) + (col & 31)][k & 7] of bfrag[j]
                        _Float16 *ph = base_h + (col >> 5) * (2 * 64 * 8) + (col & 31) * 8;
                        ph[0] = __builtin_bit_cast(_Float16, (unsigned short)hi);
                        ph[64 * 8] = __builtin_bit_cast(_Float16, (unsigned short)lo);
                        touched |= 1 + (col >> 5);
                    }
                }
                // psi1 table [row][point] and axis-0 table [tap][point] (x' psi0), both as f16 splits: the owners form the
                // split of the product psi1 * (x' psi0) from them in packed f16 arithmetic (split_product_f16x4).  Odd planes
                // accumulate the negated sum (undone at the flush): the sign-independent part of the MFMA accumulation's
                // truncation bias then alternates from plane to plane
                {
                    const int row = c1v - m + l;
                    const float d1 = f1v + (float)(m - l);
                    const float v1 = __builtin_amdgcn_exp2f(sc * d1 * d1) * kPsiScale;
                    const float d0 = f0v + (float)(m - l);
                    const float sgn = ((sl + l) & 1) ? -kOpScale : kOpScale;
                    const float va = xv * __builtin_amdgcn_exp2f(sc * d0 * d0) * sgn;
                    unsigned hi, lo;  // low halves: psi1, high halves: x' psi0
                    split_pair(v1, va, hi, lo);
                    if (l < W) {
                        O.a0[j][l][0][k] = __builtin_bit_cast(_Float16, (unsigned short)(hi >> 16));
                        O.a0[j][l][1][k] = __builtin_bit_cast(_Float16, (unsigned short)(lo >> 16));
                        if ((unsigned)row < 32u) {
                            O.p1[j][0][k >> 3][row][k & 7] = __builtin_bit_cast(_Float16, (unsigned short)hi);
                            O.p1[j][1][k >> 3][row][k & 7] = __builtin_bit_cast(_Float16, (unsigned short)lo);
                        }
                    }
                    if constexpr (PAIR) {
                        const float vb = xv1 * __builtin_amdgcn_exp2f(sc * d0 * d0) * sgn;
                        unsigned hi1, lo1;
                        split_pair(vb, 0.0f, hi1, lo1);
                        if (l < W) {
                            a1[(l * 2 + 0) * kKB + k] = __builtin_bit_cast(_Float16, (unsigned short)hi1);
                            a1[(l * 2 + 1) * kKB + k] = __builtin_bit_cast(_Float16, (unsigned short)lo1);
                        }
                    }
                }
            }
            // the plan orders a slab's points by column group (common.h): most K-blocks touch one tile only
            const int t0 = __builtin_amdgcn_ballot_w64((touched & 1) != 0) != 0ull;
            const int t1 = __builtin_amdgcn_ballot_w64((touched & 2) != 0) != 0ull;
            if (lane == 0) {
                O.halves[j] = t0 + 2 * t1;
                O.slab[j] = sl;
            }
        }
    };

    // ---- every wave adds the K-blocks that reach its plane -------------------------------------------------
    auto accumulate = [&](const MfmaOps<W> &O, const int nkb) {
        // K-block j's slab and touched column tiles, packed into one word per lane: ONE readlane per K-block and owner (every
        // owner walks every K-block of the batch, in or out of its window: the second readlane was 2 % of the kernel's VALU)
        const int packed = O.slab[lane & (kNKB - 1)] * 4 + O.halves[lane & (kNKB - 1)];
        for (int j = 0; j < nkb; ++j) {
            const int pk = __builtin_amdgcn_readlane(packed, j);
            const int s = pk >> 2;
            const int hv = pk & 3;
            // the sweep has passed plane myz once the current slab is beyond myz + m
            while (myz + m < s) {
                flush();
                myz += NOWN;
            }
            const int l0 = myz - s + m;  // axis-0 tap of this K-block's points that lands on my plane
            if ((unsigned)l0 < (unsigned)W && (!OWNED || myz < se)) {
                // A fragment of my plane: (hi, lo) of psi1[row][point] * (x' psi0)[tap l0][point] for the lane's row and
                // 8 points, from the f16 splits of the two factors -- 4 ds_read_b128 and 16 packed f16 instructions
                const u32x4 ph = *(const u32x4 *)&O.p1[j][0][h][r32][0], pl = *(const u32x4 *)&O.p1[j][1][h][r32][0];
                const u32x4 xh = *(const u32x4 *)&O.a0[j][l0][0][8 * h], xl = *(const u32x4 *)&O.a0[j][l0][1][8 * h];
                u32x4 uh, ul;
                split_product_f16x4(ph, pl, xh, xl, uh, ul);
                const f16x8 ah = __builtin_bit_cast(f16x8, uh), al = __builtin_bit_cast(f16x8, ul);
                if constexpr (PAIR) {
                    // one column tile, two columns: the B fragments serve both chains
                    const f16x8 b0h = O.bfrag[j][0][0][lane], b0l = O.bfrag[j][0][1][lane];
                    if (two) {
                        const _Float16 *const a1 = (const _Float16 *)&O.bfrag[j][1][0][0];
                        const u32x4 yh = *(const u32x4 *)&a1[(l0 * 2 + 0) * kKB + 8 * h], yl = *(const u32x4 *)&a1[(l0 * 2 + 1) * kKB + 8 * h];
                        u32x4 vh, vl;
                        split_product_f16x4(ph, pl, yh, yl, vh, vl);
                        const f16x8 ch = __builtin_bit_cast(f16x8, vh), cl = __builtin_bit_cast(f16x8, vl);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0h, ah, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0h, ch, acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0l, ah, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0l, ch, acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0h, al, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0h, cl, acc1, 0, 0, 0);
                    } else {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0h, ah, acc0, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0l, ah, acc0, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0h, al, acc0, 0, 0, 0);
                    }
                    dirty = true;
                    continue;
                }
                // one column tile after the other, each a chain of three MFMAs on its own accumulator (requesting the B
                // fragments ahead of the packed arithmetic, alternating the two chains, or both: no gain, profiles/r04_experiments.md)
                if (hv & 1) {
                    const f16x8 b0h = O.bfrag[j][0][0][lane], b0l = O.bfrag[j][0][1][lane];
                    if constexpr (OWNED) {
                        // transposed product (columns x rows): the fragments of the two operands have the same lane layout
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0h, ah, acc0, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0l, ah, acc0, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0h, al, acc0, 0, 0, 0);
                    } else {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b0h, acc0, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b0l, acc0, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b0h, acc0, 0, 0, 0);
                    }
                }
                if (hv & 2) {
                    const f16x8 b1h = O.bfrag[j][1][0][lane], b1l = O.bfrag[j][1][1][lane];
                    if constexpr (OWNED) {
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1h, ah, acc1, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1l, ah, acc1, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1h, al, acc1, 0, 0, 0);
                    } else {
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b1h, acc1, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b1l, acc1, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b1h, acc1, 0, 0, 0);
                    }
                }
                dirty = true;
            }
        }
    };

    // ---- software pipeline, one barrier per batch.  Step i: convert the points of batch i + 2 (requested in step
    // i - 2) into staging buffer i & 1 (its last reader, build_tasks of batch i in step i - 1, is behind the previous
    // barrier), request batch i + 4, build the operands of batch i + 1 (buffer (i+1) & 1) and run the MFMAs of batch i.
    // Requests past the last batch are dummies that keep the count of outstanding DMA instructions uniform.
    if (tid < 2) L.task_counter[tid] = 0;
    if (stager) {
        // fill the pipeline: records of batches 0 .. 4, then (they have landed) the coefficients of 0 .. 2, then batch 0
        for (int q = 0; q < 5; ++q) request_records(q);
        wait_lds_dma();
        for (int q = 0; q < 3; ++q) request_coefficients(q);
        wait_lds_dma();
        stage_convert(L.stag[0], 0, false);
    }
    __syncthreads();
    NFFT_TRACE(2, __builtin_amdgcn_s_memrealtime());
    NFFT_TRACE(5, (unsigned long long)(unsigned)total | ((unsigned long long)(unsigned)(tile_offsets[bin0 + wrap(s_lo + nslab - 1, g.M) + 1] - tile_offsets[bin0 + wrap(s_lo, g.M)]) << 32));
    for (int i = -1; i < nbatch; ++i) {
        NFFT_STEP(i, 0);
        if (stager) {
            if (i + 2 < nbatch) stage_convert(L.stag[i & 1], i + 2, true);
            request_coefficients(i + 4);
            request_records(i + 6);
        }
        if (tid == 0) L.task_counter[i & 1] = 0;  // for the next step; its last user is behind the previous barrier
        if (i >= 0 && owner) accumulate(L.ops[i & 1], min(kNKB, total - i * kNKB));
        NFFT_STEP(i, 1);
        if (i + 1 < nbatch)
            build_tasks(L.stag[(i + 1) & 1], L.ops[(i + 1) & 1], min(kNKB, total - (i + 1) * kNKB),
                        &L.task_counter[(i + 1) & 1]);
        NFFT_STEP_V(i, 2, (unsigned long long)tasks_taken << 56);
        barrier_lds_only();
        NFFT_STEP(i, 3);
    }
    // the dummy requests of the last steps must have landed before the workgroup gives its LDS back
    if (stager) wait_lds_dma();
    NFFT_TRACE(6, __builtin_amdgcn_s_memrealtime());
    flush();
    if constexpr (OWNED) {
        // owned planes behind the last slab that holds points
        for (myz += NOWN; myz < se; myz += NOWN) flush();
    }
    NFFT_TRACE(3, __builtin_amdgcn_s_memrealtime());
    }  // work items
}

} // namespace

bool spread_mfma_supported(const Geom &g) { return g.dim == 3 && g.wide; }

#ifdef NFFT_HIP_TRACE
extern "C" int nfft_dbg_set_spread_trace(void *device_buffer)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_spread_trace), &device_buffer, sizeof(device_buffer));
}
extern "C" int nfft_dbg_set_step_trace(void *device_buffer)  // 16 workgroups x 16 waves x 64 steps x 4 stamps x 8 bytes
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_step_trace), &device_buffer, sizeof(device_buffer));
}
#endif

// Largest |x| of every (point set, real column) plane: the power-of-two operand scale of the spreading kernel.
// `xr` is the caller's row-major [point][Cr] array; the batch vector is sorted, so point set b is the row range
// [rows[b], rows[b + 1]) with rows[b] = plan entries in front of set b in the (halo) plan.  Non-negative floats order
// like their bit patterns, so the maxima are taken with integer atomics.
constexpr int kAbsmaxThreads = 1024;
__global__ void __launch_bounds__(kAbsmaxThreads)
plane_absmax_kernel(const int *__restrict__ tile_offsets, const int64_t bins_per_set, const float *__restrict__ xr,
                    const int Cr, const int64_t B, unsigned *__restrict__ xmax)
{
    __shared__ unsigned lmax[kAbsmaxThreads];
    constexpr int NT = kAbsmaxThreads;
    // (the y grid holds at most 65 535 point sets: the workgroups stride over the rest)
    for (int64_t b = blockIdx.y; b < B; b += gridDim.y) {
    if (b != (int64_t)blockIdx.y) __syncthreads();  // lmax is reused
    const int64_t e0 = (int64_t)tile_offsets[b * bins_per_set] * Cr, e1 = (int64_t)tile_offsets[(b + 1) * bins_per_set] * Cr;
    const int64_t chunk = (e1 - e0 + gridDim.x - 1) / gridDim.x;
    const int64_t lo = e0 + chunk * blockIdx.x, hi = min(e1, lo + chunk);
    const bool fixed_col = NT % Cr == 0;  // then a thread sees one column only: e = lo' + t + NT k
    if (!fixed_col && Cr > NT) {
        for (int64_t e = lo + threadIdx.x; e < hi; e += NT) {
            const unsigned v = __float_as_uint(fabsf(xr[e]));
            unsigned *const dst = &xmax[(int64_t)b * Cr + (int)(e % Cr)];
            if (v > __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dst, v);
        }
        continue;
    }
    lmax[threadIdx.x] = 0u;
    __syncthreads();
    if (fixed_col) {
        const int64_t lo_al = lo - lo % Cr;  // start on a row boundary so that column = thread % Cr
        // four loads in flight per thread (a dependent one-load loop is latency-bound)
        float m0 = 0.0f, m1 = 0.0f, m2 = 0.0f, m3 = 0.0f;
        int64_t e = lo_al + threadIdx.x;
        for (; e + 3 * NT < hi; e += 4 * NT) {
            const float v0 = e >= lo ? xr[e] : 0.0f, v1 = xr[e + NT], v2 = xr[e + 2 * NT], v3 = xr[e + 3 * NT];
            m0 = fmaxf(m0, fabsf(v0)); m1 = fmaxf(m1, fabsf(v1)); m2 = fmaxf(m2, fabsf(v2)); m3 = fmaxf(m3, fabsf(v3));
        }
        for (; e < hi; e += NT)
            if (e >= lo) m0 = fmaxf(m0, fabsf(xr[e]));
        float mx = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
        if (Cr == 1) {
            for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
            if ((threadIdx.x & 63) == 0) atomicMax(&lmax[0], __float_as_uint(mx));
        } else {
            atomicMax(&lmax[threadIdx.x % Cr], __float_as_uint(mx));
        }
    } else {
        for (int64_t e = lo + threadIdx.x; e < hi; e += NT) atomicMax(&lmax[(int)(e % Cr)], __float_as_uint(fabsf(xr[e])));
    }
    __syncthreads();
    // Few, big workgroups: every one ends with an access to the plane's ONE word, and same-address traffic serialises at
    // its L2 channel (measured, scripts/ubench/absmax_bench.hip: 2 442 workgroups 42 us, 1 024 22 us for the same 40 MB).
    // A look before the atomic: only ~ln(workgroups) of them ever raise the maximum.
    if ((int)threadIdx.x < Cr && lmax[threadIdx.x]) {
        unsigned *const dst = &xmax[(int64_t)b * Cr + threadIdx.x];
        if (lmax[threadIdx.x] > __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dst, lmax[threadIdx.x]);
    }
    }  // point sets
}

int launch_plane_absmax(const Geom &g_halo, const PlanLayout &L_halo, const void *plan_halo, const float *xr, int64_t n,
                        int64_t B, int64_t Cr, unsigned *xmax, hipStream_t stream)
{
    if (B * Cr <= 0) return 0;
    NFFT_HIP_CHECK(hipMemsetAsync(xmax, 0, (size_t)(B * Cr * 4), stream));
    if (n <= 0) return 0;
    const int *to = (const int *)((const char *)plan_halo + L_halo.off_offsets);
    int64_t blocks = (n * Cr / B + kAbsmaxThreads * 32 - 1) / (kAbsmaxThreads * 32);  // ~32 elements per thread
    blocks = blocks < 1 ? 1 : (blocks > 512 ? 512 : blocks);
    hipLaunchKernelGGL(plane_absmax_kernel, dim3((unsigned)blocks, (unsigned)(B < 65535 ? B : 65535)), dim3(kAbsmaxThreads), 0, stream, to,
                       (int64_t)g_halo.tiles_per_batch * g_halo.SB, xr, (int)Cr, B, xmax);
    NFFT_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int W, bool OWNED, bool PAIR>
static int launch_mfma_t(const Geom &g, const PlanLayout &L, const void *plan, const int *to, const float *spos,
                         const float *xr, const float *xs, const unsigned *xmax, int64_t n, int64_t Cr, int64_t plane0,
                         int64_t nplanes, float *grid, hipStream_t stream)
{
    // Ranges per pencil: about 5-6 workgroups per CU balance the tail of the launch against the 2m+1 halo planes
    // every range flushes on top of its own (measured at C3: 6 ranges 7 % faster than 4, 12 in between).  The count
    // is a function of the plan's sizes only, because the plan's load-balance tables are built for it (common.h).
    const int64_t pencils = (int64_t)g.nta[1] * g.nta[2];
    int64_t nsets = g.tiles_per_batch > 0 ? L.ntiles / g.tiles_per_batch : 1;
    if (nsets < 1) nsets = 1;
    const int nsegm = seg_base_runs(n, nsets, pencils, g.M, device_cu_count());
    const int seg_slabs = (g.M + nsegm - 1) / nsegm;
    // y grid: planes, or the pair slots the chunk of planes touches (paired variant)
    int64_t ny = nplanes;
    if (PAIR) {
        const int64_t P = (Cr + 1) / 2, last = plane0 + nplanes - 1;
        const int64_t s0 = (plane0 / Cr) * P + (plane0 % Cr) / 2, s1 = (last / Cr) * P + (last % Cr) / 2;
        ny = s1 - s0 + 1;
    }
    const dim3 blocks((unsigned)(pencils * nsegm), (unsigned)ny);
    static DeviceOnce attr_done;  // one workgroup per CU: the double-buffered operands take most of the 160 KB LDS
    if (attr_done.first_use()) {
        NFFT_HIP_CHECK(hipFuncSetAttribute((const void *)spread_mfma_kernel<W, false, OWNED, PAIR>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MfmaLds<W>)));
        NFFT_HIP_CHECK(hipFuncSetAttribute((const void *)spread_mfma_kernel<W, true, OWNED, PAIR>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MfmaLds<W>)));
        attr_done.mark();
    }
    const char *base = (const char *)plan;
    const int4 *work = (const int4 *)(base + L.off_work), *sorted = work + L.work_head + L.work_cap;
    int *const status = device_status_block();
    hipLaunchKernelGGL((spread_mfma_kernel<W, false, OWNED, PAIR>), blocks, dim3(kMfmaThreads), sizeof(MfmaLds<W>), stream, g, to,
                       spos, xr, xs, L.cap, xmax, (int)Cr, (int)plane0, (int)nplanes, grid, seg_slabs, nsegm, work, sorted,
                       WorkTickets{nullptr, 0u}, status);
    // the persistent launch over the work list (unbalanced plans; its workgroups leave at once otherwise); entries are
    // handed out by tickets when the launch's planes fit its share of the ticket ring, else round robin
    const WorkTickets tickets{ny <= kTicketPlanes ? device_ticket_ring() : nullptr, next_launch_number()};
    const dim3 oblocks(work_list_workgroups(n, nsets, pencils, nsegm, device_cu_count()), (unsigned)ny);
    hipLaunchKernelGGL((spread_mfma_kernel<W, true, OWNED, PAIR>), oblocks, dim3(kMfmaThreads), sizeof(MfmaLds<W>), stream, g, to,
                       spos, xr, xs, L.cap, xmax, (int)Cr, (int)plane0, (int)nplanes, grid, seg_slabs, nsegm, work, sorted, tickets,
                       status);
    NFFT_HIP_CHECK(hipGetLastError());
    return 0;
}

template <bool OWNED, bool PAIR>
static int launch_mfma_w(const Geom &g, const PlanLayout &L, const void *plan, const float *xr, const float *xs,
                         const unsigned *xmax, int64_t n, int64_t Cr, int64_t plane0, int64_t nplanes, float *grid,
                         hipStream_t stream)
{
    const char *base = (const char *)plan;
    const int *to = (const int *)(base + L.off_offsets);
    const float *spos = (const float *)(base + L.off_spos);
    switch (g.m) {
    case 1: return launch_mfma_t<4, OWNED, PAIR>(g, L, plan, to, spos, xr, xs, xmax, n, Cr, plane0, nplanes, grid, stream);
    case 2: return launch_mfma_t<6, OWNED, PAIR>(g, L, plan, to, spos, xr, xs, xmax, n, Cr, plane0, nplanes, grid, stream);
    case 3: return launch_mfma_t<8, OWNED, PAIR>(g, L, plan, to, spos, xr, xs, xmax, n, Cr, plane0, nplanes, grid, stream);
    case 4: return launch_mfma_t<10, OWNED, PAIR>(g, L, plan, to, spos, xr, xs, xmax, n, Cr, plane0, nplanes, grid, stream);
    case 5: return launch_mfma_t<12, OWNED, PAIR>(g, L, plan, to, spos, xr, xs, xmax, n, Cr, plane0, nplanes, grid, stream);
    case 6: return launch_mfma_t<14, OWNED, PAIR>(g, L, plan, to, spos, xr, xs, xmax, n, Cr, plane0, nplanes, grid, stream);
    case 7: return launch_mfma_t<16, OWNED, PAIR>(g, L, plan, to, spos, xr, xs, xmax, n, Cr, plane0, nplanes, grid, stream);
    }
    set_error("matrix-core spreading supports cutoff 1..7");
    return 1;
}

// `n` is the problem's point count (it fixes the work decomposition the plan was built for); the plan may hold more
// entries than that (owned tiling).  The owned variant writes every cell of the planes: no zero-fill needed.
// Coefficients: xr != nullptr: the caller's row-major [point][Cr] array, read through the index in the plan records
// (one or two real columns); else xs, the copy in plan order, column after column with stride L.cap (gather_rows).
// xmax: largest |x| of every plane [B * Cr] (launch_plane_absmax), indexed by the global plane number plane0 + p.
int launch_spread_mfma(const Geom &g, const PlanLayout &L, const void *plan, const float *xr, const float *xs,
                       const unsigned *xmax, int64_t n, int64_t Cr, int64_t plane0, int64_t nplanes, float *grid,
                       hipStream_t stream)
{
    if (nplanes <= 0) return 0;
    if (n <= 0) {
        if (g.owned) NFFT_HIP_CHECK(hipMemsetAsync(grid, 0, (size_t)(nplanes * g.cells * 4), stream));
        return 0;
    }
    if (g.pair && Cr < 2) {
        set_error("Input mismatch: a plan of the paired owned tiling (num_columns >= 2) used with one real column");
        return 1;
    }
    return g.pair    ? launch_mfma_w<true, true>(g, L, plan, xr, xs, xmax, n, Cr, plane0, nplanes, grid, stream)
           : g.owned ? launch_mfma_w<true, false>(g, L, plan, xr, xs, xmax, n, Cr, plane0, nplanes, grid, stream)
                     : launch_mfma_w<false, false>(g, L, plan, xr, xs, xmax, n, Cr, plane0, nplanes, grid, stream);
}

} // namespace nfft
