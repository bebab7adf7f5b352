// Matrix-core spreading without the lock step (cutoffs m <= 4 of the wide tiling; spread_mfma.hip keeps the wider windows).
//
// Same arithmetic as spread_mfma.hip -- per grid plane z of a pencil G_z = A_z B on v_mfma_f32_32x32x16_f16 with f16-split
// operands, plane-owner waves holding the 32 x 64 accumulator tile in registers -- and the same result as the reference's
// adjoint_window_convolution kernels (csrc/cuda/spatial_window_operations.cu:103-211).  What differs is how the waves of a
// workgroup hand work to each other.  spread_mfma.hip runs one software-pipelined loop for all 16 waves with a workgroup
// barrier per batch of 8 K-blocks; its step trace (profiles/r04_spread_steps.txt) shows a third of every step in arrival
// skew: the owner that has to flush its plane (32 atomics per lane) or to build a late operand table is ~2 000 cycles behind
// and fifteen waves wait for it, every step.  Here every wave runs the loop of its ROLE and the roles meet through flags in LDS:
//   * 2 stager waves   fetch the plan records and coefficients of a batch by LDS-DMA and convert them to (cell, fraction,
//                      scaled value) form -- four batches of staging buffers, `staged[wave]` counts the batches converted;
//   * 4 builder waves  take the next K-block from a counter, wait until its points are staged and its ring slot is free
//                      (every owner is past the K-block that used the slot 16 K-blocks earlier), build the three operand
//                      tables of the K-block in the slot and publish `ready[slot] = K-block + 1`;
//   * 10 owner waves   (one plane of the sliding window of 2m + 2 <= 10 planes each) walk the K-blocks in order: one poll
//                      reads the 16 ready words, then every K-block known to be there is accumulated without further flag
//                      traffic; `done[wave]` counts the K-blocks the owner is finished with.
// A wave that is late (a flush, a slow table) delays nobody until the ring of 16 K-blocks is used up, and the late wave
// catches up at its own pace.  The loops are separate code paths, so the owner loop keeps only its own state in registers
// (the lock-step kernel sits at the 128-VGPR limit with spills).  Every spin is bounded (kSpinLimit): the wave that runs
// out raises the workgroup's abort word and reports the fault to the host (common.h: report_fault).
#include <algorithm>
#include <climits>
#include <cstddef>
#include <cstdlib>

#include "common.h"
#include "kernels.h"
#include "mfma_split.h"

namespace nfft {

#ifdef NFFT_HIP_TRACE
// Developer instrumentation (variant builds only, scripts/exp_build.sh -DNFFT_HIP_TRACE): per wave of the first 16
// workgroups {cycles in the role's loop, cycles of it spent waiting for a flag, units of work, second wait class}.
__device__ unsigned long long *g_ring_trace = nullptr;
#define RING_T0() const unsigned long long t_begin_ = __builtin_amdgcn_s_memtime(); unsigned long long t_wait_ = 0, t_wait2_ = 0, n_work_ = 0
#define RING_WAIT_BEGIN() const unsigned long long tw_ = __builtin_amdgcn_s_memtime()
#define RING_WAIT_END() t_wait_ += __builtin_amdgcn_s_memtime() - tw_
#define RING_WAIT2_END() t_wait2_ += __builtin_amdgcn_s_memtime() - tw_
#define RING_WORK() ++n_work_
__device__ unsigned long long *g_ring_phase = nullptr;  // [16 workgroups][16 waves][8 phases]
#define RING_PH_DECL() unsigned long long ph_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long pt_ = __builtin_amdgcn_s_memtime()
#define RING_PH(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); ph_[k] += n_ - pt_; pt_ = n_; } while (0)
#define RING_PH_OUT()                                                                                             \
    do {                                                                                                          \
        if (!OVERFLOW && g_ring_phase && blockIdx.y == 0 && blockIdx.x < 16 && lane == 0)                        \
            for (int k_ = 0; k_ < 8; ++k_) g_ring_phase[((size_t)blockIdx.x * 16 + wave) * 8 + k_] = ph_[k_];     \
    } while (0)
#define RING_T1()                                                                                                 \
    do {                                                                                                          \
        if (!OVERFLOW && g_ring_trace && blockIdx.y == 0 && blockIdx.x < 16 && lane == 0) {                      \
            unsigned long long *t_ = g_ring_trace + ((size_t)blockIdx.x * 16 + wave) * 4;                         \
            t_[0] = __builtin_amdgcn_s_memtime() - t_begin_;                                                      \
            t_[1] = t_wait_;                                                                                      \
            t_[2] = n_work_;                                                                                      \
            t_[3] = t_wait2_;                                                                                     \
        }                                                                                                         \
    } while (0)
#else
#define RING_T0() do { } while (0)
#define RING_WAIT_BEGIN() do { } while (0)
#define RING_WAIT_END() do { } while (0)
#define RING_WAIT2_END() do { } while (0)
#define RING_WORK() do { } while (0)
#define RING_T1() do { } while (0)
#define RING_PH_DECL() do { } while (0)
#define RING_PH(k) do { } while (0)
#define RING_PH_OUT() do { } while (0)
#endif

// Timing-only switches for scripts/exp_build.sh variants (results are WRONG with any of them set): what the kernel does
// not wait for.  RING_NO_MFMA: owners issue no MFMAs; RING_NO_SPLIT: no packed arithmetic; RING_NO_LOADS: no operand loads
// from LDS; RING_NO_BUILD: builders publish empty tables; RING_NO_ATOMICS: flush without the global atomics.
#ifndef RING_NO_MFMA
#define RING_NO_MFMA 0
#endif
#ifndef RING_NO_SPLIT
#define RING_NO_SPLIT 0
#endif
#ifndef RING_NO_LOADS
#define RING_NO_LOADS 0
#endif
#ifndef RING_NO_BUILD
#define RING_NO_BUILD 0
#endif
#ifndef RING_NO_ATOMICS
#define RING_NO_ATOMICS 0
#endif
#ifndef RING_NO_STAGE
#define RING_NO_STAGE 0  // stagers publish without fetching or converting anything
#endif

namespace {

constexpr int kKB = 16;         // points per K-block (the MFMA K dimension)
constexpr int kNKB = 8;         // K-blocks per staging batch
constexpr int kSlots = kKB * kNKB;
constexpr int kThreads = 1024;
constexpr int kRing = 16;       // operand slots: K-block q lives in slot q & 15
constexpr int kStageRing = 4;   // staging buffers: batch b lives in buffer b & 3
constexpr int kOwners = 12;     // plane-owner waves; a K-block reaches 2m + 2 <= 10 of them (three owners on every SIMD, each in
                                // 10 of 12 K-blocks: with 10 owners two SIMDs carry three hits per K-block, two carry two, and the
                                // slower pair sets the pace); waves 12, 13 stage (and build while they wait), waves 14, 15 build
constexpr int kStagers = 2;
constexpr int kMaxSegSlabs = 128;
constexpr int kMaxSweep = kMaxSegSlabs + 2 * kMaxCutoff;
constexpr float kPsiScale = 16.0f;  // as in spread_mfma.hip: psi1 * (x' psi0 * 2^11) <= 2^15 fits f16
#ifndef NFFT_HIP_SPIN_LIMIT
#define NFFT_HIP_SPIN_LIMIT (1 << 22)
#endif
constexpr int kSpinLimit = NFFT_HIP_SPIN_LIMIT;

// operands of ONE K-block
template <int W>
struct __align__(16) RingOps {
    f16x8 bfrag[2][2][64];      // [column tile][hi/lo][lane]
    _Float16 p1[2][2][32][8];   // [hi/lo][point >> 3][row][point & 7]   16 psi1, f16 split
    _Float16 a0[W][2][kKB];     // [axis-0 tap][hi/lo][point]            2^11 x' psi0, f16 split
};

struct __align__(16) RingStage {
    float f0[kSlots], f1[kSlots], f2[kSlots], x[kSlots];
    int c1[kSlots], c2[kSlots];
    int slab[kNKB];
};

constexpr int kRecRing = 8;
constexpr int kXRing = 4;
template <int W>
struct __align__(16) RingLds {
    RingOps<W> ops[kRing];
    RingStage stag[kStageRing];
    f32x4 raw[kRecRing][kSlots];          // landing zones of the LDS-DMA: plan records {p0, p1, p2, index}
    float rawx[kXRing][kSlots];           // ... and of the coefficients
    int raw_idx[kRecRing][kSlots];        // plan entry of the slot (plan-ordered coefficient copy only)
    signed char raw_have[kRecRing][kSlots];
    int raw_slab[kRecRing][kNKB];
    int2 sched[kMaxSweep + 8];            // per slab: {K-blocks before it, point offset}; padded with the totals
    int sched_end[kMaxSweep + 8];
    int ready[kRing];                     // slot s holds K-block q (q & 15 == s) once the high half of ready[s] is (q + 1) & 0xffff;
                                          // the low half is the K-block's slab * 4 + touched column tiles (one word: one look)
    int done[16];                         // owner w is finished with the K-blocks [0, done[w])
    int staged[kStagers];                 // stager wave s has converted its half of the batches [0, staged[s])
    int built[kStageRing];                // K-blocks built so far of the batches = i (mod 4), in units of 64 (all lanes add)
    int next_task;                        // next K-block to build, in units of 64
    int abort;                            // a spin loop ran out: everybody leaves
    int ticket;                           // work-list entry of the workgroup (persistent launch)
    float inv_xscale;
};
static_assert(sizeof(RingLds<10>) <= 160 * 1024, "LDS budget");

// Ordering of the hand-overs.  The LDS unit serves the requests of ONE wave in the order they were issued, so "data, then
// flag" on the producer side and "flag, then data" on the consumer side need no wait -- only the compiler must keep the
// program order (an empty asm with a memory clobber).  A workgroup-scope fence would do as well but also drains vmcnt: an
// owner would wait ~3 000 cycles for its flush atomics at every K-block, a stager for the LDS-DMA it has just issued.
__device__ __forceinline__ void lds_order() { asm volatile("" ::: "memory"); }
__device__ __forceinline__ int lds_load(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_store(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

template <int W, bool OVERFLOW, bool OWNED>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(4, 4)))
spread_ring_kernel(const Geom g, const int *__restrict__ tile_offsets, const float *__restrict__ spos,
                   const float *__restrict__ xr, const float *__restrict__ xs, const int64_t xs_stride,
                   const unsigned *__restrict__ xmax, const int Cr, const int plane0, float *__restrict__ grid,
                   const int seg_slabs, const int nsegm, const int4 *__restrict__ work, const int4 *__restrict__ sorted,
                   const WorkTickets tickets, int *__restrict__ status)
{
    static_assert(W + 2 <= kOwners, "one owner wave per plane of the window, two spare (their planes are being flushed)");
    constexpr int m = W / 2 - 1;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    RingLds<W> &L = *reinterpret_cast<RingLds<W> *>(smem_raw);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, h = lane >> 5;

    const int plane_local = blockIdx.y;
    const int plane = plane0 + plane_local;
    const int b = plane / Cr;
    const int cr = plane - b * Cr;
    const int pencils = g.nta[1] * g.nta[2];

    // ---- work items: as in spread_mfma.hip (balanced plan: one workgroup per range; else the plan's sorted work list)
    const int listed = work[0].z;
    if (OVERFLOW ? !listed : listed) return;
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
    const int nplane = se - sb;
    const int bin0 = b * g.tiles_per_batch + pencil * g.np0;  // one plan bin per slab
    if (nplane <= 0) continue;
    if (!OWNED && tile_offsets[bin0 + sb] == tile_offsets[bin0 + se]) continue;
    const int s_lo = OWNED ? sb - m - 1 : sb;
    const int nslab = OWNED ? nplane + W - 1 : nplane;
    const int tb1 = j1 * g.Ta[1], tb2 = j2 * g.Ta[2];
    float *const gplane = grid + (int64_t)plane_local * g.cells;

    // ---- flags, then the K-block schedule: slab s holds ceil(count / 16) K-blocks; sched[s] = {K-blocks before s, offset}
    if (tid >= 64 && tid < 64 + kRing) {
        L.ready[tid - 64] = 0;
        L.done[tid - 64] = 0;
        if (tid - 64 < kStagers) L.staged[tid - 64] = 0;
        if (tid - 64 < kStageRing) L.built[tid - 64] = 0;
        if (tid == 64) {
            L.next_task = 0;
            L.abort = 0;
        }
    }
    if (wave == 0) {
        int ob[3], oe[3], nk[3];
        int sum = 0;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int k = 3 * lane + q;
            const int sw = wrap(s_lo + min(k, nslab), g.M);
            ob[q] = tile_offsets[bin0 + sw];
            oe[q] = k < nslab ? tile_offsets[bin0 + sw + 1] : ob[q];
            nk[q] = (oe[q] - ob[q] + kKB - 1) / kKB;
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
    }
    // the plane's operand scale (plane_absmax_kernel): see spread_mfma.hip
    float xscale = 1.0f;
    {
        const float mx = __uint_as_float(xmax[plane]);
        if (mx > 1.0e-30f && mx < 3.0e38f) {
            int e;
            frexpf(mx, &e);
            xscale = ldexpf(1.0f, e > 127 ? 127 : e);
        }
    }
    xscale = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(xscale)));
    if (tid == 0) L.inv_xscale = 1.0f / xscale;
    __syncthreads();
    const int total = L.sched[nslab].x;
    const int nbatch = (total + kNKB - 1) / kNKB;
    if (OWNED && total == 0) {
        for (int e = tid; e < nplane * 512; e += kThreads) {
            const int pz = e >> 9, row = (e >> 4) & 31, c4 = e & 15;
            *(f32x4 *)(gplane + ((int64_t)(sb + pz) * g.M + tb1 + row) * g.M + tb2 + 4 * c4) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        continue;
    }
    const float sc = win_exp_scale(m);

    // ---- the three operand tables of K-block q into its ring slot (builder waves; stager waves while they wait): zero fills,
    // then the 2m + 2 taps per axis scattered into the tables (as spread_mfma.hip: build_tasks), then the hand-over
    auto build_kblock = [&](const int q) __attribute__((always_inline)) {
        const int batch = q >> 3, j = q & (kNKB - 1), slot = q & (kRing - 1);
        const RingStage &S = L.stag[batch & (kStageRing - 1)];
        RingOps<W> &O = L.ops[slot];
        // ---- zero fills, then the 2m + 2 taps per axis scattered into the tables (spread_mfma.hip: build_tasks)
        {
            const f16x8 zero = (_Float16)0.0f;
            O.bfrag[0][0][lane] = zero;
            O.bfrag[0][1][lane] = zero;
            O.bfrag[1][0][lane] = zero;
            O.bfrag[1][1][lane] = zero;
            f32x4 *pz = (f32x4 *)&O.p1[0][0][0][0];
            const f32x4 zero4 = 0.0f;
            pz[lane] = zero4;
            pz[lane + 64] = zero4;
        }
        asm volatile("" ::: "memory");
        if (RING_NO_BUILD) {
            lds_order();
            if (lane == 0) lds_store(&L.ready[slot], (int)(((unsigned)(q + 1) << 16) | (unsigned)((L.stag[batch & (kStageRing - 1)].slab[j] * 4 + 3) & 0xffff)));
            atomicAdd(&L.built[batch & (kStageRing - 1)], 1);
            return;
        }
        const int k = lane >> 2, g4 = lane & 3;
        const int pslot = j * kKB + k;
        const int c2v = S.c2[pslot], c1v = S.c1[pslot];
        const float f2v = S.f2[pslot], f1v = S.f1[pslot], f0v = S.f0[pslot], xv = S.x[pslot];
        const int sl = S.slab[j];
        int touched = 0;
        _Float16 *const base_h = (_Float16 *)&O.bfrag[0][0][32 * (k >> 3)] + (k & 7);
#pragma unroll
        for (int t = 0; t < (W + 3) / 4; ++t) {
            const int l = g4 + 4 * t;
            {
                const int col = c2v - m + l;
                const float d = f2v + (float)(m - l);
                const float v = __builtin_amdgcn_exp2f(sc * d * d) * kOpScale;
                unsigned hi, lo;
                split_pair(v, 0.0f, hi, lo);
                if (l < W && (unsigned)col < 64u) {
                    _Float16 *ph = base_h + (col >> 5) * (2 * 64 * 8) + (col & 31) * 8;
                    ph[0] = __builtin_bit_cast(_Float16, (unsigned short)hi);
                    ph[64 * 8] = __builtin_bit_cast(_Float16, (unsigned short)lo);
                    touched |= 1 + (col >> 5);
                }
            }
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
                    O.a0[l][0][k] = __builtin_bit_cast(_Float16, (unsigned short)(hi >> 16));
                    O.a0[l][1][k] = __builtin_bit_cast(_Float16, (unsigned short)(lo >> 16));
                    if ((unsigned)row < 32u) {
                        O.p1[0][k >> 3][row][k & 7] = __builtin_bit_cast(_Float16, (unsigned short)hi);
                        O.p1[1][k >> 3][row][k & 7] = __builtin_bit_cast(_Float16, (unsigned short)lo);
                    }
                }
            }
        }
        const int t0 = __builtin_amdgcn_ballot_w64((touched & 1) != 0) != 0ull;
        const int t1 = __builtin_amdgcn_ballot_w64((touched & 2) != 0) != 0ull;
        lds_order();
        if (lane == 0) lds_store(&L.ready[slot], (int)(((unsigned)(q + 1) << 16) | (unsigned)((sl * 4 + t0 + 2 * t1) & 0xffff)));
        atomicAdd(&L.built[batch & (kStageRing - 1)], 1);  // (all lanes: units of 64)
    };
    // K-block q can be built now: its points are staged and every owner is past the K-block that used its slot
    auto buildable = [&](const int q) __attribute__((always_inline)) -> bool {
        const int staged = lds_load(&L.staged[(q & (kNKB - 1)) >> 2]);
        const int dn = lane < kOwners ? lds_load(&L.done[lane]) : INT_MAX;
        const bool behind = __builtin_amdgcn_ballot_w64(dn < q - (kRing - 1)) != 0ull;
        return staged > (q >> 3) && !behind;
    };

    if (wave < kOwners) {
        // ================================================================ plane owners
        float norm = win_norm(m);
        norm = norm * norm * norm;
        const float unscale = __int_as_float(__builtin_amdgcn_readfirstlane(
            __float_as_int(xscale * norm * (1.0f / (kOpScale * kOpScale * kPsiScale)))));
        f32x16 acc0 = 0.0f, acc1 = 0.0f;
        bool dirty = false;
        const int z_lo = OWNED ? sb : sb - m;  // first plane any owner holds
        int myz = z_lo + (((wave - z_lo) % kOwners) + kOwners) % kOwners;

        auto flush = [&]() __attribute__((always_inline)) {
            if constexpr (OWNED) {
                if (myz >= sb && myz < se) {
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
                // (the tile origin passes through an empty asm: everything derived from it -- the 32 wrapped row offsets of
                // the boundary pencils above all -- is then computed HERE, once per flush, instead of being hoisted out of
                // the K-block loop into ~35 registers that the accumulation has no room for)
                int o1 = tb1 - m, o2 = tb2 - m, M = g.M, hh = h, rr = r32;
                asm volatile("" : "+s"(o1), "+s"(o2), "+s"(M), "+v"(hh), "+v"(rr));
                const int gz = wrap(myz, M);
                const float zscale = ((myz + m) & 1) ? -unscale : unscale;
                if (o1 >= 0 && o1 + 32 <= M) {
                    // the tile's rows do not cross the periodic boundary: one offset per lane and column tile, the rows
                    // are wave-uniform strides from it
                    float *const pbase = gplane + (int64_t)gz * M * M;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int gc = wrap_near(o2 + 32 * t + rr, M);
                        const unsigned off0 = (unsigned)((o1 + 4 * hh) * M + gc);
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) {
                            const unsigned row_off = (unsigned)(((reg & 3) + 8 * (reg >> 2)) * M);
                            if (!RING_NO_ATOMICS) atomicAdd(pbase + (off0 + row_off), (t == 0 ? acc0[reg] : acc1[reg]) * zscale);
                            else asm volatile("" :: "v"((t == 0 ? acc0[reg] : acc1[reg]) * zscale), "v"(off0 + row_off));
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

        // The owner's K-block is a chain LDS read (the f16 splits of psi1 for the lane's row and 8 points, and of x' psi0 for
        // the tap that lands on the plane: `raw`) -> packed arithmetic -> LDS read (B fragments) -> three dependent MFMAs per
        // touched column tile.  The requests for the NEXT hit are issued as soon as the current one has given up the registers:
        // the raw inputs right after the packed arithmetic (their latency passes under the MFMAs), the B fragments right after
        // the MFMAs (under the next packed arithmetic).  The compiler's s_waitcnt placement cannot express that (at the merge
        // points of this loop it waits for lgkmcnt(0), i.e. for the requests just issued), so these eight loads and their waits
        // are asm statements; the LDS unit returns a wave's requests in order, hence a wait for "all but the N newest".  Loads
        // update their destination in place ("+v") at ONE site each: no copy of a register whose data is still on its way.
        // Everything else in the loop (flags, progress) stays compiler-visible LDS traffic: extra requests in the queue only
        // make a counted wait conservative.
        u32x4 rph = 0u, rpl = 0u, rxh = 0u, rxl = 0u;
        f16x8 b0h = (_Float16)0.0f, b0l = (_Float16)0.0f, b1h = (_Float16)0.0f, b1l = (_Float16)0.0f;
        const unsigned ops_base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void *)&L.ops[0];
        constexpr unsigned kOffB = (unsigned)offsetof(RingOps<W>, bfrag), kOffP = (unsigned)offsetof(RingOps<W>, p1),
                           kOffA = (unsigned)offsetof(RingOps<W>, a0);
        const unsigned lane16 = (unsigned)lane * 16u, h16 = (unsigned)h * 16u;
        auto mma = [&](f32x16 &acc, const f16x8 ah, const f16x8 al, const f16x8 fh, const f16x8 fl) __attribute__((always_inline)) {
            if constexpr (OWNED) {
                // transposed product (columns x rows): the fragments of the two operands have the same lane layout
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, ah, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl, ah, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, al, acc, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, fh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, fl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, fh, acc, 0, 0, 0);
            }
        };
        // The 16 ready words travel with the operand requests: every pass of the loop ends with a request for them (lane l
        // looks at slot l & 15) and the next pass starts from what came back -- which K-blocks are there in a row from qn on,
        // and their (slab, tiles) halves -- so the owner never pays a flag round trip of its own while the builders keep up.
        int qn = 0, avail = 0, flg = 0;
        const unsigned flag_addr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void *)&L.ready[lane & (kRing - 1)];
        auto extend = [&]() __attribute__((always_inline)) {
            // lane l < 16 holds slot l, where K-block qn + ((l - qn) & 15) would be
            const int ql = qn + ((lane - qn) & (kRing - 1));
            unsigned okm = (unsigned)__builtin_amdgcn_ballot_w64(lane < kRing && ((unsigned)flg >> 16) == ((unsigned)(ql + 1) & 0xffffu));
            const int rot = qn & (kRing - 1);
            okm = ((okm >> rot) | (okm << (kRing - rot))) & 0xffffu;  // bit k: K-block qn + k is there
            avail = max(avail, min(qn + (int)__builtin_ctz(~okm), total));
        };

        bool bail = false, cur = false;  // cur: a hit is in the pipeline (its raw inputs and B fragments requested)
        int cur_hv = 0, published = 0;
        RING_T0();
        RING_PH_DECL();
        while (true) {
            // (1) the current hit's A fragment
            f16x8 ah, al;
            RING_PH(7);
            if (cur) {
                RING_WORK();
                // (5 newer requests: the ready words and its B fragments)
                if (!RING_NO_LOADS) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(rph), "+v"(rpl), "+v"(rxh), "+v"(rxl));
                u32x4 uh, ul;
                if (!RING_NO_SPLIT) split_product_f16x4(rph, rpl, rxh, rxl, uh, ul);
                else { uh = rph ^ rxh; ul = rpl ^ rxl; }
                ah = __builtin_bit_cast(f16x8, uh);
                al = __builtin_bit_cast(f16x8, ul);
            }
            RING_PH(0);
            // (2) the ready words requested at the end of the previous pass (4 newer requests -- the B fragments -- if a hit is
            // in the pipeline, else none), then the next hit among the K-blocks known to be there; stops at one that needs a
            // flush first
            asm volatile("s_cmp_eq_u32 %1, 0\n\ts_cbranch_scc1 1f\n\ts_waitcnt lgkmcnt(4)\n\ts_branch 2f\n"
                         "1:\n\ts_waitcnt lgkmcnt(0)\n2:"
                         : "+v"(flg)
                         : "s"(__builtin_amdgcn_readfirstlane((int)cur))
                         : "scc");
            extend();
            RING_PH(1);
            bool nxt = false;
            int n_hv = 0;
            unsigned vA = 0, vB = 0;
            while (qn < avail) {
                const int pk = __builtin_amdgcn_readlane(flg, qn & (kRing - 1)) & 0xffff;
                const int s = pk >> 2;
                if (myz + m < s) break;  // my plane is complete: flush before going on (below, once the pipeline is empty)
                const int l0 = myz - s + m;
                if ((unsigned)l0 < (unsigned)W && (!OWNED || myz < se)) {
                    nxt = true;
                    n_hv = pk & 3;
                    const unsigned sb_ = ops_base + (unsigned)(qn & (kRing - 1)) * (unsigned)sizeof(RingOps<W>);
                    vA = sb_ + lane16;
                    vB = sb_ + (unsigned)l0 * 64u + h16;
                    ++qn;
                    break;
                }
                ++qn;  // my plane is outside this K-block's window
            }
            RING_PH(2);
            if (nxt && !RING_NO_LOADS) {
                asm volatile("ds_read_b128 %0, %4 offset:%6\n\tds_read_b128 %1, %4 offset:%7\n\t"
                             "ds_read_b128 %2, %5 offset:%8\n\tds_read_b128 %3, %5 offset:%9"
                             : "+v"(rph), "+v"(rpl), "+v"(rxh), "+v"(rxl)
                             : "v"(vA), "v"(vB), "i"(kOffP), "i"(kOffP + 1024u), "i"(kOffA), "i"(kOffA + 32u));
            }
            RING_PH(3);
            // (3) the current hit's MFMAs
            if (cur) {
                // all but the 4 newest requests (the next hit's raw inputs) if there are such, else all -- ONE statement: two
                // would be two definitions of the fragments, merged by register copies
                if (!RING_NO_LOADS) asm volatile("s_cmp_eq_u32 %4, 0\n\ts_cbranch_scc1 1f\n\ts_waitcnt lgkmcnt(4)\n\ts_branch 2f\n"
                             "1:\n\ts_waitcnt lgkmcnt(0)\n2:"
                             : "+v"(b0h), "+v"(b0l), "+v"(b1h), "+v"(b1l)
                             : "s"(__builtin_amdgcn_readfirstlane((int)nxt))
                             : "scc");
                // (two plain ifs, one chain of three MFMAs per touched tile: with a three-way branch that alternates the two
                // chains the compiler moves the accumulator tiles between register sets, 16 copies per K-block)
                if (!RING_NO_MFMA) {
                    if (cur_hv & 1) mma(acc0, ah, al, b0h, b0l);
                    if (cur_hv & 2) mma(acc1, ah, al, b1h, b1l);
                } else {
                    asm volatile("" :: "v"(ah), "v"(al), "v"(b0h), "v"(b0l), "v"(b1h), "v"(b1l));
                }
                dirty = true;
            }
            RING_PH(4);
            // the ready words for the next pass, then the next hit's B fragments
            asm volatile("ds_read_b32 %0, %1" : "+v"(flg) : "v"(flag_addr));
            if (nxt && !RING_NO_LOADS) {
                // both column tiles, touched or not: a fixed number of requests keeps the counted waits simple
                asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\t"
                             "ds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8"
                             : "+v"(b0h), "+v"(b0l), "+v"(b1h), "+v"(b1l)
                             : "v"(vA), "i"(kOffB), "i"(kOffB + 1024u), "i"(kOffB + 2048u), "i"(kOffB + 3072u));
            }
            // (4) progress: this owner is finished with every K-block in front of the one in the pipeline
            {
                const int fin = nxt ? qn - 1 : qn;
                if (fin != published) {
                    published = fin;
                    lds_order();
                    if (lane == 0) lds_store(&L.done[wave], fin);
                }
            }
            cur = nxt;
            cur_hv = n_hv;
            RING_PH(5);
            if (!cur) {
                // pipeline empty: the place to wait, for the builders or for this wave's flush
                if (qn >= total) break;
                if (qn >= avail) {
                    RING_WAIT_BEGIN();
                    int spins = 0;
                    while (true) {
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(flg));
                        extend();
                        if (qn < avail) break;
                        if (lds_load(&L.abort) || ++spins > kSpinLimit) {
                            bail = true;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                        asm volatile("ds_read_b32 %0, %1" : "+v"(flg) : "v"(flag_addr));
                    }
                    if (bail) break;
                    // (the loop top expects one request for the ready words in flight)
                    asm volatile("ds_read_b32 %0, %1" : "+v"(flg) : "v"(flag_addr));
                    RING_WAIT_END();
                } else {
                    RING_WAIT_BEGIN();
                    const int s = (__builtin_amdgcn_readlane(flg, qn & (kRing - 1)) & 0xffff) >> 2;
                    while (myz + m < s) {
                        flush();
                        myz += kOwners;
                    }
                    RING_WAIT2_END();
                }
                RING_PH(6);
            }
        }
        RING_PH_OUT();
        RING_T1();
        if (bail) {
            lds_store(&L.abort, 1);
            if (lane == 0) report_fault(status, kFaultStreamStall);
        }
        flush();
        if constexpr (OWNED) {
            for (myz += kOwners; myz < se; myz += kOwners) flush();
        }
    } else if (wave < kOwners + kStagers) {
        // ================================================================ stagers
        // Thread st of the two waves stages slot st of every batch (K-block st / 16, point st % 16): while batch i is being
        // built the records of batch i + 6 are requested (LDS-DMA into raw[(i + 6) & 7]), the coefficients of batch i + 4
        // (their address comes out of the landed record), and batch i + 2 is converted.  Every step issues exactly two DMA
        // instructions per wave, so the consumer waits with a count.
        const int sw = wave - kOwners;
        const int st = tid - kOwners * 64;
        const float *const xcol = xs + (int64_t)cr * xs_stride;
        int cur = 0;
        auto locate = [&](const int batch, int &idx, int &have, int &slab) {
            const int j = st / kKB, i = st - j * kKB;
            const int q = batch * kNKB + j;
            have = 0;
            slab = INT_MAX;
            idx = 0;
            if (q < total) {
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
                if (!have) idx = e0.y;
            }
        };
        auto request_records = [&](const int batch) {
            int idx, have, slab;
            locate(batch, idx, have, slab);
            const int j = st / kKB, i = st - j * kKB;
            const int buf = batch & (kRecRing - 1);
            lds_dma_dwordx4(spos + (int64_t)idx * 4, &L.raw[buf][sw * 64]);
            L.raw_have[buf][st] = (signed char)have;
            if (i == 0) L.raw_slab[buf][j] = slab;
            if (!xr) L.raw_idx[buf][st] = idx;
        };
        auto request_coefficients = [&](const int batch) {
            const float *src;
            if (xr) {
                const int orig = __float_as_int(L.raw[batch & (kRecRing - 1)][st].w);
                src = xr + (int64_t)orig * Cr + cr;
            } else {
                src = xcol + L.raw_idx[batch & (kRecRing - 1)][st];
            }
            lds_dma_dword(src, &L.rawx[batch & (kXRing - 1)][sw * 64]);
        };
        auto stage_convert = [&](RingStage &S, const int batch, const bool newest_in_flight) {
            const int buf = batch & (kRecRing - 1);
            if (newest_in_flight) wait_lds_dma_but_newest(); else wait_lds_dma();
            float f0 = 0.f, f1 = 0.f, f2 = 0.f, xv = 0.f;
            int c1 = -1000, c2 = -1000;
            if (L.raw_have[buf][st]) {
                int c0;
                const f32x4 rec = L.raw[buf][st];
                split_cell(rec.x, g.M, c0, f0);
                split_cell(rec.y, g.M, c1, f1);
                split_cell(rec.z, g.M, c2, f2);
                if constexpr (OWNED) {
                    c1 -= tb1;
                    c2 -= tb2;
                    c1 = c1 >= 32 + m ? c1 - g.M : (c1 < -(m + 1) ? c1 + g.M : c1);
                    c2 = c2 >= 64 + m ? c2 - g.M : (c2 < -(m + 1) ? c2 + g.M : c2);
                } else {
                    c1 -= tb1 - m;
                    c2 -= tb2 - m;
                }
                xv = L.rawx[batch & (kXRing - 1)][st] * L.inv_xscale;
                if (fabsf(xv) > 1.0f && fabsf(xv) < __builtin_huge_valf()) {
                    report_fault(status, kFaultBatchOrder);
                    xv = fminf(fmaxf(xv, -1.0f), 1.0f);
                }
            }
            S.f0[st] = f0; S.f1[st] = f1; S.f2[st] = f2; S.x[st] = xv;
            S.c1[st] = c1; S.c2[st] = c2;
            if ((st & (kKB - 1)) == 0) S.slab[st / kKB] = L.raw_slab[buf][st / kKB];
        };
        auto publish = [&](const int batches) {
            lds_order();
            if (lane == 0) lds_store(&L.staged[sw], batches);
        };

        if (!RING_NO_STAGE) {
            for (int q = 0; q < 5; ++q) request_records(q);
            wait_lds_dma();
            for (int q = 0; q < 3; ++q) request_coefficients(q);
            wait_lds_dma();
            stage_convert(L.stag[0], 0, false);
        }
        publish(1);
        bool bail = false;
        RING_T0();
        for (int i = -1; i < nbatch && !bail; ++i) {
            const int c = i + 2;
            if (c < nbatch) {
                RING_WORK();
                if (c >= kStageRing) {
                    RING_WAIT_BEGIN();
                    // the buffer's previous batch, c - 4, must be built: `built` of its residue class counts 8 K-blocks for
                    // every batch up to it (only the last batch of an item is short, and nothing follows it)
                    const int want = 64 * kNKB * (c / kStageRing);
                    int spins = 0;
                    while (lds_load(&L.built[c & (kStageRing - 1)]) < want) {
                        // a stager has nothing to do until then: it builds K-blocks as well -- the next one of the queue,
                        // claimed with a compare-and-swap once it is known to be buildable (a stager must not sit on a claim
                        // that waits for its own staging).  All lanes swap: the first one's result is the wave's.
                        const int nt = lds_load(&L.next_task);
                        if ((nt >> 6) < total && buildable(nt >> 6) &&
                            __builtin_amdgcn_readfirstlane(atomicCAS(&L.next_task, nt, nt + 64)) == nt) {
                            lds_order();
                            build_kblock(nt >> 6);
                            spins = 0;
                            continue;
                        }
                        if (lds_load(&L.abort) || ++spins > kSpinLimit) {
                            bail = true;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    lds_order();
                    RING_WAIT_END();
                }
                if (!bail) {
                    if (!RING_NO_STAGE) stage_convert(L.stag[c & (kStageRing - 1)], c, true);
                    publish(c + 1);
                }
            }
            if (!RING_NO_STAGE) {
                request_coefficients(i + 4);
                request_records(i + 6);
            }
        }
        RING_T1();
        wait_lds_dma();  // the dummy requests of the last steps must have landed before the workgroup gives its LDS back
        if (bail) {
            lds_store(&L.abort, 1);
            if (lane == 0) report_fault(status, kFaultStreamStall);
        }
    } else {
        // ================================================================ builders
        bool bail = false;
        RING_T0();
        while (true) {
            // (all 64 lanes add 1: the counter runs in units of 64, lane 0 sees the wave's base value)
            const int q = __builtin_amdgcn_readfirstlane(atomicAdd(&L.next_task, 1)) >> 6;
            if (q >= total) break;
            {
                RING_WAIT_BEGIN();
                int spins = 0;
                while (true) {
                    if (buildable(q)) break;
                    if (lds_load(&L.abort) || ++spins > kSpinLimit) {
                        bail = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                if (bail) break;
                lds_order();
                RING_WAIT_END();
                RING_WORK();
            }
            build_kblock(q);
        }
        RING_T1();
        if (bail) {
            lds_store(&L.abort, 1);
            if (lane == 0) report_fault(status, kFaultStreamStall);
        }
    }
    }  // work items
}

} // namespace

#ifdef NFFT_HIP_TRACE
extern "C" int nfft_dbg_set_ring_phase(void *device_buffer)  // 16 workgroups x 16 waves x 8 words
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ring_phase), &device_buffer, sizeof(device_buffer));
}
extern "C" int nfft_dbg_set_ring_trace(void *device_buffer)  // 16 workgroups x 16 waves x 4 words
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ring_trace), &device_buffer, sizeof(device_buffer));
}
#endif

bool spread_ring_supported(const Geom &g) { return g.dim == 3 && g.wide && 2 * g.m + 2 + 2 <= kOwners; }

template <int W, bool OWNED>
static int launch_ring_t(const Geom &g, const PlanLayout &L, const void *plan, const int *to, const float *spos,
                         const float *xr, const float *xs, const unsigned *xmax, int64_t n, int64_t Cr, int64_t plane0,
                         int64_t nplanes, float *grid, hipStream_t stream)
{
    // work decomposition: exactly that of spread_mfma.hip (the plan's load-balance tables are built for it)
    const int64_t pencils = (int64_t)g.nta[1] * g.nta[2];
    int64_t nsets = g.tiles_per_batch > 0 ? L.ntiles / g.tiles_per_batch : 1;
    if (nsets < 1) nsets = 1;
    const int nsegm = seg_base_runs(n, nsets, pencils, g.M, device_cu_count());
    const int seg_slabs = (g.M + nsegm - 1) / nsegm;
    const dim3 blocks((unsigned)(pencils * nsegm), (unsigned)nplanes);
    static DeviceOnce attr_done;
    if (attr_done.first_use()) {
        NFFT_HIP_CHECK(hipFuncSetAttribute((const void *)spread_ring_kernel<W, false, OWNED>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RingLds<W>)));
        attr_done.mark();
    }
    const char *base = (const char *)plan;
    const int4 *work = (const int4 *)(base + L.off_work), *sorted = work + L.work_head + L.work_cap;
    int *const status = device_status_block();
    hipLaunchKernelGGL((spread_ring_kernel<W, false, OWNED>), blocks, dim3(kThreads), sizeof(RingLds<W>), stream, g, to,
                       spos, xr, xs, L.cap, xmax, (int)Cr, (int)plane0, grid, seg_slabs, nsegm, work, sorted, WorkTickets{nullptr, 0u}, status);
    NFFT_HIP_CHECK(hipGetLastError());
    return 0;
}

template <bool OWNED>
static int launch_ring_w(const Geom &g, const PlanLayout &L, const void *plan, const float *xr, const float *xs,
                         const unsigned *xmax, int64_t n, int64_t Cr, int64_t plane0, int64_t nplanes, float *grid,
                         hipStream_t stream)
{
    const char *base = (const char *)plan;
    const int *to = (const int *)(base + L.off_offsets);
    const float *spos = (const float *)(base + L.off_spos);
    switch (g.m) {
    case 1: return launch_ring_t<4, OWNED>(g, L, plan, to, spos, xr, xs, xmax, n, Cr, plane0, nplanes, grid, stream);
    case 2: return launch_ring_t<6, OWNED>(g, L, plan, to, spos, xr, xs, xmax, n, Cr, plane0, nplanes, grid, stream);
    case 3: return launch_ring_t<8, OWNED>(g, L, plan, to, spos, xr, xs, xmax, n, Cr, plane0, nplanes, grid, stream);
    case 4: return launch_ring_t<10, OWNED>(g, L, plan, to, spos, xr, xs, xmax, n, Cr, plane0, nplanes, grid, stream);
    }
    set_error("flag-driven matrix-core spreading supports cutoff 1..4");
    return 1;
}

// The balanced-plan launch of launch_spread_mfma (one workgroup per range of slabs; it returns at once when the plan says
// "walk the work list": that persistent launch stays with spread_mfma.hip, whose item loop this kernel does not have yet).
int launch_spread_ring(const Geom &g, const PlanLayout &L, const void *plan, const float *xr, const float *xs,
                       const unsigned *xmax, int64_t n, int64_t Cr, int64_t plane0, int64_t nplanes, float *grid,
                       hipStream_t stream)
{
    return g.owned ? launch_ring_w<true>(g, L, plan, xr, xs, xmax, n, Cr, plane0, nplanes, grid, stream)
                   : launch_ring_w<false>(g, L, plan, xr, xs, xmax, n, Cr, plane0, nplanes, grid, stream);
}

} // namespace nfft
