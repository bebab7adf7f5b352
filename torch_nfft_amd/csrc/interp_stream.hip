// Interpolation (forward gather) on the matrix cores, streamed: producer waves feed a ring of grid planes,
// consumer waves take blocks of points from a queue -- no workgroup barrier inside a work item.
//
// Same arithmetic as interp_mfma.hip (reference: csrc/cuda/spatial_window_operations.cu:214-332): per plane z of a
// pencil and block of 32 points  T_z = G_z Psi2  (MFMAs on f16-split operands), t = sum_u1 psi1[u1] T_z[u1, i],
// y_i += psi0_i[z] t.  What changes is the schedule.  interp_mfma.hip advances in lock step: stage the planes of a
// chunk (all waves wait for the loads), barrier, every wave takes one block, barrier ... -- measured at config C3:
// staging alone 0.44 ms, compute alone 1.30 ms, nothing overlaps (profiles/r02_experiments.md).  Here
//   * 4 producer waves stage planes continuously: wave p takes the ring slots s = p (mod 4) of the item's sweep,
//     loads the padded 32 x 64 tile (the next plane's loads are in flight while the current one is converted),
//     scales it by its own power of two, f16-splits it into the ring slot z & 15 and publishes ready[slot] = z;
//   * 12 consumer waves pull blocks from an LDS counter.  A block is 32 points of ONE chunk (17 - (2m+2) slabs, so its
//     window is at most the 16 planes of the ring) and, when the plan is ordered by column group (common.h), of ONE
//     group: its windows then lie inside two of the tile's four 16-column k-steps, which halves the MFMAs, the B
//     fragments and the A-fragment reads (measured: 1.43 -> 1.05 ms at C3 for half the k-steps).  The points of a
//     (chunk, group) are the group's runs of the chunk's slabs, concatenated; blocks are handed out in the order of the
//     run their first point lies in, i.e. by first slab, whatever the group -- so a wave's first plane never moves
//     backwards.  A wave publishes that plane (progress[wave]), waits until all planes of the block are staged (one
//     poll reads all 16 flags), builds its B fragments / psi1 weights and walks the planes without further checks; the
//     A fragments of the next plane are requested as soon as the MFMAs that read a buffer are issued;
//   * a wave claims its next block and fetches that block's points while it works on the current one;
//   * a producer may overwrite slot z & 15 once every consumer's progress is beyond z - 16.  The slowest consumer
//     needs planes below progress + 16 only, so the producers can always serve it: no cycle of waits.
// Every spin loop is bounded (kSpinLimit): a logic error never hangs the GPU; the wave that runs out raises the
// device's fault flag (common.h report_fault), everybody leaves the item, and the next entry point of the C ABI
// returns NFFT_HIP_EKERNEL instead of handing out the unfinished rows as a result.
#include <algorithm>
#include <climits>
#include <cstdlib>

#include "common.h"
#include "kernels.h"
#include "mfma_split.h"

namespace nfft {

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kIsThreads = 1024;
constexpr int kIsWaves = kIsThreads / 64;
constexpr int kIsProducers = 4;      // measured at C3: 3 producers 1.50 ms, 4: 1.41, 5: 1.50 (each owns 16 / 4 ring slots)
constexpr int kIsConsumers = kIsWaves - kIsProducers;
constexpr int kIsRing = 16;          // resident planes: TC + 2m+1 = 16 for every cutoff of the wide tiling
constexpr int kIsMaxSlabs = 160;     // slabs the chunks of one work item cover (<= 128 + 2 TC)
constexpr int kIsMaxRuns = 3 * kIsMaxSlabs;
#ifndef NFFT_HIP_SPIN_LIMIT
#define NFFT_HIP_SPIN_LIMIT (1 << 22)
#endif
constexpr int kSpinLimit = NFFT_HIP_SPIN_LIMIT;  // (the fault-report test builds a variant library with a limit of 0)

struct __align__(16) StreamLds {
    f16x8 frag[kIsRing][4][2][64];   // [plane slot][k-step][hi/lo][lane = 32 (column half) + row]   128 KB
    float pinv[kIsRing];             // what one unit of the scaled plane is worth, times the B operand scale
    int ready[kIsRing];              // plane number held by the slot (published after the fragments)
    int progress[kIsConsumers];      // first plane a consumer still needs
    int next_block;                  // block queue (counts in units of 64: every lane adds 1)
    int abort;                       // set when a spin loop ran out: everybody leaves
    int ticket;                      // work-list entry of the workgroup (persistent launch: next_work_item)
    // runs of the item: run e = (slab - first slab) * NG + group
    int run_start[kIsMaxRuns + 4];   // first point of run e; [runs] = end of the last one
    int run_cum[kIsMaxRuns + 4];     // points of the same chunk and group in front of run e
    int run_blk[kIsMaxRuns + 4];     // blocks whose first point lies in a run before e (prefix sums); [runs] = all blocks
};

// Ordering of the LDS hand-overs between producer and consumer waves.  The LDS unit serves the requests of ONE wave in
// the order they were issued: "data, then flag" on the producer side and "flag, then data" on the consumer side need no
// wait, only the compiler must keep the program order.  The workgroup-scope fences that stood here until round 4 compile
// to s_waitcnt vmcnt(0) as well: a consumer then waited at every block for the acknowledgement of its scattered result
// stores and for the point records of the NEXT block it had just requested (profiles/r04_experiments.md).
#ifndef NFFT_STREAM_LDS_ORDER
#define NFFT_STREAM_LDS_ORDER 1
#endif
__device__ __forceinline__ void lds_release()
{
    if (NFFT_STREAM_LDS_ORDER) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
}
__device__ __forceinline__ void lds_acquire()
{
    if (NFFT_STREAM_LDS_ORDER) asm volatile("" ::: "memory");
    else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ int lds_load(const int *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_store(int *p, int v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// NG: column groups of the plan (3, or 1 for plans without the group order: every block then spans all four k-steps)
template <int W, bool OVERFLOW, int NG>
__global__ void __launch_bounds__(kIsThreads) __attribute__((amdgpu_waves_per_eu(4, 4)))
interp_stream_kernel(const Geom g, const int *__restrict__ tile_offsets, const int *__restrict__ group_starts,
                     const float *__restrict__ spos, const float *__restrict__ grid,
                     const int Cr, const int plane0, float *__restrict__ yr, const int seg_slabs, const int nsegm,
                     const int4 *__restrict__ work, const int4 *__restrict__ sorted, const WorkTickets tickets, int *__restrict__ status)
{
    constexpr int m = W / 2 - 1;
    constexpr int TC = 17 - W;                                   // slabs per chunk
    constexpr int SPAN = TC + W - 1;                             // planes a chunk's blocks may touch
    constexpr int NKS = NG == 3 ? 2 : 4;  // k-steps of a block
    static_assert(TC >= 1 && SPAN <= kIsRing, "the ring holds a chunk's planes");
    static_assert(128 + 2 * TC <= kIsMaxSlabs, "run tables");
    extern __shared__ __align__(16) unsigned char smem_raw[];
    StreamLds &L = *reinterpret_cast<StreamLds *>(smem_raw);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, h = lane >> 5;

    const int plane_local = blockIdx.y;
    const int plane = plane0 + plane_local;
    const int b = plane / Cr;
    const int cr = plane - b * Cr;
    const int pencils = g.nta[1] * g.nta[2];
    const int M = g.M;

    // (work items as in spread_mfma.hip: one workgroup per range, or a persistent grid over the plan's work list)
    const int listed = work[0].z;
    if (OVERFLOW ? !listed : listed) return;
    // (a plane walks its own point set's part of the sorted list: set_hdr[b] = {entries, first entry})
    const int2 set_hdr = OVERFLOW ? ((const int2 *)(work + 1))[b] : make_int2(1, 0);
    const int n_items = set_hdr.x;
    const int4 *const entries = sorted + set_hdr.y;
    for (int item = OVERFLOW ? next_work_item(tickets, &L.ticket, -1, plane_local) : 0; item < n_items;
         item = OVERFLOW ? next_work_item(tickets, &L.ticket, item, plane_local) : 1) {
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
    // an item owns the chunks whose first slab lies in its range (as in interp_mfma.hip)
    const int k_begin = (sb + TC - 1) / TC;
    const int k_end = min((M + TC - 1) / TC, (se + TC - 1) / TC);
    if (k_begin >= k_end) continue;
    const int bin0 = b * g.tiles_per_batch + pencil * g.np0;
    const int s0 = k_begin * TC;                    // first slab of the item's chunks
    const int nsl = min(k_end * TC, M) - s0;        // slabs they cover (<= kIsMaxSlabs: a range holds <= 128 slabs)
    const int nruns = nsl * NG;
    if (tile_offsets[bin0 + s0] == tile_offsets[bin0 + s0 + nsl]) continue;  // no points in these chunks
    const int j2 = pencil % g.nta[2], j1 = pencil / g.nta[2];
    const int tb1 = j1 * g.Ta[1], tb2 = j2 * g.Ta[2];
    const float sc = win_exp_scale(m);
    float norm = win_norm(m);
    norm = norm * norm * norm;
    const float *const gplane = grid + (int64_t)plane_local * g.cells;

    // ---- item set-up: run tables, flags -----------------------------------------------------------------------
    __syncthreads();  // the previous item is done with the LDS
    for (int e = tid; e <= nruns; e += kIsThreads) {
        const int s = e / NG, q = e - s * NG;
        const int bin = bin0 + s0 + s;
        L.run_start[e] = q == 0 ? tile_offsets[bin] : group_starts[2 * (int64_t)bin + q - 1];
    }
    if (wave == 0) {
        if (lane < kIsRing) L.ready[lane] = INT_MIN;
        if (lane < kIsConsumers) L.progress[lane] = s0 - m;
        if (lane == 0) { L.next_block = 0; L.abort = 0; }
    }
    __syncthreads();
    for (int e = tid; e < nruns; e += kIsThreads) {
        // blocks of a (chunk, group) are cut from the concatenation of the group's runs over the chunk's slabs
        const int s = e / NG;
        const int t = s % TC;
        int cum = 0;
        for (int k = 1; k <= t; ++k) cum += L.run_start[e - k * NG + 1] - L.run_start[e - k * NG];
        const int len = L.run_start[e + 1] - L.run_start[e];
        L.run_cum[e] = cum;
        L.run_blk[e] = ((cum + len + 31) >> 5) - ((cum + 31) >> 5);  // blocks whose first point lies in this run
    }
    __syncthreads();
    if (wave == 0) {
        int carry = 0;
        for (int base = 0; base <= nruns; base += 64) {
            const int e = base + lane;
            const int nb = e < nruns ? L.run_blk[e] : 0;
            int incl = nb;
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(incl, off);
                if (lane >= off) incl += t;
            }
            if (e <= nruns) L.run_blk[e] = carry + incl - nb;
            carry += __shfl(incl, 63);
        }
    }
    __syncthreads();
    const int total_blocks = L.run_blk[nruns];

    if (wave >= kIsConsumers) {
        // ================================ producer: planes z = z_begin + p, + 4, ... ================================
        const int p = wave - kIsConsumers;
        const int z_begin = k_begin * TC - m, z_end = (k_end - 1) * TC + SPAN - m;  // planes any chunk of the item needs
        // a lane's 4 tasks of a plane: (row, group of 8 columns); 16 consecutive lanes = 16 consecutive rows of one
        // column group -> consecutive 16-byte LDS slots on the way out, half rows of 128 contiguous bytes on the way in
        auto needed = [&](const int z) {
            // plane z is used by chunk k iff k TC - m <= z <= k TC + TC + m: at most three candidates
            const int k_hi = min(k_end - 1, (z + m) / TC), k_lo = max(k_begin, (z - TC - m + TC - 1) / TC);
            for (int k = k_lo; k <= k_hi; ++k) {
                if (k < k_begin) continue;
                const int lo = (k - k_begin) * TC, hi = min(lo + TC, nsl);
                if (L.run_start[hi * NG] > L.run_start[lo * NG]) return true;
            }
            return false;
        };
        auto load_plane = [&](const int z, f32x4 (&v)[8]) {
            const int64_t gz = wrap(z, M);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = lane + 64 * i;
                const int cg = ((t >> 4) & 3) + 4 * ((t >> 7) & 1), row = (t & 15) + 16 * ((t >> 6) & 1);
                const int64_t g1 = wrap_near(tb1 - m + row, M);
                const float *const grow = gplane + (gz * M + g1) * M;
                const int c0 = tb2 - m + 8 * cg;
                if (c0 >= 0 && c0 + 8 <= M) {
                    v[2 * i] = *(const f32x4_dw *)(grow + c0);
                    v[2 * i + 1] = *(const f32x4_dw *)(grow + c0 + 4);
                } else {
                    f32x4 a, c;
                    a.x = grow[wrap_near(c0 + 0, M)]; a.y = grow[wrap_near(c0 + 1, M)];
                    a.z = grow[wrap_near(c0 + 2, M)]; a.w = grow[wrap_near(c0 + 3, M)];
                    c.x = grow[wrap_near(c0 + 4, M)]; c.y = grow[wrap_near(c0 + 5, M)];
                    c.z = grow[wrap_near(c0 + 6, M)]; c.w = grow[wrap_near(c0 + 7, M)];
                    v[2 * i] = a;
                    v[2 * i + 1] = c;
                }
            }
        };
        // Next plane >= z of this producer's sequence that some chunk needs.  A ring slot belongs to ONE producer
        // (slot mod kIsProducers): planes z and z + 16 share a slot, and only one wave staging them in order keeps a
        // late plane z from overwriting (and un-publishing) plane z + 16.
        auto next_needed = [&](int z) {
            while (z < z_end && (((z & (kIsRing - 1)) % kIsProducers) != p || !needed(z))) ++z;
            return z;
        };
        int z = next_needed(z_begin);
        f32x4 cur[8], nxt[8];
        if (z < z_end) load_plane(z, cur);
        while (z < z_end) {
            const int zn = next_needed(z + 1);
            if (zn < z_end) load_plane(zn, nxt);
            // power-of-two scale of the tile; odd planes are stored negated (un-negated through pinv): the MFMA
            // accumulation truncates with a small sign-independent bias that cancels over alternating planes
            float mx = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                mx = fmaxf(fmaxf(fmaxf(mx, fabsf(cur[e].x)), fabsf(cur[e].y)), fmaxf(fabsf(cur[e].z), fabsf(cur[e].w)));
            for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
            float scale = 1.0f, inv = 1.0f;
            if (mx > 1.0e-30f && mx < 3.0e38f) {
                int ex;
                frexpf(mx, &ex);
                scale = ldexpf(1.0f, 11 - ex);
                inv = ldexpf(1.0f, ex - 11);
            }
            const int slot = z & (kIsRing - 1);
            if (slot & 1) { scale = -scale; inv = -inv; }
            // the slot's previous plane, z - 16, must be behind every consumer
            int spins = 0;
            while (true) {
                int lo = lane < kIsConsumers ? lds_load(&L.progress[lane]) : INT_MAX;
                for (int off = 32; off >= 1; off >>= 1) lo = min(lo, __shfl_xor(lo, off));
                if (lo > z - kIsRing || lds_load(&L.abort)) break;
                if (++spins > kSpinLimit) {
                    lds_store(&L.abort, 1);
                    if (lane == 0) report_fault(status, kFaultStreamStall);
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            lds_acquire();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = lane + 64 * i;
                const int cg = ((t >> 4) & 3) + 4 * ((t >> 7) & 1), row = (t & 15) + 16 * ((t >> 6) & 1);
                const f32x4 a = cur[2 * i], c = cur[2 * i + 1];
                unsigned h0, h1, h2, h3, q0, q1, q2, q3;
                split_pair(a.x * scale, a.y * scale, h0, q0);
                split_pair(a.z * scale, a.w * scale, h1, q1);
                split_pair(c.x * scale, c.y * scale, h2, q2);
                split_pair(c.z * scale, c.w * scale, h3, q3);
                const int ln = 32 * (cg & 1) + row;
                L.frag[slot][cg >> 1][0][ln] = __builtin_bit_cast(f16x8, u32x4{h0, h1, h2, h3});
                L.frag[slot][cg >> 1][1][ln] = __builtin_bit_cast(f16x8, u32x4{q0, q1, q2, q3});
            }
            if (lane == 0) L.pinv[slot] = inv * (1.0f / kOpScale);
            lds_release();
            if (lane == 0) lds_store(&L.ready[slot], z);
            if (lds_load(&L.abort)) break;
            // (every consumer gone: nobody reads what is left of the sweep)
            {
                int lo = lane < kIsConsumers ? lds_load(&L.progress[lane]) : INT_MAX;
                for (int off = 32; off >= 1; off >>= 1) lo = min(lo, __shfl_xor(lo, off));
                if (lo == INT_MAX) break;
            }
            z = zn;
#pragma unroll
            for (int e = 0; e < 8; ++e) cur[e] = nxt[e];
        }
    } else {
        // ================================ consumer: blocks from the queue ===========================================
        // A wave claims its next block while it still works on the current one and fetches that block's points (sorted
        // position, output index) early.  Near the end of the queue blocks are claimed only when the wave is ready
        // for them, so that no wave sits on a block while others run dry.
        int run = 0;  // run of the last claimed block's first point: only moves forward
        auto claim = [&]() {
            // all 64 lanes add 1 (one ds_add of 64 per wave): the counter runs in units of 64
            return __builtin_amdgcn_readfirstlane(atomicAdd(&L.next_block, 1)) >> 6;
        };
        // my point of block `blk` (index into the plan, -1: none), the block's group
        auto fetch = [&](const int blk, int &idx, int &grp, float &a0, float &a1, float &a2, int &pm) {
            while (L.run_blk[run + 1] <= blk) ++run;
            const int s = run / NG;
            grp = run - s * NG;
            const int t = s % TC;
            const int cum = L.run_cum[run];
            // position of the block's first point inside its run, then mine: walk the group's runs of the chunk
            int x = 32 * (((cum + 31) >> 5) + blk - L.run_blk[run]) - cum + r32;
            idx = -1;
#pragma unroll
            for (int k = 0; k < TC; ++k) {
                if (t + k < TC && s + k < nsl) {  // wave-uniform
                    const int e = run + k * NG;
                    const int st = L.run_start[e], len = L.run_start[e + 1] - st;
                    if (idx < 0 && x >= 0) {
                        if (x < len) idx = st + x;
                        x -= len;  // (negative once found)
                    }
                }
            }
            a0 = a1 = a2 = 0.0f;
            pm = 0;
            if (idx >= 0) {
                const f32x4 v = *(const f32x4 *)(spos + (int64_t)idx * 4);  // plan record {p0, p1, p2, x}: one aligned load
                a0 = v.x; a1 = v.y; a2 = v.z;
                pm = __float_as_int(v.w);  // index of the point in the caller's arrays
            }
        };
        int blk = claim();
        int idx = -1, grp = 0, pj = 0;
        float q0 = 0.0f, q1 = 0.0f, q2 = 0.0f;
        if (blk < total_blocks) fetch(blk, idx, grp, q0, q1, q2, pj);
        while (blk < total_blocks && !lds_load(&L.abort)) {
            const bool ahead = blk + 2 * kIsConsumers <= total_blocks;
            int nblk = total_blocks, nidx = -1, ngrp = 0, npj = 0;
            float n0 = 0.0f, n1 = 0.0f, n2 = 0.0f;
            if (ahead) {
                nblk = claim();
                if (nblk < total_blocks) fetch(nblk, nidx, ngrp, n0, n1, n2, npj);
            }
            const bool valid = idx >= 0;
            int c0 = 0, c1 = 0, c2 = 0;
            float f0 = 0.f, f1 = 0.f, f2 = 0.f;
            if (valid) {
                split_cell(q0, M, c0, f0);
                split_cell(q1, M, c1, f1);
                split_cell(q2, M, c2, f2);
            }
            // the points of a block are ordered by slab: its planes run from the first point's window to the last one's
            const int nvalid = __builtin_popcountll(__builtin_amdgcn_ballot_w64(valid && h == 0));
            const int z_first = __builtin_amdgcn_readlane(c0, 0) - m;
            const int z_last = __builtin_amdgcn_readlane(c0, nvalid - 1) + m + 1;
            // planes below z_first are no longer mine: the fragment reads of the previous block must have completed
            // before a producer sees this and overwrites their slots
            lds_release();
            if (lane == 0) lds_store(&L.progress[wave], z_first);
            const int ks0 = NG == 3 ? grp : 0;  // first k-step of the block's group

            // B fragments: psi2 of my point on the padded columns 16 (ks0 + ks) + 8 h + jj (zero outside the window).
            // d = f2 + m - l2 with l2 = column - o2: one subtraction from a per-lane base per value; the scale 2^11 of
            // the operand rides in the exponent; padding lanes get a base far outside every window.
            u32x4 bh[NKS], bl[NKS];
            const int o2h = c2 - tb2 - 8 * h - 16 * ks0;  // padded column of tap 0, minus this lane's column offset
            const float dbase2 = valid ? f2 + (float)(m + o2h) : 1.0e4f;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                // (packed fp32 math: two values per VALU instruction for the three arithmetic steps)
                float w[8];
#pragma unroll
                for (int jj = 0; jj < 8; jj += 2) {
                    const f32x2 d = f32x2{dbase2, dbase2} - f32x2{(float)(16 * ks + jj), (float)(16 * ks + jj + 1)};
                    const f32x2 arg = __builtin_elementwise_fma(d * d, f32x2{sc, sc}, f32x2{11.0f, 11.0f});
                    const float e0 = __builtin_amdgcn_exp2f(arg.x), e1 = __builtin_amdgcn_exp2f(arg.y);  // exp2(sc d^2) * kOpScale
                    w[jj] = (unsigned)(16 * ks + jj - o2h) < (unsigned)W ? e0 : 0.0f;
                    w[jj + 1] = (unsigned)(16 * ks + jj + 1 - o2h) < (unsigned)W ? e1 : 0.0f;
                }
                unsigned h0, h1, h2, h3, p0, p1, p2, p3;
                split_pair(w[0], w[1], h0, p0);
                split_pair(w[2], w[3], h1, p1);
                split_pair(w[4], w[5], h2, p2);
                split_pair(w[6], w[7], h3, p3);
                bh[ks] = u32x4{h0, h1, h2, h3};
                bl[ks] = u32x4{p0, p1, p2, p3};
            }
            // psi1 of my point on the 16 rows this lane holds of every T_z (MFMA result layout): row = r + 8 q + 4 h;
            // kept as pairs (registers 2 p, 2 p + 1) for the packed FMAs of the reduction
            f32x2 w1[8];
            const int o1h = c1 - tb1 - 4 * h;
            const float dbase1 = f1 + (float)(m + o1h);
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int rq0 = ((2 * p) & 3) + 8 * ((2 * p) >> 2), rq1 = rq0 + 1;
                const f32x2 d = f32x2{dbase1, dbase1} - f32x2{(float)rq0, (float)rq1};
                const f32x2 arg = d * d * f32x2{sc, sc};
                const float e0 = __builtin_amdgcn_exp2f(arg.x), e1 = __builtin_amdgcn_exp2f(arg.y);
                w1[p].x = (unsigned)(rq0 - o1h) < (unsigned)W ? e0 : 0.0f;
                w1[p].y = (unsigned)(rq1 - o1h) < (unsigned)W ? e1 : 0.0f;
            }

            float y = 0.0f;
            bool bail = false;
            // All planes of the block first (one poll reads the 16 flags: lane s looks at slot s), then a loop without
            // flag round trips.  One A-fragment buffer per k-step: as soon as the MFMAs of (z, ks) are issued the
            // fragments of (z + 1, ks) are requested into the same registers -- the other k-steps' MFMAs and the
            // reduction cover the LDS latency, which under this load is several hundred cycles.
            {
                const int zs = z_first + ((lane - z_first) & (kIsRing - 1));  // the plane of my window that lives in slot `lane`
                const bool need = lane < kIsRing && zs <= z_last;
                int spins = 0;
                while (true) {
                    const int v = need ? lds_load(&L.ready[lane]) : zs;
                    if (__builtin_amdgcn_ballot_w64(v != zs) == 0ull) break;
                    if (lds_load(&L.abort) || ++spins > kSpinLimit) { bail = true; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                lds_acquire();
            }
            if (bail) {
                // the host learns of it (nfft_hip_check_status): the rows this item has not written stay undefined
                lds_store(&L.abort, 1);
                if (lane == 0) report_fault(status, kFaultStreamStall);
                break;
            }
            {
                f16x8 A[NKS][2];
                const float pinv_first = L.pinv[z_first & (kIsRing - 1)];
                {
                    // same request order as in the loop (the compiler's wait counts merge both paths into the loop head)
                    const int slot = z_first & (kIsRing - 1);
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) {
                        __builtin_amdgcn_sched_barrier(0);
                        A[ks][0] = L.frag[slot][ks0 + ks][0][lane];
                        A[ks][1] = L.frag[slot][ks0 + ks][1][lane];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                auto plane_weight = [&](const int z, const float pinv) {
                    // axis-0 weight of plane z for my point (zero outside its window), times the plane's scale
                    const int l0 = z - (c0 - m);
                    const float d0 = f0 + (float)(m - l0);
                    const float p0 = __builtin_amdgcn_exp2f(sc * d0 * d0) * pinv;
                    return (unsigned)l0 < (unsigned)W ? p0 : 0.0f;
                };
                float pw = plane_weight(z_first, pinv_first);
                for (int z = z_first; z <= z_last; ++z) {
                    const int nz = min(z + 1, z_last);  // (the last plane re-reads itself: no branch in the loop body)
                    const int nslot = nz & (kIsRing - 1);
                    // requested before the fragments: LDS answers in order, waiting for it must not wait for them
                    const float pinv_next = L.pinv[nslot];
                    __builtin_amdgcn_sched_barrier(0);
                    f32x16 acc = 0.0f;
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) {
                        const f16x8 bhk = __builtin_bit_cast(f16x8, bh[ks]), blk2 = __builtin_bit_cast(f16x8, bl[ks]);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[ks][0], bhk, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[ks][0], blk2, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[ks][1], bhk, acc, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        A[ks][0] = L.frag[nslot][ks0 + ks][0][lane];
                        A[ks][1] = L.frag[nslot][ks0 + ks][1][lane];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    const float pw_next = plane_weight(nz, pinv_next);
                    f32x2 t = {0.0f, 0.0f};
#pragma unroll
                    for (int p = 0; p < 8; ++p)
                        t = __builtin_elementwise_fma(w1[p], f32x2{acc[2 * p], acc[2 * p + 1]}, t);  // v_pk_fma_f32
                    y = fmaf(pw, t.x + t.y, y);
                    pw = pw_next;
                }
            }
            y += __shfl_xor(y, 32);  // the two row halves of the point
            if (valid && h == 0) yr[(int64_t)pj * Cr + cr] = y * norm;
            if (!ahead) {
                nblk = claim();
                if (nblk < total_blocks) fetch(nblk, nidx, ngrp, n0, n1, n2, npj);
            }
            blk = nblk; idx = nidx; grp = ngrp; pj = npj;
            q0 = n0; q1 = n1; q2 = n2;
        }
        // nothing of the ring is mine any more: the producers may run to the end of their sweep
        lds_release();
        if (lane == 0) lds_store(&L.progress[wave], INT_MAX);
    }
    }  // work items
}

} // namespace

bool interp_stream_pays(const Geom &g, const PlanLayout &L, int64_t n)
{
    int64_t nsets = g.tiles_per_batch > 0 ? L.ntiles / g.tiles_per_batch : 1;
    if (nsets < 1) nsets = 1;
    return stream_items(n, nsets, device_cu_count(), g.M);
}

bool interp_stream_supported(const Geom &g)
{
    static const bool off = [] {
        const char *env = std::getenv("NFFT_HIP_GATHER");
        return env && (env[0] == 'l' || env[0] == 'm');  // lds: lane-per-point kernel, mfma: plane-ring kernel in lock step
    }();
    return !off && g.dim == 3 && g.wide && !g.owned && g.W <= 16;
}

template <int W, int NG>
static int launch_is_t(const Geom &g, const PlanLayout &L, const void *plan, const float *grid, int64_t n, int64_t Cr,
                       int64_t plane0, int64_t nplanes, float *yr, hipStream_t stream)
{
    const char *base = (const char *)plan;
    const int *to = (const int *)(base + L.off_offsets);
    const int *gs = (const int *)(base + L.off_groups);
    const float *spos = (const float *)(base + L.off_spos);
    const int4 *work = (const int4 *)(base + L.off_work), *sorted = work + L.work_head + L.work_cap;
    const int64_t pencils = (int64_t)g.nta[1] * g.nta[2];
    int64_t nsets = g.tiles_per_batch > 0 ? L.ntiles / g.tiles_per_batch : 1;
    if (nsets < 1) nsets = 1;
    const int nsegm = seg_base_runs(n, nsets, pencils, g.M, device_cu_count());
    const int seg_slabs = (g.M + nsegm - 1) / nsegm;
    const dim3 blocks((unsigned)(pencils * nsegm), (unsigned)nplanes);
    int *const status = device_status_block();
    static DeviceOnce attr_done;
    if (attr_done.first_use()) {
        NFFT_HIP_CHECK(hipFuncSetAttribute((const void *)interp_stream_kernel<W, false, NG>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(StreamLds)));
        NFFT_HIP_CHECK(hipFuncSetAttribute((const void *)interp_stream_kernel<W, true, NG>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(StreamLds)));
        attr_done.mark();
    }
    hipLaunchKernelGGL((interp_stream_kernel<W, false, NG>), blocks, dim3(kIsThreads), sizeof(StreamLds), stream, g, to, gs,
                       spos, grid, (int)Cr, (int)plane0, yr, seg_slabs, nsegm, work, sorted, WorkTickets{nullptr, 0u}, status);
    // the persistent launch over the work list (unbalanced plans; its workgroups leave at once otherwise); entries are
    // handed out by tickets when the launch's planes fit its share of the ticket ring, else round robin
    const WorkTickets tickets{nplanes <= kTicketPlanes ? device_ticket_ring() : nullptr, next_launch_number()};
    const dim3 oblocks(work_list_workgroups(n, nsets, pencils, nsegm, device_cu_count()), (unsigned)nplanes);
    hipLaunchKernelGGL((interp_stream_kernel<W, true, NG>), oblocks, dim3(kIsThreads), sizeof(StreamLds), stream, g, to,
                       gs, spos, grid, (int)Cr, (int)plane0, yr, seg_slabs, nsegm, work, sorted, tickets, status);
    NFFT_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int W>
static int launch_is_w(const Geom &g, const PlanLayout &L, const void *plan, const float *grid, int64_t n, int64_t Cr,
                       int64_t plane0, int64_t nplanes, float *yr, hipStream_t stream)
{
    return L.grouped ? launch_is_t<W, 3>(g, L, plan, grid, n, Cr, plane0, nplanes, yr, stream)
                     : launch_is_t<W, 1>(g, L, plan, grid, n, Cr, plane0, nplanes, yr, stream);
}

int launch_interp_stream(const Geom &g, const PlanLayout &L, const void *plan, const float *grid, int64_t n, int64_t Cr,
                         int64_t plane0, int64_t nplanes, float *yr, hipStream_t stream)
{
    if (nplanes <= 0 || n <= 0) return 0;
    switch (g.m) {
    case 1: return launch_is_w<4>(g, L, plan, grid, n, Cr, plane0, nplanes, yr, stream);
    case 2: return launch_is_w<6>(g, L, plan, grid, n, Cr, plane0, nplanes, yr, stream);
    case 3: return launch_is_w<8>(g, L, plan, grid, n, Cr, plane0, nplanes, yr, stream);
    case 4: return launch_is_w<10>(g, L, plan, grid, n, Cr, plane0, nplanes, yr, stream);
    case 5: return launch_is_w<12>(g, L, plan, grid, n, Cr, plane0, nplanes, yr, stream);
    case 6: return launch_is_w<14>(g, L, plan, grid, n, Cr, plane0, nplanes, yr, stream);
    case 7: return launch_is_w<16>(g, L, plan, grid, n, Cr, plane0, nplanes, yr, stream);
    }
    set_error("matrix-core interpolation supports cutoff 1..7");
    return 1;
}

} // namespace nfft
