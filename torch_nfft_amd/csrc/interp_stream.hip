// Interpolation (forward gather) on the matrix cores, streamed: producer waves feed a ring of grid planes,
// consumer waves take blocks of points from a queue -- no workgroup barrier inside a work item.
//
// Same arithmetic as interp_mfma.hip (reference: csrc/cuda/spatial_window_operations.cu:214-332): per plane z of a
// pencil and block of 32 points  T_z = G_z Psi2  (12 MFMAs on f16-split operands), t = sum_u1 psi1[u1] T_z[u1, i],
// y_i += psi0_i[z] t.  What changes is the schedule.  interp_mfma.hip advances in lock step: stage the planes of a
// chunk (all waves wait for the loads), barrier, every wave takes one block, barrier ... -- measured at config C3:
// staging alone 0.44 ms, compute alone 1.30 ms, nothing overlaps, and a chunk's ~19 blocks leave 3 for a second round
// (profiles/r02_experiments.md).  Here
//   * 3 producer waves stage planes continuously: wave p takes the ring slots s = p (mod 3), i.e. the planes z with
//     (z & 15) % 3 == p, of the item's
//     sweep, loads the padded 32 x 64 tile (the next plane's loads are in flight while the current one is converted),
//     scales it by its own power of two, f16-splits it into the ring slot z & 15 and publishes ready[slot] = z;
//   * 13 consumer waves pull blocks from an LDS counter.  A block belongs to one chunk of 17 - (2m+2) slabs, so its
//     window is at most the 16 planes of the ring; it publishes the first plane it needs (progress[wave]), builds
//     its B fragments / psi1 weights, then walks its planes, waiting for each plane's ready flag;
//   * a producer may overwrite slot z & 15 once every consumer's progress is beyond z - 16.  The slowest consumer
//     needs planes below progress + 16 only, so the producers can always serve it: no cycle of waits.
// Waves drift apart on their own, so the fragment builds (VALU) of some waves overlap the MFMAs of others.
// Every spin loop is bounded (kSpinLimit): a logic error ends the kernel with wrong numbers, never a hung GPU.
#include <algorithm>
#include <climits>
#include <cstdlib>

#include "common.h"
#include "kernels.h"
#include "mfma_split.h"

namespace nfft {

namespace {

constexpr int kIsThreads = 1024;
constexpr int kIsWaves = kIsThreads / 64;
constexpr int kIsProducers = 3;      // measured at C3: 1 producer 2.29 ms, 2: 1.58, 3: 1.43, 4: 1.47 (interp_mfma.hip: 1.63)
constexpr int kIsConsumers = kIsWaves - kIsProducers;
constexpr int kIsRing = 16;          // resident planes: TC + 2m+1 = 16 for every cutoff of the wide tiling
constexpr int kIsMaxChunks = 64;     // chunks of one work item: one lane of the set-up wave each (<= 128 slabs / TC + 1)
constexpr int kSpinLimit = 1 << 22;

struct __align__(16) StreamLds {
    f16x8 frag[kIsRing][4][2][64];   // [plane slot][k-step][hi/lo][lane = 32 (column half) + row]   128 KB
    float pinv[kIsRing];             // what one unit of the scaled plane is worth, times the B operand scale
    int ready[kIsRing];              // plane number held by the slot (published after the fragments)
    int progress[kIsConsumers];      // first plane a consumer still needs
    int next_block;                  // block queue (counts in units of 64: every lane adds 1)
    int abort;                       // set when a spin loop ran out: everybody leaves
    int nchunks;
    int chunk_blk[kIsMaxChunks + 1]; // blocks before chunk c of the item (prefix sums)
    int chunk_pt[kIsMaxChunks + 1];  // first point of chunk c
};

__device__ __forceinline__ int lds_load(const int *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_store(int *p, int v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int W, bool OVERFLOW>
__global__ void __launch_bounds__(kIsThreads) __attribute__((amdgpu_waves_per_eu(4, 4)))
interp_stream_kernel(const Geom g, const int *__restrict__ tile_offsets, const int *__restrict__ perm,
                     const float *__restrict__ spos, const float *__restrict__ grid, const int Cr, const int plane0,
                     float *__restrict__ yr, const int seg_slabs, const int nsegm, const int *__restrict__ first_end,
                     const int4 *__restrict__ overflow)
{
    constexpr int m = W / 2 - 1;
    constexpr int TC = 17 - W;
    static_assert(TC >= 1 && TC + W - 1 == kIsRing, "ring holds exactly one chunk's planes");
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

    const int n_items = OVERFLOW ? overflow[0].x : 1;
    for (int item = OVERFLOW ? (int)blockIdx.x : 0; item < n_items; item += OVERFLOW ? (int)gridDim.x : 1) {
    int pencil, sb, se;
    if constexpr (OVERFLOW) {
        const int4 it = overflow[1 + item];
        if (it.x / pencils != b) continue;  // another point set's piece
        pencil = it.x % pencils;
        sb = it.y;
        se = it.z;
    } else {
        pencil = blockIdx.x / nsegm;
        const int seg = blockIdx.x % nsegm;
        sb = min(seg * seg_slabs, M);
        se = sb < M ? first_end[(int64_t)(b * pencils + pencil) * kSegMax + seg] : sb;
    }
    // an item owns the chunks whose first slab lies in its range (as in interp_mfma.hip)
    const int k_begin = (sb + TC - 1) / TC;
    const int k_end = min(g.nta[0], (se + TC - 1) / TC);
    if (k_begin >= k_end) continue;
    const int bin0 = b * g.tiles_per_batch + pencil * g.np0;
    const int nchunks = k_end - k_begin;  // <= kIsMaxChunks (launcher)
    {
        int s0, e0, s1, e1;
        chunk_range(g, tile_offsets, bin0, k_begin, s0, e0);
        chunk_range(g, tile_offsets, bin0, k_end - 1, s1, e1);
        if (s0 == e1) continue;  // no points in these chunks
    }
    const int j2 = pencil % g.nta[2], j1 = pencil / g.nta[2];
    const int tb1 = j1 * g.Ta[1], tb2 = j2 * g.Ta[2];
    const float sc = win_exp_scale(m);
    float norm = win_norm(m);
    norm = norm * norm * norm;
    const float *const gplane = grid + (int64_t)plane_local * g.cells;

    // ---- item set-up: chunk table, flags ----------------------------------------------------------------------
    __syncthreads();  // the previous item is done with the LDS
    if (wave == 0) {
        // blocks per chunk -> prefix sums (nchunks <= 64: one lane per chunk)
        int s = 0, e = 0;
        if (lane < nchunks) chunk_range(g, tile_offsets, bin0, k_begin + lane, s, e);
        const int nb = (e - s + 31) >> 5;
        int incl = nb;
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (lane < nchunks) {
            L.chunk_blk[lane] = incl - nb;
            L.chunk_pt[lane] = s;
        }
        if (lane == nchunks - 1) {
            L.chunk_blk[nchunks] = incl;
            L.chunk_pt[nchunks] = e;
        }
        if (lane < kIsRing) L.ready[lane] = INT_MIN;
        if (lane < kIsConsumers) L.progress[lane] = k_begin * TC - m;
        if (lane == 0) { L.next_block = 0; L.abort = 0; L.nchunks = nchunks; }
    }
    __syncthreads();
    const int total_blocks = L.chunk_blk[nchunks];

    if (wave >= kIsConsumers) {
        // ================================ producer: planes z = z_begin + p, + 4, ... ================================
        const int p = wave - kIsConsumers;
        const int z_begin = k_begin * TC - m, z_end = (k_end - 1) * TC + kIsRing - m;  // planes any chunk of the item needs
        // a lane's 4 tasks of a plane: (row, group of 8 columns); 16 consecutive lanes = 16 consecutive rows of one
        // column group -> consecutive 16-byte LDS slots on the way out, half rows of 128 contiguous bytes on the way in
        auto needed = [&](const int z) {
            // plane z is used by chunk k iff k TC - m <= z <= k TC + TC + m: at most three candidates
            const int k_hi = min(k_end - 1, (z + m) / TC), k_lo = max(k_begin, (z - TC - m + TC - 1) / TC);
            for (int k = k_lo; k <= k_hi; ++k) {
                if (k < k_begin) continue;
                const int c = k - k_begin;
                if (L.chunk_blk[c + 1] > L.chunk_blk[c]) return true;
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
                    v[2 * i] = *(const f32x4 *)(grow + c0);
                    v[2 * i + 1] = *(const f32x4 *)(grow + c0 + 4);
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
                if (++spins > kSpinLimit) { lds_store(&L.abort, 1); break; }
                __builtin_amdgcn_s_sleep(4);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
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
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
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
        int chunk = 0;  // chunks only move forward for a wave
        while (true) {
            // all 64 lanes add 1 (one ds_add of 64 per wave): the counter runs in units of 64
            const int blk = __builtin_amdgcn_readfirstlane(atomicAdd(&L.next_block, 1)) >> 6;
            if (blk >= total_blocks || lds_load(&L.abort)) break;
            while (L.chunk_blk[chunk + 1] <= blk) ++chunk;
            const int s = L.chunk_pt[chunk], e = L.chunk_pt[chunk + 1];
            const int j0 = s + (blk - L.chunk_blk[chunk]) * 32;
            const int j = j0 + r32;
            const bool valid = j < e;
            int c0 = 0, c1 = 0, c2 = 0;
            float f0 = 0.f, f1 = 0.f, f2 = 0.f;
            if (valid) {
                split_cell(spos[(int64_t)j * 3 + 0], M, c0, f0);
                split_cell(spos[(int64_t)j * 3 + 1], M, c1, f1);
                split_cell(spos[(int64_t)j * 3 + 2], M, c2, f2);
            }
            // the plan is sorted by slab: the block's planes run from the first point's window to the last one's
            const int nvalid = min(32, e - j0);
            const int z_first = __builtin_amdgcn_readlane(c0, 0) - m;
            const int z_last = __builtin_amdgcn_readlane(c0, nvalid - 1) + m + 1;
            if (lane == 0) lds_store(&L.progress[wave], z_first);  // planes below are no longer mine

            // B fragments: psi2 of my point on the padded columns 16 ks + 8 h + jj (zero outside the window).
            // d = f2 + m - l2 with l2 = column - o2: one subtraction from a per-lane base per value; the scale 2^11 of
            // the operand rides in the exponent; padding lanes get a base far outside every window.
            u32x4 bh[4], bl[4];
            const int o2h = c2 - tb2 - 8 * h;  // padded column of tap 0, minus this lane's column offset
            const float dbase2 = valid ? f2 + (float)(m + o2h) : 1.0e4f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                float w[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const float d = dbase2 - (float)(16 * ks + jj);
                    const float ev = __builtin_amdgcn_exp2f(fmaf(d * d, sc, 11.0f));  // exp2(sc d^2) * kOpScale
                    w[jj] = (unsigned)(16 * ks + jj - o2h) < (unsigned)W ? ev : 0.0f;
                }
                unsigned h0, h1, h2, h3, q0, q1, q2, q3;
                split_pair(w[0], w[1], h0, q0);
                split_pair(w[2], w[3], h1, q1);
                split_pair(w[4], w[5], h2, q2);
                split_pair(w[6], w[7], h3, q3);
                bh[ks] = u32x4{h0, h1, h2, h3};
                bl[ks] = u32x4{q0, q1, q2, q3};
            }
            // 16-column groups of the padded tile that no point of the block touches: their B fragments are all zero and
            // the k-step is skipped (the plan orders the points of a slab by column quarter: ~2.5 of 4 groups are live)
            bool live[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                live[ks] = __builtin_amdgcn_ballot_w64((bh[ks].x | bh[ks].y | bh[ks].z | bh[ks].w) != 0u) != 0ull;
            // psi1 of my point on the 16 rows this lane holds of every T_z (MFMA result layout): row = r + 8 q + 4 h
            float w1[16];
            const int o1h = c1 - tb1 - 4 * h;
            const float dbase1 = f1 + (float)(m + o1h);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int rq = (reg & 3) + 8 * (reg >> 2);
                const float d = dbase1 - (float)rq;
                const float ev = __builtin_amdgcn_exp2f(sc * d * d);
                w1[reg] = (unsigned)(rq - o1h) < (unsigned)W ? ev : 0.0f;
            }

            float y = 0.0f;
            bool bail = false;
            for (int z = z_first; z <= z_last; ++z) {
                const int slot = z & (kIsRing - 1);
                int spins = 0;
                while (lds_load(&L.ready[slot]) != z) {
                    if (lds_load(&L.abort) || ++spins > kSpinLimit) { bail = true; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                if (bail) break;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                f32x16 acc = 0.0f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    if (!live[ks]) continue;  // wave-uniform
                    const f16x8 ah = L.frag[slot][ks][0][lane], al = L.frag[slot][ks][1][lane];
                    const f16x8 bhk = __builtin_bit_cast(f16x8, bh[ks]), blk2 = __builtin_bit_cast(f16x8, bl[ks]);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bhk, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, blk2, acc, 0, 0, 0);
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
            if (bail) { lds_store(&L.abort, 1); break; }
            y += __shfl_xor(y, 32);  // the two row halves of the point
            if (valid && h == 0) yr[(int64_t)perm[j] * Cr + cr] = y * norm;
        }
        // nothing of the ring is mine any more: the producers may run to the end of their sweep
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) lds_store(&L.progress[wave], INT_MAX);
    }
    }  // work items
}

} // namespace

// Worth it when the work items are big: the pipeline's warm-up and tail cost ~10 us per item.  The size of an item in
// a populated region is the plan's target (seg_target_points: n / (5.4 CUs), at least 2048 points): 7 237 at config C3
// (1.63 -> 1.43 ms), 2 048 at C5, where the lock-step kernel stays ahead (0.62 vs 0.69 ms).
bool interp_stream_pays(const Geom &g, const PlanLayout &L, int64_t n)
{
    int64_t nsets = g.tiles_per_batch > 0 ? L.ntiles / g.tiles_per_batch : 1;
    if (nsets < 1) nsets = 1;
    return seg_target_points(n, nsets, device_cu_count()) >= 4000;
}

bool interp_stream_supported(const Geom &g)
{
    static const bool off = [] {
        const char *env = std::getenv("NFFT_HIP_GATHER");
        return env && (env[0] == 'l' || env[0] == 'm');  // lds: lane-per-point kernel, mfma: plane-ring kernel in lock step
    }();
    // chunk table: a work item has at most ceil(128 / TC) + 1 chunks
    const int TC = 17 - g.W;
    return !off && g.dim == 3 && g.wide && !g.owned && TC >= 1 && (128 + TC - 1) / TC + 1 <= kIsMaxChunks;
}

template <int W>
static int launch_is_t(const Geom &g, const PlanLayout &L, const void *plan, const float *grid, int64_t n, int64_t Cr,
                       int64_t plane0, int64_t nplanes, float *yr, hipStream_t stream)
{
    const char *base = (const char *)plan;
    const int *to = (const int *)(base + L.off_offsets);
    const int *perm = (const int *)(base + L.off_perm);
    const float *spos = (const float *)(base + L.off_spos);
    const int *first_end = (const int *)(base + L.off_cursor);
    const int64_t pencils = (int64_t)g.nta[1] * g.nta[2];
    int64_t nsets = g.tiles_per_batch > 0 ? L.ntiles / g.tiles_per_batch : 1;
    if (nsets < 1) nsets = 1;
    const int nsegm = seg_base_runs(n, nsets, pencils, g.M, device_cu_count());
    const int seg_slabs = (g.M + nsegm - 1) / nsegm;
    const dim3 blocks((unsigned)(pencils * nsegm), (unsigned)nplanes);
    static DeviceOnce attr_done;
    if (attr_done.first_use()) {
        NFFT_HIP_CHECK(hipFuncSetAttribute((const void *)interp_stream_kernel<W, false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(StreamLds)));
        NFFT_HIP_CHECK(hipFuncSetAttribute((const void *)interp_stream_kernel<W, true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(StreamLds)));
        attr_done.mark();
    }
    hipLaunchKernelGGL((interp_stream_kernel<W, false>), blocks, dim3(kIsThreads), sizeof(StreamLds), stream, g, to, perm,
                       spos, grid, (int)Cr, (int)plane0, yr, seg_slabs, nsegm, first_end, (const int4 *)nullptr);
    if (L.two_level) {
        const int4 *overflow = (const int4 *)(base + L.off_tmp);
        const dim3 oblocks((unsigned)std::min(device_cu_count(), 1024), (unsigned)nplanes);
        hipLaunchKernelGGL((interp_stream_kernel<W, true>), oblocks, dim3(kIsThreads), sizeof(StreamLds), stream, g, to,
                           perm, spos, grid, (int)Cr, (int)plane0, yr, seg_slabs, nsegm, first_end, overflow);
    }
    NFFT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_interp_stream(const Geom &g, const PlanLayout &L, const void *plan, const float *grid, int64_t n, int64_t Cr,
                         int64_t plane0, int64_t nplanes, float *yr, hipStream_t stream)
{
    if (nplanes <= 0 || n <= 0) return 0;
    switch (g.m) {
    case 1: return launch_is_t<4>(g, L, plan, grid, n, Cr, plane0, nplanes, yr, stream);
    case 2: return launch_is_t<6>(g, L, plan, grid, n, Cr, plane0, nplanes, yr, stream);
    case 3: return launch_is_t<8>(g, L, plan, grid, n, Cr, plane0, nplanes, yr, stream);
    case 4: return launch_is_t<10>(g, L, plan, grid, n, Cr, plane0, nplanes, yr, stream);
    case 5: return launch_is_t<12>(g, L, plan, grid, n, Cr, plane0, nplanes, yr, stream);
    case 6: return launch_is_t<14>(g, L, plan, grid, n, Cr, plane0, nplanes, yr, stream);
    case 7: return launch_is_t<16>(g, L, plan, grid, n, Cr, plane0, nplanes, yr, stream);
    }
    set_error("matrix-core interpolation supports cutoff 1..7");
    return 1;
}

} // namespace nfft
