// Spreading on the matrix cores (3-D grids), selected by NFFT_HIP_SPREAD=mfma.
//
// Same result as the reference's adjoint_window_convolution kernels (csrc/cuda/spatial_window_operations.cu:
// 103-211).  For one grid plane z of a pencil the window sum is an outer-product accumulation
//     G_z[u1, u2] = sum_i (x_i psi0_i[z] psi1_i[u1]) * psi2_i[u2]  =  (A_z B)[u1, u2],
// A_z = [32 rows x K points], B = [K points x 64 columns] -- a GEMM whose operands hold only 32 + 64 window
// values per point instead of the (2m+2)^2 products, and whose 32 x 64 result is exactly the padded pencil
// (T1 + 2m+1 = 32 rows, T2 + 2m+1 = 64 columns).  It runs on v_mfma_f32_32x32x16_f16 with two-way split
// operands (2^11 x = hi + lo in f16; hi*hi + hi*lo + lo*hi keeps ~22 bits), fp32 accumulators in registers:
//   * the plan is sorted by single planes ("slabs"); a K-block is 16 points of ONE slab, so the axis-0 weight of
//     a plane is a wave-uniform row of a small table;
//   * one workgroup (16 waves) sweeps a segment of a pencil; wave w owns the resident plane z = w (mod 16): it
//     needs no LDS accumulator and no barrier to accumulate, flushes its 32 x 64 tile with global atomics (each
//     instruction = two 128-byte row segments) as soon as the sweep has passed it, and moves on to plane z + 16;
//   * per batch of 8 K-blocks the workgroup builds the operands once in LDS (psi1 table, f16-split B fragments in
//     MFMA register order, axis-0 table); every wave whose plane lies in a K-block's window turns the psi1 rows
//     into its A fragment (8 multiplies + f16 split) and issues 6 MFMAs.
// Why: ds_add_f32 is unusable on gfx950 and the f64 LDS atomic bounds spread.hip at ~3 ms for 1e10 taps
// (DESIGN.md section 4); here the taps cost 30 CU-cycles per point on the matrix pipe.
#include <climits>

#include "common.h"
#include "kernels.h"

namespace nfft {

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kKB = 16;        // points per K-block (the MFMA K dimension)
constexpr int kNKB = 8;        // K-blocks per batch
constexpr int kSlots = kKB * kNKB;
constexpr int kMfmaThreads = 1024;
constexpr int kMaxSegSlabs = 128;
// Both operands are scaled by 2^11 before the f16 split (values <= 2048 fit f16): the lo parts, ~2^-11 of the
// value, are then normal f16 numbers instead of subnormals; the flush multiplies by 2^-22.
constexpr float kOpScale = 2048.0f;
constexpr int kPsiStride = 20; // floats per row of the psi1 table (16 + 4: conflict-free ds_read_b128 over rows)

// Operands of one batch of K-blocks, double-buffered: while the waves run the MFMAs of batch i they already build
// the operands of batch i + 1.
template <int W>
struct __align__(16) MfmaOps {
    f16x8 bfrag[kNKB][2][2][64];          // [K-block][column tile][hi/lo][lane]
    float psi1[kNKB][32][kPsiStride];     // [K-block][row][point]
    float atab[kNKB][W][kKB];             // [K-block][axis-0 tap][point]   x' * psi0
    int slab[kNKB];
};

// Points of one batch (cell fractions, in-pencil cells, scaled value), double-buffered as well.
struct __align__(16) MfmaStage {
    float f0[kSlots], f1[kSlots], f2[kSlots], x[kSlots];
    int c1[kSlots], c2[kSlots];
    int slab[kNKB];
};

template <int W>
struct __align__(16) MfmaLds {
    MfmaOps<W> ops[2];
    MfmaStage stag[2];
    int soff[kMaxSegSlabs + 1];           // point offsets of the segment's slabs
    int kbp[kMaxSegSlabs + 1];            // K-blocks before each slab (a slab's last K-block may be partial)
};

template <int W>
__global__ void __launch_bounds__(kMfmaThreads)
spread_mfma_kernel(const Geom g, const int *__restrict__ tile_offsets, const float *__restrict__ spos,
                   const float *__restrict__ xs, const float *__restrict__ maxabs, const int64_t n, const int Cr,
                   const int plane0, float *__restrict__ grid, const int seg_slabs, const int nsegm)
{
    constexpr int m = W / 2 - 1;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    MfmaLds<W> &L = *reinterpret_cast<MfmaLds<W> *>(smem_raw);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r32 = lane & 31, h = lane >> 5;

    const int seg = blockIdx.x % nsegm;
    const int pencil = blockIdx.x / nsegm;
    const int j2 = pencil % g.nta[2];
    const int j1 = pencil / g.nta[2];
    const int plane_local = blockIdx.y;
    const int plane = plane0 + plane_local;
    const int b = plane / Cr;
    const int cr = plane - b * Cr;

    const int sb = seg * seg_slabs;
    const int se = min(g.M, sb + seg_slabs);
    const int nslab = se - sb;
    const int bin0 = b * g.tiles_per_batch + pencil * g.np0;  // np0 == M: one plan bin per slab
    if (tile_offsets[bin0 + sb] == tile_offsets[bin0 + se]) return;

    const int tb1 = j1 * g.Ta[1], tb2 = j2 * g.Ta[2];
    const float sc = win_exp_scale(m);
    float norm = win_norm(m);
    norm = norm * norm * norm;
    // x is scaled into [-1, 1] by a power of two so that every operand fits f16; undone at the flush
    float xscale = 1.0f;
    {
        const float mx = *maxabs;
        if (mx > 0.0f && mx < 3.0e38f) {
            int e;
            frexpf(mx, &e);
            xscale = ldexpf(1.0f, e);
        }
    }
    const float inv_xscale = 1.0f / xscale;
    const float unscale = xscale * norm * (1.0f / (kOpScale * kOpScale));
    const float *const xcol = xs + (int64_t)cr * n;
    float *const gplane = grid + (int64_t)plane_local * g.cells;

    f32x16 acc0 = 0.0f, acc1 = 0.0f;
    bool dirty = false;
    // wave w owns the plane congruent to w (mod 16) inside the sliding window that starts at sb - m
    int myz = (sb - m) + (((wave - (sb - m)) % 16) + 16) % 16;

    auto flush = [&]() {
        if (dirty) {
            const int gz = wrap(myz, g.M);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int gc = wrap_near(tb2 - m + 32 * t + r32, g.M);
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
                    const float v = (t == 0 ? acc0[reg] : acc1[reg]) * unscale;
                    if (v != 0.0f) {
                        const int gr = wrap_near(tb1 - m + row, g.M);
                        atomicAdd(gplane + ((int64_t)gz * g.M + gr) * g.M + gc, v);
                    }
                }
            }
            acc0 = 0.0f;
            acc1 = 0.0f;
            dirty = false;
        }
    };

    // ---- K-block schedule: slab s holds ceil(count / 16) K-blocks; kbp = their exclusive prefix sums -----------
    for (int i = tid; i <= nslab; i += kMfmaThreads) L.soff[i] = tile_offsets[bin0 + sb + i];
    __syncthreads();
    if (wave == 0) {
        const int s0 = 2 * lane, s1 = s0 + 1;
        const int n0 = s0 < nslab ? (L.soff[s0 + 1] - L.soff[s0] + kKB - 1) / kKB : 0;
        const int n1 = s1 < nslab ? (L.soff[s1 + 1] - L.soff[s1] + kKB - 1) / kKB : 0;
        int incl = n0 + n1;
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        const int excl = incl - (n0 + n1);
        L.kbp[s0] = excl;
        L.kbp[s1] = excl + n0;
        if (lane == 63) L.kbp[kMaxSegSlabs] = incl;
    }
    __syncthreads();
    const int total = L.kbp[nslab];
    const int nbatch = (total + kNKB - 1) / kNKB;

    // ---- staging of a batch: thread -> (K-block, point); the loads are issued one pipeline step ahead ---------
    float r0 = 0.f, r1 = 0.f, r2 = 0.f, rx = 0.f;
    int rslab = INT_MAX;
    bool rhave = false;
    auto stage_load = [&](const int batch) {
        const int j = tid / kKB, i = tid - j * kKB;
        const int q = batch * kNKB + j;
        rslab = INT_MAX;
        rhave = false;
        if (q < total) {
            int lo = 0, hi = nslab;  // kbp[lo] <= q < kbp[hi]
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (L.kbp[mid] <= q) lo = mid; else hi = mid;
            }
            const int start = L.soff[lo] + kKB * (q - L.kbp[lo]);
            rslab = sb + lo;
            if (start + i < L.soff[lo + 1]) {
                const int64_t idx = (int64_t)start + i;
                r0 = spos[idx * 3 + 0];
                r1 = spos[idx * 3 + 1];
                r2 = spos[idx * 3 + 2];
                rx = xcol[idx];
                rhave = true;
            }
        }
    };
    auto stage_store = [&](MfmaStage &S) {
        float f0 = 0.f, f1 = 0.f, f2 = 0.f, xv = 0.f;
        int c1 = -1000, c2 = -1000;  // padding slots: outside every window
        if (rhave) {
            int c0;
            split_cell(r0, g.M, c0, f0);
            split_cell(r1, g.M, c1, f1);
            split_cell(r2, g.M, c2, f2);
            c1 -= tb1 - m;  // row of the point's cell inside the padded pencil (tap l1 sits at row c1 - m + l1)
            c2 -= tb2 - m;
            xv = rx * inv_xscale;
        }
        S.f0[tid] = f0; S.f1[tid] = f1; S.f2[tid] = f2; S.x[tid] = xv;
        S.c1[tid] = c1; S.c2[tid] = c2;
        if ((tid & (kKB - 1)) == 0) S.slab[tid / kKB] = rslab;
    };

    // ---- operands of a batch ---------------------------------------------------------------------------------
    auto build_operands = [&](const MfmaStage &S, MfmaOps<W> &O, const int nkb) {
        // B fragments: thread -> (K-block, column tile, lane): 8 points of one column, split into f16 hi / lo
        for (int task = tid; task < nkb * 128; task += kMfmaThreads) {
            const int j = task >> 7, t = (task >> 6) & 1, ln = task & 63;
            const int col = 32 * t + (ln & 31), k0 = 8 * (ln >> 5);
            f16x8 hi, lo;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int slot = j * kKB + k0 + jj;
                const int dc = S.c2[slot] - col;  // distance = fraction + whole cells, exact in fp32
                const float d = S.f2[slot] + (float)dc;
                const int l2 = m - dc;
                float v = __builtin_amdgcn_exp2f(sc * d * d) * kOpScale;
                v = (unsigned)l2 < (unsigned)W ? v : 0.0f;
                asm volatile("" : "+v"(v));  // see the A fragments below
                const _Float16 vh = (_Float16)v;
                hi[jj] = vh;
                lo[jj] = (_Float16)(v - (float)vh);
            }
            O.bfrag[j][t][0][ln] = hi;
            O.bfrag[j][t][1][ln] = lo;
        }
        // psi1 table: thread -> (K-block, row, half): 8 points
        for (int task = tid; task < nkb * 64; task += kMfmaThreads) {
            const int j = task >> 6, row = task & 31, k0 = 8 * ((task >> 5) & 1);
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int slot = j * kKB + k0 + jj;
                const int dc = S.c1[slot] - row;
                const float d = S.f1[slot] + (float)dc;
                const int l1 = m - dc;
                const float v = __builtin_amdgcn_exp2f(sc * d * d);
                O.psi1[j][row][k0 + jj] = (unsigned)l1 < (unsigned)W ? v : 0.0f;
            }
        }
        // axis-0 table: thread -> (K-block, tap, point)
        for (int task = tid; task < nkb * W * kKB; task += kMfmaThreads) {
            const int j = task / (W * kKB), rem = task - j * (W * kKB);
            const int l0 = rem / kKB, k = rem - l0 * kKB;
            const int slot = j * kKB + k;
            const float d = S.f0[slot] + (float)(m - l0);
            O.atab[j][l0][k] = S.x[slot] * __builtin_amdgcn_exp2f(sc * d * d) * kOpScale;
        }
        if (tid < kNKB) O.slab[tid] = S.slab[tid];
    };

    // ---- every wave adds the K-blocks that reach its plane -------------------------------------------------
    auto accumulate = [&](const MfmaOps<W> &O, const int nkb) {
        for (int j = 0; j < nkb; ++j) {
            const int s = O.slab[j];
            // the sweep has passed plane myz once the current slab is beyond myz + m
            while (myz + m < s) {
                flush();
                myz += 16;
            }
            const int l0 = myz - s + m;  // axis-0 tap of this K-block's points that lands on my plane
            if ((unsigned)l0 < (unsigned)W) {
                const f32x4 *pp = (const f32x4 *)&O.psi1[j][r32][8 * h];
                const f32x4 *pa = (const f32x4 *)&O.atab[j][l0][8 * h];
                const f32x4 p0 = pp[0], p1 = pp[1], a0 = pa[0], a1 = pa[1];
                float v[8] = {p0.x * a0.x, p0.y * a0.y, p0.z * a0.z, p0.w * a0.w,
                              p1.x * a1.x, p1.y * a1.y, p1.z * a1.z, p1.w * a1.w};
                f16x8 ah, al;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    // opaque to the optimiser: hi must be the f16 rounding of the SAME fp32 value the residual is
                    // taken from (under -ffp-contract=fast the residual is otherwise fused against a separately
                    // rounded product and ends up one f16 ulp off near ties)
                    asm volatile("" : "+v"(v[jj]));
                    const _Float16 vh = (_Float16)v[jj];
                    ah[jj] = vh;
                    al[jj] = (_Float16)(v[jj] - (float)vh);
                }
                const f16x8 b0h = O.bfrag[j][0][0][lane], b0l = O.bfrag[j][0][1][lane];
                const f16x8 b1h = O.bfrag[j][1][0][lane], b1l = O.bfrag[j][1][1][lane];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b0h, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b1h, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b0l, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b1l, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b0h, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b1h, acc1, 0, 0, 0);
                dirty = true;
            }
        }
    };

    // ---- software pipeline, one barrier per batch: step i loads the points of batch i + 2 into registers, builds
    // the operands of batch i + 1 (buffer (i+1)&1) and runs the MFMAs of batch i (buffer i&1); the loaded points go
    // to the staging buffer batch i used (its last reader, step i - 1, is behind the previous barrier).
    if (tid < kSlots) {
        stage_load(0);
        stage_store(L.stag[0]);
    }
    __syncthreads();
    for (int i = -1; i < nbatch; ++i) {
        const bool have2 = i + 2 < nbatch && tid < kSlots;
        if (have2) stage_load(i + 2);
        if (i + 1 < nbatch) build_operands(L.stag[(i + 1) & 1], L.ops[(i + 1) & 1], min(kNKB, total - (i + 1) * kNKB));
        if (i >= 0) accumulate(L.ops[i & 1], min(kNKB, total - i * kNKB));
        if (have2) stage_store(L.stag[i & 1]);
        __syncthreads();
    }
    flush();
}

} // namespace

bool spread_mfma_supported(const Geom &g) { return g.dim == 3 && g.wide; }

template <int W>
static int launch_mfma_t(const Geom &g, const int *to, const float *spos, const float *xs, const float *maxabs,
                         int64_t n, int64_t Cr, int64_t plane0, int64_t nplanes, float *grid, hipStream_t stream)
{
    int seg_slabs = kMaxSegSlabs;
    if (seg_slabs > g.M) seg_slabs = g.M;
    const int nsegm = (g.M + seg_slabs - 1) / seg_slabs;
    const dim3 blocks((unsigned)(g.nta[1] * g.nta[2] * nsegm), (unsigned)nplanes);
    static bool attr_done = false;  // one workgroup per CU: the double-buffered operands take most of the 160 KB LDS
    if (!attr_done) {
        NFFT_HIP_CHECK(hipFuncSetAttribute((const void *)spread_mfma_kernel<W>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)sizeof(MfmaLds<W>)));
        attr_done = true;
    }
    hipLaunchKernelGGL((spread_mfma_kernel<W>), blocks, dim3(kMfmaThreads), sizeof(MfmaLds<W>), stream, g, to, spos, xs, maxabs, n,
                       (int)Cr, (int)plane0, grid, seg_slabs, nsegm);
    NFFT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_spread_mfma(const Geom &g, const PlanLayout &L, const void *plan, const float *xs, const float *maxabs,
                       int64_t n, int64_t Cr, int64_t plane0, int64_t nplanes, float *grid, hipStream_t stream)
{
    const char *base = (const char *)plan;
    const int *to = (const int *)(base + L.off_offsets);
    const float *spos = (const float *)(base + L.off_spos);
    if (nplanes <= 0 || n <= 0) return 0;
    switch (g.m) {
    case 1: return launch_mfma_t<4>(g, to, spos, xs, maxabs, n, Cr, plane0, nplanes, grid, stream);
    case 2: return launch_mfma_t<6>(g, to, spos, xs, maxabs, n, Cr, plane0, nplanes, grid, stream);
    case 3: return launch_mfma_t<8>(g, to, spos, xs, maxabs, n, Cr, plane0, nplanes, grid, stream);
    case 4: return launch_mfma_t<10>(g, to, spos, xs, maxabs, n, Cr, plane0, nplanes, grid, stream);
    case 5: return launch_mfma_t<12>(g, to, spos, xs, maxabs, n, Cr, plane0, nplanes, grid, stream);
    case 6: return launch_mfma_t<14>(g, to, spos, xs, maxabs, n, Cr, plane0, nplanes, grid, stream);
    case 7: return launch_mfma_t<16>(g, to, spos, xs, maxabs, n, Cr, plane0, nplanes, grid, stream);
    case 8: return launch_mfma_t<18>(g, to, spos, xs, maxabs, n, Cr, plane0, nplanes, grid, stream);
    }
    set_error("cutoff m must be in 1..8");
    return 1;
}

} // namespace nfft
