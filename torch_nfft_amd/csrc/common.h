// Shared host/device definitions of the MI355X NFFT hot path (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <string>

namespace nfft {

// ---- error plumbing -------------------------------------------------------
void set_error(const std::string &msg);
#define NFFT_HIP_CHECK(expr)                                                                      \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            nfft::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                   \
            return 4; /* NFFT_HIP_EHIP */                                                         \
        }                                                                                         \
    } while (0)

// ---- device-side failure reports ------------------------------------------
// A kernel that has to give up (a bounded spin loop of the streamed gather ran out, a batch index outside [0, B))
// cannot return an error code: it raises a flag in a host-mapped status block instead, one block per device, which
// the entry points of the C ABI look at (nfft_hip_check_status; every compute entry point refuses to run while a
// fault is pending).  The reference aborts the process on a device error (csrc/cuda/cuda_utils.cu:5-16); here the
// next call returns NFFT_HIP_EKERNEL.  One int per fault kind: plain system-scope stores, no read-modify-write
// across the bus.
enum DeviceFault { kFaultStreamStall = 0, kFaultBatchIndex = 1, kFaultBatchOrder = 2, kFaultStalePlan = 3, kNumFaults = 4 };
constexpr int kStatusInts = 16;  // ints per device block (a 64-byte line)
int *device_status_block();      // api.hip: device-visible address of the current device's block (nullptr: none)

// ---- tiling of the oversampled grid --------------------------------------
// Internally every problem is 3-D with axes (a0, a1, a2); a2 is the fastest
// (last) axis of the grid.  For dim < 3 the leading axes are degenerate
// (extent 1, one tap of weight 1).  User axis u maps to internal axis u + 3 - dim.
//
// A "pencil" is a T1 x T2 cross-section in (a1, a2) swept along a0 in chunks of
// TC planes; (pencil, chunk) is the binning tile.  One workgroup sweeps a
// segment of SEG chunks of one pencil.
struct TileCfg {
    int T1, T2; // pencil cross-section (axis 1, axis 2)
    int NP;     // planes resident (LDS or registers) while a chunk is processed
    int TC;     // planes per chunk along axis 0: NP - (W - 1), so a chunk's taps fit the resident planes
};
// Narrow tiling (LDS spreading kernel): it accumulates in 8-byte cells (ds_add_f64: the 4-byte float LDS atomic is
// ~45x slower on gfx950, scripts/ubench/lds_ops.hip), which sets the budget NP * (T1+W-1) * (T2+W) * 8 B.
// Wide tiling (MFMA spreading kernel, 3-D): the padded pencil is exactly one 32 x 64 accumulator tile pair,
// T1 + W - 1 = 32 rows and T2 + W - 1 = 64 columns, and the plan is sorted by single planes ("slabs").
// Owned tiling (owner-computes variant of the MFMA spreading kernel, sparse inputs): the accumulator tile IS the owned
// region, T1 = 32, T2 = 64, and a point is entered into every tile its window touches (1, 2 or 4 plan entries).
// Paired owned tiling (problems with two or more coefficient columns): T2 = 32 -- a wave's two accumulator tiles hold the
// same 32 x 32 cells of TWO columns' grids, so that one sweep over the points (one set of operand tables) serves both.
constexpr TileCfg tile_cfg(int dim, int W, bool wide = false, bool owned = false, bool pair = false)
{
    return dim == 3 ? (owned ? TileCfg{32, pair ? 32 : 64, (W <= 16 ? 17 - W : 1) + W - 1, W <= 16 ? 17 - W : 1}
                     : wide ? TileCfg{33 - W, 65 - W, (W <= 16 ? 17 - W : 1) + W - 1, W <= 16 ? 17 - W : 1}
                     : W <= 12 ? TileCfg{16, 32, 16, 16 - (W - 1)}
                     : W <= 14 ? TileCfg{8, 32, 16, 16 - (W - 1)}
                               : TileCfg{8, 16, 20, 20 - (W - 1)})
         : dim == 2 ? TileCfg{32, 32, 1, 1}
                    : TileCfg{1, 256, 1, 1};
}
constexpr int kSegChunks = 8;   // chunks swept by one workgroup
constexpr int kSub = 8;         // edge of a sub-block (cells): one wave of the register-tile spreading kernel owns kSub x kSub cells
constexpr int kMaxW = 18;       // 2m+2 for m <= 8
constexpr int kMaxCutoff = 8;

// Spreading kernel for 3-D problems, chosen once per process by NFFT_HIP_SPREAD:
//   mfma (spread_mfma.hip)  default: per-plane outer products on the matrix cores, accumulators in registers
//                           (grids of 64^3 and up, m <= 7; everything else takes the lds kernel)
//   lds  (spread.hip)       ds_add_f64 into LDS-resident planes (also all 1-D / 2-D problems)
//   reg  (spread_reg.hip)   atomics-free register tiles, bitwise reproducible
// The choice fixes the pencil tiling and the order of the point plan, so it cannot change between calls.
enum SpreadMode { kSpreadLds = 0, kSpreadMfma = 1, kSpreadReg = 2 };
SpreadMode spread_mode();
inline bool subblock_plan_enabled() { return spread_mode() == kSpreadReg; }

struct Geom {
    int dim;      // user dimension 1..3
    int M;        // oversampled grid size per axis, 2N
    int N;
    int m;        // cutoff
    int W;        // taps per axis, 2m+2
    int Ma[3];    // extent per internal axis (1 when degenerate)
    int Wa[3];    // taps per internal axis (1 when degenerate)
    int Ta[3];    // tile extent per internal axis (Ta[0] = planes per chunk)
    int nta[3];   // tiles per internal axis (nta[0] = chunks)
    int nseg;     // segments per pencil
    int wide;     // wide (MFMA) tiling; the plan then has one bin per plane along axis 0
    int owned;    // owned variant of the wide tiling: 32 x 64 tiles without halo, a point has an entry in every tile
                  // its window touches (spreading by owner-computes: plain stores, no zero-fill, no atomics)
    int pair;     // owned tiling with 32 x 32 tiles: the spreading kernel sweeps the points once for two coefficient columns
    int bin0;     // planes per plan bin along axis 0: Ta[0], or 1 for the wide tiling
    int np0;      // plan bins per pencil along axis 0
    int tiles_per_batch;  // plan bins per point set: np0 * nta[1] * nta[2]
    int l1seg;    // first-level sort bins per pencil (segments along axis 0)
    int l1bins;   // plan bins along axis 0 per first-level segment
    // Every tile is further split into sb1 x sb2 sub-blocks of kSub x kSub cells in (axis 1, axis 2); the point
    // plan is sorted down to (tile, sub-block), so a tile's points are contiguous AND grouped by sub-block.
    // tile_offsets has one entry per (tile, sub-block): index tile * SB + s1 * sb2 + s2.
    int sb1, sb2, SB;
    // Column groups of the wide tiling (CG = 3, else 1): inside a slab the plan orders the points of a pencil by the
    // group of their window in the padded 64-column tile -- 0: inside columns [0, 32), 1: inside [16, 48), 2: inside
    // [32, 64) -- and records where groups 1 and 2 start (PlanLayout::off_groups, two ints per plan bin).  A K-block
    // or gather block of one group needs two of the tile's four 16-column k-steps / one of its two 32-column halves:
    // the matrix-core kernels skip the rest.  tile_offsets keeps one entry per slab.
    int CG;
    // floats per point of the plan's tile-ordered copy: 3-D points are 16-byte records {p0, p1, p2, index of the point
    // in the caller's arrays (int bits)} -- one aligned 16-byte access per point for every consumer, one scattered store
    // per point for the sort, no separate permutation array; 1-D / 2-D keep dim floats + the permutation array
    int pstride;
    int64_t cells; // M^dim
};

inline bool owned_supported(int dim, int64_t N, int64_t m);
// (4096 launch numbers before a row of the ring is used again: a launch would have to outlive 4096 later ones -- on other
// streams -- for two to meet in a row; 8 MB per device)
constexpr int kTicketPlanes = 256, kTicketLaunches = 4096, kTicketSlots = kTicketPlanes * kTicketLaunches;
unsigned long long *device_ticket_ring();  // api.hip: per-device ring of ticket words (WorkTickets below); nullptr on failure
unsigned next_launch_number();             // api.hip: process-wide, never 0
bool work_list_forced();       // api.hip: NFFT_HIP_WORK_LIST=1 runs every wide plan from its work list
bool column_groups_enabled();  // api.hip: NFFT_HIP_COLGROUPS=0 turns the column-group order of the plan off

inline Geom make_geom(int dim, int64_t N, int64_t m, bool owned = false, bool pair = false, bool narrow = false)
{
    Geom g;
    g.dim = dim;
    g.N = (int)N;
    g.M = (int)(2 * N);
    g.m = (int)m;
    g.W = (int)(2 * m + 2);
    g.cells = 1;
    for (int a = 0; a < 3; ++a) {
        const bool live = a >= 3 - dim;
        g.Ma[a] = live ? g.M : 1;
        g.Wa[a] = live ? g.W : 1;
        if (live) g.cells *= g.M;
    }
    // 16 waves hold 16 resident planes: the axis-0 window (2m+2 planes) has to fit
    // (`narrow`: the caller knows the problem is better served by the narrow tiling and its LDS kernels: api.hip prefer_narrow)
    g.wide = dim == 3 && spread_mode() == kSpreadMfma && g.M >= 64 && g.M <= 1024 && g.W <= 16 && !narrow;
    // at least two tiles per axis, so that a window never touches the same tile from both sides of the torus
    g.owned = owned && g.wide && g.M >= 128 && g.M % 64 == 0;
    g.pair = g.owned && pair;
    const TileCfg tc = tile_cfg(dim, g.W, g.wide != 0, g.owned != 0, g.pair != 0);
    g.Ta[0] = tc.TC;
    g.Ta[1] = tc.T1;
    g.Ta[2] = tc.T2;
    for (int a = 0; a < 3; ++a) {
        if (g.Ta[a] > g.Ma[a]) g.Ta[a] = g.Ma[a];
        g.nta[a] = (g.Ma[a] + g.Ta[a] - 1) / g.Ta[a];
    }
    g.nseg = (g.nta[0] + kSegChunks - 1) / kSegChunks;
    g.bin0 = g.wide ? 1 : g.Ta[0];
    g.np0 = g.wide ? g.Ma[0] : g.nta[0];
    g.tiles_per_batch = g.np0 * g.nta[1] * g.nta[2];
    // the wide tiling has few, long pencils: its second-level sort works on segments of 128 slabs
    g.l1bins = g.wide ? 128 : g.np0;
    g.l1seg = (g.np0 + g.l1bins - 1) / g.l1bins;
    // only the opt-in register-tile spreading kernel needs the sub-block order (it costs ~0.25 ms of sorting at C3)
    const bool sub = subblock_plan_enabled() && dim == 3 && g.M % kSub == 0 && g.Ta[1] % kSub == 0 &&
                     g.Ta[2] % kSub == 0;
    g.sb1 = sub ? g.Ta[1] / kSub : 1;
    g.sb2 = sub ? g.Ta[2] / kSub : 1;
    g.SB = g.sb1 * g.sb2;
    g.CG = (g.wide && !g.owned && g.SB == 1 && column_groups_enabled()) ? 3 : 1;
    g.pstride = dim == 3 ? 4 : dim;
    return g;
}

inline bool owned_supported(int dim, int64_t N, int64_t m) { return make_geom(dim, N, m, true).owned != 0; }

// Sparse inputs take the owner-computes spreading kernel: below ~0.026 points per grid cell the atomic flush of the
// padded tiles (and the zero-fill in front of it) costs more than spreading 1.46x as many plan entries
// (measured crossover on MI355X; NFFT_HIP_OWNED=0 / 1 forces the choice).
int owned_override();  // api.hip: -1 auto, 0 never, 1 whenever supported
// `occupied`: fraction of the grid the points are known to live in (1/8 for the fastsum geometry, else 1): what
// counts is the density where the points are.
inline bool choose_owned(int dim, int64_t N, int64_t m, int64_t n, int64_t B, double occupied)
{
    if (!owned_supported(dim, N, m)) return false;
    if (n >= (int64_t(1) << 28)) return false;  // the owned plan holds up to 4 n entries behind 32-bit offsets
    const int ov = owned_override();
    if (ov >= 0) return ov != 0;
    const double cells = 8.0 * (double)N * (double)N * (double)N * (double)(B > 0 ? B : 1) * occupied;
    return n > 0 && (double)n < 0.026 * cells;
}

// ---- device helpers --------------------------------------------------------
#if defined(__HIPCC__)

__device__ __forceinline__ void report_fault(int *status, DeviceFault kind)
{
    if (status) __hip_atomic_store(status + (int)kind, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Window constants of the reference (Gaussian, oversampling 2):
//   phi(t) = exp(-t^2 * (3 pi / 4) / m) * sqrt(0.75 / m)
// (csrc/cuda/spatial_window_operations.cu:1-28).  We evaluate exp2 on the
// pre-scaled exponent.
__device__ __forceinline__ float win_exp_scale(int m)
{
    return -(2.356194490192345f / (float)m) * 1.4426950408889634f; // -(3pi/4)/m * log2(e)
}
__device__ __forceinline__ float win_norm(int m) { return sqrtf(0.75f / (float)m); }

// Split pos*M into its integer cell in [0, M) and the fractional offset in [0, 1).
// The reference computes shift = floor(pos*M) - m in fp32 (spatial_window_operations.cu:50) and
// evaluates the window at pos*2N - shift - l formed in double (":85", the 2.0 literal), i.e. at the
// exactly rounded offset.  pos*M is exact in fp32 for power-of-two M; for other M the fma residual
// recovers the bits the fp32 product drops.
__device__ __forceinline__ void split_cell(float pos, int M, int &cell, float &frac)
{
    const float Mf = (float)M;
    // The rounded product, through an asm statement so that nothing can be fused into it.  Under -ffp-contract=fast the
    // compiler turned `hi - fl` below into fma(pos, M, -fl) -- the EXACT product again -- and adding the residual `lo`
    // counted it twice: a position error of half an ulp of pos * M, i.e. a relative error of ~6e-9 M in the transform
    // for grids that are not a power of two (1.2e-5 at M = 2000; found by scripts/fuzz_more.py).  Neither HIP's
    // __fmul_rn (a plain product) nor `#pragma clang fp contract(off)` kept the backend from fusing.  Power-of-two grids
    // take the exact fast path and never saw it.
    float hi;
    asm("v_mul_f32 %0, %1, %2" : "=v"(hi) : "v"(pos), "v"(Mf));
    // Power-of-two M (every benchmark configuration) and an ordinary coordinate: pos * M is exact, so floor, fraction
    // and the periodic wrap are one instruction each -- the same (cell, frac) as the general path below in 7 instead of
    // ~25 vector instructions (three of these per point in the sort passes, in the spreading kernel's staging and in
    // every gather block).  Anything else -- other M, |pos * M| >= 2^22, NaN -- takes the general path.
    if ((M & (M - 1)) == 0 && fabsf(hi) < 4194304.0f) {
        const float flp = floorf(hi);
        const float frp = hi - flp;  // exact; == 1 only for a negative hi below half an ulp of 1 (general path: next cell)
        if (frp < 1.0f) {
            frac = frp;
            cell = (int)flp & (M - 1);
            return;
        }
    }
    const float lo = fmaf(pos, Mf, -hi);
    float fl = floorf(hi);
    float fr = (hi - fl) + lo;
    if (fr < 0.0f) { fr += 1.0f; fl -= 1.0f; }
    if (fr >= 1.0f) { fr -= 1.0f; fl += 1.0f; }
    if (!(fr >= 0.0f && fr < 1.0f)) fr = 0.0f;  // NaN / inf input: keep indices in range
    // reduce the (possibly huge) integer part before converting: q is in [-M, 2M) whatever the rounding of the quotient
    // (then one conditional step fixes it); only garbage inputs (|pos| beyond 2^24 cells) take the integer modulo, which
    // has no hardware instruction (~25 VALU operations that every point of every kernel would pay)
    float q = fl - floorf(fl * __builtin_amdgcn_rcpf(Mf)) * Mf;  // (a quotient off by one is one step off: fixed below)
    int c = (int)q;
    if (c < 0) c += M;
    else if (c >= M) c -= M;
    if ((unsigned)c >= (unsigned)M) {
        c %= M;
        if (c < 0) c += M;
    }
    cell = c;
    frac = fr;
}

__device__ __forceinline__ int wrap(int v, int M)
{
    if ((M & (M - 1)) == 0) return v & (M - 1);  // (power-of-two grids: one instruction instead of an integer modulo)
    v %= M;
    return v < 0 ? v + M : v;
}

// wrap for values known to lie within one period of [0, M) on either side (falls back to % otherwise)
__device__ __forceinline__ int wrap_near(int v, int M)
{
    if (v < 0) v += M;
    else if (v >= M) v -= M;
    if (v < 0 || v >= M) v = wrap(v, M);
    return v;
}

// a / b for cell indices by tile sizes (a >= 0, b >= 1) without the integer division (no hardware instruction: ~30 VALU
// operations): for a < 2^17, (a + 0.5) / b lies at least 0.5 / b away from every integer and the rounding of the
// reciprocal and of the product moves it by less than 0.04 / b.  Larger indices (1-D grids beyond 2^16 points) divide.
__device__ __forceinline__ int div_small(int a, int b)
{
    if (a >= (1 << 17)) return a / b;
    return (int)(((float)a + 0.5f) * __builtin_amdgcn_rcpf((float)b));
}

// Plan bin of a point inside its batch: ((j1 * nt2 + j2) * np0 + k0); the bins of one pencil are contiguous.
__device__ __forceinline__ int tile_of_cells(const Geom &g, const int cell[3])
{
    const int k0 = g.bin0 == 1 ? cell[0] : div_small(cell[0], g.bin0);
    const int j1 = div_small(cell[1], g.Ta[1]);
    const int j2 = div_small(cell[2], g.Ta[2]);
    return (j1 * g.nta[2] + j2) * g.np0 + k0;
}

// Owned tiling: the (up to four) pencils whose 32 x 64 tile the window of a point in cell (c1, c2) touches.
// Returns their count; pencil index = j1 * nta[2] + j2.
__device__ __forceinline__ int owned_pencils(const Geom &g, int c1, int c2, int pencil[4])
{
    const int a0 = div_small(wrap_near(c1 - g.m, g.M), g.Ta[1]), a1 = div_small(wrap_near(c1 + g.m + 1, g.M), g.Ta[1]);
    const int b0 = div_small(wrap_near(c2 - g.m, g.M), g.Ta[2]), b1 = div_small(wrap_near(c2 + g.m + 1, g.M), g.Ta[2]);
    int k = 0;
    pencil[k++] = a0 * g.nta[2] + b0;
    if (b1 != b0) pencil[k++] = a0 * g.nta[2] + b1;
    if (a1 != a0) {
        pencil[k++] = a1 * g.nta[2] + b0;
        if (b1 != b0) pencil[k++] = a1 * g.nta[2] + b1;
    }
    return k;
}

// Point range of chunk k (Ta[0] planes) of a pencil: [s, e).  first_bin = index of the pencil's first plan bin.
__device__ __forceinline__ void chunk_range(const Geom &g, const int *__restrict__ tile_offsets, int first_bin, int k,
                                            int &s, int &e)
{
    const int r = g.Ta[0] / g.bin0;  // plan bins per chunk
    const int lo = k * r;
    const int hi = min(lo + r, g.np0);
    s = tile_offsets[(first_bin + lo) * g.SB];
    e = tile_offsets[(first_bin + hi) * g.SB];
}

// Column group of a point of the wide tiling whose cell lies `col` columns into its pencil (window = padded columns
// [col, col + W), W <= 16): 0 if it ends by column 32, 2 if it starts at 32 or later, else 1 (then inside [16, 48)).
__device__ __forceinline__ int column_group(const int col, const int W) { return col + W <= 32 ? 0 : (col >= 32 ? 2 : 1); }

// Sub-block of a point inside its tile: s1 * sb2 + s2 (0 when the tile is not subdivided).
__device__ __forceinline__ int sub_of_cells(const Geom &g, const int cell[3])
{
    if (g.SB == 1) return 0;
    const int s1 = (cell[1] % g.Ta[1]) / kSub;
    const int s2 = (cell[2] % g.Ta[2]) / kSub;
    return s1 * g.sb2 + s2;
}

// Dynamic hand-out of the sorted work list to the workgroups of a persistent launch: a workgroup starts with entry
// blockIdx.x and then takes the next free entry whenever it is done (the list is sorted biggest first: longest-processing-
// time-first scheduling).  One 64-bit word {launch number, entries handed out} per plane of the launch in the library's
// ring (api.hip: device_ticket_ring).  A ticket is ONE atomic add; only the first arrivals of a launch, which find an
// older launch's number in the word, claim it with a compare-and-swap (a CAS per ticket was quadratic under the
// stampede of a launch's first round: +2.3 ms at C3-clustered).
struct WorkTickets {
    unsigned long long *ring;  // nullptr: static round robin (more planes than a launch's share of the ring)
    unsigned launch;
};
__device__ __forceinline__ int take_ticket(const WorkTickets &t, const int plane_local)
{
    unsigned long long *slot = t.ring + (((t.launch & (kTicketLaunches - 1)) * kTicketPlanes) + plane_local);
    const unsigned long long mine = (unsigned long long)t.launch << 32;
    while (true) {
        const unsigned long long old = atomicAdd(slot, 1ull);
        if ((old >> 32) == t.launch) return (int)(unsigned)old;
        // an older launch's word (plus the increment just made): install {launch, 1} and take ticket 0 -- unless another
        // workgroup of this launch gets there first
        unsigned long long cur = old + 1ull;
        while ((cur >> 32) != t.launch) {
            const unsigned long long seen = atomicCAS(slot, cur, mine | 1ull);
            if (seen == cur) return 0;
            cur = seen;
        }
    }
}

// Next entry of the work list for this workgroup of a persistent launch (n_items or more: none left): its own index
// first, then tickets -- or the static round robin.  Called by all threads of the workgroup together; `word` is an LDS
// int of the workgroup.
__device__ __forceinline__ int next_work_item(const WorkTickets &t, int *word, const int prev /* < 0: first call */,
                                              const int plane_local)
{
    if (prev < 0) return (int)blockIdx.x;
    if (!t.ring) return prev + (int)gridDim.x;
    __syncthreads();  // every wave is done with the previous item (and has read the previous ticket)
    if (threadIdx.x == 0) *word = (int)gridDim.x + take_ticket(t, plane_local);
    __syncthreads();
    return *word;
}

// Entry `item` (= round * gridDim.x + blockIdx.x) of a persistent launch over the plan's sorted work list: a static
// round robin, every other round in reverse -- the workgroup that took the biggest item of one round takes the
// smallest of the next.  (A partial last round stays in order.)
__device__ __forceinline__ int4 listed_item(const int4 *__restrict__ sorted, const int item, const int n_items)
{
    const int G = (int)gridDim.x, round = item / G;
    const bool reverse = (round & 1) && (round + 1) * G <= n_items;
    return sorted[reverse ? (round + 1) * G - 1 - (int)blockIdx.x : item];
}

#endif // __HIPCC__

// ---- plan layout -----------------------------------------------------------
// [ tile_offset int32[ntiles*SB+1] | cursor int32[ntiles] | perm int32[n] | spos float[n*pstride] | scan temp | sort scratch ]
// Wide tiling, load balance of the matrix-core kernels: a pencil is swept in `runs` equal ranges of slabs (as many as
// give ~5.4 workgroups per CU for an average pencil).  Balanced inputs (every uniform one) run one workgroup per range.
// Ranges that hold far more points than average (clustered inputs) are cut further by point count at plan time, and all
// pieces go to a work list in the plan:
//   [0] = {entries, any range cut, 1 = walk the list, 0};  then one int2 {entries, first entry} per point set;
//   then the entries {point set * pencils + pencil, first slab, end slab, points} as they were produced;
//   then the same entries grouped by point set, every set's biggest first.
// ONE persistent launch walks a set's part of the sorted list instead (entries handed out by tickets, below).  Both
// launches are always enqueued; the one that is not the plan's returns at once.
constexpr int kSegMax = 32;      // most ranges per pencil
constexpr int kSegPieces = 16;   // most pieces a range is cut into
double items_per_cu();  // api.hip: work items per CU the ranges of a pencil are sized for (5.4; NFFT_HIP_ITEMS_PER_CU: tuning)
inline int64_t seg_target_points(int64_t n, int64_t nsets, int ncu)
{
    const double per_set = (double)n / (double)(nsets > 0 ? nsets : 1);
    const int64_t t = (int64_t)(per_set / (items_per_cu() * (ncu > 0 ? ncu : 256)) + 0.5);
    return t < 2048 ? 2048 : t;
}
inline int seg_base_runs(int64_t n, int64_t nsets, int64_t pencils, int M, int ncu)
{
    const double avg = (double)n / (double)((nsets > 0 ? nsets : 1) * (pencils > 0 ? pencils : 1));
    int64_t r = (int64_t)(avg / (double)seg_target_points(n, nsets, ncu) + 0.5);
    const int64_t lo = (M + 127) / 128, hi = M / 32 > lo ? M / 32 : lo;  // a range holds 32 .. 128 slabs
    r = r < lo ? lo : (r > hi ? hi : r);
    // Small problems: a range is ONE workgroup's serial chain of chunks, so few long ranges leave most CUs idle behind a long
    // chain (N = 64, 2e4 points: 18 workgroups of 128 slabs took 0.26 ms in the gather, 72 of 32 slabs 0.08):
    // as many ranges as the CUs take in ONE round (a count chosen by a cost model, rounds x (slabs + halo), also cut the ranges
    // of problems that fill the CUs more than once: measured slower, N = 64, four sets of 10^5 points 0.484 -> 0.532 ms)
    const int64_t groups = (nsets > 0 ? nsets : 1) * (pencils > 0 ? pencils : 1);
    while (r < hi && groups * (r + 1) <= (ncu > 0 ? ncu : 256)) ++r;
    return (int)(r > kSegMax ? kSegMax : r);
}
// Workgroups of the persistent launch over a plan's work list: one per CU, but no more than a point set can have entries
// (its ranges plus one cut per `target` points) -- a batch of many small point sets launches a few per plane, not 256.
inline unsigned work_list_workgroups(int64_t n, int64_t nsets, int64_t pencils, int runs, int ncu)
{
    const int64_t most = pencils * runs + n / seg_target_points(n, nsets, ncu) + 1;
    const int64_t cus = ncu > 0 ? (ncu < 1024 ? ncu : 1024) : 256;
    return (unsigned)(most < cus ? (most < 1 ? 1 : most) : cus);
}
// Work items big enough for the streamed gather (interp_stream.hip: its pipeline costs ~10 us of warm-up and tail per
// item): 7 237 points per item at config C3, the minimum of 2 048 at C5, where the lock-step kernel stays ahead
// (0.60 vs 0.83 ms); at 5e6 points (3 617 per item) the streamed kernel + column groups win by 0.1 ms per step.
int stream_min_item_points();  // api.hip: 3000, or NFFT_HIP_STREAM_MIN (tuning)
// ... and dense mid-size point sets (round 4): from 10^6 points per set at >= 0.1 points per grid cell the streamed gather and
// the column-group order win although the items hold only the minimum of 2 048 points (N = 64, 10^6 points: plan + spreading +
// gather 0.428 -> 0.402 ms, 2e6: 0.749 -> 0.678; N = 128, 4e6: 1.30 -> 1.19; a sparse 10^6 on a 512^3 grid loses: 0.89 -> 1.13)
inline bool stream_items(int64_t n, int64_t nsets, int ncu, int M)
{
    if (seg_target_points(n, nsets, ncu) >= stream_min_item_points()) return true;
    const double per_set = (double)n / (double)(nsets > 0 ? nsets : 1);
    return stream_min_item_points() == 3000 /* (not under a tuning override) */ && per_set >= 1.0e6 &&
           per_set >= 0.1 * (double)M * (double)M * (double)M;
}
int device_cu_count();  // api.hip: CU count of the current device
int current_device();
// "done once per device" flag for per-kernel set-up (hipFuncSetAttribute is per device): a process may drive several
// devices, so a plain static bool would skip the set-up on every device but the first.
constexpr int kMaxDevices = 64;
struct DeviceOnce {
    std::atomic<bool> done[kMaxDevices];
    bool first_use();
    void mark();
};

constexpr int64_t kSealBytes = 256;  // seal block of a plan: [0] checksum, [1 .. 8] verification accumulators, then their counters
struct PlanLayout {
    int64_t cap;  // entries the plan can hold: n, or 4 n for the owned tiling (an entry per touched tile)
    int64_t ntiles;
    int64_t npencils, nblocks, block_points;  // two-level sort geometry
    bool two_level;
    int64_t off_offsets, off_cursor, off_perm, off_spos, off_scan, scan_bytes;
    int64_t off_seal, off_sealpart;  // the plan's seal (checksum of pos / batch: kernels.h) and the count pass's partial sums
    int64_t off_hist, off_hscan, off_tmp, off_hist2;
    int64_t off_groups;  // column-group starts (two ints per plan bin) when `grouped`
    int64_t off_work, work_cap, work_head;  // wide tiling: work list {work_head 16-byte words: header + one int2 per point
                                            // set, work_cap entries, work_cap entries in launch order} (binning.hip)
    int64_t off_key1, off_key2;  // sort scratch: first-level bin of every point, fine key of every record (16 bits each)
    bool grouped;        // the plan is ordered by column group inside the slabs (Geom::CG == 3, two-level sort)
    int64_t total;
};
inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }
PlanLayout plan_layout(const Geom &g, int64_t n, int64_t B);

// LDS-tile kernels of the narrow tilings (spread.hip, interp.hip): workgroups that share one (pencil, segment, plane).
// Enough to start about two workgroups per CU, never fewer than ~128 points per workgroup, at most 32.
inline int point_splits(const Geom &g, const PlanLayout &L, int64_t n, int64_t nplanes)
{
    const int64_t blocks = (int64_t)g.nta[1] * g.nta[2] * g.nseg;
    if (blocks <= 0 || nplanes <= 0) return 1;
    const int64_t nsets = g.tiles_per_batch > 0 ? std::max<int64_t>(1, L.ntiles / ((int64_t)g.tiles_per_batch * g.SB)) : 1;
    int64_t s = (2 * (int64_t)device_cu_count() + blocks * nplanes - 1) / (blocks * nplanes);
    s = std::min(s, n / (nsets * blocks * 128));
    return (int)std::max<int64_t>(1, std::min<int64_t>(s, 32));
}

} // namespace nfft
