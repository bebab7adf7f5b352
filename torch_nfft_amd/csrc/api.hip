// C-ABI entry points (include/nfft_hip.h): validation, workspace carving and stage orchestration.
//
// Host drivers being replaced: nfft_adjoint_cuda (csrc/cuda/core_cuda.cu:144-336) and
// nfft_forward_cuda (:340-531) of the reference, including its validators check_point_input (:38-66),
// check_spatial_coeffs_input (:69-86), check_spectral_coeffs_input (:89-115) and
// setup_spectral_dimensions (:118-137).  Differences in behaviour, all on the safe side:
//   * nothing synchronises the device and nothing is allocated or freed here (the reference issues six
//     cudaDeviceSynchronize and four cudaMalloc/cudaFree per call);
//   * errors are returned (the reference calls exit() on a CUDA error, cuda_utils.cu:7-14);
//   * the preconditions the reference only documents (1 <= m, 2m+2 <= 2N, N even) are checked.
#include "../../include/nfft_hip.h"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace nfft {

static thread_local std::string g_last_error;
void set_error(const std::string &msg) { g_last_error = msg; }

// ---- optional stage timing -------------------------------------------------------------------
namespace {
struct TimedSpan { int stage; hipEvent_t start, stop; };
std::atomic<bool> g_profile{false};
std::atomic<unsigned> g_profile_stages{~0u};  // stages that are timed while g_profile is set (bit s = stage s)
std::mutex g_spans_mutex;  // guards g_spans / g_free_events (the entry points may be called from several threads)
std::vector<TimedSpan> g_spans;
std::vector<hipEvent_t> g_free_events;
hipEvent_t get_event()
{
    if (!g_free_events.empty()) { hipEvent_t e = g_free_events.back(); g_free_events.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
} // namespace

StageTimer::StageTimer(Stage stage, hipStream_t s) : slot(-1), stream(s)
{
    if (!g_profile.load(std::memory_order_relaxed)) return;
    if (!((g_profile_stages.load(std::memory_order_relaxed) >> (int)stage) & 1u)) return;
    std::lock_guard<std::mutex> lock(g_spans_mutex);
    TimedSpan sp{(int)stage, get_event(), get_event()};
    (void)hipEventRecord(sp.start, stream);
    slot = (int)g_spans.size();
    g_spans.push_back(sp);
}
StageTimer::~StageTimer()
{
    if (slot < 0) return;
    std::lock_guard<std::mutex> lock(g_spans_mutex);
    if (slot < (int)g_spans.size()) (void)hipEventRecord(g_spans[slot].stop, stream);
}

int current_device()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return dev;
}

// CU count of the CURRENT device (cached per device: a process may drive several)
int device_cu_count()
{
    static std::atomic<int> cache[kMaxDevices];
    const int dev = current_device();
    if (dev >= kMaxDevices) return 256;
    int v = cache[dev].load(std::memory_order_relaxed);
    if (v <= 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cache[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

bool DeviceOnce::first_use()
{
    const int dev = current_device();
    if (dev >= kMaxDevices) return true;
    return !done[dev].load(std::memory_order_acquire);
}
void DeviceOnce::mark()
{
    const int dev = current_device();
    if (dev < kMaxDevices) done[dev].store(true, std::memory_order_release);
}

// ---- device-side failure reports (common.h) ---------------------------------------------------------------
namespace {
std::mutex g_status_mutex;
int *g_status_host = nullptr;      // pinned, portable: kMaxDevices blocks of kStatusInts ints
bool g_status_failed = false;
int *status_host_blocks()
{
    std::lock_guard<std::mutex> lock(g_status_mutex);
    if (!g_status_host && !g_status_failed) {
        void *p = nullptr;
        if (hipHostMalloc(&p, sizeof(int) * kStatusInts * kMaxDevices, hipHostMallocPortable | hipHostMallocMapped) == hipSuccess && p) {
            std::memset(p, 0, sizeof(int) * kStatusInts * kMaxDevices);
            g_status_host = (int *)p;
        } else {
            (void)hipGetLastError();
            g_status_failed = true;  // no reports on this platform; the kernels get a null pointer
        }
    }
    return g_status_host;
}
const char *const kFaultText[kNumFaults] = {
    "device fault: the streamed interpolation kernel gave up waiting inside a work item (bounded spin ran out); "
    "the result of that forward transform is invalid",
    "Input mismatch: batch holds an index outside [0, batch_size) (the batch vector must be sorted with "
    "batch[-1] + 1 == batch_size)",
    "Input mismatch: the batch vector is not sorted (a row carries another index than its point set's row range, or a "
    "coefficient exceeds the largest one of its range)",
    "stale point plan: pos or batch were modified (behind the version counter) after the cached plan was built; the "
    "transform that used it is invalid -- clear or disable the plan cache (torch_nfft_amd.ops.plan_cache_clear)",
};
}  // namespace

int *device_status_block()
{
    int *host = status_host_blocks();
    const int dev = current_device();
    if (!host || dev >= kMaxDevices) return nullptr;
    void *d = nullptr;
    if (hipHostGetDevicePointer(&d, host + dev * kStatusInts, 0) != hipSuccess || !d) {
        (void)hipGetLastError();
        return nullptr;
    }
    return (int *)d;
}

// Ticket counters of the persistent launches over a plan's work list (common.h: WorkTickets): a ring of 64-bit words
// {launch number, tickets taken} per device, zeroed once.  A launch owns the 256 words of its number modulo 4096 -- one per
// plane -- and claims a word by overwriting whatever an earlier launch (4096 launches ago: long finished) left there, so
// nothing is reset between launches and the plan itself stays read-only (it may be in use on several streams).
unsigned long long *device_ticket_ring()
{
    static std::mutex mutex;
    static unsigned long long *ring[kMaxDevices] = {};
    const int dev = current_device();
    if (dev >= kMaxDevices) return nullptr;
    std::lock_guard<std::mutex> lock(mutex);
    if (!ring[dev]) {
        void *p = nullptr;
        if (hipMalloc(&p, (size_t)kTicketSlots * 8) != hipSuccess || hipMemset(p, 0, (size_t)kTicketSlots * 8) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        ring[dev] = (unsigned long long *)p;
    }
    return ring[dev];
}

unsigned next_launch_number()
{
    static std::atomic<unsigned> counter{0};
    unsigned v = ++counter;
    if (v == 0) v = ++counter;  // (0 is what the zeroed ring holds)
    // A launch takes over the ring row that the launch kTicketLaunches numbers earlier used, and must not do so while that
    // one is still running (two launches claiming one word from each other would hand work-list entries out twice).
    // Numbers come in epochs of half a ring; a device that sees a new epoch first waits for everything it has in flight:
    // whatever ran in the epoch before the previous one -- the only launches that can own this epoch's rows -- is then
    // finished.  One hipDeviceSynchronize per kTicketLaunches / 2 persistent launches and device (the reference
    // synchronises the device six times per call, cuda_utils.cu:16).
    static std::mutex mutex;
    static unsigned last_epoch[kMaxDevices] = {};
    const int dev = current_device();
    if (dev >= 0 && dev < kMaxDevices) {
        const unsigned epoch = v / (kTicketLaunches / 2) + 1;
        std::lock_guard<std::mutex> lock(mutex);
        if (last_epoch[dev] != epoch) {
            if (last_epoch[dev] != 0) (void)hipDeviceSynchronize();
            last_epoch[dev] = epoch;
        }
    }
    return v;
}

// Faults the kernels of the current device have reported since the last look: sets the error text, clears the flags and
// returns NFFT_HIP_EKERNEL (NFFT_HIP_EINVAL for a bad batch vector); 0 when there is none.
static int take_pending_fault()
{
    int *host = status_host_blocks();
    const int dev = current_device();
    if (!host || dev >= kMaxDevices) return 0;
    volatile int *blk = host + dev * kStatusInts;
    for (int k = 0; k < kNumFaults; ++k) {
        if (blk[k]) {
            blk[k] = 0;
            set_error(kFaultText[k]);
            return k == kFaultStreamStall ? NFFT_HIP_EKERNEL : NFFT_HIP_EINVAL;
        }
    }
    return 0;
}

SpreadMode spread_mode()
{
    static const SpreadMode mode = [] {
        const char *env = std::getenv("NFFT_HIP_SPREAD");
        if (env && env[0] == 'r') return kSpreadReg;
        if (env && env[0] == 'l') return kSpreadLds;
        return kSpreadMfma;
    }();
    return mode;
}

bool column_groups_enabled()
{
    static const bool on = [] {
        const char *env = std::getenv("NFFT_HIP_COLGROUPS");
        return !(env && env[0] == '0');
    }();
    return on;
}

// NFFT_HIP_WORK_LIST=1: every wide plan is run from its work list (the persistent launch), balanced or not -- for tests
// and for timing the two forms of the matrix-core kernels against each other
bool work_list_forced()
{
    static const bool on = [] {
        const char *env = std::getenv("NFFT_HIP_WORK_LIST");
        return env && env[0] == '1';
    }();
    return on;
}

double items_per_cu()
{
    static const double v = [] {
        const char *env = std::getenv("NFFT_HIP_ITEMS_PER_CU");
        const double t = env ? std::atof(env) : 0.0;
        return t >= 1.0 && t <= 64.0 ? t : 5.4;
    }();
    return v;
}

int stream_min_item_points()
{
    static const int v = [] {
        const char *env = std::getenv("NFFT_HIP_STREAM_MIN");
        const int t = env ? std::atoi(env) : 0;
        return t > 0 ? t : 3000;
    }();
    return v;
}

// NFFT_HIP_OWNED_PAIR=0: owned plans keep the 32 x 64 tiles and the one-column sweeps for every column count
bool owned_pair_enabled()
{
    static const bool on = [] {
        const char *env = std::getenv("NFFT_HIP_OWNED_PAIR");
        return !(env && env[0] == '0');
    }();
    return on;
}

int owned_override()
{
    static const int v = [] {
        const char *env = std::getenv("NFFT_HIP_OWNED");
        if (env && env[0] == '0') return 0;
        if (env && env[0] == '1') return 1;
        return -1;
    }();
    return v;
}

namespace {

// The point plan of a problem: the tile-sorted points (pencils with halo: what the interpolation kernels and the
// scatter spreading kernels walk) and, for sparse 3-D problems, a second sort by owned 32 x 64 tiles with an entry
// per touched tile for the owner-computes spreading kernel (common.h choose_owned), stored behind the first.
struct PlanSet {
    Geom g;
    PlanLayout L;
    bool owned;
    Geom go;
    PlanLayout Lo;
    int64_t off_own, total;
    const Geom &spread_geom() const { return owned ? go : g; }
    const PlanLayout &spread_layout() const { return owned ? Lo : L; }
    const void *spread_plan(const void *plan) const { return owned ? (const char *)plan + off_own : (const char *)plan; }
};
// Geometry of a problem's (halo-tiling) plan.  The column-group order of the plan (common.h Geom::CG) pays where the
// streamed gather runs, i.e. for big work items; below that it only costs sorting time (+0.03 ms at 10^6 points).
// The 64^3 grid (N = 32, the reference's default bandwidth: torch_nfft/nfft.py:150-156) is 3 x 2 pencils of the wide tiling,
// two of which hold 70 % of the cells: the matrix-core kernels' work items (ranges of slabs of one pencil, at most 16 pieces
// per range) cannot spread a dense point set over 256 CUs -- 10^6 points, m = 3: 1.46 ms per adjoint + forward against 0.51
// with the narrow tiling, whose LDS kernels share a (pencil, segment) among up to 32 workgroups.  What the narrow path pays is
// one LDS atomic per window tap: it wins for the narrow windows and for few points (sweep: profiles/r04_experiments.md).
// NFFT_HIP_SMALL_NARROW=0 keeps the wide tiling.
bool prefer_narrow(const nfft_hip_problem *p)
{
    static const bool off = [] {
        const char *env = std::getenv("NFFT_HIP_SMALL_NARROW");
        return env && env[0] == '0';
    }();
    if (off || p->dim != 3) return false;
    // 128^3 grid, narrow window, many point sets: every (set, pencil) is a chain of nearly empty K-blocks (2e4 points per set:
    // 9 per pencil and slab), and from eight sets up there are more chains than CUs -- 16 sets x 2e4 points, m = 2: 1.17 ms
    // against 0.90 on the narrow tiling; with four sets the matrix-core path is still ahead (0.42 against 0.45)
    // (one column only: with eight columns and 2e4 points per set the paired owner-computes spreading is 5x ahead of the LDS
    // kernel, 0.24 against 1.28 ms)
    if (p->N == 64) return p->m <= 3 && p->batch_size >= 8 && p->num_columns == 1;
    if (p->N != 32) return false;
    return p->m <= 3 || p->num_points <= 30000;
}
Geom problem_geom(const nfft_hip_problem *p)
{
    Geom g = make_geom(p->dim, p->N, p->m, false, false, prefer_narrow(p));
    if (g.CG > 1 && !stream_items(p->num_points, p->batch_size, device_cu_count(), g.M)) g.CG = 1;
    return g;
}
PlanSet plan_set(const nfft_hip_problem *p)
{
    PlanSet ps;
    ps.g = problem_geom(p);
    ps.L = plan_layout(ps.g, p->num_points, p->batch_size);
    ps.owned = choose_owned(p->dim, p->N, p->m, p->num_points, p->batch_size,
                            (p->flags & NFFT_HIP_POINTS_IN_QUARTER_BALL) ? 0.125 : 1.0);
    // (a single column on a 128^3 grid: what the owner-computes kernel saves -- 10 us of zero-fill, the atomics of a few
    // thousand K-blocks -- is less than its second sort costs: plan 0.074 against 0.038 ms at 2e4 points, round 4)
    if (p->N < 128 && p->num_columns < 2 && owned_override() < 0) ps.owned = false;
    if (!ps.g.wide) ps.owned = false;  // (the narrow tiling was preferred: no matrix-core spreading, no owned plan)
    ps.go = ps.g;
    ps.Lo = ps.L;
    ps.off_own = 0;
    ps.total = ps.L.total;
    if (ps.owned) {
        // two or more coefficient columns: 32 x 32 tiles, one sweep of the points per PAIR of columns (spread_mfma.hip)
        ps.go = make_geom(p->dim, p->N, p->m, true, p->num_columns >= 2 && owned_pair_enabled());
        ps.Lo = plan_layout(ps.go, p->num_points, p->batch_size);
        ps.off_own = align_up(ps.L.total, 256);
        ps.total = ps.off_own + ps.Lo.total;
    }
    return ps;
}
int build_plans(const PlanSet &ps, const float *pos, const int64_t *batch, int64_t n, int64_t B, void *plan, hipStream_t s)
{
    if (int rc = launch_plan_points(ps.g, ps.L, pos, batch, n, B, plan, s)) return rc;
    if (ps.owned) return launch_plan_points(ps.go, ps.Lo, pos, batch, n, B, (char *)plan + ps.off_own, s);
    return 0;
}

// interpolation: matrix-core kernels for the wide 3-D tiling (the wave-per-column one from 4 real columns up) unless
// NFFT_HIP_GATHER=lds (lane-per-point kernel) or =mfma (always the plane-ring kernel)
int gather_any(const Geom &g, const PlanLayout &L, const void *plan, const float *grid, int64_t n, int64_t Cr,
               int64_t plane0, int64_t nplanes, float *yr, hipStream_t s)
{
    static const bool lds_only = [] {
        const char *env = std::getenv("NFFT_HIP_GATHER");
        return env && env[0] == 'l';
    }();
    if (!lds_only && interp_cols_supported(g, Cr)) return launch_interp_cols(g, L, plan, grid, n, Cr, plane0, nplanes, yr, s);
    if (!lds_only && interp_stream_supported(g) && interp_stream_pays(g, L, n)) return launch_interp_stream(g, L, plan, grid, n, Cr, plane0, nplanes, yr, s);
    if (!lds_only && interp_mfma_supported(g)) return launch_interp_mfma(g, L, plan, grid, n, Cr, plane0, nplanes, yr, s);
    return launch_interp(g, L, plan, grid, n, Cr, plane0, nplanes, yr, s);
}

int validate(const nfft_hip_problem *p)
{
    if (!p) { set_error("Input mismatch: null problem"); return NFFT_HIP_EINVAL; }
    if (p->dim < 1 || p->dim > 3) { set_error("Input mismatch: dim must be 1, 2 or 3"); return NFFT_HIP_EINVAL; }
    if (p->num_points < 0 || p->num_columns < 0 || p->batch_size < 1) {
        set_error("Input mismatch: negative size");
        return NFFT_HIP_EINVAL;
    }
    if (p->N < 2 || (p->N & 1)) { set_error("Input mismatch: bandwidth N must be even and >= 2"); return NFFT_HIP_EINVAL; }
    if (p->m < 1 || p->m > kMaxCutoff) { set_error("Input mismatch: cutoff m must be in 1..8"); return NFFT_HIP_EINVAL; }
    if (2 * p->m + 2 > 2 * p->N) { set_error("Input mismatch: window 2m+2 exceeds the oversampled grid 2N"); return NFFT_HIP_EINVAL; }
    if (p->N > (1 << 20)) { set_error("Input mismatch: bandwidth too large"); return NFFT_HIP_EINVAL; }
    if (p->num_points >= (int64_t(1) << 31)) { set_error("Input mismatch: too many points"); return NFFT_HIP_EINVAL; }
    {
        // the point plan indexes its (point set, tile) bins with 32-bit integers
        const Geom g = problem_geom(p);
        if ((double)g.tiles_per_batch * (double)p->batch_size * (double)g.SB >= 2.0e9) {
            set_error("Input mismatch: too many point sets for this grid (plan bins exceed 2^31)");
            return NFFT_HIP_EINVAL;
        }
    }
    return 0;
}

int64_t grid_budget_bytes()
{
    // upper bound for (real grid + half spectrum) of one chunk of planes; the rest of the 288 GB stays free
    // for the caller.  Overridable for tests of the chunked path.
    const char *env = std::getenv("NFFT_HIP_CHUNK_BYTES");
    if (env) {
        const long long v = std::atoll(env);
        if (v > 0) return (int64_t)v;
    }
    return int64_t(16) << 30;
}

struct Carve {
    PlanSet ps;
    Geom g;        // = ps.g
    PlanLayout L;  // = ps.L
    int64_t n, B, C, Cr, total_planes, chunk_planes;
    int64_t half_cells;
    bool colfft;  // pruned column passes (colfft.hip) instead of the full dim-dimensional rocFFT transform
    int64_t off_plan, off_xs, off_xmax, off_grid, off_spec, off_col, off_work, work_bytes, total;
};

bool spread_reg_enabled() { return subblock_plan_enabled(); }

// The matrix-core spreading kernel reads one or two real coefficient columns in place: every plan record carries the
// index of its point in the caller's arrays and the staging pipeline fetches x[index] two steps ahead of its use -- no
// permutation pass (0.18 ms at C3 as a kernel of its own, a 23 us latency-bound prologue per work item inside the
// spreading kernel).  With more columns one pass (gather_rows) reads every row once for all of them.
// NFFT_HIP_XGATHER=1 always runs the separate pass.
bool spread_permutes(const Geom &g, int64_t Cr)
{
    static const bool off = [] {
        const char *env = std::getenv("NFFT_HIP_XGATHER");
        return env && env[0] == '1';
    }();
    if (off || Cr > 2) return false;
    // (the LDS-tile kernel of the narrow tilings reads x through the plan's permutation as well; the opt-in register-tile
    // kernel does not)
    return spread_mfma_supported(g) || !(spread_reg_supported(g) && spread_reg_enabled());
}

// xs: the planar copy in plan order; xr: nullptr when the caller has filled xs, else what spread_permutes() reads;
// xmax: per-plane largest |x| (matrix-core kernel only: launch_plane_absmax)
int spread_any(const Geom &g, const PlanLayout &L, const void *plan, const float *xr, float *xs, const unsigned *xmax,
               int64_t n, int64_t Cr, int64_t p0, int64_t np, float *grid, hipStream_t s)
{
    if (spread_mfma_supported(g)) {
        // (the owner-computes variant writes every cell itself)
        if (!g.owned) { StageTimer t(kStageZero, s); NFFT_HIP_CHECK(hipMemsetAsync(grid, 0, (size_t)(np * g.cells * 4), s)); }
        StageTimer t(kStageSpread, s);
        return launch_spread_mfma(g, L, plan, xr, xs, xmax, n, Cr, p0, np, grid, s);
    }
    if (spread_reg_supported(g) && spread_reg_enabled()) {
        StageTimer t(kStageSpread, s);
        return launch_spread_reg(g, L, plan, xs, n, Cr, p0, np, grid, s);
    }
    { StageTimer t(kStageZero, s); NFFT_HIP_CHECK(hipMemsetAsync(grid, 0, (size_t)(np * g.cells * 4), s)); }
    StageTimer t(kStageSpread, s);
    return launch_spread(g, L, plan, xr, xs, n, Cr, p0, np, grid, s);
}

// own pruned row passes instead of rocFFT's for the contiguous axis (NFFT_HIP_ROCFFT_ROWS=1 keeps rocFFT)
bool own_row_passes(const Geom &g)
{
    static const bool off = [] {
        const char *env = std::getenv("NFFT_HIP_ROCFFT_ROWS");
        return env && env[0] == '1';
    }();
    return !off && rowfft_supported(g);
}

// Several coefficient columns: column-innermost passes (no planar copy, no layout transposes) for chunks of at least two
// groups of planes; NFFT_HIP_COL_PLANAR=1 keeps the planar passes + transposes of rounds 1-2.
bool column_innermost(const Geom &g, int64_t C, int64_t nplanes)
{
    static const bool off = [] {
        const char *env = std::getenv("NFFT_HIP_COL_PLANAR");
        return env && env[0] == '1';
    }();
    return !off && C > 1 && nplanes >= 32 && colfft_ci_supported(g);
}

bool colfft_enabled()
{
    const char *env = std::getenv("NFFT_HIP_NO_COLFFT");
    return !(env && env[0] == '1');
}

// planes_per_col: 2 when a column owns a (re, im) pair of real planes, else 1
int make_carve(const nfft_hip_problem *p, int planes_per_col, bool need_xs, FftKind kind, Carve &c)
{
    c.ps = plan_set(p);
    c.g = c.ps.g;
    c.L = c.ps.L;
    c.n = p->num_points;
    c.B = p->batch_size;
    c.C = p->num_columns;
    c.Cr = c.C * planes_per_col;
    c.total_planes = c.B * c.Cr;
    c.half_cells = (int64_t)(c.g.M / 2 + 1);
    for (int a = 0; a < 2; ++a) c.half_cells *= c.g.Ma[a];
    c.colfft = colfft_supported(c.g) && colfft_enabled();
    const FftKind fkind = c.colfft ? (kind == kR2C ? kR2CRows : kC2RRows) : kind;
    const int64_t plane_bytes = c.g.cells * 4 + c.half_cells * 8 + (c.colfft ? colfft_scratch_bytes(c.g, 1) : 0);
    // (the budget is a soft one: the group padding of the column-innermost passes, <= 15 planes of scratch, comes on top)
    int64_t chunk = grid_budget_bytes() / plane_bytes;
    chunk -= chunk % planes_per_col;
    if (chunk < planes_per_col) chunk = planes_per_col;
    if (chunk > c.total_planes) chunk = c.total_planes;
    if (chunk > 32768) chunk = 32768 - 32768 % planes_per_col;  // blockIdx.y limit
    c.chunk_planes = chunk;
    c.work_bytes = 0;
    if (c.total_planes > 0) {
        // plans for the full chunk and for the remainder chunk
        int64_t w = fft_work_bytes(fkind, c.g.dim, c.g.M, chunk);
        if (w < 0) return NFFT_HIP_EFFT;
        c.work_bytes = w;
        const int64_t rem = c.total_planes % chunk;
        if (rem) {
            w = fft_work_bytes(fkind, c.g.dim, c.g.M, rem);
            if (w < 0) return NFFT_HIP_EFFT;
            if (w > c.work_bytes) c.work_bytes = w;
        }
    }
    int64_t o = 0;
    c.off_plan = o; o = align_up(o + c.ps.total, 256);
    c.off_xs = o;   o = align_up(o + (need_xs ? (align_up(c.ps.spread_layout().cap * c.Cr, 64) + 64) * 4 : 0), 256);
    c.off_xmax = o; o = align_up(o + (need_xs ? c.total_planes * 4 : 0), 256);
    c.off_grid = o; o = align_up(o + chunk * c.g.cells * 4, 256);
    c.off_spec = o; o = align_up(o + chunk * c.half_cells * 8, 256);
    // (several columns: the column-innermost passes work on whole groups of 16 planes)
    c.off_col = o;  o = align_up(o + (c.colfft ? colfft_scratch_bytes(c.g, c.C > 1 ? colfft_ci_planes(chunk) : chunk) : 0), 256);
    c.off_work = o; o = align_up(o + c.work_bytes, 256);
    c.total = o + 256;
    return 0;
}

} // namespace
} // namespace nfft

using namespace nfft;

extern "C" {

int nfft_hip_abi_version(void) { return NFFT_HIP_ABI_VERSION; }
const char *nfft_hip_last_error(void) { return g_last_error.c_str(); }

int nfft_hip_check_status(void *stream, int synchronize)
{
    if (synchronize) NFFT_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return take_pending_fault();
}

void nfft_hip_profile_enable(int enable)
{
    g_profile.store(enable != 0);
}

void nfft_hip_profile_stages(unsigned stage_mask)
{
    g_profile_stages.store(stage_mask);
}

int nfft_hip_profile_collect(double *ms_per_stage, int64_t *launches_per_stage, int num_stages)
{
    for (int i = 0; i < num_stages; ++i) { ms_per_stage[i] = 0.0; launches_per_stage[i] = 0; }
    std::lock_guard<std::mutex> lock(g_spans_mutex);
    for (const TimedSpan &sp : g_spans) {
        float ms = 0.f;
        NFFT_HIP_CHECK(hipEventSynchronize(sp.stop));
        NFFT_HIP_CHECK(hipEventElapsedTime(&ms, sp.start, sp.stop));
        if (sp.stage < num_stages) { ms_per_stage[sp.stage] += ms; launches_per_stage[sp.stage] += 1; }
        g_free_events.push_back(sp.start);
        g_free_events.push_back(sp.stop);
    }
    g_spans.clear();
    return 0;
}

int64_t nfft_hip_adjoint_workspace_bytes(const nfft_hip_problem *p, int x_is_complex, int real_output)
{
    (void)real_output;
    if (validate(p)) return -1;
    Carve c;
    if (make_carve(p, x_is_complex ? 2 : 1, true, kR2C, c)) return -1;
    return c.total;
}

int64_t nfft_hip_forward_workspace_bytes(const nfft_hip_problem *p, int x_is_complex, int real_output)
{
    (void)x_is_complex;
    if (validate(p)) return -1;
    Carve c;
    if (make_carve(p, real_output ? 1 : 2, false, kC2R, c)) return -1;
    return c.total;
}

int64_t nfft_hip_plan_bytes(const nfft_hip_problem *p)
{
    if (validate(p)) return -1;
    return plan_set(p).total;
}

int nfft_hip_plan_points(const nfft_hip_problem *p, const float *pos, const int64_t *batch, void *plan,
                         int64_t plan_bytes, void *stream)
{
    if (int rc = take_pending_fault()) return rc;  // a kernel of an earlier call gave up: say so
    if (int rc = validate(p)) return rc;
    const PlanSet ps = plan_set(p);
    if (!plan || plan_bytes < ps.total) { set_error("plan buffer too small"); return NFFT_HIP_EWORKSPACE; }
    if (p->num_points > 0 && !pos) { set_error("Input mismatch: pos is null"); return NFFT_HIP_EINVAL; }
    StageTimer t(kStagePlan, (hipStream_t)stream);
    return build_plans(ps, pos, batch, p->num_points, p->batch_size, plan, (hipStream_t)stream);
}

int nfft_hip_plan_verify(const nfft_hip_problem *p, const float *pos, const int64_t *batch, void *plan, void *stream)
{
    if (int rc = validate(p)) return rc;
    if (!plan || (p->num_points > 0 && !pos)) { set_error("Input mismatch: null plan or pos"); return NFFT_HIP_EINVAL; }
    static std::atomic<unsigned> slot{0};  // eight verifications of one plan may be in flight (on different streams)
    return launch_points_verify(pos, batch, p->num_points, p->dim, (char *)plan + plan_set(p).L.off_seal, (int)(slot++ & 7u),
                                (hipStream_t)stream);
}

int64_t nfft_hip_spread_scratch_bytes(const nfft_hip_problem *p, int64_t real_columns)
{
    if (validate(p) || real_columns < 0) return -1;
    // plan-ordered copy of the coefficients + one word per plane (its largest |x|)
    return (align_up(plan_set(p).spread_layout().cap * real_columns, 64) + 64 + align_up(p->batch_size * real_columns, 64)) * 4;
}

int nfft_hip_spread(const nfft_hip_problem *p, const void *plan, const float *xr, int64_t real_columns, float *grid,
                    float *scratch, void *stream)
{
    if (int rc = take_pending_fault()) return rc;  // a kernel of an earlier call gave up: say so
    if (int rc = validate(p)) return rc;
    const PlanSet ps = plan_set(p);
    const Geom &g = ps.spread_geom();
    const PlanLayout &L = ps.spread_layout();
    const void *const halo_plan = plan;
    plan = ps.spread_plan(plan);
    hipStream_t s = (hipStream_t)stream;
    const int64_t planes = p->batch_size * real_columns;
    if (planes > 32768) { set_error("Input mismatch: too many planes for one spread call"); return NFFT_HIP_EINVAL; }
    unsigned *xmax = (unsigned *)(scratch + align_up(L.cap * real_columns, 64) + 64);
    if (spread_mfma_supported(g))
        if (int rc = launch_plane_absmax(ps.g, ps.L, halo_plan, xr, p->num_points, p->batch_size, real_columns, xmax, s)) return rc;
    if (spread_permutes(g, real_columns)) return spread_any(g, L, plan, xr, scratch, xmax, p->num_points, real_columns, 0, planes, grid, s);
    if (int rc = launch_gather_rows(g, L, plan, p->num_points, xr, real_columns, scratch, s)) return rc;
    return spread_any(g, L, plan, nullptr, scratch, xmax, p->num_points, real_columns, 0, planes, grid, s);
}

int nfft_hip_interpolate(const nfft_hip_problem *p, const void *plan, const float *grid, int64_t real_columns,
                         float *yr, void *stream)
{
    if (int rc = take_pending_fault()) return rc;  // a kernel of an earlier call gave up: say so
    if (int rc = validate(p)) return rc;
    const Geom g = problem_geom(p);
    const PlanLayout L = plan_layout(g, p->num_points, p->batch_size);
    const int64_t planes = p->batch_size * real_columns;
    if (planes > 32768) { set_error("Input mismatch: too many planes for one interpolate call"); return NFFT_HIP_EINVAL; }
    return gather_any(g, L, plan, grid, p->num_points, real_columns, 0, planes, yr, (hipStream_t)stream);
}

static int adjoint_impl(const nfft_hip_problem *p, const float *pos, const int64_t *batch, const void *ext_plan,
                        const void *x, int x_is_complex, int real_output, void *y, void *workspace,
                        int64_t workspace_bytes, void *stream, const void *mult = nullptr, int mult_kind = 0)
{
    if (int rc = take_pending_fault()) return rc;  // a kernel of an earlier call gave up: say so
    if (int rc = validate(p)) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (!ext_plan && small_grid_supported(p)) {
        // grid in one workgroup's LDS: one kernel, no plan, no workspace (smallgrid.hip)
        if (p->batch_size * p->num_columns == 0) return 0;
        if (!y) { set_error("Input mismatch: y is null"); return NFFT_HIP_EINVAL; }
        if (p->num_points > 0 && (!pos || !x)) { set_error("Input mismatch: null input"); return NFFT_HIP_EINVAL; }
        StageTimer t(kStageSpread, s);
        return launch_small_grid_adjoint(p, pos, batch, x, x_is_complex, real_output, y, mult, mult_kind, s);
    }
    const int ppc = x_is_complex ? 2 : 1;
    Carve c;
    if (int rc = make_carve(p, ppc, true, kR2C, c)) return rc;
    if (c.total_planes == 0) return 0;
    if (!y) { set_error("Input mismatch: y is null"); return NFFT_HIP_EINVAL; }
    if (c.n > 0 && ((!pos && !ext_plan) || !x)) { set_error("Input mismatch: null input"); return NFFT_HIP_EINVAL; }
    if (!workspace || workspace_bytes < c.total) { set_error("workspace too small"); return NFFT_HIP_EWORKSPACE; }
    char *ws = (char *)(((uintptr_t)workspace + 255) & ~uintptr_t(255));
    const void *plan = ext_plan;
    float *xs = (float *)(ws + c.off_xs);
    float *grid = (float *)(ws + c.off_grid);
    float2 *spec = (float2 *)(ws + c.off_spec);
    void *work = ws + c.off_work;

    if (!ext_plan) {
        StageTimer t(kStagePlan, s);
        if (int rc = build_plans(c.ps, pos, batch, c.n, c.B, ws + c.off_plan, s)) return rc;
        plan = ws + c.off_plan;
    }
    const Geom &gs = c.ps.spread_geom();
    const PlanLayout &Ls = c.ps.spread_layout();
    const void *plan_s = c.ps.spread_plan(plan);
    const bool fused = spread_permutes(gs, c.Cr);
    unsigned *xmax = (unsigned *)(ws + c.off_xmax);
    if (spread_mfma_supported(gs)) {
        // operand scales of the matrix-core kernel: one streaming pass over x for all planes of the call
        StageTimer t(kStageGather, s);
        if (int rc = launch_plane_absmax(c.g, c.L, plan, (const float *)x, c.n, c.B, c.Cr, xmax, s)) return rc;
    }
    if (!fused) { StageTimer t(kStageGather, s); if (int rc = launch_gather_rows(gs, Ls, plan_s, c.n, (const float *)x, c.Cr, xs, s)) return rc; }
    for (int64_t p0 = 0; p0 < c.total_planes; p0 += c.chunk_planes) {
        const int64_t np = std::min(c.chunk_planes, c.total_planes - p0);
        if (int rc = spread_any(gs, Ls, plan_s, fused ? (const float *)x : nullptr, xs, xmax, c.n, c.Cr, p0, np, grid, s)) return rc;
        if (c.colfft) {
            const bool own_rows = own_row_passes(c.g);
            if (own_rows && column_innermost(c.g, c.C, np)) {
                // several columns: the planes travel in groups of 16, plane index innermost, and the last pass writes the
                // reference's [B, N^3, C] layout directly
                { StageTimer t(kStageFft, s); if (int rc = launch_row_r2c_ci(c.g, grid, np, spec, s)) return rc; }
                StageTimer t(kStageDeconv, s);
                if (int rc = launch_colfft_adjoint_ci(c.g, spec, ws + c.off_col, c.C, x_is_complex, real_output, p0, np, y, mult, mult_kind, s)) return rc;
                continue;
            }
            if (own_rows) { StageTimer t(kStageFft, s); if (int rc = launch_row_r2c(c.g, grid, ws + c.off_col, c.chunk_planes, np, spec, s)) return rc; }
            else { StageTimer t(kStageFft, s); if (int rc = fft_execute(kR2CRows, c.g.dim, c.g.M, np, grid, spec, work, c.work_bytes, s)) return rc; }
            if (c.C > 1) {
                // several columns: the passes write a planar copy (into the grid buffer, free by now), one tiled
                // transpose brings it into the reference's column-interleaved layout
                StageTimer t(kStageDeconv, s);
                const int ppc_a = x_is_complex ? 2 : 1;
                const int64_t K = c.g.dim == 2 ? c.g.N * (int64_t)c.g.N : c.g.N * (int64_t)c.g.N * c.g.N;
                if (int rc = launch_colfft_adjoint(c.g, spec, own_rows, ws + c.off_col, c.chunk_planes, 1, x_is_complex, real_output, 0, np, grid, mult, mult_kind, s)) return rc;
                if (int rc = launch_column_layout(true, grid, y, K, c.C, p0 / ppc_a, np / ppc_a, real_output ? 4 : 8, s)) return rc;
            } else {
                StageTimer t(kStageDeconv, s);
                if (int rc = launch_colfft_adjoint(c.g, spec, own_rows, ws + c.off_col, c.chunk_planes, c.C, x_is_complex, real_output, p0, np, y, mult, mult_kind, s)) return rc;
            }
        } else {
            { StageTimer t(kStageFft, s); if (int rc = fft_execute(kR2C, c.g.dim, c.g.M, np, grid, spec, work, c.work_bytes, s)) return rc; }
            { StageTimer t(kStageDeconv, s); if (int rc = launch_deconv_adjoint(c.g, spec, c.C, x_is_complex, real_output, p0, np, y, mult, mult_kind, s)) return rc; }
        }
    }
    return 0;
}

static int forward_impl(const nfft_hip_problem *p, const float *pos, const int64_t *batch, const void *ext_plan,
                        const void *xhat, int x_is_complex, int real_output, void *y, void *workspace,
                        int64_t workspace_bytes, void *stream)
{
    if (int rc = take_pending_fault()) return rc;  // a kernel of an earlier call gave up: say so
    if (int rc = validate(p)) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (!ext_plan && small_grid_supported(p)) {
        if (p->batch_size * p->num_columns == 0 || p->num_points == 0) return 0;
        if (!y || !pos || !xhat) { set_error("Input mismatch: null input"); return NFFT_HIP_EINVAL; }
        StageTimer t(kStageInterp, s);
        return launch_small_grid_forward(p, pos, batch, xhat, x_is_complex, real_output, y, s);
    }
    const int ppc = real_output ? 1 : 2;
    Carve c;
    if (int rc = make_carve(p, ppc, false, kC2R, c)) return rc;
    if (c.total_planes == 0 || c.n == 0) return 0;
    if (!y || (!pos && !ext_plan) || !xhat) { set_error("Input mismatch: null input"); return NFFT_HIP_EINVAL; }
    if (!workspace || workspace_bytes < c.total) { set_error("workspace too small"); return NFFT_HIP_EWORKSPACE; }
    char *ws = (char *)(((uintptr_t)workspace + 255) & ~uintptr_t(255));
    const void *plan = ext_plan;
    float *grid = (float *)(ws + c.off_grid);
    float2 *spec = (float2 *)(ws + c.off_spec);
    void *work = ws + c.off_work;

    if (!ext_plan) {
        // (only the first sort: the forward transform never spreads)
        StageTimer t(kStagePlan, s);
        if (int rc = launch_plan_points(c.g, c.L, pos, batch, c.n, c.B, ws + c.off_plan, s)) return rc;
        plan = ws + c.off_plan;
    }
    for (int64_t p0 = 0; p0 < c.total_planes; p0 += c.chunk_planes) {
        const int64_t np = std::min(c.chunk_planes, c.total_planes - p0);
        if (c.colfft) {
            const bool own_rows = own_row_passes(c.g);
            if (own_rows && column_innermost(c.g, c.C, np)) {
                { StageTimer t(kStageDeconv, s); if (int rc = launch_colfft_forward_ci(c.g, xhat, spec, ws + c.off_col, c.C, x_is_complex, real_output, p0, np, s)) return rc; }
                { StageTimer t(kStageFft, s); if (int rc = launch_row_c2r_ci(c.g, spec, np, grid, s)) return rc; }
            } else {
                if (c.C > 1) {
                    // several columns: planar copy of this chunk's columns first (the grid buffer is free until the row pass)
                    StageTimer t(kStageDeconv, s);
                    const int64_t K = c.g.dim == 2 ? c.g.N * (int64_t)c.g.N : c.g.N * (int64_t)c.g.N * c.g.N;
                    if (int rc = launch_column_layout(false, xhat, grid, K, c.C, p0 / ppc, np / ppc, x_is_complex ? 8 : 4, s)) return rc;
                    if (int rc = launch_colfft_forward(c.g, grid, ws + c.off_col, c.chunk_planes, 1, x_is_complex, real_output, 0, np, spec, own_rows, s)) return rc;
                } else {
                    StageTimer t(kStageDeconv, s);
                    if (int rc = launch_colfft_forward(c.g, xhat, ws + c.off_col, c.chunk_planes, c.C, x_is_complex, real_output, p0, np, spec, own_rows, s)) return rc;
                }
                if (own_rows) { StageTimer t(kStageFft, s); if (int rc = launch_row_c2r(c.g, spec, ws + c.off_col, c.chunk_planes, np, grid, s)) return rc; }
                else { StageTimer t(kStageFft, s); if (int rc = fft_execute(kC2RRows, c.g.dim, c.g.M, np, spec, grid, work, c.work_bytes, s)) return rc; }
            }
        } else {
            { StageTimer t(kStageDeconv, s); if (int rc = launch_deconv_forward(c.g, xhat, c.C, x_is_complex, real_output, p0, np, spec, s)) return rc; }
            { StageTimer t(kStageFft, s); if (int rc = fft_execute(kC2R, c.g.dim, c.g.M, np, spec, grid, work, c.work_bytes, s)) return rc; }
        }
        { StageTimer t(kStageInterp, s); if (int rc = gather_any(c.g, c.L, plan, grid, c.n, c.Cr, p0, np, (float *)y, s)) return rc; }
    }
    return 0;
}

int nfft_hip_adjoint(const nfft_hip_problem *p, const float *pos, const void *x, int x_is_complex,
                     const int64_t *batch, int real_output, void *y, void *workspace, int64_t workspace_bytes,
                     void *stream)
{
    return adjoint_impl(p, pos, batch, nullptr, x, x_is_complex, real_output, y, workspace, workspace_bytes, stream);
}

int nfft_hip_forward(const nfft_hip_problem *p, const float *pos, const void *xhat, int x_is_complex,
                     const int64_t *batch, int real_output, void *y, void *workspace, int64_t workspace_bytes,
                     void *stream)
{
    return forward_impl(p, pos, batch, nullptr, xhat, x_is_complex, real_output, y, workspace, workspace_bytes, stream);
}

int nfft_hip_plan_needed(const nfft_hip_problem *p)
{
    if (validate(p)) return 1;
    return small_grid_supported(p) ? 0 : 1;
}

int nfft_hip_adjoint_planned(const nfft_hip_problem *p, const void *plan, const void *x, int x_is_complex,
                             int real_output, void *y, void *workspace, int64_t workspace_bytes, void *stream)
{
    if (!plan) { set_error("Input mismatch: plan is null"); return NFFT_HIP_EINVAL; }
    return adjoint_impl(p, nullptr, nullptr, plan, x, x_is_complex, real_output, y, workspace, workspace_bytes, stream);
}

int nfft_hip_forward_planned(const nfft_hip_problem *p, const void *plan, const void *xhat, int x_is_complex,
                             int real_output, void *y, void *workspace, int64_t workspace_bytes, void *stream)
{
    if (!plan) { set_error("Input mismatch: plan is null"); return NFFT_HIP_EINVAL; }
    return forward_impl(p, nullptr, nullptr, plan, xhat, x_is_complex, real_output, y, workspace, workspace_bytes, stream);
}

// ---- fast summation -------------------------------------------------------------------------
namespace {
struct FastsumCarve {
    int64_t band_bytes, plan_s, plan_t, inner, off_band, off_plan_s, off_plan_t, off_inner, total;
};
int fastsum_check(const nfft_hip_problem *src, const nfft_hip_problem *tgt)
{
    if (int rc = validate(src)) return rc;
    if (int rc = validate(tgt)) return rc;
    if (src->dim != tgt->dim || src->N != tgt->N || src->m != tgt->m || src->batch_size != tgt->batch_size ||
        src->num_columns != tgt->num_columns) {
        set_error("Input mismatch: sources and targets disagree in dim / N / m / batch size / columns");
        return NFFT_HIP_EINVAL;
    }
    return 0;
}
int fastsum_carve(const nfft_hip_problem *src, const nfft_hip_problem *tgt, int x_is_complex, bool own_plans,
                  bool shared_points, FastsumCarve &f)
{
    Carve a, b;
    if (int rc = make_carve(src, x_is_complex ? 2 : 1, true, kR2C, a)) return rc;
    if (int rc = make_carve(tgt, x_is_complex ? 2 : 1, false, kC2R, b)) return rc;
    int64_t band = src->batch_size * src->num_columns * 8;
    for (int d = 0; d < src->dim; ++d) band *= src->N;
    f.band_bytes = align_up(band, 256);
    f.plan_s = own_plans ? align_up(a.ps.total, 256) : 0;
    f.plan_t = own_plans && !shared_points ? align_up(b.ps.total, 256) : 0;
    f.inner = std::max(a.total, b.total);
    int64_t o = 0;
    f.off_band = o;   o += f.band_bytes;
    f.off_plan_s = o; o += f.plan_s;
    f.off_plan_t = o; o += f.plan_t;
    f.off_inner = o;  o += f.inner;
    f.total = o + 256;
    return 0;
}
int fastsum_impl(const nfft_hip_problem *src_in, const float *sources, const int64_t *source_batch,
                 const void *source_plan, const nfft_hip_problem *tgt_in, const float *targets,
                 const int64_t *target_batch, const void *target_plan, const void *x, int x_is_complex,
                 const void *coeffs, int coeffs_are_complex, void *y, void *workspace, int64_t workspace_bytes,
                 void *stream)
{
    if (!src_in || !tgt_in) { set_error("Input mismatch: null problem"); return NFFT_HIP_EINVAL; }
    // fastsum geometry: every point within radius 1/4 (the kernel is only defined there)
    nfft_hip_problem src_q = *src_in, tgt_q = *tgt_in;
    src_q.flags |= NFFT_HIP_POINTS_IN_QUARTER_BALL;
    tgt_q.flags |= NFFT_HIP_POINTS_IN_QUARTER_BALL;
    const nfft_hip_problem *src = &src_q, *tgt = &tgt_q;
    if (int rc = fastsum_check(src, tgt)) return rc;
    const bool own_plans = source_plan == nullptr;
    const bool shared = own_plans ? (sources == targets && source_batch == target_batch && src->num_points == tgt->num_points)
                                  : (source_plan == target_plan);
    FastsumCarve f;
    if (int rc = fastsum_carve(src, tgt, x_is_complex, own_plans, shared, f)) return rc;
    if (tgt->num_points == 0 || src->num_columns == 0) return 0;
    if (!coeffs || !y) { set_error("Input mismatch: null input"); return NFFT_HIP_EINVAL; }
    if (!workspace || workspace_bytes < f.total) { set_error("workspace too small"); return NFFT_HIP_EWORKSPACE; }
    char *ws = (char *)(((uintptr_t)workspace + 255) & ~uintptr_t(255));
    hipStream_t s = (hipStream_t)stream;
    // problems whose grid fits one workgroup's LDS: adjoint (with the kernel's coefficients folded into its roll-off)
    // and forward transform are one fused kernel each on the caller's points -- no plans (smallgrid.hip)
    const bool fused1d = own_plans && small_grid_supported(src) && small_grid_supported(tgt);
    if (fused1d) {
        if (src->num_points > 0 && !sources) { set_error("Input mismatch: sources is null"); return NFFT_HIP_EINVAL; }
        if (!targets) { set_error("Input mismatch: targets is null"); return NFFT_HIP_EINVAL; }
        void *band1 = ws + f.off_band;
        if (src->num_points == 0) {
            NFFT_HIP_CHECK(hipMemsetAsync(band1, 0, (size_t)f.band_bytes, s));
        } else {
            if (int rc = adjoint_impl(src, sources, source_batch, nullptr, x, x_is_complex, 0, band1, nullptr, 0, stream,
                                      coeffs, coeffs_are_complex ? 2 : 1)) return rc;
        }
        return forward_impl(tgt, targets, target_batch, nullptr, band1, 1, x_is_complex ? 0 : 1, y, nullptr, 0, stream);
    }
    if (own_plans) {
        if (src->num_points > 0 && !sources) { set_error("Input mismatch: sources is null"); return NFFT_HIP_EINVAL; }
        if (!targets) { set_error("Input mismatch: targets is null"); return NFFT_HIP_EINVAL; }
        {
            StageTimer t(kStagePlan, s);
            if (int rc = build_plans(plan_set(src), sources, source_batch, src->num_points, src->batch_size,
                                     ws + f.off_plan_s, s)) return rc;
        }
        source_plan = ws + f.off_plan_s;
        target_plan = source_plan;
        if (!shared) {
            StageTimer t(kStagePlan, s);
            if (int rc = build_plans(plan_set(tgt), targets, target_batch, tgt->num_points, tgt->batch_size,
                                     ws + f.off_plan_t, s)) return rc;
            target_plan = ws + f.off_plan_t;
        }
    }
    void *band = ws + f.off_band;
    if (src->num_points == 0) {
        NFFT_HIP_CHECK(hipMemsetAsync(band, 0, (size_t)f.band_bytes, s));
    } else {
        // adjoint of the sources; the kernel's Fourier coefficients are multiplied in by the last spectral pass
        if (int rc = adjoint_impl(src, nullptr, nullptr, source_plan, x, x_is_complex, 0, band, ws + f.off_inner, f.inner,
                                  stream, coeffs, coeffs_are_complex ? 2 : 1)) return rc;
    }
    // forward transform at the targets; real coefficients give a real result (core_cuda.cu:817-821)
    return forward_impl(tgt, nullptr, nullptr, target_plan, band, 1, x_is_complex ? 0 : 1, y, ws + f.off_inner, f.inner,
                        stream);
}
} // namespace

int64_t nfft_hip_fastsum_workspace_bytes(const nfft_hip_problem *src, const nfft_hip_problem *tgt, int x_is_complex,
                                         int shared_points, int planned)
{
    if (!src || !tgt) return -1;
    nfft_hip_problem s = *src, t = *tgt;
    s.flags |= NFFT_HIP_POINTS_IN_QUARTER_BALL;
    t.flags |= NFFT_HIP_POINTS_IN_QUARTER_BALL;
    if (fastsum_check(&s, &t)) return -1;
    FastsumCarve f;
    if (fastsum_carve(&s, &t, x_is_complex, planned == 0, shared_points != 0, f)) return -1;
    return f.total;
}

int nfft_hip_fastsum(const nfft_hip_problem *src, const float *sources, const int64_t *source_batch,
                     const nfft_hip_problem *tgt, const float *targets, const int64_t *target_batch, const void *x,
                     int x_is_complex, const void *coeffs, int coeffs_are_complex, void *y, void *workspace,
                     int64_t workspace_bytes, void *stream)
{
    return fastsum_impl(src, sources, source_batch, nullptr, tgt, targets, target_batch, nullptr, x, x_is_complex, coeffs,
                        coeffs_are_complex, y, workspace, workspace_bytes, stream);
}

int nfft_hip_fastsum_planned(const nfft_hip_problem *src, const void *source_plan, const nfft_hip_problem *tgt,
                             const void *target_plan, const void *x, int x_is_complex, const void *coeffs,
                             int coeffs_are_complex, void *y, void *workspace, int64_t workspace_bytes, void *stream)
{
    if (!source_plan || !target_plan) { set_error("Input mismatch: plan is null"); return NFFT_HIP_EINVAL; }
    return fastsum_impl(src, nullptr, nullptr, source_plan, tgt, nullptr, nullptr, target_plan, x, x_is_complex, coeffs,
                        coeffs_are_complex, y, workspace, workspace_bytes, stream);
}

} // extern "C"
