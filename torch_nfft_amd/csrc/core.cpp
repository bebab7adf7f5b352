// core.so -- the native operator registry of the package: the eight torch_nfft::* operators of the reference
// (csrc/core.cpp:43-121, 176-184; prototypes csrc/core.h:6-65), registered from C++ and loaded by the package's
// __init__ with torch.ops.load_library exactly like the reference's core.so (torch_nfft/__init__.py:11).
//
// This file is host glue only: input checks and error texts of the reference's validators
// (csrc/cuda/core_cuda.cu:38-137), output allocation from torch's caching allocator, the one blocking read of
// batch[-1] (core_cuda.cu:60), the point-plan cache, and the call into the C ABI of include/nfft_hip.h
// (libnfft_hip.so), which does all the arithmetic on torch's current stream.  There is no CPU path.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/core/DeviceGuard.h>
#include <hip/hip_runtime_api.h>
#include <torch/library.h>

#include <mutex>
#include <vector>

#include "../../include/nfft_hip.h"

namespace {

#define CHECK_INPUT(cond) TORCH_CHECK((cond), "Input mismatch")  // csrc/cuda/cuda_utils.cu:3

void check_rc(int rc)
{
    if (rc == NFFT_HIP_OK) return;
    std::string msg = nfft_hip_last_error();
    if (rc == NFFT_HIP_EINVAL && msg.rfind("Input mismatch", 0) != 0) msg = "Input mismatch: " + msg;
    TORCH_CHECK(false, msg);
}

void *stream_of(const at::Tensor &t)
{
    return (void *)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream();
}

at::Tensor byte_buffer(int64_t nbytes, const at::Tensor &like)
{
    return at::empty({nbytes}, like.options().dtype(at::kByte));
}

struct Points {
    int dim;
    int64_t n, B;
    at::Tensor pos, batch;  // contiguous; batch undefined when the caller passed None
};

// ---- ends of the batch vector -------------------------------------------------------------------------------
// B = batch[-1] + 1 needs a blocking read-back (as in the reference, core_cuda.cu:60): ~35 us per operator call -- more
// than a small transform takes, and paid again by the forward transform, the backward pass and the next step on the very
// same vector.  The two ends of the last few batch vectors are remembered by tensor identity + version counter: the same
// rules, the same limitation (writes that bypass the version counter) and the same switches as the point-plan cache
// below (ops.plan_cache_enabled / plan_cache_clear).  Inference tensors carry no version counter: always read back.
struct BatchEnds {
    const void *ptr = nullptr;
    int64_t version = -1, n = -1, first = 0, last = 0;
    int device = -1;
    // The entry belongs to ONE tensor object, held weakly: it neither pins the vector's memory (80 MB at 10^7 points, four
    // entries) nor survives the tensor -- an address the allocator hands to another tensor is a miss, not a stale hit.
    c10::weak_intrusive_ptr<c10::TensorImpl> owner{c10::intrusive_ptr<c10::TensorImpl>()};
    uint64_t last_use = 0;
    bool owned_by(const at::Tensor &t) const
    {
        const auto alive = owner.lock();
        return alive && alive.get() == t.unsafeGetTensorImpl();
    }
};
struct BatchEndsCache {
    std::mutex mutex;
    BatchEnds entries[4];
    bool enabled = true;
    uint64_t tick = 0;
    void clear() { for (BatchEnds &e : entries) e = BatchEnds(); }
};
BatchEndsCache &g_ends = *new BatchEndsCache;  // (never destroyed: see the plan cache)

void batch_ends(const at::Tensor &batch, int64_t n, int64_t &first, int64_t &last)
{
    const bool cacheable = !batch.is_inference();
    const void *ptr = batch.data_ptr();
    const int device = batch.device().index();
    int64_t version = -1;
    if (cacheable) {
        std::lock_guard<std::mutex> lock(g_ends.mutex);
        if (g_ends.enabled) {
            version = (int64_t)batch._version();
            for (BatchEnds &e : g_ends.entries) {
                if (e.ptr == ptr && e.version == version && e.n == n && e.device == device && e.owned_by(batch)) {
                    e.last_use = ++g_ends.tick;
                    first = e.first;
                    last = e.last;
                    return;
                }
            }
        }
    }
    const at::Tensor ends = at::stack({batch[0], batch[n - 1]}).cpu();
    first = ends[0].item<int64_t>();
    last = ends[1].item<int64_t>();
    if (cacheable && version >= 0) {
        std::lock_guard<std::mutex> lock(g_ends.mutex);
        if (!g_ends.enabled) return;
        BatchEnds *slot = &g_ends.entries[0];
        for (BatchEnds &e : g_ends.entries)
            if (e.last_use < slot->last_use) slot = &e;
        slot->ptr = ptr; slot->version = version; slot->n = n; slot->device = device;
        slot->first = first; slot->last = last; slot->last_use = ++g_ends.tick;
        slot->owner = c10::weak_intrusive_ptr<c10::TensorImpl>(batch.getIntrusivePtr());
    }
}

// check_point_input (core_cuda.cu:38-66).  One blocking read-back, as in the reference (:60) -- it also fetches
// batch[0], so that a negative first entry of the (sorted) batch vector is rejected instead of being clamped -- unless
// the ends of this very vector are remembered (batch_ends above).
Points check_points(const at::Tensor &pos, const c10::optional<at::Tensor> &opt_batch, const char *batch_name)
{
    TORCH_CHECK(pos.is_cuda(), "pos must be CUDA tensor");
    CHECK_INPUT(pos.dim() == 2);
    CHECK_INPUT(pos.scalar_type() == at::kFloat);
    Points p;
    p.n = pos.size(0);
    p.dim = (int)pos.size(1);
    CHECK_INPUT(p.dim >= 1 && p.dim <= 3);
    p.pos = pos.contiguous();
    p.B = 1;
    if (opt_batch.has_value() && opt_batch->defined()) {
        const at::Tensor &batch = *opt_batch;
        TORCH_CHECK(batch.is_cuda(), batch_name, " must be CUDA tensor");
        CHECK_INPUT(batch.dim() == 1);
        CHECK_INPUT(batch.scalar_type() == at::kLong);
        CHECK_INPUT(batch.numel() == p.n);
        CHECK_INPUT(batch.device() == pos.device());
        p.batch = batch.contiguous();
        if (p.n > 0) {
            int64_t first = 0, last = 0;
            batch_ends(p.batch, p.n, first, last);
            CHECK_INPUT(first >= 0 && last >= first);
            p.B = last + 1;
        }
    }
    return p;
}

bool real_dtype(const at::Tensor &t)
{
    if (t.scalar_type() == at::kFloat) return true;
    CHECK_INPUT(t.scalar_type() == at::kComplexFloat);
    return false;
}

nfft_hip_problem problem(const Points &p, int64_t C, int64_t N, int64_t m, int32_t flags = 0)
{
    nfft_hip_problem q;
    q.dim = p.dim;
    q.flags = flags;
    q.num_points = p.n;
    q.num_columns = C;
    q.batch_size = p.B;
    q.N = N;
    q.m = m;
    return q;
}

// ---- point-plan cache -------------------------------------------------------------------------------------
// The tile-sorted copy of the points depends only on (pos, batch, N, m).  Adjoint <-> forward pairs on the same
// points (autograd backward, a forward fed by an adjoint, fastsum -- the reference exploits
// sources.is_same(targets), core_cuda.cu:552-564) reuse it instead of re-binning.  Two entries (sources and
// targets of a fastsum), least recently used out.  Keyed on tensor identity + version counter: in-place edits
// through the tensor invalidate a plan; writes that bypass the version counter (pos.data, foreign kernels,
// DLPack aliases) do NOT -- callers who do that turn the cache off (torch_nfft_amd.ops.plan_cache_enabled(False))
// or clear it.  A plan is built on one stream and may be consumed on another: the consumer's stream waits for the
// build event and the plan's storage is recorded on it, so the allocator does not recycle it early.
struct PlanKey {
    const void *pos_ptr = nullptr, *batch_ptr = nullptr;
    int64_t pos_version = -1, batch_version = -1, n = -1, B = -1, N = -1, m = -1;
    int dim = 0, device = -1, flags = 0;  // (a plan is only valid for the geometry hints it was built with)
    bool operator==(const PlanKey &o) const
    {
        return pos_ptr == o.pos_ptr && batch_ptr == o.batch_ptr && pos_version == o.pos_version &&
               batch_version == o.batch_version && n == o.n && B == o.B && N == o.N && m == o.m && dim == o.dim &&
               device == o.device && flags == o.flags;
    }
};
struct PlanEntry {
    PlanKey key;
    at::Tensor plan, pos, batch;  // pos / batch kept alive so that their addresses cannot be recycled
    void *stream = nullptr;
    hipEvent_t built = nullptr;
    uint64_t last_use = 0;
};
struct PlanCache {
    std::mutex mutex;
    PlanEntry entries[2];
    bool enabled = true;
    bool verify = true;  // check a cached plan's seal on every hit (plan_cache_control 5 / 6)
    int64_t hits = 0, misses = 0;
    uint64_t tick = 0;
    void clear()
    {
        for (PlanEntry &e : entries) {
            if (e.built) (void)hipEventDestroy(e.built);
            e = PlanEntry();
        }
    }
};
// (never destroyed: at process exit the tensors it holds would otherwise be released after torch's allocator and
// the HIP runtime have been torn down)
PlanCache &g_cache = *new PlanCache;

// Cache key of a point set, or std::nullopt-like `cacheable == false` for points the cache must not hold.
struct PlanLookup {
    PlanKey key;
    bool use_cache = false;
    void *stream = nullptr;
};

PlanLookup plan_lookup_key(const Points &p, const nfft_hip_problem &q)
{
    PlanLookup lk;
    lk.stream = stream_of(p.pos);
    // Inference tensors carry no version counter (Tensor::_version() throws): such points are planned afresh in
    // every call and never enter the cache -- an in-place edit could not be told from the cached state.
    const bool cacheable = !p.pos.is_inference() && !(p.batch.defined() && p.batch.is_inference());
    PlanKey &key = lk.key;
    key.pos_ptr = p.pos.data_ptr();
    key.batch_ptr = p.batch.defined() ? p.batch.data_ptr() : nullptr;
    key.n = p.n; key.B = p.B; key.N = q.N; key.m = q.m; key.dim = p.dim; key.device = p.pos.device().index();
    // (the owner-computes spreading plan of a sparse problem has another tiling from two columns up: api.hip plan_set)
    key.flags = q.flags | (q.num_columns >= 2 ? (1 << 30) : 0);
    lk.use_cache = g_cache.enabled && cacheable;  // (read under the lock by the callers below)
    if (lk.use_cache) {
        key.pos_version = (int64_t)p.pos._version();
        key.batch_version = p.batch.defined() ? (int64_t)p.batch._version() : -1;
    }
    return lk;
}

// a cached plan for these points, made safe to use on the caller's stream; undefined tensor on a miss (lock held)
at::Tensor cache_find(const PlanLookup &lk)
{
    if (!lk.use_cache) return at::Tensor();
    for (PlanEntry &e : g_cache.entries) {
        if (e.plan.defined() && e.key == lk.key) {
            ++g_cache.hits;
            e.last_use = ++g_cache.tick;
            if (e.stream != lk.stream) {
                TORCH_CHECK(hipStreamWaitEvent((hipStream_t)lk.stream, e.built, 0) == hipSuccess,
                            "hipStreamWaitEvent failed");
                e.plan.record_stream(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(lk.key.device));
            }
            return e.plan;
        }
    }
    return at::Tensor();
}

// enter a plan that has just been enqueued on lk.stream (lock held)
void cache_insert(const PlanLookup &lk, const Points &p, const at::Tensor &plan)
{
    if (!lk.use_cache) return;
    PlanEntry *slot = &g_cache.entries[0];
    if (g_cache.entries[0].plan.defined() &&
        (!g_cache.entries[1].plan.defined() || g_cache.entries[1].last_use < g_cache.entries[0].last_use))
        slot = &g_cache.entries[1];
    if (slot->built && slot->key.device != lk.key.device) {  // events belong to the device they were created on
        (void)hipEventDestroy(slot->built);
        slot->built = nullptr;
    }
    if (!slot->built)
        TORCH_CHECK(hipEventCreateWithFlags(&slot->built, hipEventDisableTiming) == hipSuccess, "hipEventCreate failed");
    TORCH_CHECK(hipEventRecord(slot->built, (hipStream_t)lk.stream) == hipSuccess, "hipEventRecord failed");
    slot->key = lk.key;
    slot->plan = plan;
    slot->pos = p.pos;
    slot->batch = p.batch;
    slot->stream = lk.stream;
    slot->last_use = ++g_cache.tick;
}

at::Tensor new_plan_buffer(const Points &p, const nfft_hip_problem &q, int64_t &nbytes)
{
    nbytes = nfft_hip_plan_bytes(&q);
    if (nbytes < 0) check_rc(NFFT_HIP_EINVAL);
    return byte_buffer(nbytes, p.pos);
}

at::Tensor get_plan(const Points &p, const nfft_hip_problem &q)
{
    std::lock_guard<std::mutex> lock(g_cache.mutex);
    const PlanLookup lk = plan_lookup_key(p, q);
    const float *pos = p.pos.data_ptr<float>();
    const int64_t *batch = p.batch.defined() ? p.batch.data_ptr<int64_t>() : nullptr;
    at::Tensor plan = cache_find(lk);
    if (plan.defined()) {
        // A plan from an earlier call: identity + version say the points are the same, the seal checks that they ARE (a
        // write behind the version counter -- pos.data, a foreign kernel, a DLPack alias -- would otherwise give a wrong
        // transform silently; the reference recomputes per call, core_cuda.cu:188-211).  One streaming pass over pos /
        // batch on the caller's stream; a mismatch is reported like every device-side fault: the next operator raises.
        if (g_cache.verify) check_rc(nfft_hip_plan_verify(&q, pos, batch, plan.data_ptr(), lk.stream));
        return plan;
    }
    ++g_cache.misses;
    int64_t nbytes = 0;
    plan = new_plan_buffer(p, q, nbytes);
    check_rc(nfft_hip_plan_points(&q, pos, batch, plan.data_ptr(), nbytes, lk.stream));
    cache_insert(lk, p, plan);
    return plan;
}

// action: 0 clear, 1 enable, 2 disable (and clear), 3 -> hits, 4 -> misses, 5 / 6 seal verification on / off
int64_t plan_cache_control(int64_t action)
{
    std::lock_guard<std::mutex> lock(g_cache.mutex);
    if (action >= 0 && action <= 2) {  // the remembered batch-vector ends follow the same switches
        std::lock_guard<std::mutex> lock2(g_ends.mutex);
        g_ends.clear();
        if (action != 0) g_ends.enabled = action == 1;
    }
    switch (action) {
    case 0: g_cache.clear(); return 0;
    case 1: g_cache.enabled = true; return 0;
    case 2: g_cache.enabled = false; g_cache.clear(); return 0;
    case 3: return g_cache.hits;
    case 4: return g_cache.misses;
    case 5: g_cache.verify = true; return 0;
    case 6: g_cache.verify = false; return 0;  // (for callers who guarantee they never write behind the version counter)
    }
    TORCH_CHECK(false, "unknown plan cache action");
}

// Faults a kernel reported since the last look (include/nfft_hip.h nfft_hip_check_status): raises, or returns 0.
// synchronize != 0 drains the current stream of the current device first.
int64_t check_status(int64_t synchronize)
{
    int dev = 0;
    TORCH_CHECK(hipGetDevice(&dev) == hipSuccess, "hipGetDevice failed");
    void *stream = (void *)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA((c10::DeviceIndex)dev).stream();
    check_rc(nfft_hip_check_status(stream, synchronize != 0 ? 1 : 0));
    return 0;
}

// ---- operators ----------------------------------------------------------------------------------------------
// torch_nfft::nfft_adjoint (csrc/core.cpp:43-55; driver core_cuda.cu:144-336)
at::Tensor nfft_adjoint(at::Tensor pos, at::Tensor x, c10::optional<at::Tensor> opt_batch, int64_t N, int64_t m,
                        int64_t real_output)
{
    TORCH_CHECK(x.is_cuda(), "torch_nfft.nfft_adjoint is currently only implemented for GPU tensors");
    const Points p = check_points(pos, opt_batch, "(*out_batch)");
    const bool real_input = real_dtype(x);  // check_spatial_coeffs_input, core_cuda.cu:69-86
    CHECK_INPUT(x.dim() >= 1);
    CHECK_INPUT(x.size(0) == p.n);
    CHECK_INPUT(x.device() == pos.device());
    int64_t C = 1;
    for (int64_t d = 1; d < x.dim(); ++d) C *= x.size(d);
    std::vector<int64_t> shape{p.B};  // core_cuda.cu:298-304
    for (int d = 0; d < p.dim; ++d) shape.push_back(N);
    for (int64_t d = 1; d < x.dim(); ++d) shape.push_back(x.size(d));
    at::Tensor y = at::empty(shape, x.options().dtype(real_output ? at::kFloat : at::kComplexFloat));
    if (y.numel() == 0) return y;
    const at::Tensor xc = x.contiguous();
    const nfft_hip_problem q = problem(p, C, N, m);
    c10::DeviceGuard guard(x.device());
    if (!nfft_hip_plan_needed(&q)) {  // one fused kernel on the caller's points: no plan, no workspace
        check_rc(nfft_hip_adjoint(&q, p.pos.data_ptr<float>(), xc.data_ptr(), real_input ? 0 : 1,
                                  p.batch.defined() ? p.batch.data_ptr<int64_t>() : nullptr, real_output ? 1 : 0,
                                  y.data_ptr(), nullptr, 0, stream_of(x)));
        return y;
    }
    const int64_t ws_bytes = nfft_hip_adjoint_workspace_bytes(&q, real_input ? 0 : 1, real_output ? 1 : 0);
    if (ws_bytes < 0) check_rc(std::string(nfft_hip_last_error()).rfind("Input mismatch", 0) == 0 ? NFFT_HIP_EINVAL : NFFT_HIP_EFFT);
    at::Tensor ws = byte_buffer(ws_bytes, x);
    const at::Tensor plan = get_plan(p, q);
    check_rc(nfft_hip_adjoint_planned(&q, plan.data_ptr(), xc.data_ptr(), real_input ? 0 : 1, real_output ? 1 : 0,
                                      y.data_ptr(), ws.data_ptr(), ws_bytes, stream_of(x)));
    return y;
}

// torch_nfft::nfft_forward (csrc/core.cpp:94-105; driver core_cuda.cu:340-531)
at::Tensor nfft_forward(at::Tensor pos, at::Tensor x, c10::optional<at::Tensor> opt_batch, int64_t m, int64_t real_output)
{
    TORCH_CHECK(x.is_cuda(), "torch_nfft.nfft_forward is currently only implemented for GPU tensors");
    const Points p = check_points(pos, opt_batch, "(*out_batch)");
    const bool real_input = real_dtype(x);  // check_spectral_coeffs_input, core_cuda.cu:89-115
    CHECK_INPUT(x.dim() >= p.dim + 1);
    CHECK_INPUT(x.size(0) == p.B);
    CHECK_INPUT(x.device() == pos.device());
    const int64_t N = x.size(1);
    CHECK_INPUT(N >= 2);
    for (int d = 2; d <= p.dim; ++d) CHECK_INPUT(x.size(d) == N);
    int64_t C = 1;
    std::vector<int64_t> shape{p.n};
    for (int64_t d = p.dim + 1; d < x.dim(); ++d) {
        C *= x.size(d);
        shape.push_back(x.size(d));
    }
    at::Tensor y = at::empty(shape, x.options().dtype(real_output ? at::kFloat : at::kComplexFloat));
    if (y.numel() == 0) return y;
    const at::Tensor xc = x.contiguous();
    const nfft_hip_problem q = problem(p, C, N, m);
    c10::DeviceGuard guard(x.device());
    if (!nfft_hip_plan_needed(&q)) {
        check_rc(nfft_hip_forward(&q, p.pos.data_ptr<float>(), xc.data_ptr(), real_input ? 0 : 1,
                                  p.batch.defined() ? p.batch.data_ptr<int64_t>() : nullptr, real_output ? 1 : 0,
                                  y.data_ptr(), nullptr, 0, stream_of(x)));
        return y;
    }
    const int64_t ws_bytes = nfft_hip_forward_workspace_bytes(&q, real_input ? 0 : 1, real_output ? 1 : 0);
    if (ws_bytes < 0) check_rc(std::string(nfft_hip_last_error()).rfind("Input mismatch", 0) == 0 ? NFFT_HIP_EINVAL : NFFT_HIP_EFFT);
    at::Tensor ws = byte_buffer(ws_bytes, x);
    const at::Tensor plan = get_plan(p, q);
    check_rc(nfft_hip_forward_planned(&q, plan.data_ptr(), xc.data_ptr(), real_input ? 0 : 1, real_output ? 1 : 0,
                                      y.data_ptr(), ws.data_ptr(), ws_bytes, stream_of(x)));
    return y;
}

// torch_nfft::nfft_fastsum (csrc/core.cpp:108-121; driver core_cuda.cu:535-852)
at::Tensor nfft_fastsum(at::Tensor sources, at::Tensor targets, at::Tensor x, at::Tensor coeffs,
                        c10::optional<at::Tensor> opt_source_batch, c10::optional<at::Tensor> opt_target_batch, int64_t m)
{
    TORCH_CHECK(x.is_cuda(), "torch_nfft.nfft_fastsum is currently only implemented for GPU tensors");
    TORCH_CHECK(coeffs.is_cuda(), "coeffs must be CUDA tensor");
    const Points ps = check_points(sources, opt_source_batch, "(*out_batch)");
    const bool same_tensor = sources.is_same(targets);
    const bool same_batch = (!opt_source_batch.has_value() && !opt_target_batch.has_value()) ||
                            (opt_source_batch.has_value() && opt_target_batch.has_value() &&
                             opt_source_batch->is_same(*opt_target_batch));
    const bool shared = same_tensor && same_batch;  // core_cuda.cu:552-564
    const Points pt = shared ? ps : check_points(targets, opt_target_batch, "(*out_batch)");
    CHECK_INPUT(pt.dim == ps.dim);
    CHECK_INPUT(pt.B == ps.B);  // core_cuda.cu:566-568
    CHECK_INPUT(coeffs.dim() == ps.dim);  // core_cuda.cu:585-590
    const int64_t N = coeffs.size(0);
    for (int d = 1; d < ps.dim; ++d) CHECK_INPUT(coeffs.size(d) == N);
    const bool real_coeffs = real_dtype(coeffs);
    const bool real_input = real_dtype(x);
    CHECK_INPUT(x.dim() >= 1);
    CHECK_INPUT(x.size(0) == ps.n);
    CHECK_INPUT(x.device() == sources.device() && targets.device() == sources.device() &&
                coeffs.device() == sources.device());
    int64_t C = 1;
    std::vector<int64_t> shape{pt.n};
    for (int64_t d = 1; d < x.dim(); ++d) {
        C *= x.size(d);
        shape.push_back(x.size(d));
    }
    at::Tensor y = at::empty(shape, x.options());  // same dtype as x (core_cuda.cu:817-821)
    if (y.numel() == 0) return y;
    const at::Tensor xc = x.contiguous(), cc = coeffs.contiguous();
    const nfft_hip_problem qs = problem(ps, C, N, m, NFFT_HIP_POINTS_IN_QUARTER_BALL),
                           qt = problem(pt, C, N, m, NFFT_HIP_POINTS_IN_QUARTER_BALL);
    c10::DeviceGuard guard(x.device());
    const bool planned = nfft_hip_plan_needed(&qs) != 0 || nfft_hip_plan_needed(&qt) != 0;
    const int64_t ws_bytes = nfft_hip_fastsum_workspace_bytes(&qs, &qt, real_input ? 0 : 1, shared ? 1 : 0, planned ? 1 : 0);
    if (ws_bytes < 0) check_rc(std::string(nfft_hip_last_error()).rfind("Input mismatch", 0) == 0 ? NFFT_HIP_EINVAL : NFFT_HIP_EFFT);
    at::Tensor ws = byte_buffer(ws_bytes, x);
    if (!planned) {  // two fused kernels on the caller's points (1-D, grid in LDS): no plans
        check_rc(nfft_hip_fastsum(&qs, ps.pos.data_ptr<float>(), ps.batch.defined() ? ps.batch.data_ptr<int64_t>() : nullptr,
                                  &qt, pt.pos.data_ptr<float>(), pt.batch.defined() ? pt.batch.data_ptr<int64_t>() : nullptr,
                                  xc.data_ptr(), real_input ? 0 : 1, cc.data_ptr(), real_coeffs ? 0 : 1, y.data_ptr(),
                                  ws.data_ptr(), ws_bytes, stream_of(x)));
        return y;
    }
    const at::Tensor plan_s = get_plan(ps, qs);
    const at::Tensor plan_t = shared ? plan_s : get_plan(pt, qt);
    check_rc(nfft_hip_fastsum_planned(&qs, plan_s.data_ptr(), &qt, plan_t.data_ptr(), xc.data_ptr(), real_input ? 0 : 1,
                                      cc.data_ptr(), real_coeffs ? 0 : 1, y.data_ptr(), ws.data_ptr(), ws_bytes,
                                      stream_of(x)));
    return y;
}

// coefficient operators (csrc/core.cpp:124-171; drivers core_cuda.cu:855-1064): outputs live on the current device
at::TensorOptions current_device_options(at::ScalarType dtype)
{
    int dev = 0;
    TORCH_CHECK(hipGetDevice(&dev) == hipSuccess, "hipGetDevice failed");
    return at::TensorOptions().device(c10::Device(c10::kCUDA, (c10::DeviceIndex)dev)).dtype(dtype);
}

std::vector<int64_t> cube(int64_t N, int64_t dim)
{
    CHECK_INPUT(dim >= 1 && dim <= 3 && N >= 2);
    return std::vector<int64_t>((size_t)dim, N);
}

at::Tensor coeffs_workspace(int64_t N, int64_t dim, const at::Tensor &like, int64_t &nbytes)
{
    nbytes = nfft_hip_coeffs_workspace_bytes(N, (int32_t)dim);
    if (nbytes < 0) check_rc(std::string(nfft_hip_last_error()).rfind("Input mismatch", 0) == 0 ? NFFT_HIP_EINVAL : NFFT_HIP_EFFT);
    return byte_buffer(nbytes, like);
}

at::Tensor gaussian_analytic_coeffs(double sigma, int64_t N, int64_t dim)
{
    at::Tensor out = at::empty(cube(N, dim), current_device_options(at::kFloat));
    check_rc(nfft_hip_gaussian_analytic_coeffs(sigma, N, (int32_t)dim, out.data_ptr<float>(), stream_of(out)));
    return out;
}

at::Tensor gaussian_interpolated_coeffs(double sigma, int64_t N, int64_t dim, int64_t p, double eps)
{
    TORCH_CHECK(p <= 0, "Gaussian interpolated coeffs are currently only implemented for p<=0");    // core_cuda.cu:890
    TORCH_CHECK(eps == 0.0, "Gaussian interpolated coeffs are currently only implemented for eps=0");  // :891
    at::Tensor out = at::empty(cube(N, dim), current_device_options(at::kComplexFloat));
    int64_t nbytes = 0;
    at::Tensor ws = coeffs_workspace(N, dim, out, nbytes);
    check_rc(nfft_hip_gaussian_interpolated_coeffs(sigma, N, (int32_t)dim, p, eps, out.data_ptr(), ws.data_ptr(), nbytes,
                                                   stream_of(out)));
    return out;
}

at::Tensor interpolation_grid(int64_t N, int64_t dim)
{
    std::vector<int64_t> shape = cube(N, dim);
    shape.push_back(dim);
    at::Tensor out = at::empty(shape, current_device_options(at::kFloat));
    check_rc(nfft_hip_interpolation_grid(N, (int32_t)dim, 0, out.data_ptr<float>(), stream_of(out)));
    return out;
}

at::Tensor radial_interpolation_grid(int64_t N, int64_t dim)
{
    at::Tensor out = at::empty(cube(N, dim), current_device_options(at::kFloat));
    check_rc(nfft_hip_interpolation_grid(N, (int32_t)dim, 1, out.data_ptr<float>(), stream_of(out)));
    return out;
}

at::Tensor interpolated_kernel_coeffs(at::Tensor grid_values)
{
    TORCH_CHECK(grid_values.is_cuda(),
                "torch_nfft.interpolated_kernel_coeffs is currently only implemented for GPU tensors");
    const int64_t dim = grid_values.dim();
    CHECK_INPUT(dim >= 1 && dim <= 3);
    const int64_t N = grid_values.size(0);
    for (int64_t d = 1; d < dim; ++d) CHECK_INPUT(grid_values.size(d) == N);
    const bool real = real_dtype(grid_values);
    const at::Tensor vals = grid_values.contiguous();
    c10::DeviceGuard guard(vals.device());
    at::Tensor out = at::empty(cube(N, dim), vals.options().dtype(at::kComplexFloat));
    int64_t nbytes = 0;
    at::Tensor ws = coeffs_workspace(N, dim, out, nbytes);
    check_rc(nfft_hip_interpolated_kernel_coeffs(vals.data_ptr(), real ? 0 : 1, N, (int32_t)dim, out.data_ptr(),
                                                 ws.data_ptr(), nbytes, stream_of(out)));
    return out;
}

} // namespace

// Same eight schemas as the reference's RegisterOperators block (csrc/core.cpp:176-184); every caller is positional
// (torch_nfft/nfft.py:14-86, coeffs.py:11-27), so naming the arguments changes nothing for them.
TORCH_LIBRARY(torch_nfft, m)
{
    m.def("nfft_adjoint(Tensor pos, Tensor x, Tensor? batch, int N, int m, int real_output) -> Tensor", &nfft_adjoint);
    m.def("nfft_forward(Tensor pos, Tensor x, Tensor? batch, int m, int real_output) -> Tensor", &nfft_forward);
    m.def("nfft_fastsum(Tensor sources, Tensor targets, Tensor x, Tensor coeffs, Tensor? source_batch, "
          "Tensor? target_batch, int m) -> Tensor", &nfft_fastsum);
    m.def("gaussian_analytic_coeffs(float sigma, int N, int dim) -> Tensor", &gaussian_analytic_coeffs);
    m.def("gaussian_interpolated_coeffs(float sigma, int N, int dim, int p, float eps) -> Tensor",
          &gaussian_interpolated_coeffs);
    m.def("interpolation_grid(int N, int dim) -> Tensor", &interpolation_grid);
    m.def("radial_interpolation_grid(int N, int dim) -> Tensor", &radial_interpolation_grid);
    m.def("interpolated_kernel_coeffs(Tensor grid_values) -> Tensor", &interpolated_kernel_coeffs);
    // not in the reference: control of the point-plan cache (torch_nfft_amd.ops.plan_cache_*)
    m.def("_plan_cache(int action) -> int", &plan_cache_control);
    // not in the reference: device-side fault reports (torch_nfft_amd.ops.check_status)
    m.def("_check_status(int synchronize) -> int", &check_status);
}
