// rocFFT plan cache + execution on the caller's stream.
//
// Replaces the reference's per-call cufftPlanMany / cufftExecC2C / cufftDestroy
// (csrc/cuda/core_cuda.cu:254-272, 432-450).  The reference transforms a complex (2N)^d grid per
// (batch, column); here every grid plane is real, so the adjoint uses a real-to-complex transform and
// the forward a complex-to-real one (half the data and work), batched over the planes of a chunk.
// Plans are created once per (kind, dim, M, batch, device) and kept in a bounded least-recently-used cache
// (kMaxPlans entries: workloads whose number of point sets varies from call to call would otherwise grow it
// without limit); execution is bound to the caller's stream (the reference leaves cuFFT on the default stream
// and synchronises the device).  An evicted plan is not destroyed on the spot: rocfft_plan_destroy frees device
// buffers (hipFree: a device-wide synchronisation in the middle of the hot path, and the plan's kernels may still be
// queued); it is parked with the event recorded behind its last execution and destroyed by a later call once that
// event has completed.
#include <rocfft/rocfft.h>

#include <map>
#include <memory>
#include <mutex>
#include <tuple>
#include <vector>

#include "kernels.h"

namespace nfft {

namespace {
struct PlanEntry {
    // shared ownership: an entry evicted from the cache stays alive until the call that is executing it returns
    std::shared_ptr<rocfft_plan_t> plan;
    std::shared_ptr<ihipEvent_t> done;  // recorded behind every execution of the plan (on the executing stream)
    size_t work_bytes = 0;
    uint64_t last_use = 0;
};
std::vector<PlanEntry> g_retired;  // evicted plans whose last execution may still be running
constexpr size_t kMaxPlans = 32;
std::mutex g_mutex;
bool g_setup = false;
uint64_t g_tick = 0;
std::map<std::tuple<int, int, int, int, int64_t>, PlanEntry> g_plans;

const char *status_name(rocfft_status s)
{
    switch (s) {
    case rocfft_status_success: return "success";
    case rocfft_status_failure: return "failure";
    case rocfft_status_invalid_arg_value: return "invalid_arg_value";
    case rocfft_status_invalid_dimensions: return "invalid_dimensions";
    case rocfft_status_invalid_array_type: return "invalid_array_type";
    case rocfft_status_invalid_strides: return "invalid_strides";
    case rocfft_status_invalid_distance: return "invalid_distance";
    case rocfft_status_invalid_offset: return "invalid_offset";
    case rocfft_status_invalid_work_buffer: return "invalid_work_buffer";
    default: return "unknown";
    }
}

int get_plan(FftKind kind, int dim, int M, int64_t nplanes, PlanEntry &out)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        set_error("hipGetDevice failed");
        return 4;
    }
    std::lock_guard<std::mutex> lock(g_mutex);
    if (!g_setup) {
        rocfft_setup();
        g_setup = true;
    }
    // evicted plans whose work has drained can go now (nobody waits for the device here)
    for (size_t i = 0; i < g_retired.size();) {
        if (!g_retired[i].done || hipEventQuery(g_retired[i].done.get()) != hipErrorNotReady) {
            g_retired[i] = g_retired.back();
            g_retired.pop_back();
        } else {
            ++i;
        }
    }
    const auto key = std::make_tuple(dev, (int)kind, dim, M, nplanes);
    auto it = g_plans.find(key);
    if (it != g_plans.end()) {
        it->second.last_use = ++g_tick;
        out = it->second;
        return 0;
    }
    size_t lengths[3] = {(size_t)M, (size_t)M, (size_t)M};
    PlanEntry e;
    const bool rows = kind == kR2CRows || kind == kC2RRows;
    const bool fwd = kind == kR2C || kind == kR2CRows;
    size_t batch = (size_t)nplanes;
    if (rows)  // one 1-D transform per grid row: planes * M^(dim-1) rows
        for (int a = 1; a < dim; ++a) batch *= (size_t)M;
    rocfft_status st;
    rocfft_plan raw = nullptr;
    if (kind == kC2CForward)
        st = rocfft_plan_create(&raw, rocfft_placement_inplace, rocfft_transform_type_complex_forward,
                                rocfft_precision_single, (size_t)dim, lengths, batch, nullptr);
    else
        st = rocfft_plan_create(&raw, rocfft_placement_notinplace,
                                fwd ? rocfft_transform_type_real_forward : rocfft_transform_type_real_inverse,
                                rocfft_precision_single, rows ? 1 : (size_t)dim, lengths, batch, nullptr);
    if (st != rocfft_status_success) {
        set_error(std::string("Failed to create rocFFT plan: ") + status_name(st));
        return 3;
    }
    e.plan = std::shared_ptr<rocfft_plan_t>(raw, [](rocfft_plan_t *p) { rocfft_plan_destroy(p); });
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess && ev)
        e.done = std::shared_ptr<ihipEvent_t>(ev, [](ihipEvent_t *p) { (void)hipEventDestroy(p); });
    st = rocfft_plan_get_work_buffer_size(raw, &e.work_bytes);
    if (st != rocfft_status_success) {
        set_error(std::string("rocfft_plan_get_work_buffer_size: ") + status_name(st));
        return 3;
    }
    e.last_use = ++g_tick;
    if (g_plans.size() >= kMaxPlans) {
        auto victim = g_plans.begin();
        for (auto jt = g_plans.begin(); jt != g_plans.end(); ++jt)
            if (jt->second.last_use < victim->second.last_use) victim = jt;
        g_retired.push_back(victim->second);
        g_plans.erase(victim);
    }
    g_plans[key] = e;
    out = e;
    return 0;
}
} // namespace

int64_t fft_work_bytes(FftKind kind, int dim, int M, int64_t nplanes)
{
    PlanEntry e;
    if (get_plan(kind, dim, M, nplanes, e)) return -1;
    return (int64_t)e.work_bytes;
}

int fft_execute(FftKind kind, int dim, int M, int64_t nplanes, void *in, void *out, void *work, int64_t work_bytes,
                hipStream_t stream)
{
    PlanEntry e;
    if (int rc = get_plan(kind, dim, M, nplanes, e)) return rc;
    if ((int64_t)e.work_bytes > work_bytes) {
        set_error("rocFFT work buffer too small");
        return 2;
    }
    rocfft_execution_info info = nullptr;
    rocfft_status st = rocfft_execution_info_create(&info);
    if (st == rocfft_status_success) st = rocfft_execution_info_set_stream(info, stream);
    if (st == rocfft_status_success && e.work_bytes)
        st = rocfft_execution_info_set_work_buffer(info, work, e.work_bytes);
    void *ins[1] = {in};
    void *outs[1] = {out};
    if (st == rocfft_status_success) st = rocfft_execute(e.plan.get(), ins, outs, info);
    if (e.done) (void)hipEventRecord(e.done.get(), stream);
    if (info) rocfft_execution_info_destroy(info);
    if (st != rocfft_status_success) {
        set_error(std::string("Failed to execute rocFFT plan: ") + status_name(st));
        return 3;
    }
    return 0;
}

} // namespace nfft
