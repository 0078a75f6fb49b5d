// Optional per-kernel timing: when enabled, the library launches the selected kernel through
// hipExtLaunchKernel with a start and a stop event, which receive the time stamps of the dispatch itself, so a
// benchmark can read the kernel's own duration (what a kernel trace reports; not host-side wall time, and not the
// distance between marker events, which also contains the gaps around the kernel) without an external profiler.
// Off by default; costs one relaxed atomic load per launch when off.
// The selection, the stride and the event pools are state of the process (not of a stream or a device): every thread
// that launches through the library sees the same selection, and od_profile_read hands out what all of them recorded.
// Events are created on whichever device is current at the launch, so a timed kernel must be read while only one device
// has been used with profiling on.
#include "od_common.h"
#include "../../include/od_boxes.h"

#include <atomic>
#include <mutex>
#include <vector>

namespace od {

struct EventPair { hipEvent_t a, b; };

static std::mutex g_mu;
static std::atomic<int> g_selected{OD_PROFILE_OFF};   // read without the lock on every launch
static int g_every = 1;                                  // time every g_every-th launch of a selected kernel
static long long g_seen[OD_KERNEL_COUNT];
static std::vector<EventPair> g_pool[OD_KERNEL_COUNT];   // created once, reused after each read
static size_t g_used[OD_KERNEL_COUNT];

bool profile_events(int kernel_id, hipEvent_t* start, hipEvent_t* stop) {
    const int selected = g_selected.load(std::memory_order_relaxed);
    if (selected == OD_PROFILE_OFF) return false;
    if (selected != OD_PROFILE_ALL && selected != kernel_id) return false;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_seen[kernel_id]++ % g_every != 0) return false;
    if (g_used[kernel_id] == g_pool[kernel_id].size()) {
        EventPair p;
        if (hipEventCreate(&p.a) != hipSuccess) return false;
        if (hipEventCreate(&p.b) != hipSuccess) { (void)hipEventDestroy(p.a); return false; }
        g_pool[kernel_id].push_back(p);
    }
    const EventPair& p = g_pool[kernel_id][g_used[kernel_id]++];
    *start = p.a;
    *stop = p.b;
    return true;
}

}  // namespace od

extern "C" int od_profile_select(int32_t kernel_id) {
    if (kernel_id != OD_PROFILE_OFF && kernel_id != OD_PROFILE_ALL && (kernel_id < 0 || kernel_id >= OD_KERNEL_COUNT))
        return OD_ERR_ARG;
    std::lock_guard<std::mutex> lk(od::g_mu);
    od::g_selected.store(kernel_id, std::memory_order_relaxed);
    return OD_OK;
}

extern "C" int od_profile_stride(int32_t every) {
    if (every < 1) return OD_ERR_ARG;
    std::lock_guard<std::mutex> lk(od::g_mu);
    od::g_every = every;
    for (int k = 0; k < OD_KERNEL_COUNT; ++k) od::g_seen[k] = 0;
    return OD_OK;
}

extern "C" int od_profile_read(int32_t kernel_id, double* total_ms, int64_t* launches) {
    if (kernel_id < 0 || kernel_id >= OD_KERNEL_COUNT || !total_ms || !launches) return OD_ERR_ARG;
    std::lock_guard<std::mutex> lk(od::g_mu);
    double sum = 0.0;
    const size_t n = od::g_used[kernel_id];
    for (size_t i = 0; i < n; ++i) {
        od::EventPair& p = od::g_pool[kernel_id][i];
        OD_HIP(hipEventSynchronize(p.b));
        float ms = 0.f;
        OD_HIP(hipEventElapsedTime(&ms, p.a, p.b));
        sum += ms;
    }
    od::g_used[kernel_id] = 0;
    *total_ms = sum;
    *launches = (int64_t)n;
    return OD_OK;
}
