// Optional per-kernel timing: when enabled, the library brackets the launches of the selected kernel
// with hipEvents recorded on the caller's stream, so a benchmark can read the kernel's own duration
// (not host-side wall time) without an external profiler.  Off by default; costs nothing when off.
#include "od_common.h"
#include "../../include/od_boxes.h"

#include <mutex>
#include <vector>

namespace od {

struct EventPair { hipEvent_t a, b; };

static std::mutex g_mu;
static int g_selected = OD_PROFILE_OFF;
static std::vector<EventPair> g_pool[OD_KERNEL_COUNT];   // created once, reused after each read
static size_t g_used[OD_KERNEL_COUNT];

bool profile_begin(int kernel_id, hipStream_t s, void** token) {
    *token = nullptr;
    if (g_selected == OD_PROFILE_OFF) return false;
    if (g_selected != OD_PROFILE_ALL && g_selected != kernel_id) return false;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_used[kernel_id] == g_pool[kernel_id].size()) {
        EventPair p;
        if (hipEventCreate(&p.a) != hipSuccess) return false;
        if (hipEventCreate(&p.b) != hipSuccess) { (void)hipEventDestroy(p.a); return false; }
        g_pool[kernel_id].push_back(p);
    }
    EventPair& p = g_pool[kernel_id][g_used[kernel_id]++];
    (void)hipEventRecord(p.a, s);
    *token = &p;   // stable until the next push_back of the same id, which cannot happen before profile_end
    return true;
}

void profile_end(void* token, hipStream_t s) {
    if (token) (void)hipEventRecord(static_cast<EventPair*>(token)->b, s);
}

}  // namespace od

extern "C" int od_profile_select(int32_t kernel_id) {
    if (kernel_id != OD_PROFILE_OFF && kernel_id != OD_PROFILE_ALL && (kernel_id < 0 || kernel_id >= OD_KERNEL_COUNT))
        return OD_ERR_ARG;
    std::lock_guard<std::mutex> lk(od::g_mu);
    od::g_selected = kernel_id;
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
