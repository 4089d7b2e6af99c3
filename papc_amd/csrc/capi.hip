// capi.hip -- version, thread-local error string, event profiler of libpapc_hip.
#include <stdarg.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace papc {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return PAPC_E_LAUNCH;
    }
    return PAPC_OK;
}

// ---- profiler: event pairs per enabled kernel family, resolved lazily in papc_prof_read ---------
struct ProfState {
    std::mutex mu;
    unsigned mask = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending[PAPC_K_COUNT];
    std::vector<hipEvent_t> pool;
    double total_ms[PAPC_K_COUNT] = {0};
    int64_t launches[PAPC_K_COUNT] = {0};
};
static ProfState g_prof;

static hipEvent_t prof_get_event()
{
    if (!g_prof.pool.empty()) {
        hipEvent_t e = g_prof.pool.back();
        g_prof.pool.pop_back();
        return e;
    }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

ProfScope::ProfScope(int k, hipStream_t s) : kernel(k), stream(s), on(false), e0(nullptr)
{
    if (!(g_prof.mask & (1u << k))) return;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    e0 = prof_get_event();
    if (!e0) return;
    on = hipEventRecord(e0, stream) == hipSuccess;
}

ProfScope::~ProfScope()
{
    if (!on) return;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    hipEvent_t e1 = prof_get_event();
    if (!e1) return;
    if (hipEventRecord(e1, stream) == hipSuccess) g_prof.pending[kernel].push_back({e0, e1});
}

static void prof_drain(int k)
{
    for (auto &pr : g_prof.pending[k]) {
        float ms = 0.f;
        if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
            g_prof.total_ms[k] += ms;
            g_prof.launches[k] += 1;
        }
        g_prof.pool.push_back(pr.first);
        g_prof.pool.push_back(pr.second);
    }
    g_prof.pending[k].clear();
}

}  // namespace papc

extern "C" {

int papc_version(void) { return 100; /* 0.1.0 */ }

const char *papc_last_error_string(void) { return papc::g_err; }

int papc_prof_enable(unsigned mask)
{
    std::lock_guard<std::mutex> lk(papc::g_prof.mu);
    papc::g_prof.mask = mask;
    return PAPC_OK;
}

int papc_prof_reset(void)
{
    std::lock_guard<std::mutex> lk(papc::g_prof.mu);
    for (int k = 0; k < PAPC_K_COUNT; ++k) {
        papc::prof_drain(k);
        papc::g_prof.total_ms[k] = 0;
        papc::g_prof.launches[k] = 0;
    }
    return PAPC_OK;
}

int papc_prof_read(int kernel, double *total_ms, int64_t *launches)
{
    PAPC_REQUIRE(kernel >= 0 && kernel < PAPC_K_COUNT, PAPC_E_INVALID, "papc_prof_read: bad kernel id %d", kernel);
    std::lock_guard<std::mutex> lk(papc::g_prof.mu);
    papc::prof_drain(kernel);
    if (total_ms) *total_ms = papc::g_prof.total_ms[kernel];
    if (launches) *launches = papc::g_prof.launches[kernel];
    return PAPC_OK;
}

}  // extern "C"
