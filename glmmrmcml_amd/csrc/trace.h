// Named ranges around the phases of an MCML iteration (tracing), visible in
// `rocprofv3 --marker-trace`: sample / beta-step / theta-step / refresh.  libroctx64.so is resolved with dlopen the
// first time a range is opened and only when GLMMR_MCML_ROCTX=1, so that a host without the profiler libraries
// (or a run that does not ask) pays one branch per phase and nothing else.
#pragma once
#include <dlfcn.h>
#include <atomic>
#include <chrono>
#include <mutex>
#include <cstdlib>
#include <cstring>

namespace mcml {

struct RoctxApi {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
};

inline const RoctxApi& roctx_api()
{
    static RoctxApi api = [] {
        RoctxApi a;
        const char* e = getenv("GLMMR_MCML_ROCTX");
        if (!e || strcmp(e, "1") != 0) return a;
        void* h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return a;
        a.push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        a.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (!a.push || !a.pop) a = RoctxApi{};
        return a;
    }();
    return api;
}

// Host wall-clock time per phase, summed since the last reset (glmmr_mcml_dbg_phase_ms; bench.py reports it next to the
// iteration time of the launch-bound configurations, so that a slow repetition says WHICH phase was slow).  Every phase
// ends on a host synchronisation of its own (a result comes back), except the refresh, whose tail runs under the next
// sampler call.  Process-wide and off by default: one relaxed load per phase.
struct PhaseClock {
    std::atomic<bool> on{false};
    std::mutex mu;
    double ms[4] = {0, 0, 0, 0};          // sample, beta-step, theta-step, refresh
    long long n[4] = {0, 0, 0, 0};
};
inline PhaseClock& phase_clock() { static PhaseClock pc; return pc; }
inline int phase_id(const char* name)
{
    if (!strcmp(name, "mcml:sample")) return 0;
    if (!strcmp(name, "mcml:beta-step")) return 1;
    if (!strcmp(name, "mcml:theta-step")) return 2;
    return 3;
}

// RAII: the range closes on every exit path (the MCML_TRY early returns included)
struct PhaseRange {
    bool open = false;
    int id = -1;
    std::chrono::steady_clock::time_point t0;
    explicit PhaseRange(const char* name)
    {
        const RoctxApi& a = roctx_api();
        if (a.push) { a.push(name); open = true; }
        if (phase_clock().on.load(std::memory_order_relaxed)) { id = phase_id(name); t0 = std::chrono::steady_clock::now(); }
    }
    ~PhaseRange()
    {
        if (id >= 0) {
            const double dt = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            PhaseClock& pc = phase_clock();
            std::lock_guard<std::mutex> g(pc.mu);
            pc.ms[id] += dt; pc.n[id] += 1;
        }
        if (open) roctx_api().pop();
    }
    PhaseRange(const PhaseRange&) = delete;
    PhaseRange& operator=(const PhaseRange&) = delete;
};

}  // namespace mcml
