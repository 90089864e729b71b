// Named ranges around the phases of an MCML iteration (tracing), visible in
// `rocprofv3 --marker-trace`: sample / beta-step / theta-step / refresh.  libroctx64.so is resolved with dlopen the
// first time a range is opened and only when GLMMR_MCML_ROCTX=1, so that a host without the profiler libraries
// (or a run that does not ask) pays one branch per phase and nothing else.
#pragma once
#include <dlfcn.h>
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

// RAII: the range closes on every exit path (the MCML_TRY early returns included)
struct PhaseRange {
    bool open = false;
    explicit PhaseRange(const char* name)
    {
        const RoctxApi& a = roctx_api();
        if (a.push) { a.push(name); open = true; }
    }
    ~PhaseRange() { if (open) roctx_api().pop(); }
    PhaseRange(const PhaseRange&) = delete;
    PhaseRange& operator=(const PhaseRange&) = delete;
};

}  // namespace mcml
