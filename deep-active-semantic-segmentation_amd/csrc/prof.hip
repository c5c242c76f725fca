// Kernel-exact timing of the library's OWN launches, for bench.py's in-step roofline ("measured live inside
// bench.py with HIP events ... on the stream the kernel is launched on").  While a profile is open (dass_prof_begin ..
// dass_prof_end) every launch of this library goes out through hipExtLaunchKernelGGL with a start and a stop event bound to
// that one dispatch, on whatever stream the launch uses -- the elapsed time of the pair is the kernel's own begin -> end, the same
// quantity a rocprofv3 kernel trace reports (no marker packets, no inter-launch gaps).  Closed: the plain launch path, one
// predictable branch per launch.  Test / measurement infrastructure only: nothing on the product path depends on it.
#include "dass_common.h"
#include <cxxabi.h>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

namespace {
struct Rec {
    const void *fn;
    hipStream_t stream;
    long long grid;
    hipEvent_t e0, e1;
};
std::mutex g_mu;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_free;  // events of closed profiles, reused
std::unordered_map<const void *, std::string> g_names;
}  // namespace

int g_dass_prof_on = 0;

void dass_prof_slot(const void *fn, long long grid, hipStream_t st, hipEvent_t *e0, hipEvent_t *e1) {
    std::lock_guard<std::mutex> lk(g_mu);
    hipEvent_t ev[2];
    for (int i = 0; i < 2; ++i) {
        if (!g_free.empty()) {
            ev[i] = g_free.back();
            g_free.pop_back();
        } else if (hipEventCreate(&ev[i]) != hipSuccess) {
            ev[i] = nullptr;
        }
    }
    *e0 = ev[0];
    *e1 = ev[1];
    g_recs.push_back(Rec{fn, st, grid, ev[0], ev[1]});
}

/* open a profile: forget the previous one's launches (its events are kept for reuse) */
extern "C" int dass_prof_begin(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (const Rec &r : g_recs) {
        if (r.e0) g_free.push_back(r.e0);
        if (r.e1) g_free.push_back(r.e1);
    }
    g_recs.clear();
    g_dass_prof_on = 1;
    return DASS_OK;
}

/* close it: later launches take the plain path again; the records stay readable until the next dass_prof_begin */
extern "C" int dass_prof_end(void) {
    g_dass_prof_on = 0;
    return DASS_OK;
}

/* launches recorded so far in the open (or last closed) profile */
extern "C" int dass_prof_count(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    return (int)g_recs.size();
}

/* record i: demangled kernel name (truncated to name_bytes - 1), duration in ms (waits for the launch to finish), grid size in
 * workgroups, the stream it ran on */
extern "C" int dass_prof_get(int i, char *name, int name_bytes, float *ms, int64_t *grid, void **stream) {
    Rec r;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (i < 0 || i >= (int)g_recs.size() || !name || name_bytes < 2 || !ms) return DASS_ERR_ARG;
        r = g_recs[i];
    }
    if (!r.e0 || !r.e1) return DASS_ERR_LAUNCH;
    if (hipEventSynchronize(r.e1) != hipSuccess) return DASS_ERR_LAUNCH;
    if (hipEventElapsedTime(ms, r.e0, r.e1) != hipSuccess) return DASS_ERR_LAUNCH;
    std::string nm;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_names.find(r.fn);
        if (it == g_names.end()) {
            const char *raw = hipKernelNameRefByPtr(r.fn, r.stream);
            std::string s = raw ? raw : "?";
            int status = 1;
            char *dem = raw ? abi::__cxa_demangle(raw, nullptr, nullptr, &status) : nullptr;
            if (status == 0 && dem) s = dem;
            free(dem);
            it = g_names.emplace(r.fn, s).first;
        }
        nm = it->second;
    }
    strncpy(name, nm.c_str(), (size_t)name_bytes - 1);
    name[name_bytes - 1] = 0;
    if (grid) *grid = r.grid;
    if (stream) *stream = (void *)r.stream;
    return DASS_OK;
}
