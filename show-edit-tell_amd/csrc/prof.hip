// Opt-in per-kernel timing with HIP events recorded on the launch stream (used by bench.py for
// the live roofline figure and the per-kernel breakdown).  Off by default: when off, ProfScope
// is two predictable branches and nothing is recorded.  Not thread-safe (a debug facility).
#include <cstring>
#include <vector>
#include "set_common.h"

namespace set {

struct ProfRec {
    const char* tag;
    hipEvent_t a, b;
    double flops, bytes;
};
static bool g_prof_on = false;
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;

static hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

ProfScope::ProfScope(const char* tag, hipStream_t s, double flops, double bytes) : idx(-1), st(s) {
    if (!g_prof_on || !tag) return;
    ProfRec r{tag, get_event(), get_event(), flops, bytes};
    if (!r.a || !r.b) return;
    (void)hipEventRecord(r.a, s);
    g_recs.push_back(r);
    idx = (int)g_recs.size() - 1;
}
ProfScope::~ProfScope() {
    if (idx >= 0) (void)hipEventRecord(g_recs[idx].b, st);
}

}  // namespace set

using namespace set;

extern "C" {

int set_profile_enable(int on) {
    g_prof_on = on != 0;
    if (on) {
        for (auto& r : g_recs) { g_pool.push_back(r.a); g_pool.push_back(r.b); }
        g_recs.clear();
    }
    return SET_OK;
}

// Synchronises the recorded events and aggregates by tag.  Returns the number of entries written
// (<= max_entries) or a negative error code.
int set_profile_report(SetProfileEntry* out, int max_entries) {
    if (!out || max_entries <= 0) return -SET_ERR_ARG;
    int n = 0;
    for (auto& r : g_recs) {
        if (hipEventSynchronize(r.b) != hipSuccess) return -SET_ERR_HIP;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) return -SET_ERR_HIP;
        int k = -1;
        for (int i = 0; i < n; ++i)
            if (!strcmp(out[i].tag, r.tag)) { k = i; break; }
        if (k < 0) {
            if (n >= max_entries) continue;
            k = n++;
            memset(&out[k], 0, sizeof(out[k]));
            strncpy(out[k].tag, r.tag, sizeof(out[k].tag) - 1);
        }
        out[k].launches += 1;
        out[k].ms += ms;
        out[k].flops += r.flops;
        out[k].bytes += r.bytes;
    }
    return n;
}

}  // extern "C"
