// Error reporting and per-kernel-family event timing for libtranscodon_hip.so.
#include "common.h"

#include <mutex>
#include <vector>

namespace {

thread_local char g_err[512] = "";

struct ProfSlot {
    hipEvent_t start, stop;
    int kind;
    int count;          // 1 for the first launch of a scope, 0 for its further launches
    double work;
};

struct ProfState {
    std::mutex mu;
    bool enabled = false;
    std::vector<ProfSlot *> slots;      // recorded since the last reset
    std::vector<ProfSlot *> free_slots; // events are recycled, never destroyed while loaded
    int64_t launches[TC_PROF_NKINDS] = {};
    double ms[TC_PROF_NKINDS] = {};
    double work[TC_PROF_NKINDS] = {};
} g_prof;

}  // namespace

void tc_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *tc_last_error(void) { return g_err; }
extern "C" int tc_version(void) { return TC_VERSION; }
extern "C" size_t tc_struct_bytes(int which) {
    switch (which) {
        case 0: return sizeof(tc_layer_weights);
        case 1: return sizeof(tc_encoder_desc);
        case 2: return sizeof(tc_layer_grads);
        case 3: return sizeof(tc_encoder_grads);
        default: return 0;
    }
}

static thread_local TcProfScope *t_scope = nullptr;

TcProfScope::TcProfScope(int kind_, hipStream_t, double work_)
    : kind(kind_), work(work_), first(true), active(g_prof.enabled), outer(nullptr) {
    if (!active) return;
    outer = t_scope;
    t_scope = this;
}

TcProfScope::~TcProfScope() {
    if (active) t_scope = outer;
}

bool tc_prof_launch_events(hipEvent_t *start, hipEvent_t *stop) {
    TcProfScope *sc = t_scope;
    if (!sc) return false;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    ProfSlot *s;
    if (!g_prof.free_slots.empty()) {
        s = g_prof.free_slots.back();
        g_prof.free_slots.pop_back();
    } else {
        s = new ProfSlot();
        if (hipEventCreate(&s->start) != hipSuccess) {
            delete s;
            return false;
        }
        if (hipEventCreate(&s->stop) != hipSuccess) {
            (void)hipEventDestroy(s->start);
            delete s;
            return false;
        }
    }
    s->kind = sc->kind;
    s->count = sc->first ? 1 : 0;
    s->work = sc->first ? sc->work : 0.0;
    sc->first = false;
    g_prof.slots.push_back(s);
    *start = s->start;
    *stop = s->stop;
    return true;
}

static void prof_drain_locked() {
    for (ProfSlot *s : g_prof.slots) {
        if (hipEventSynchronize(s->stop) == hipSuccess) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, s->start, s->stop) == hipSuccess) {
                g_prof.launches[s->kind] += s->count;
                g_prof.ms[s->kind] += ms;
                g_prof.work[s->kind] += s->work;
            }
        }
        g_prof.free_slots.push_back(s);
    }
    g_prof.slots.clear();
}

extern "C" int tc_prof_enable(int enabled) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.enabled = enabled != 0;
    return TC_OK;
}

extern "C" int tc_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    prof_drain_locked();
    for (int k = 0; k < TC_PROF_NKINDS; ++k) {
        g_prof.launches[k] = 0;
        g_prof.ms[k] = 0.0;
        g_prof.work[k] = 0.0;
    }
    return TC_OK;
}

extern "C" int tc_prof_collect(int kind, int64_t *launches, double *ms_total, double *work_total) {
    TC_REQUIRE(kind >= 0 && kind < TC_PROF_NKINDS, "tc_prof_collect: bad kind %d", kind);
    std::lock_guard<std::mutex> lk(g_prof.mu);
    prof_drain_locked();
    if (launches) *launches = g_prof.launches[kind];
    if (ms_total) *ms_total = g_prof.ms[kind];
    if (work_total) *work_total = g_prof.work[kind];
    return TC_OK;
}
