// prof.hip - optional per-kernel-class timing with HIP events on the launch stream (bench.py's
// roofline leg).  Off by default; when off the hooks cost one branch.  Not thread-safe by design:
// one process per GPU, one host thread drives the path.
#include <vector>
#include "common.h"
#include "prof.h"

namespace {
struct Rec { int id; hipEvent_t a, b; };
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t g_open = nullptr;
hipEvent_t take() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
const char* kNames[EH_K_COUNT] = {"degree", "select", "offsets", "time_embed", "node_in", "graph_update", "gemm_qt",
                                  "attention", "gemm_out", "gru", "ctx", "gather_active", "gemm_tail", "pair",
                                  "q_sample", "layer", "misc", "train_ops"};
}  // namespace

bool eh_prof_on() { return g_on; }

void eh_prof_begin(hipStream_t s) {
    if (!g_on) return;
    g_open = take();
    if (g_open) (void)hipEventRecord(g_open, s);
}

void eh_prof_end(int id, hipStream_t s) {
    if (!g_on || !g_open) return;
    hipEvent_t b = take();
    if (b) { (void)hipEventRecord(b, s); g_recs.push_back(Rec{id, g_open, b}); }
    g_open = nullptr;
}

extern "C" int edgehip_prof_enable(int on) {
    g_on = on != 0;
    return EDGEHIP_OK;
}

extern "C" int edgehip_prof_collect(double* ms_host, int64_t* launches_host, int32_t n) {
    if (!ms_host || !launches_host || n < EH_K_COUNT) return EDGEHIP_ERR_ARG;
    for (int i = 0; i < n; ++i) { ms_host[i] = 0.0; launches_host[i] = 0; }
    for (auto& r : g_recs) {
        if (hipEventSynchronize(r.b) != hipSuccess) return EDGEHIP_ERR_LAUNCH;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) return EDGEHIP_ERR_LAUNCH;
        ms_host[r.id] += ms;
        launches_host[r.id] += 1;
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    g_recs.clear();
    return EDGEHIP_OK;
}

extern "C" const char* edgehip_prof_kernel_name(int32_t id) { return (id >= 0 && id < EH_K_COUNT) ? kNames[id] : ""; }
extern "C" int32_t edgehip_prof_num_kernels(void) { return EH_K_COUNT; }
