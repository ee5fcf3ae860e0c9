// Optional per-launch HIP-event profiling of the convolution kernels (bench.py; see dvd_prof_enable in include/dvdgan_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include <mutex>

namespace dvdprof {
struct ProfRec { hipEvent_t a, b; double flops; int kind; long long M; int C, Cout, taps, split, flags, variant; };
extern bool g_prof;
extern std::vector<ProfRec> g_recs;
extern std::mutex g_prof_mu;
struct ProfScope {
    ProfRec r; bool on; hipStream_t s;
    ProfScope(int kind, double flops, void* stream, long long M, int C, int Cout, int taps, int split, int flags)
        : on(g_prof), s((hipStream_t)stream) {
        if (!on) return;
        r.kind = kind; r.flops = flops; r.M = M; r.C = C; r.Cout = Cout; r.taps = taps; r.split = split; r.flags = flags;
        r.variant = 0;
        hipEventCreate(&r.a); hipEventCreate(&r.b);
        hipEventRecord(r.a, s);
    }
    ~ProfScope() {
        if (!on) return;
        hipEventRecord(r.b, s);
        std::lock_guard<std::mutex> l(g_prof_mu);
        g_recs.push_back(r);
    }
};
}  // namespace dvdprof
using dvdprof::ProfScope;
