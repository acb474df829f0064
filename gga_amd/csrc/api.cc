// Error reporting, ABI version and the bench-only kernel timing sessions of libgga_hip.so
// (include/gga_hip.h).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include <mutex>
#include <vector>

#include "../../include/gga_hip.h"

static thread_local char g_err[512] = "";

void gga_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* gga_last_error(void) { return g_err; }
extern "C" int gga_abi_version(void) { return 23; }

// ---- timing sessions -----------------------------------------------------------------------
// One session per site. While a session is armed, the entry point of that site brackets its
// launches with a HIP event pair on the caller's stream; nothing synchronises until
// gga_timing_collect. All state sits behind one mutex (launches of different host threads may
// interleave; each takes its own event pair).
struct TimingSite {
    std::vector<hipEvent_t> ev;      // 2 per sample
    int cap = 0, count = 0;
    int64_t key = 0;
};
static TimingSite g_sites[GGA_TIME_SITES];
static std::mutex g_timing_mu;

hipEvent_t* gga_timing_acquire(int site, int64_t key) {
    if (site < 0 || site >= GGA_TIME_SITES) return nullptr;
    TimingSite& s = g_sites[site];
    if (s.cap == 0) return nullptr;                      // unarmed: no lock on the hot path
    std::lock_guard<std::mutex> lk(g_timing_mu);
    if (s.count >= s.cap || (s.key != 0 && s.key != key)) return nullptr;
    return &s.ev[2 * (size_t)s.count++];
}

extern "C" int gga_timing_begin(int site, int max_samples, int64_t key) {
    if (site < 0 || site >= GGA_TIME_SITES || max_samples < 0 || max_samples > 65536) {
        gga_set_error("gga_timing_begin: site %d / max_samples %d out of range", site, max_samples);
        return GGA_ERR_INVALID_ARG;
    }
    std::lock_guard<std::mutex> lk(g_timing_mu);
    TimingSite& s = g_sites[site];
    while ((int)s.ev.size() < 2 * max_samples) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) {
            gga_set_error("gga_timing_begin: hipEventCreate failed");
            return GGA_ERR_LAUNCH;
        }
        s.ev.push_back(e);
    }
    s.cap = max_samples;
    s.count = 0;
    s.key = key;
    return GGA_OK;
}

extern "C" int gga_timing_collect(int site, float* ms_host, int cap) {
    if (site < 0 || site >= GGA_TIME_SITES || (!ms_host && cap != 0)) {
        gga_set_error("gga_timing_collect: bad arguments");
        return GGA_ERR_INVALID_ARG;
    }
    std::lock_guard<std::mutex> lk(g_timing_mu);
    TimingSite& s = g_sites[site];
    const int n = s.count < cap ? s.count : cap;
    for (int i = 0; i < n; ++i) {
        if (hipEventSynchronize(s.ev[2 * i + 1]) != hipSuccess ||
            hipEventElapsedTime(&ms_host[i], s.ev[2 * i], s.ev[2 * i + 1]) != hipSuccess) {
            gga_set_error("gga_timing_collect: event %d of site %d did not complete", i, site);
            s.cap = s.count = 0;
            return GGA_ERR_LAUNCH;
        }
    }
    s.cap = s.count = 0;
    return n;
}
