// x265_hip_debug.h — diagnostics shared by the binding translation units (x265_amd/host/*.cpp); nothing here changes what the encoder computes.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
#include <unistd.h>
#include "x265hip.h"

namespace X265_NS {

// X265HIP_DEBUG_STARTUP=1: milliseconds since the process started (/proc/self/stat) at the points where the bindings first touch the device — where
// the wall clock of a short encode goes before and after the encoder's own fps clock runs
inline void x265hip_debug_mark(const char* what)
{
    static const bool on = getenv("X265HIP_DEBUG_STARTUP") != NULL;
    if (!on)
        return;
    static double procStart = -1;
    struct timespec ts;
    clock_gettime(CLOCK_BOOTTIME, &ts);
    const double now = ts.tv_sec + ts.tv_nsec * 1e-9;
    if (procStart < 0)
    {
        procStart = now;
        FILE* f = fopen("/proc/self/stat", "r");
        if (f)
        {
            char buf[1024];
            if (fgets(buf, sizeof(buf), f))
            {
                const char* p = strrchr(buf, ')');
                unsigned long long start = 0;
                int field = 2;
                for (p = p ? p + 1 : buf; *p && field < 22; p++)
                    if (*p == ' ') field++;
                if (sscanf(p, "%llu", &start) == 1)
                    procStart = (double)start / sysconf(_SC_CLK_TCK);
            }
            fclose(f);
        }
    }
    fprintf(stderr, "x265hip-startup: %8.1f ms  %s\n", (now - procStart) * 1e3, what);
}

// SURVEY.md §8b "Errors": a failing GPU call must never take the encoder down — every seam has the reference's own host function one branch away, and
// that is where it goes: the failing module says so ONCE, switches itself (or the object concerned) off, and the encode continues with the reference's
// bytes.  X265HIP=require (what bench.py, the tools and the GPU tests run with) makes the same event fatal: a number measured on a fallback is worthless.
inline bool x265hip_required()
{
    static const bool r = getenv("X265HIP") && !strcmp(getenv("X265HIP"), "require");
    return r;
}
inline void x265hip_device_failure(const char* module, const char* what)
{
    static std::mutex lock;
    static char said[8][32];
    static int n = 0;
    {
        std::lock_guard<std::mutex> g(lock);
        bool first = true;
        for (int i = 0; i < n; i++)
            if (!strncmp(said[i], module, 31)) first = false;
        if (first)
        {
            if (n < 8) { strncpy(said[n], module, 31); said[n][31] = 0; n++; }
            fprintf(stderr, "x265hip: %s: %s: %s — this part of the GPU path is OFF from here on, the encoder's own host code takes over%s\n", module, what, x265hip_last_error(),
                    x265hip_required() ? " (X265HIP=require: fatal)" : "");
        }
    }
    if (x265hip_required())
        abort();
}

// X265HIP_DEVICES=0,1,2,3: the encoder's device work is spread over several GPUs — place p (include/x265hip.h, x265hip_places) lives on the p-th
// device of the list; the same device may be listed more than once (two places on one GPU: the way the exchange is exercised on a one-GPU box).
// Reference-picture mirrors and source pictures take their places in turn (x265's frame encoders work on consecutive frames at the same time:
// consecutive pictures -> different GPUs); a SAD surface is built where its source picture lives, from a replica of the reference picture that the
// library feeds device to device.  Returns the number of places (0: the variable is not set — everything lives on the calling thread's device).
inline int x265hip_places_configured()
{
    static int n = -1;
    static std::mutex lock;
    std::lock_guard<std::mutex> g(lock);
    if (n >= 0)
        return n;
    n = 0;
    const char* env = getenv("X265HIP_DEVICES");
    if (!env || !*env)
        return 0;
    int devs[64], count = 0;
    for (const char* p = env; *p && count < 64;)
    {
        char* end;
        const long v = strtol(p, &end, 10);
        if (end == p) break;
        devs[count++] = (int)v;
        p = *end == ',' ? end + 1 : end;
        if (*end && *end != ',') break;
    }
    if (count < 1 || x265hip_places(count, devs))
    {
        // a configuration error, not a device failure: places stay unconfigured (everything lives on one device) unless the run requires the GPU path
        fprintf(stderr, "x265hip: X265HIP_DEVICES=%s: %s — ignored, one device is used\n", env, count < 1 ? "not a list of device numbers" : x265hip_last_error());
        if (x265hip_required())
            abort();
        return 0;
    }
    n = count;
    return n;
}

} // namespace X265_NS
