// x265_hip_debug.h — diagnostics shared by the binding translation units (x265_amd/host/*.cpp); nothing here changes what the encoder computes.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <unistd.h>

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

} // namespace X265_NS
