/* cpusample.c — LD_PRELOAD sampling profiler for the encoder binaries (profiling aid, not part of the product).
 *   gcc -O2 -fPIC -shared -o tools/prof/libcpusample.so tools/prof/cpusample.c
 *   X265HIP_CPUSAMPLE_OUT=/tmp/s.bin LD_PRELOAD=tools/prof/libcpusample.so oracle/_ref/x265_hip_8bit ...
 *   python tools/prof/resolve.py /tmp/s.bin
 * ITIMER_PROF ticks with the CPU time of the whole process (all threads); the kernel delivers SIGPROF to a thread that is running, so the
 * program counters recorded by the handler sample where the process's CPU time goes.  The file: /proc/self/maps as text, a line "PCS", then raw
 * uint64 program counters. */
#define _GNU_SOURCE
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <ucontext.h>
#include <unistd.h>
#include <execinfo.h>

#define CAP (8u << 20)
static uint64_t* g_pc;
static volatile uint32_t g_n;
static const char* g_out;
/* X265HIP_CPUSAMPLE_STACK=1: 10 return addresses per sample as well (backtrace(); file <out>.stacks): who calls the ioctls? */
#define DEPTH 24
static uint64_t* g_stack;
static int g_wantStack;

static void on_prof(int sig, siginfo_t* si, void* ctx)
{
    (void)sig; (void)si;
    const ucontext_t* uc = (const ucontext_t*)ctx;
    const uint32_t i = __atomic_fetch_add(&g_n, 1, __ATOMIC_RELAXED);
    if (i < CAP)
    {
        g_pc[i] = (uint64_t)uc->uc_mcontext.gregs[REG_RIP];
        if (g_wantStack && i < (CAP >> 4))
        {
            void* fr[DEPTH + 4];
            const int n = backtrace(fr, DEPTH + 4);
            /* frames 0..: this handler, the signal trampoline, then the interrupted function and its callers */
            for (int k = 0; k < DEPTH; k++)
                g_stack[(size_t)i * DEPTH + k] = k + 2 < n ? (uint64_t)fr[k + 2] : 0;
        }
    }
}

static void dump(void)
{
    struct itimerval off;
    memset(&off, 0, sizeof(off));
    setitimer(ITIMER_PROF, &off, NULL);
    FILE* o = fopen(g_out, "wb");
    if (!o) return;
    FILE* m = fopen("/proc/self/maps", "r");
    if (m)
    {
        char line[1024];
        while (fgets(line, sizeof(line), m)) fputs(line, o);
        fclose(m);
    }
    fputs("PCS\n", o);
    const uint32_t n = g_n < CAP ? g_n : CAP;
    fwrite(g_pc, 8, n, o);
    fclose(o);
    if (g_wantStack)
    {
        char path[1024];
        snprintf(path, sizeof(path), "%s.stacks", g_out);
        FILE* st = fopen(path, "wb");
        if (st)
        {
            const uint32_t ns = n < (CAP >> 4) ? n : (CAP >> 4);
            fwrite(g_stack, 8, (size_t)ns * DEPTH, st);
            fclose(st);
        }
    }
}

__attribute__((constructor)) static void start(void)
{
    g_out = getenv("X265HIP_CPUSAMPLE_OUT");
    if (!g_out) return;
    g_pc = (uint64_t*)calloc(CAP, 8);
    if (!g_pc) return;
    g_wantStack = getenv("X265HIP_CPUSAMPLE_STACK") != NULL;
    if (g_wantStack)
    {
        g_stack = (uint64_t*)calloc((size_t)(CAP >> 4) * DEPTH, 8);
        void* warm[4];
        backtrace(warm, 4);                 /* loads libgcc now, not inside the first signal */
        if (!g_stack) g_wantStack = 0;
    }
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = on_prof;
    sa.sa_flags = SA_SIGINFO | SA_RESTART;
    sigaction(SIGPROF, &sa, NULL);
    const int us = getenv("X265HIP_CPUSAMPLE_US") ? atoi(getenv("X265HIP_CPUSAMPLE_US")) : 1000;
    struct itimerval it;
    it.it_interval.tv_sec = 0; it.it_interval.tv_usec = us;
    it.it_value = it.it_interval;
    setitimer(ITIMER_PROF, &it, NULL);
    atexit(dump);
}
