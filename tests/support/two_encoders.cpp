// two_encoders.cpp — the x265 API driven the way a long-lived process drives it: several encoders opened and closed one after another, with
// different picture sizes, in ONE process.  Test infrastructure for the lifetime rules of the x265-side bindings (x265_amd/host/*.cpp): a closed
// encoder's picture buffers are freed, malloc hands the same addresses to the next encoder's buffers, and neither a reference-picture mirror
// nor a source-picture entry of the first may answer for them (PicYuv::destroy seam).
// Linked three ways by oracle/Makefile (reference objects only / + bindings + emulated ABI / + bindings + libx265hip.so); the outputs must agree.
//
//   two_encoders <out-prefix> [par]   writes <out-prefix>_<k>.hevc for each session k; `par`: sessions run two (then three) at a time on
//                                     their own threads — encoders alive at the same time, the way an ABR ladder runs them
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <dirent.h>
#include <execinfo.h>
#include <signal.h>
#include <ucontext.h>
#include <sys/syscall.h>
#include <unistd.h>

#include "x265.h"

struct Session { int w, h, frames, seed; const char* preset; int bframes; };

// a textured scene that moves: deterministic, no files
static void make_frame(std::vector<uint8_t>& buf, int w, int h, int t, uint32_t seed)
{
    buf.resize((size_t)w * h * 3 / 2);
    uint32_t s = seed * 2654435761u + (uint32_t)t * 40503u;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            const int sx = x + 3 * t, sy = y + t;
            int v = 96 + ((sx * 7 + sy * 13) & 63) + (((sx >> 4) ^ (sy >> 4)) & 1) * 48 + ((sx / 24 + sy / 40) % 3) * 9;
            s = s * 1664525u + 1013904223u;
            v += (int)((s >> 28) & 7) - 3;
            buf[(size_t)y * w + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    uint8_t* cb = buf.data() + (size_t)w * h;
    uint8_t* cr = cb + (size_t)(w / 2) * (h / 2);
    for (int y = 0; y < h / 2; y++)
        for (int x = 0; x < w / 2; x++)
        {
            cb[(size_t)y * (w / 2) + x] = (uint8_t)(128 + (((x + t) >> 3) & 7) * 3);
            cr[(size_t)y * (w / 2) + x] = (uint8_t)(120 + (((y + 2 * t) >> 3) & 7) * 4);
        }
}

static bool run(const Session& ss, const char* path)
{
    x265_param* p = x265_param_alloc();
    if (x265_param_default_preset(p, ss.preset, NULL) < 0) return false;
    p->sourceWidth = ss.w; p->sourceHeight = ss.h; p->fpsNum = 30; p->fpsDenom = 1; p->internalCsp = X265_CSP_I420;
    p->totalFrames = ss.frames; p->bframes = ss.bframes; p->frameNumThreads = 2; p->decodedPictureHashSEI = 1;
    p->logLevel = X265_LOG_ERROR; p->bRepeatHeaders = 1;
    x265_param_parse(p, "pools", "4");
    x265_encoder* enc = x265_encoder_open(p);
    if (!enc) { fprintf(stderr, "two_encoders: open failed\n"); return false; }
    FILE* f = fopen(path, "wb");
    if (!f) return false;
    x265_picture* pic = x265_picture_alloc();
    x265_picture_init(p, pic);
    std::vector<uint8_t> buf;
    x265_nal* nal; uint32_t nnal;
    for (int t = 0; t < ss.frames; t++)
    {
        make_frame(buf, ss.w, ss.h, t, (uint32_t)ss.seed);
        pic->planes[0] = buf.data(); pic->planes[1] = buf.data() + (size_t)ss.w * ss.h; pic->planes[2] = (uint8_t*)pic->planes[1] + (size_t)(ss.w / 2) * (ss.h / 2);
        pic->stride[0] = ss.w; pic->stride[1] = pic->stride[2] = ss.w / 2;
        pic->bitDepth = 8; pic->colorSpace = X265_CSP_I420; pic->pts = t;
        if (x265_encoder_encode(enc, &nal, &nnal, pic, NULL) < 0) return false;
        for (uint32_t i = 0; i < nnal; i++) fwrite(nal[i].payload, 1, nal[i].sizeBytes, f);
    }
    while (x265_encoder_encode(enc, &nal, &nnal, NULL, NULL) > 0)
        for (uint32_t i = 0; i < nnal; i++) fwrite(nal[i].payload, 1, nal[i].sizeBytes, f);
    fclose(f);
    x265_picture_free(pic);
    x265_encoder_close(enc);
    x265_param_free(p);
    return true;
}

// TWO_ENCODERS_WATCHDOG=<seconds>: if the program is still running after that long, every thread prints its stack (module + offset,
// resolvable with addr2line against the binaries as built) and the process exits with code 9 — a hang on the GPU box must leave evidence
static void dump_stack(int)
{
    void* frames[48];
    const int n = backtrace(frames, 48);
    char head[64];
    const int len = snprintf(head, sizeof(head), "---- thread %ld\n", (long)syscall(SYS_gettid));
    if (write(2, head, len) < 0) {}
    backtrace_symbols_fd(frames, n, 2);
}
static void watchdog(int seconds)
{
    sleep(seconds);
    fprintf(stderr, "two_encoders: watchdog after %d s, stacks of all threads:\n", seconds);
    const long self = (long)syscall(SYS_gettid);
    for (int round = 0; round < 3; round++)             // three snapshots half a second apart: stuck or merely slow?
    {
        fprintf(stderr, "==== snapshot %d\n", round);
        DIR* d = opendir("/proc/self/task");
        for (dirent* e = d ? readdir(d) : NULL; e; e = readdir(d))
        {
            const long tid = atol(e->d_name);
            if (tid > 0 && tid != self)
            {
                syscall(SYS_tgkill, getpid(), tid, SIGUSR1);
                usleep(20000);
            }
        }
        if (d) closedir(d);
        usleep(500000);
    }
    _exit(9);
}

int main(int argc, char** argv)
{
    if (const char* w = getenv("TWO_ENCODERS_WATCHDOG"))
    {
        signal(SIGUSR1, dump_stack);
        struct sigaction sa;
        memset(&sa, 0, sizeof(sa));
        sa.sa_flags = SA_SIGINFO;
        sa.sa_sigaction = [](int, siginfo_t* si, void* ucv) {
            // a call through a null pointer leaves nothing for the unwinder: print where it came from (the return address on top of the stack)
            ucontext_t* uc = (ucontext_t*)ucv;
            const uintptr_t rip = (uintptr_t)uc->uc_mcontext.gregs[REG_RIP], rsp = (uintptr_t)uc->uc_mcontext.gregs[REG_RSP];
            void* ret[1] = { rip ? (void*)rip : *(void**)rsp };
            char head[128];
            const int len = snprintf(head, sizeof(head), "---- SIGSEGV at %p (fault address %p), thread %ld, called from:\n", (void*)rip, si->si_addr, (long)syscall(SYS_gettid));
            if (write(2, head, len) < 0) {}
            backtrace_symbols_fd(ret, 1, 2);
            dump_stack(0);
            _exit(11);
        };
        sigaction(SIGSEGV, &sa, NULL);
        signal(SIGABRT, [](int) { dump_stack(0); _exit(12); });
        std::thread(watchdog, atoi(w)).detach();
    }
    if (argc < 2) { fprintf(stderr, "usage: two_encoders <out-prefix> [par]\n"); return 2; }
    // sizes chosen so that freed buffers of one session are likely to be handed out again in the next, whole or in part
    const Session sessions[] = {
        { 352, 288, 10, 11, "medium", 3 },
        { 640, 360, 8, 22, "fast", 2 },
        { 352, 288, 10, 33, "medium", 3 },     // the first geometry again: same allocation sizes, other pictures
        { 176, 144, 12, 44, "slow", 4 },
        { 640, 368, 8, 55, "medium", 0 },
    };
    const int n = (int)(sizeof(sessions) / sizeof(sessions[0]));
    const bool par = argc > 2 && !strcmp(argv[2], "par");
    bool ok[n];
    auto one = [&](int k) {
        char path[512];
        snprintf(path, sizeof(path), "%s_%d.hevc", argv[1], k);
        ok[k] = run(sessions[k], path);
    };
    // TWO_ENCODERS_ONLY=<k>: session k alone (telling an encoder's own timing dependence from interference between live encoders)
    if (const char* only = getenv("TWO_ENCODERS_ONLY"))
    {
        for (int k = 0; k < n; k++) ok[k] = true;
        one(atoi(only));
    }
    else if (par)
    {
        const int groups[2][2] = { { 0, 2 }, { 2, 5 } };
        for (const auto& g : groups)
        {
            std::vector<std::thread> ts;
            for (int k = g[0]; k < g[1]; k++) ts.emplace_back(one, k);
            for (auto& t : ts) t.join();
        }
    }
    else
        for (int k = 0; k < n; k++) one(k);
    for (int k = 0; k < n; k++)
        if (!ok[k]) { fprintf(stderr, "two_encoders: session %d failed\n", k); return 1; }
    x265_cleanup();
    return 0;
}
