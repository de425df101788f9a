// handoff.hip — what does it cost a host thread to hand one CU's worth of work to the MI355X and get the result back?
// (measurement aid for DESIGN.md §4f, not part of the product; the product's own path is measured by tools/cuserve_bench.py)
//
// One "job" = IN bytes the device must read out of page-locked host memory + OUT bytes it must write back + a completion word.  Sizes of
// interest: 3 KB in / 6 KB out (the residual chain of a 32x32 CU: source + prediction in, levels + reconstructed residual out),
// 12 KB / 24 KB (64x64), 320 B / 160 B (a 35-mode intra scan).  The device work itself is a copy, so the figures are pure hand-off.
//
// Variants:
//   sync      hipLaunchKernelGGL on the thread's own stream + hipStreamSynchronize
//   flag      hipLaunchKernelGGL on the thread's own stream; the kernel ends by storing a sequence number to host memory; the host spins on it
//   mailbox   a resident kernel, one workgroup per slot, polls the slot's doorbell word in coherent host memory (system-scope atomic load,
//             s_sleep between polls), does the job and stores the completion word; the host writes the doorbell and spins
// each with T host threads submitting concurrently (T = 1, 4, 16), each thread on its own stream / slot.
// Also reports: what hipFree / hipHostFree / hipMalloc do while the resident kernel is running (a resident kernel must not deadlock the
// library's own memory management).
#include <hip/hip_runtime.h>
#include <atomic>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Slot
{
    volatile uint32_t doorbell;      // host -> device: sequence number of the job in `in`
    uint32_t inBytes, outBytes, quit;
    char pad0[48];
    volatile uint32_t done;          // device -> host
    char pad1[60];
    char in[32 * 1024];
    char out[64 * 1024];
};

__device__ __forceinline__ void do_job(Slot* s, uint32_t seq)
{
    const int inN = (int)(s->inBytes >> 4), outN = (int)(s->outBytes >> 4);
    const uint4* in = (const uint4*)s->in;
    uint4* out = (uint4*)s->out;
    uint4 acc = make_uint4(seq, 0, 0, 0);
    for (int i = threadIdx.x; i < inN; i += blockDim.x)
    {
        const uint4 v = in[i];
        acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
    }
    for (int i = threadIdx.x; i < outN; i += blockDim.x)
        out[i] = acc;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_store((uint32_t*)&s->done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void job_kernel(Slot* s, uint32_t seq) { do_job(s, seq); }
__global__ void empty_kernel() {}

// resident: workgroup b serves slot b until its quit word is set or `maxPolls` polls have gone by without work (safety net)
__global__ __launch_bounds__(256) void server_kernel(Slot* slots, uint32_t sleepArg, uint64_t maxIdlePolls)
{
    Slot* s = slots + blockIdx.x;
    __shared__ uint32_t sh;
    uint32_t last = 0;
    uint64_t idle = 0;
    for (;;)
    {
        if (threadIdx.x == 0)
        {
            uint32_t v;
            for (;;)
            {
                v = __hip_atomic_load((uint32_t*)&s->doorbell, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
                if (v != last) break;
                if (++idle > maxIdlePolls) { v = 0xffffffffu; break; }       // safety net: a forgotten server ends by itself
                if (sleepArg) __builtin_amdgcn_s_sleep(8);
            }
            sh = v;
        }
        __syncthreads();
        const uint32_t v = sh;
        __syncthreads();
        if (v == 0xffffffffu)
            return;
        idle = 0;
        last = v;
        do_job(s, v);
    }
}

struct Stats { double mean, p50, p99, max; };
static Stats stats(std::vector<double>& v)
{
    std::sort(v.begin(), v.end());
    double s = 0; for (double x : v) s += x;
    return { s / v.size(), v[v.size() / 2], v[(size_t)(v.size() * 0.99)], v.back() };
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    CK(hipSetDevice(0));
    const int maxT = 16;
    Slot* slots;
    CK(hipHostMalloc((void**)&slots, sizeof(Slot) * maxT, hipHostMallocCoherent | hipHostMallocMapped));
    memset(slots, 0, sizeof(Slot) * maxT);
    Slot* dslots;
    CK(hipHostGetDevicePointer((void**)&dslots, slots, 0));
    std::vector<hipStream_t> st(maxT);
    for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipStream_t serverStream;
    CK(hipStreamCreateWithFlags(&serverStream, hipStreamNonBlocking));

    // launch latency floor
    {
        for (int i = 0; i < 100; i++) { hipLaunchKernelGGL(empty_kernel, 1, 1, 0, st[0]); CK(hipStreamSynchronize(st[0])); }
        std::vector<double> v;
        for (int i = 0; i < iters; i++) { double t0 = now_us(); hipLaunchKernelGGL(empty_kernel, 1, 1, 0, st[0]); CK(hipStreamSynchronize(st[0])); v.push_back(now_us() - t0); }
        Stats a = stats(v);
        printf("empty kernel + hipStreamSynchronize, 1 thread: mean %.1f us, median %.1f, p99 %.1f\n", a.mean, a.p50, a.p99);
        std::vector<double> w;
        for (int i = 0; i < iters; i++) { double t0 = now_us(); hipLaunchKernelGGL(empty_kernel, 1, 1, 0, st[0]); w.push_back(now_us() - t0); if ((i & 63) == 63) CK(hipStreamSynchronize(st[0])); }
        CK(hipStreamSynchronize(st[0]));
        Stats b = stats(w);
        printf("hipLaunchKernelGGL alone (host-side cost of one launch): mean %.1f us, median %.1f, p99 %.1f\n", b.mean, b.p50, b.p99);
    }

    struct Size { const char* name; uint32_t in, out; } sizes[] = { { "32x32 CU residual chain (3 KB in, 6 KB out)", 3072, 6144 },
                                                                     { "64x64 CU residual chain (12 KB in, 24 KB out)", 12288, 24576 },
                                                                     { "35-mode intra scan (320 B in, 160 B out)", 320, 160 } };
    const int threadCounts[] = { 1, 4, 16 };
    for (const Size& sz : sizes)
    {
        printf("--- %s\n", sz.name);
        for (int variant = 0; variant < 3; variant++)
        {
            const char* vname = variant == 0 ? "sync   " : variant == 1 ? "flag   " : "mailbox";
            for (int T : threadCounts)
            {
                for (int t = 0; t < maxT; t++) { slots[t].inBytes = sz.in; slots[t].outBytes = sz.out; slots[t].quit = 0; slots[t].doorbell = 0; slots[t].done = 0; }
                std::atomic_thread_fence(std::memory_order_seq_cst);
                if (variant == 2)
                    hipLaunchKernelGGL(server_kernel, T, 256, 0, serverStream, dslots, 1u, (uint64_t)4 * 1000 * 1000);
                std::vector<std::vector<double>> lat(T);
                std::atomic<int> go(0);
                auto body = [&](int t)
                {
                    CK(hipSetDevice(0));
                    Slot* s = slots + t;
                    while (!go.load()) {}
                    std::vector<double>& v = lat[t];
                    for (int i = 1; i <= iters + 200; i++)
                    {
                        memset(s->in, i & 255, sz.in);               // the host's own staging of the job
                        const double t0 = now_us();
                        if (variant == 0)
                        {
                            hipLaunchKernelGGL(job_kernel, 1, 256, 0, st[t], dslots + t, (uint32_t)i);
                            CK(hipStreamSynchronize(st[t]));
                        }
                        else if (variant == 1)
                        {
                            hipLaunchKernelGGL(job_kernel, 1, 256, 0, st[t], dslots + t, (uint32_t)i);
                            while (s->done != (uint32_t)i) __builtin_ia32_pause();
                        }
                        else
                        {
                            __atomic_store_n((uint32_t*)&s->doorbell, (uint32_t)i, __ATOMIC_RELEASE);
                            while (__atomic_load_n((uint32_t*)&s->done, __ATOMIC_ACQUIRE) != (uint32_t)i) __builtin_ia32_pause();
                        }
                        volatile char sink = s->out[sz.out - 1]; (void)sink;
                        const double t1 = now_us();
                        if (i > 200) v.push_back(t1 - t0);
                    }
                    if (variant == 1) CK(hipStreamSynchronize(st[t]));
                };
                std::vector<std::thread> th;
                for (int t = 0; t < T; t++) th.emplace_back(body, t);
                const double w0 = now_us();
                go = 1;
                for (auto& x : th) x.join();
                const double wall = now_us() - w0;
                if (variant == 2)
                {
                    for (int t = 0; t < T; t++) __atomic_store_n((uint32_t*)&slots[t].doorbell, 0xffffffffu, __ATOMIC_RELEASE);
                    CK(hipStreamSynchronize(serverStream));
                }
                std::vector<double> all;
                for (auto& v : lat) all.insert(all.end(), v.begin(), v.end());
                Stats a = stats(all);
                printf("  %s %2d thread%s: round trip mean %6.1f us, median %6.1f, p99 %6.1f, max %7.1f; %.0f jobs/s in total\n", vname, T, T > 1 ? "s" : " ", a.mean, a.p50, a.p99, a.max,
                       (double)T * (iters + 200) / (wall * 1e-6));
                fflush(stdout);
            }
        }
    }

    // memory management beside a resident kernel: does hipFree / hipHostFree / hipMalloc wait for it?
    {
        for (int t = 0; t < maxT; t++) { slots[t].quit = 0; slots[t].doorbell = 0; slots[t].done = 0; }
        void* d0; void* h0;
        CK(hipMalloc(&d0, 1 << 20));
        CK(hipHostMalloc(&h0, 1 << 20, hipHostMallocDefault));
        hipLaunchKernelGGL(server_kernel, 4, 256, 0, serverStream, dslots, 1u, (uint64_t)4 * 1000 * 1000);   // bounded: gives up after 4 M polls
        std::atomic<int> stage(0);
        std::thread mm([&]
        {
            CK(hipSetDevice(0));
            void* d1; double t0 = now_us();
            CK(hipMalloc(&d1, 1 << 20)); stage = 1;
            printf("beside a resident kernel: hipMalloc returned after %.0f us\n", now_us() - t0); fflush(stdout);
            t0 = now_us(); CK(hipFree(d0)); stage = 2;
            printf("beside a resident kernel: hipFree returned after %.0f us\n", now_us() - t0); fflush(stdout);
            t0 = now_us(); CK(hipHostFree(h0)); stage = 3;
            printf("beside a resident kernel: hipHostFree returned after %.0f us\n", now_us() - t0); fflush(stdout);
            void* reg = aligned_alloc(4096, 1 << 20); memset(reg, 0, 1 << 20);
            t0 = now_us(); CK(hipHostRegister(reg, 1 << 20, hipHostRegisterDefault)); CK(hipHostUnregister(reg)); stage = 4;
            printf("beside a resident kernel: hipHostRegister + hipHostUnregister returned after %.0f us\n", now_us() - t0); fflush(stdout);
            (void)hipFree(d1);
        });
        const double t0 = now_us();
        while (stage.load() < 4 && now_us() - t0 < 2e6) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        const int reached = stage.load();
        if (reached < 4)
            printf("beside a resident kernel: memory management call number %d (1 hipMalloc, 2 hipFree, 3 hipHostFree, 4 hipHostRegister) did NOT return within 2 s "
                   "-> it waits for the resident kernel; releasing the kernel now\n", reached + 1);
        for (int t = 0; t < 4; t++) __atomic_store_n((uint32_t*)&slots[t].doorbell, 0xffffffffu, __ATOMIC_RELEASE);
        mm.join();
        CK(hipStreamSynchronize(serverStream));
    }
    printf("done\n");
    return 0;
}
