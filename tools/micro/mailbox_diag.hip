// mailbox_diag.hip — why did the resident-kernel variant of tools/micro/handoff hang on the MI355X box (round 4, first GPU call)?
// A resident kernel (bounded to ~1.5 s by the device's own 100 MHz clock) that
//   * stores a heartbeat counter to host-coherent memory every iteration (is device -> host visible while the kernel runs?)
//   * reads a doorbell word four ways: system-scope atomic load, agent-scope atomic load, volatile load, plain load after __threadfence_system()
//     and reports back what each saw (is host -> device visible while the kernel runs, and through which kind of load?)
// Run for hipHostMallocCoherent, hipHostMallocDefault, hipHostMallocNonCoherent and hipHostRegister'ed memory.  Every wait on the host is bounded.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)

struct Box
{
    uint32_t doorbell; uint32_t pad0[15];
    uint32_t heartbeat; uint32_t sawSystem, sawAgent, sawVolatile, sawPlain; uint32_t started; uint32_t pad1[10];
    uint64_t firstSeenTick[4]; uint64_t startTick;
};

__global__ void diag_kernel(Box* b, uint64_t maxTicks)
{
    const uint64_t t0 = wall_clock64();
    __hip_atomic_store(&b->started, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    b->startTick = t0;
    uint32_t beat = 0;
    bool seen[4] = { false, false, false, false };
    for (;;)
    {
        const uint64_t now = wall_clock64();
        if (now - t0 > maxTicks) break;
        __hip_atomic_store(&b->heartbeat, ++beat, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint32_t a = __hip_atomic_load(&b->doorbell, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint32_t c = __hip_atomic_load(&b->doorbell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t v = *(volatile uint32_t*)&b->doorbell;
        __threadfence_system();
        const uint32_t p = b->doorbell;
        if (a && !seen[0]) { seen[0] = true; b->sawSystem = a; b->firstSeenTick[0] = now - t0; }
        if (c && !seen[1]) { seen[1] = true; b->sawAgent = c; b->firstSeenTick[1] = now - t0; }
        if (v && !seen[2]) { seen[2] = true; b->sawVolatile = v; b->firstSeenTick[2] = now - t0; }
        if (p && !seen[3]) { seen[3] = true; b->sawPlain = p; b->firstSeenTick[3] = now - t0; }
        if (a == 0xffffffffu) break;
        __builtin_amdgcn_s_sleep(8);
    }
    __threadfence_system();
}

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    CK(hipSetDevice(0));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    struct { const char* name; unsigned flags; int reg; } kinds[] = { { "hipHostMallocCoherent|Mapped", hipHostMallocCoherent | hipHostMallocMapped, 0 },
                                                                       { "hipHostMallocDefault", hipHostMallocDefault, 0 },
                                                                       { "hipHostMallocNonCoherent", hipHostMallocNonCoherent, 0 },
                                                                       { "malloc + hipHostRegister", 0, 1 } };
    for (auto& k : kinds)
    {
        Box* h = nullptr;
        if (k.reg) { h = (Box*)aligned_alloc(4096, 4096); CK(hipHostRegister(h, 4096, hipHostRegisterMapped)); }
        else CK(hipHostMalloc((void**)&h, 4096, k.flags));
        memset(h, 0, 4096);
        Box* d = nullptr;
        CK(hipHostGetDevicePointer((void**)&d, h, 0));
        const double t0 = now_ms();
        hipLaunchKernelGGL(diag_kernel, 1, 1, 0, st, d, (uint64_t)150 * 1000 * 1000);       // 1.5 s of the device's 100 MHz clock
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
        const uint32_t started = __atomic_load_n(&h->started, __ATOMIC_ACQUIRE), hb1 = __atomic_load_n(&h->heartbeat, __ATOMIC_ACQUIRE);
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
        const uint32_t hb2 = __atomic_load_n(&h->heartbeat, __ATOMIC_ACQUIRE);
        const double tRing = now_ms() - t0;
        __atomic_store_n(&h->doorbell, 7u, __ATOMIC_RELEASE);
        std::this_thread::sleep_for(std::chrono::milliseconds(200));
        const uint32_t s1 = h->sawSystem, s2 = h->sawAgent, s3 = h->sawVolatile, s4 = h->sawPlain;
        __atomic_store_n(&h->doorbell, 0xffffffffu, __ATOMIC_RELEASE);
        const double tq = now_ms();
        CK(hipStreamSynchronize(st));
        const double left = now_ms() - tq;
        printf("%-30s host ptr %p device ptr %p\n", k.name, (void*)h, (void*)d);
        printf("   while the kernel ran: started word %u, heartbeat after 100 ms %u, after 200 ms %u  (device -> host %s)\n", started, hb1, hb2,
               hb2 > hb1 && hb1 > 0 ? "VISIBLE" : "NOT visible until the kernel ends");
        printf("   doorbell rung at %.0f ms; 200 ms later the device had reported: system-scope load %u, agent-scope load %u, volatile load %u, plain load after fence %u\n", tRing,
               s1, s2, s3, s4);
        printf("   after the kernel: saw %u / %u / %u / %u, first seen at %.1f / %.1f / %.1f / %.1f ms of device time; heartbeat %u; the kernel left %.1f ms after the quit ring\n",
               h->sawSystem, h->sawAgent, h->sawVolatile, h->sawPlain, h->firstSeenTick[0] * 1e-5, h->firstSeenTick[1] * 1e-5, h->firstSeenTick[2] * 1e-5,
               h->firstSeenTick[3] * 1e-5, h->heartbeat, left);
        fflush(stdout);
        if (k.reg) { CK(hipHostUnregister(h)); free(h); } else CK(hipHostFree(h));
    }
    // memory management beside a resident kernel (bounded: the kernel leaves by itself after 1.5 s of its own clock)
    {
        Box* h = nullptr;
        CK(hipHostMalloc((void**)&h, 4096, hipHostMallocCoherent | hipHostMallocMapped));
        memset(h, 0, 4096);
        Box* d = nullptr;
        CK(hipHostGetDevicePointer((void**)&d, h, 0));
        void* d0; void* h0;
        CK(hipMalloc(&d0, 1 << 20));
        CK(hipHostMalloc(&h0, 1 << 20, hipHostMallocDefault));
        hipStream_t other;
        CK(hipStreamCreateWithFlags(&other, hipStreamNonBlocking));
        hipLaunchKernelGGL(diag_kernel, 1, 1, 0, st, d, (uint64_t)150 * 1000 * 1000);
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
        void* d1;
        double t0 = now_ms(); CK(hipMalloc(&d1, 1 << 20));
        printf("beside a resident kernel: hipMalloc %.2f ms", now_ms() - t0);
        t0 = now_ms(); CK(hipMemsetAsync(d1, 0, 1 << 20, other)); CK(hipStreamSynchronize(other));
        printf(", memset + synchronise on another stream %.2f ms", now_ms() - t0);
        t0 = now_ms(); CK(hipFree(d0));
        printf(", hipFree %.2f ms", now_ms() - t0);
        t0 = now_ms(); CK(hipHostFree(h0));
        printf(", hipHostFree %.2f ms", now_ms() - t0);
        void* reg = aligned_alloc(4096, 1 << 20); memset(reg, 0, 1 << 20);
        t0 = now_ms(); CK(hipHostRegister(reg, 1 << 20, hipHostRegisterDefault)); CK(hipHostUnregister(reg));
        printf(", hipHostRegister + Unregister %.2f ms", now_ms() - t0);
        printf("  (heartbeat %u: the kernel %s)\n", h->heartbeat, __atomic_load_n(&h->heartbeat, __ATOMIC_ACQUIRE) ? "was running" : "was NOT running");
        fflush(stdout);
        __atomic_store_n(&h->doorbell, 0xffffffffu, __ATOMIC_RELEASE);
        CK(hipStreamSynchronize(st));
        CK(hipFree(d1));
    }
    return 0;
}
