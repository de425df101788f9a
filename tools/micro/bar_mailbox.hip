// bar_mailbox.hip — can the host write a job straight into DEVICE memory (large BAR), and what does that buy the hand-off?
// (measurement aid for DESIGN.md §4f: the resident server polls its doorbell and reads header + pixels over PCIe — two round trips of the 8.4 us a
// 32x32 CU's first unit takes; a mailbox in device memory turns both into posted writes by the host and local reads by the device)
//   usage: bar_mailbox <kind>   kind 0: doorbell + payload in host-coherent pinned memory (round 4's layout, the baseline)
//                               kind 1: hipMalloc memory written by the host through its device pointer
//                               kind 2: hipExtMallocWithFlags(hipDeviceMallocFinegrained)
//                               kind 3: hipExtMallocWithFlags(hipDeviceMallocUncached)
// One run = one kind (a host store to unmapped device memory is a SIGSEGV: the caller's shell sees the exit status).  Ping-pong: the host writes a 6 KB
// payload + doorbell, a resident single-workgroup kernel (bounded to 3 s of the device clock) sees the doorbell, sums the payload (256 threads),
// writes sum + echo to host-coherent memory; the host measures doorbell -> echo.  Every host wait is bounded.
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

constexpr int kPayload = 6144;
struct In { uint32_t doorbell; uint32_t pad[31]; uint8_t payload[kPayload]; };
struct Out { uint32_t echo; uint32_t sum; uint32_t seenTicks; uint32_t pad[29]; };

__global__ __launch_bounds__(256) void pong_kernel(In* in, Out* out, uint64_t maxTicks)
{
    __shared__ uint32_t sTicket, sSum;
    const uint64_t t0 = wall_clock64();
    uint32_t last = 0;
    for (;;)
    {
        if (threadIdx.x == 0)
        {
            uint32_t v;
            for (;;)
            {
                v = __hip_atomic_load(&in->doorbell, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
                if (v != last || wall_clock64() - t0 > maxTicks) break;
                __builtin_amdgcn_s_sleep(2);
            }
            sTicket = v; sSum = 0;
        }
        __syncthreads();
        const uint32_t v = sTicket;
        if (v == last || v == 0xffffffffu) return;
        last = v;
        uint32_t s = 0;
        const uint32_t* p = (const uint32_t*)in->payload;
        for (int i = threadIdx.x; i < kPayload / 4; i += 256)
            s += __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        atomicAdd(&sSum, s);
        __syncthreads();
        if (threadIdx.x == 0)
        {
            out->sum = sSum;
            __hip_atomic_store(&out->echo, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __syncthreads();
    }
}

int main(int argc, char** argv)
{
    const int kind = argc > 1 ? atoi(argv[1]) : 0;
    CK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("kind %d; device %s, isLargeBar %d\n", kind, prop.name, prop.isLargeBar);
    fflush(stdout);
    In* in = nullptr; In* inDev = nullptr;
    if (kind == 0) { CK(hipHostMalloc((void**)&in, sizeof(In), hipHostMallocCoherent | hipHostMallocMapped)); CK(hipHostGetDevicePointer((void**)&inDev, in, 0)); }
    else if (kind == 1) { CK(hipMalloc((void**)&in, sizeof(In))); inDev = in; }
    else if (kind == 2) { CK(hipExtMallocWithFlags((void**)&in, sizeof(In), hipDeviceMallocFinegrained)); inDev = in; }
    else { CK(hipExtMallocWithFlags((void**)&in, sizeof(In), hipDeviceMallocUncached)); inDev = in; }
    if (kind) CK(hipMemset(in, 0, sizeof(In))); else memset(in, 0, sizeof(In));
    CK(hipDeviceSynchronize());
    Out* out; Out* outDev;
    CK(hipHostMalloc((void**)&out, sizeof(Out), hipHostMallocCoherent | hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void**)&outDev, out, 0));
    memset(out, 0, sizeof(Out));
    printf("mailbox at %p; first host store ...\n", (void*)in); fflush(stdout);
    ((volatile uint32_t*)in->payload)[0] = 1;                       // SIGSEGV here = the host cannot reach this memory
    _mm_sfence();
    printf("host store went through; host load reads %u\n", ((volatile uint32_t*)in->payload)[0]); fflush(stdout);
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipLaunchKernelGGL(pong_kernel, 1, 256, 0, st, inDev, outDev, (uint64_t)300 * 1000 * 1000);
    std::vector<uint8_t> src(kPayload);
    std::vector<double> rt, wr;
    bool ok = true;
    for (uint32_t n = 1; n <= 4000 && ok; n++)
    {
        uint32_t want = 0;
        for (int i = 0; i < kPayload; i++) src[i] = (uint8_t)(i * 7 + n);
        for (int i = 0; i < kPayload / 4; i++) want += ((uint32_t*)src.data())[i];
        const double t0 = now_us();
        memcpy(in->payload, src.data(), kPayload);
        _mm_sfence();
        const double t1 = now_us();
        __atomic_store_n(&in->doorbell, n, __ATOMIC_RELEASE);
        _mm_sfence();
        while (__atomic_load_n(&out->echo, __ATOMIC_ACQUIRE) != n)
        {
            _mm_pause();
            if (now_us() - t1 > 500000) { printf("no echo for ticket %u within 0.5 s\n", n); ok = false; break; }
        }
        const double t2 = now_us();
        if (ok && out->sum != want) { printf("ticket %u: payload sum %u, expected %u (the device read stale bytes)\n", n, out->sum, want); ok = false; }
        if (n > 200) { rt.push_back(t2 - t1); wr.push_back(t1 - t0); }
    }
    __atomic_store_n(&in->doorbell, 0xffffffffu, __ATOMIC_RELEASE);
    _mm_sfence();
    CK(hipStreamSynchronize(st));
    if (ok)
    {
        std::sort(rt.begin(), rt.end()); std::sort(wr.begin(), wr.end());
        printf("kind %d: %zu ping-pongs: 6 KB payload write median %.2f us (p90 %.2f); doorbell -> echo (device sums the payload) median %.2f us, p10 %.2f, p90 %.2f\n", kind,
               rt.size(), wr[wr.size() / 2], wr[wr.size() * 9 / 10], rt[rt.size() / 2], rt[rt.size() / 10], rt[rt.size() * 9 / 10]);
    }
    return ok ? 0 : 3;
}
