// bar_mailbox.hip — can the host write a job straight into DEVICE memory (large BAR), and what does that buy the hand-off?
// (measurement aid for DESIGN.md §4f: the resident server polls its doorbell and reads header + pixels over PCIe — two round trips of the 8.4 us a
// 32x32 CU's first unit takes; a mailbox in device memory turns both into posted writes by the host and local reads by the device)
//   usage: bar_mailbox <kind> [payload bytes = 6144] [hdp flush = 0]
//                               hdp flush 1: after ringing, the host writes the device's HDP_MEM_FLUSH_CNTL register (HSA_AMD_AGENT_INFO_HDP_FLUSH): host
//                               writes into VRAM pass through the HDP block, which may hold them back for a while
//                               kind 0: doorbell + payload in host-coherent pinned memory (round 4's layout, the baseline)
//                               kind 1: hipMalloc memory written by the host through its device pointer
//                               kind 2: hipExtMallocWithFlags(hipDeviceMallocFinegrained)
//                               kind 3: hipExtMallocWithFlags(hipDeviceMallocUncached)
// One run = one kind (a host store to unmapped device memory is a SIGSEGV: the caller's shell sees the exit status).  Ping-pong: the host writes a 6 KB
// payload + doorbell, a resident single-workgroup kernel (bounded to 3 s of the device clock) sees the doorbell, sums the payload (256 threads),
// writes sum + echo to host-coherent memory; the host measures doorbell -> echo.  Every host wait is bounded.
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
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
struct Out { uint32_t echo; uint32_t sum; uint32_t spanTicks; uint32_t pad[29]; };

__global__ __launch_bounds__(256) void pong_kernel(In* in, Out* out, uint64_t maxTicks, int payloadBytes)
{
    __shared__ uint32_t sTicket, sSum;
    __shared__ uint64_t sSeen;
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
            sTicket = v; sSum = 0; sSeen = wall_clock64();
        }
        __syncthreads();
        const uint32_t v = sTicket;
        if (v == last || v == 0xffffffffu) return;
        last = v;
        uint32_t s = 0;
        const uint32_t* p = (const uint32_t*)in->payload;
        for (int i = threadIdx.x; i < payloadBytes / 4; i += 256)
            s += __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        atomicAdd(&sSum, s);
        __syncthreads();
        if (threadIdx.x == 0)
        {
            out->sum = sSum;
            out->spanTicks = (uint32_t)(wall_clock64() - sSeen);
            __hip_atomic_store(&out->echo, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __syncthreads();
    }
}

int main(int argc, char** argv)
{
    const int kind = argc > 1 ? atoi(argv[1]) : 0;
    const int payload = argc > 2 ? atoi(argv[2]) & ~63 : kPayload;
    const int hdpFlush = argc > 3 ? atoi(argv[3]) : 0;
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
    volatile uint32_t* hdpReg = nullptr;
    if (hdpFlush)
    {
        // the first GPU agent = HIP device 0 on a one-GPU box
        hsa_agent_t gpu = { 0 };
        hsa_iterate_agents([](hsa_agent_t a, void* data) {
            hsa_device_type_t t;
            if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) == HSA_STATUS_SUCCESS && t == HSA_DEVICE_TYPE_GPU && !((hsa_agent_t*)data)->handle) *(hsa_agent_t*)data = a;
            return HSA_STATUS_SUCCESS; }, &gpu);
        hsa_amd_hdp_flush_t regs = { nullptr, nullptr };
        if (gpu.handle && hsa_agent_get_info(gpu, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_HDP_FLUSH, &regs) == HSA_STATUS_SUCCESS) hdpReg = regs.HDP_MEM_FLUSH_CNTL;
        printf("HDP_MEM_FLUSH_CNTL at %p\n", (void*)hdpReg); fflush(stdout);
    }
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipLaunchKernelGGL(pong_kernel, 1, 256, 0, st, inDev, outDev, (uint64_t)300 * 1000 * 1000, payload);
    std::vector<uint8_t> src(kPayload);
    std::vector<double> rt, wr, dv;
    bool ok = true;
    for (uint32_t n = 1; n <= 4000 && ok; n++)
    {
        uint32_t want = 0;
        for (int i = 0; i < payload; i++) src[i] = (uint8_t)(i * 7 + n);
        for (int i = 0; i < payload / 4; i++) want += ((uint32_t*)src.data())[i];
        const double t0 = now_us();
        memcpy(in->payload, src.data(), payload);
        _mm_sfence();
        const double t1 = now_us();
        __atomic_store_n(&in->doorbell, n, __ATOMIC_RELEASE);
        _mm_sfence();
        if (hdpReg) { *hdpReg = 1u; _mm_sfence(); }
        while (__atomic_load_n(&out->echo, __ATOMIC_ACQUIRE) != n)
        {
            _mm_pause();
            if (now_us() - t1 > 500000) { printf("no echo for ticket %u within 0.5 s\n", n); ok = false; break; }
        }
        const double t2 = now_us();
        if (ok && out->sum != want) { printf("ticket %u: payload sum %u, expected %u (the device read stale bytes)\n", n, out->sum, want); ok = false; }
        if (n > 200) { rt.push_back(t2 - t1); wr.push_back(t1 - t0); dv.push_back(out->spanTicks * 0.01); }
    }
    __atomic_store_n(&in->doorbell, 0xffffffffu, __ATOMIC_RELEASE);
    _mm_sfence();
    CK(hipStreamSynchronize(st));
    if (ok)
    {
        std::sort(rt.begin(), rt.end()); std::sort(wr.begin(), wr.end()); std::sort(dv.begin(), dv.end());
        printf("kind %d, %d-byte payload, hdp flush %d: %zu ping-pongs: payload write median %.2f us (p90 %.2f); doorbell -> echo median %.2f us, p10 %.2f, p90 %.2f; of it on the "
               "device (doorbell seen -> echo issued) median %.2f us: transport both ways %.2f us\n", kind, payload, hdpFlush ? (hdpReg ? 1 : -1) : 0,
               rt.size(), wr[wr.size() / 2], wr[wr.size() * 9 / 10], rt[rt.size() / 2], rt[rt.size() / 10], rt[rt.size() * 9 / 10], dv[dv.size() / 2],
               rt[rt.size() / 2] - dv[dv.size() / 2]);
    }
    return ok ? 0 : 3;
}
