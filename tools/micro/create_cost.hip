// create_cost.hip — what does creating the per-picture device objects cost?  (measurement aid for DESIGN.md §5 "start-up": a device copy of a source
// picture takes 4-10 ms to create in round 4's start-up marks, 36 of them before the first frame is decided)
// Times, 40 times each like a 1080p encoder's start-up: hipStreamCreateWithFlags, hipMalloc of 2 MB, a 2 MB page-locked block both ways
// (hipHostMalloc; mmap + MADV_HUGEPAGE + touch + hipHostRegister), hipEventCreate pairs — and the same hipMalloc as one 40-picture slab.
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const int N = 40;
    const size_t bytes = (size_t)2048 * 1080;
    CK(hipSetDevice(0));
    void* warm; CK(hipMalloc(&warm, 1 << 20)); CK(hipFree(warm));
    hipStream_t st[N]; void* d[N]; void* h[N]; hipEvent_t ev[2 * N];
    double t0 = now_ms();
    for (int i = 0; i < N; i++) CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
    double t1 = now_ms();
    printf("hipStreamCreateWithFlags x %d: %.3f ms each\n", N, (t1 - t0) / N);
    t0 = now_ms();
    for (int i = 0; i < N; i++) CK(hipMalloc(&d[i], bytes + 256));
    t1 = now_ms();
    printf("hipMalloc(%zu) x %d: %.3f ms each\n", bytes, N, (t1 - t0) / N);
    t0 = now_ms();
    for (int i = 0; i < N; i++) CK(hipHostMalloc(&h[i], bytes, hipHostMallocDefault));
    t1 = now_ms();
    printf("hipHostMalloc(%zu) x %d: %.3f ms each\n", bytes, N, (t1 - t0) / N);
    t0 = now_ms();
    for (int i = 0; i < N; i++) CK(hipHostFree(h[i]));
    t1 = now_ms();
    printf("hipHostFree x %d: %.3f ms each\n", N, (t1 - t0) / N);
    const size_t huge = (size_t)2 << 20, span = (bytes + huge - 1) & ~(huge - 1);
    t0 = now_ms();
    for (int i = 0; i < N; i++)
    {
        char* m = (char*)mmap(NULL, span + huge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        char* a = (char*)(((uintptr_t)m + huge - 1) & ~(uintptr_t)(huge - 1));
        madvise(a, span, MADV_HUGEPAGE);
        memset(a, 0, bytes);
        CK(hipHostRegister(a, span, hipHostRegisterDefault));
        h[i] = a;
    }
    t1 = now_ms();
    printf("mmap + MADV_HUGEPAGE + touch + hipHostRegister(%zu) x %d: %.3f ms each\n", span, N, (t1 - t0) / N);
    t0 = now_ms();
    for (int i = 0; i < 2 * N; i++) CK(hipEventCreateWithFlags(&ev[i], hipEventDefault));
    t1 = now_ms();
    printf("hipEventCreate x %d: %.3f ms each\n", 2 * N, (t1 - t0) / (2 * N));
    t0 = now_ms();
    void* slab; CK(hipMalloc(&slab, (bytes + 256) * N));
    t1 = now_ms();
    printf("one hipMalloc of %d pictures (%zu bytes): %.3f ms\n", N, (bytes + 256) * N, t1 - t0);
    t0 = now_ms();
    char* hs; CK(hipHostMalloc((void**)&hs, bytes * N, hipHostMallocDefault));
    t1 = now_ms();
    printf("one hipHostMalloc of %d pictures: %.3f ms\n", N, t1 - t0);
    // a first copy through each stream (queue creation is lazy in some runtimes)
    t0 = now_ms();
    for (int i = 0; i < N; i++) { CK(hipMemcpyAsync(d[i], hs + (size_t)i * bytes, bytes, hipMemcpyHostToDevice, st[i])); CK(hipStreamSynchronize(st[i])); }
    t1 = now_ms();
    printf("first 2 MB h2d + synchronise on each new stream: %.3f ms each\n", (t1 - t0) / N);
    t0 = now_ms();
    for (int i = 0; i < N; i++) { CK(hipMemcpyAsync(d[i], hs + (size_t)i * bytes, bytes, hipMemcpyHostToDevice, st[i])); CK(hipStreamSynchronize(st[i])); }
    t1 = now_ms();
    printf("second: %.3f ms each\n", (t1 - t0) / N);
    return 0;
}
