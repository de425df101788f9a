// pin_thp.hip — what does page-locking cost, and do transparent huge pages make it cheaper?  (measurement aid for DESIGN.md §5 "start-up and teardown")
// 64 MB (one reference-picture mirror's host planes at 1080p): hipHostMalloc + hipHostFree vs anonymous mmap (+ MADV_HUGEPAGE) + first touch +
// hipHostRegister + hipHostUnregister + munmap; and whether a device-to-host copy into each runs at the same speed.
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
    const size_t bytes = 64 << 20;
    CK(hipSetDevice(0));
    char* d; CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 3, bytes));
    FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
    char line[256] = "?";
    if (f) { if (!fgets(line, sizeof(line), f)) line[0] = 0; fclose(f); }
    printf("transparent_hugepage/enabled: %s", line);
    for (int rep = 0; rep < 3; rep++)
    {
        double t0 = now_ms();
        char* h; CK(hipHostMalloc((void**)&h, bytes, hipHostMallocDefault));
        double t1 = now_ms();
        CK(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost));
        double t2 = now_ms();
        CK(hipHostFree(h));
        double t3 = now_ms();
        printf("hipHostMalloc %.2f ms, 64 MB d2h %.2f ms (%.1f GB/s), hipHostFree %.2f ms\n", t1 - t0, t2 - t1, bytes / (t2 - t1) * 1e-6, t3 - t2);
        for (int huge = 0; huge < 2; huge++)
        {
            t0 = now_ms();
            char* m = (char*)mmap(NULL, bytes + (2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (m == MAP_FAILED) { perror("mmap"); return 2; }
            char* a = (char*)(((uintptr_t)m + (2 << 20) - 1) & ~(uintptr_t)((2 << 20) - 1));
            if (huge && madvise(a, bytes, MADV_HUGEPAGE)) perror("madvise");
            memset(a, 0, bytes);
            t1 = now_ms();
            CK(hipHostRegister(a, bytes, hipHostRegisterDefault));
            t2 = now_ms();
            CK(hipMemcpy(a, d, bytes, hipMemcpyDeviceToHost));
            double t3b = now_ms();
            CK(hipHostUnregister(a));
            t3 = now_ms();
            munmap(m, bytes + (2 << 20));
            double t4 = now_ms();
            printf("mmap%s + touch %.2f ms, hipHostRegister %.2f ms, 64 MB d2h %.2f ms (%.1f GB/s), hipHostUnregister %.2f ms, munmap %.2f ms\n", huge ? " + MADV_HUGEPAGE" : "", t1 - t0, t2 - t1,
                   t3b - t2, bytes / (t3b - t2) * 1e-6, t3 - t3b, t4 - t3);
        }
    }
    return 0;
}
