// tools/micro/launch_cost.hip — what a kernel launch costs the CALLING thread (CPU time inside hipLaunchKernelGGL / hipMemcpyAsync / hipEventRecord) while a
// resident kernel occupies a stream of the same process, as the bound encoder's job server does.  The CPU profile of the bound encoder shows ~2.6 % of its
// samples below hipModuleLaunchKernel on the library's worker threads for ~2 700 launches per second: ~100 us per launch if it were all launch cost.
//   launch_cost            runs: no resident kernel / resident kernel on a high-priority stream / resident kernel on a default-priority stream
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <unistd.h>

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double cpu_us()
{
    timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

__global__ void resident(volatile int* leave)
{
    while (!*leave)
        __builtin_amdgcn_s_sleep(2);
}
__global__ void tiny(int* p) { if (threadIdx.x == 0 && p) p[blockIdx.x] = 1; }

static void measure(const char* what, hipStream_t st, int* d, char* h, char* dbuf)
{
    const int N = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipStreamSynchronize(st);
    double wall = 0, cpu = 0, wallC = 0, cpuC = 0, wallE = 0, cpuE = 0;
    for (int i = 0; i < N; i++)
    {
        double w0 = now_us(), c0 = cpu_us();
        hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, st, d);
        wall += now_us() - w0; cpu += cpu_us() - c0;
        w0 = now_us(); c0 = cpu_us();
        (void)hipMemcpyAsync(h, dbuf, 65536, hipMemcpyDeviceToHost, st);
        wallC += now_us() - w0; cpuC += cpu_us() - c0;
        w0 = now_us(); c0 = cpu_us();
        (void)hipEventRecord(i & 1 ? e1 : e0, st);
        wallE += now_us() - w0; cpuE += cpu_us() - c0;
        if (i % 50 == 49) (void)hipStreamSynchronize(st);
    }
    (void)hipStreamSynchronize(st);
    printf("%-52s launch %6.1f us wall / %6.1f us CPU   64 KB d2h copy %6.1f / %6.1f   event record %6.1f / %6.1f\n", what, wall / N, cpu / N, wallC / N, cpuC / N, wallE / N, cpuE / N);
    fflush(stdout);
}

int main()
{
    if (hipSetDevice(0) != hipSuccess) { fprintf(stderr, "no device\n"); return 2; }
    (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
    int* d; char* h; char* dbuf; int* flag;
    (void)hipMalloc((void**)&d, 64 * sizeof(int));
    (void)hipMalloc((void**)&dbuf, 65536);
    (void)hipHostMalloc((void**)&h, 65536, hipHostMallocDefault);
    (void)hipHostMalloc((void**)&flag, 64, hipHostMallocCoherent | hipHostMallocMapped);
    hipStream_t st;
    (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    measure("no resident kernel", st, d, h, dbuf);
    for (int prio = 0; prio < 2; prio++)
    {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        hipStream_t rs;
        if (prio == 0) (void)hipStreamCreateWithPriority(&rs, hipStreamNonBlocking, hi);
        else (void)hipStreamCreateWithFlags(&rs, hipStreamNonBlocking);
        *flag = 0;
        hipLaunchKernelGGL(resident, dim3(64), dim3(256), 0, rs, (volatile int*)flag);
        usleep(20000);
        measure(prio == 0 ? "resident kernel (64 workgroups) on a high-priority stream" : "resident kernel on a default-priority stream", st, d, h, dbuf);
        // a second and third launching stream, as the library's workers have
        hipStream_t s2;
        (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
        measure("   ... launching from another new stream", s2, d, h, dbuf);
        *flag = 1;
        (void)hipStreamSynchronize(rs);
        (void)hipStreamDestroy(rs);
    }
    return 0;
}
