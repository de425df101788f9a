// tools/micro/exit_cost.hip — what a process that has used the GPU pays between "its work is done" and "its parent sees it exit" (the bench's `value` is
// frames / wall clock of the encoder PROCESS; the bound encoder ends ~0.2 s after its last exit handler, the reference ~0 s).  The parent runs itself as a
// child in several shapes and prints, per shape: the child's own clock when main() returned, and the parent's clock when waitpid() came back.
//   exit_cost                runs every shape three times
//   exit_cost child <shape>  (internal)
// shapes: none (no HIP call at all), init (hipFree(0)), vram (1 GiB hipMalloc, touched by a memset), pinned (1 GiB mmap + huge pages + hipHostRegister),
// hostmalloc (1 GiB hipHostMalloc), streams (16 streams, a memset on each), kernel (a resident kernel that is told to leave before exit), all (everything)
#include <hip/hip_runtime.h>
#include <spawn.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <unistd.h>

extern char** environ;

static double now()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

__global__ void resident(volatile int* leave)
{
    while (!*leave)
        __builtin_amdgcn_s_sleep(8);
}

static int child(const char* shape)
{
    const bool all = !strcmp(shape, "all");
    const size_t G = (size_t)1 << 30;
    const double t0 = now();
    if (strcmp(shape, "none"))
        (void)hipFree(0);
    const double tInit = now();
    if (all || !strcmp(shape, "vram"))
    {
        void* d = nullptr;
        (void)hipMalloc(&d, G);
        (void)hipMemset(d, 1, G);
        (void)hipDeviceSynchronize();
    }
    if (all || !strcmp(shape, "pinned"))
    {
        void* a = mmap(nullptr, G, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        madvise(a, G, MADV_HUGEPAGE);
        memset(a, 1, G);
        (void)hipHostRegister(a, G, hipHostRegisterDefault);
    }
    if (all || !strcmp(shape, "hostmalloc"))
    {
        void* h = nullptr;
        (void)hipHostMalloc(&h, G, hipHostMallocDefault);
        memset(h, 1, G);
    }
    if (all || !strcmp(shape, "streams"))
    {
        void* d = nullptr;
        (void)hipMalloc(&d, 1 << 20);
        for (int i = 0; i < 16; i++)
        {
            hipStream_t s;
            (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            (void)hipMemsetAsync(d, i, 1 << 20, s);
            (void)hipStreamSynchronize(s);
        }
    }
    if (all || !strcmp(shape, "kernel"))
    {
        int* flag = nullptr;
        (void)hipHostMalloc((void**)&flag, 64, hipHostMallocCoherent | hipHostMallocMapped);
        *flag = 0;
        hipStream_t s;
        (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        hipLaunchKernelGGL(resident, dim3(64), dim3(256), 0, s, (volatile int*)flag);
        usleep(20000);
        *flag = 1;
        (void)hipStreamSynchronize(s);
    }
    const double t1 = now();
    printf("%.6f %.6f %.6f\n", tInit - t0, t1 - tInit, t1);
    fflush(stdout);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc >= 3 && !strcmp(argv[1], "child"))
        return child(argv[2]);
    const char* shapes[] = { "none", "init", "vram", "pinned", "hostmalloc", "streams", "kernel", "all" };
    printf("# shape: child's hipFree(0) ms | its work ms | from main()'s return to the parent's waitpid() ms | whole process ms   (three runs)\n");
    for (const char* shape : shapes)
        for (int k = 0; k < 3; k++)
        {
            int fds[2];
            if (pipe(fds)) return 1;
            posix_spawn_file_actions_t fa;
            posix_spawn_file_actions_init(&fa);
            posix_spawn_file_actions_adddup2(&fa, fds[1], 1);
            posix_spawn_file_actions_addclose(&fa, fds[0]);
            char* args[] = { argv[0], (char*)"child", (char*)shape, nullptr };
            pid_t pid;
            const double t0 = now();
            if (posix_spawn(&pid, "/proc/self/exe", &fa, nullptr, args, environ)) return 1;
            close(fds[1]);
            char buf[256] = { 0 };
            ssize_t n = read(fds[0], buf, sizeof(buf) - 1);
            (void)n;
            int st;
            waitpid(pid, &st, 0);
            const double t1 = now();
            close(fds[0]);
            double a = 0, b = 0, c = 0;
            sscanf(buf, "%lf %lf %lf", &a, &b, &c);
            printf("%-10s init %7.1f  work %7.1f  exit %7.1f  process %7.1f\n", shape, a * 1e3, b * 1e3, (t1 - c) * 1e3, (t1 - t0) * 1e3);
            fflush(stdout);
        }
    return 0;
}
