// runtime.hip — device selection, memory/stream/event plumbing and error reporting of the C ABI (include/x265hip.h).
// There is deliberately no CPU fallback anywhere in this library: without a gfx950-class device every entry point
// returns X265HIP_ENODEV.
#include "common.h"
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>
#include <mutex>
#include <sys/mman.h>

namespace xh {

static thread_local char t_err[512] = "";
static thread_local int t_dev_ready = 0;   // 0 = not tried, 1 = ok, -1 = failed
static thread_local int t_device = 0;

int set_error(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_err, sizeof(t_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_hip(hipError_t e, const char* what)
{
    if (e == hipSuccess)
        return X265HIP_OK;
    return set_error(e == hipErrorOutOfMemory ? X265HIP_ENOMEM : X265HIP_EHIP, "%s: %s", what, hipGetErrorString(e));
}

static int init_device(int device)
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
    {
        t_dev_ready = -1;
        (void)hipGetLastError();
        return set_error(X265HIP_ENODEV, "libx265hip: no HIP device available (%s); there is no CPU fallback",
                         e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    }
    if (device < 0 || device >= count)
        return set_error(X265HIP_EINVAL, "libx265hip: device %d out of range (have %d)", device, count);
    e = hipSetDevice(device);
    if (e != hipSuccess)
    {
        t_dev_ready = -1;
        return check_hip(e, "hipSetDevice");
    }
    // How the threads that wait for a stream wait (hipSetDeviceFlags): blocking — on the interrupt — by default.  HIP's own default (`auto`) spins while there are
    // fewer contexts than CPUs, and on a host whose CPUs the encoder needs that spin is taken from the encoder: 1080p medium, 5 interleaved rounds on the MI355X
    // box, 45.9 fps / 30.2 CPU-s blocking vs 45.1 / 30.8 auto, 45.6 yield, 45.1 spin (profiles/r06_v1_sync_ab.txt).  X265HIP_SYNC=auto | spin | yield | blocking
    {
        const char* how = getenv("X265HIP_SYNC");
        const unsigned flag = !how || !strcmp(how, "blocking") ? hipDeviceScheduleBlockingSync : !strcmp(how, "spin") ? hipDeviceScheduleSpin : !strcmp(how, "yield") ? hipDeviceScheduleYield : hipDeviceScheduleAuto;
        if (hipSetDeviceFlags(flag) != hipSuccess) (void)hipGetLastError();
    }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess)
        return check_hip(e, "hipGetDeviceProperties");
    if (prop.warpSize != kWave)
    {
        t_dev_ready = -1;
        return set_error(X265HIP_ENODEV, "libx265hip: device %d (%s) has wavefront %d; kernels are built for wave64 gfx950",
                         device, prop.gcnArchName, prop.warpSize);
    }
    t_device = device;
    t_dev_ready = 1;
    return X265HIP_OK;
}

int ensure_device()
{
    if (t_dev_ready == 1)
        return X265HIP_OK;
    if (t_dev_ready == -1)
        return X265HIP_ENODEV;
    // honour a device already chosen by the host framework (e.g. torch.cuda.set_device) on this thread
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess)
        cur = 0;
    return init_device(cur);
}

// places (x265hip_places): place p lives on HIP device g_places[p]; several places may share a device
static int g_places[64];
static std::atomic<int> g_nPlaces{ 0 };
int place_device(int place)
{
    if (place < 0) return -(place + 1);
    return place < g_nPlaces.load() ? g_places[place] : -1;
}
int place_of_device(int device) { return -(device + 1); }

static std::atomic<uint64_t> g_clkSpans[X265HIP_CLK_COUNT], g_clkNs[X265HIP_CLK_COUNT], g_clkBytes[X265HIP_CLK_COUNT];

static std::atomic<int> g_residentWgs[64];
void resident_workgroups(int device, int delta) { if (device >= 0 && device < 64) g_residentWgs[device] += delta; }
int free_compute_units(int device)
{
    static std::atomic<int> cus[64];
    if (device < 0 || device >= 64) return 256;
    int n = cus[device].load();
    if (!n)
    {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || v < 1) { (void)hipGetLastError(); v = 256; }
        cus[device] = n = v;
    }
    const int left = n - g_residentWgs[device].load();
    return left > n / 4 ? left : n / 4;
}

static std::mutex g_streamLock;
static std::vector<hipStream_t> g_streamPool[64];
hipStream_t stream_lease(int device)
{
    if (device < 0 || device >= 64) return nullptr;
    {
        std::lock_guard<std::mutex> g(g_streamLock);
        auto& v = g_streamPool[device];
        if (!v.empty()) { hipStream_t st = v.back(); v.pop_back(); return st; }
    }
    hipStream_t st = nullptr;                       // the caller has made `device` current
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return st;
}
void stream_return(int device, hipStream_t st)
{
    if (!st || device < 0 || device >= 64) return;
    std::lock_guard<std::mutex> g(g_streamLock);
    g_streamPool[device].push_back(st);
}

hipError_t device_free(void* p)
{
    if (!p) return hipSuccess;
    servers_pause();
    const hipError_t e = hipFree(p);
    servers_resume();
    return e;
}

// ---- page-locked host memory for the big buffers (phase planes, SAD tables, staging): anonymous memory on transparent huge pages + hipHostRegister.
// Measured on the MI355X box (tools/micro/pin_thp, 64 MB): hipHostMalloc 10.4 ms + hipHostFree 4.9 ms; mmap + MADV_HUGEPAGE + first touch 3.3 ms +
// hipHostRegister 0.33 ms, unregister + munmap 2.5 ms; device-to-host copies run at the same 56 GB/s into either.  X265HIP_PINNED=hip: hipHostMalloc.
static std::mutex g_pinLock;
static std::map<void*, std::pair<void*, size_t>> g_pinned;            // user pointer -> (mapping, bytes of the mapping); absent: a hipHostMalloc pointer
hipError_t pinned_alloc(void** out, size_t bytes)
{
    static const bool useHip = getenv("X265HIP_PINNED") && !strcmp(getenv("X265HIP_PINNED"), "hip");
    const size_t huge = (size_t)2 << 20;
    if (!useHip && bytes >= huge)
    {
        const size_t len = ((bytes + huge - 1) & ~(huge - 1)) + huge;
        void* m = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m != MAP_FAILED)
        {
            char* a = (char*)(((uintptr_t)m + huge - 1) & ~(uintptr_t)(huge - 1));
            (void)madvise(a, len - huge, MADV_HUGEPAGE);
            if (hipHostRegister(a, bytes, hipHostRegisterDefault) == hipSuccess)
            {
                std::lock_guard<std::mutex> g(g_pinLock);
                g_pinned[a] = { m, len };
                *out = a;
                return hipSuccess;
            }
            (void)hipGetLastError();
            munmap(m, len);
        }
    }
    return hipHostMalloc(out, bytes, hipHostMallocDefault);
}
hipError_t pinned_free(void* p)
{
    if (!p) return hipSuccess;
    std::pair<void*, size_t> m{ nullptr, 0 };
    {
        std::lock_guard<std::mutex> g(g_pinLock);
        auto it = g_pinned.find(p);
        if (it != g_pinned.end()) { m = it->second; g_pinned.erase(it); }
    }
    if (!m.first) return hipHostFree(p);
    const hipError_t e = hipHostUnregister(p);
    munmap(m.first, m.second);
    return e;
}

void clock_add(int clk, uint64_t spans, uint64_t ns, uint64_t bytes)
{
    if (clk < 0 || clk >= X265HIP_CLK_COUNT) return;
    g_clkSpans[clk] += spans; g_clkNs[clk] += ns; g_clkBytes[clk] += bytes;
}

DevSpan::DevSpan(int clock, hipStream_t stream) : clk(clock), st(stream)
{
    static thread_local hipEvent_t ev[X265HIP_CLK_COUNT][2];
    static thread_local int evDevice[X265HIP_CLK_COUNT];
    static thread_local bool have[X265HIP_CLK_COUNT];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (have[clk] && evDevice[clk] != dev)
    {
        (void)hipEventDestroy(ev[clk][0]); (void)hipEventDestroy(ev[clk][1]);
        have[clk] = false;
    }
    if (!have[clk])
    {
        if (hipEventCreate(&ev[clk][0]) != hipSuccess || hipEventCreate(&ev[clk][1]) != hipSuccess) { (void)hipGetLastError(); return; }
        have[clk] = true; evDevice[clk] = dev;
    }
    e0 = ev[clk][0]; e1 = ev[clk][1];
    if (hipEventRecord(e0, st) != hipSuccess) { (void)hipGetLastError(); e0 = e1 = nullptr; }
}

void DevSpan::end()
{
    if (e1 && hipEventRecord(e1, st) != hipSuccess) { (void)hipGetLastError(); e0 = e1 = nullptr; }
}

void DevSpan::commit()
{
    float ms = 0;
    if (e0 && e1 && hipEventElapsedTime(&ms, e0, e1) == hipSuccess)
    {
        g_clkNs[clk] += (uint64_t)((double)ms * 1e6);
        g_clkSpans[clk]++;
        g_clkBytes[clk] += bytes;
    }
    else
        (void)hipGetLastError();
    e0 = e1 = nullptr;
}

bool valid_depth(int depth) { return depth == 8 || depth == 10 || depth == 12; }
bool valid_block(int w, int h) { return w >= 2 && h >= 2 && w <= 64 && h <= 64 && !(w & 1) && !(h & 1); }

} // namespace xh

using namespace xh;

extern "C" {

int x265hip_init(int device) { return init_device(device); }

int x265hip_device_count(void)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess)
    {
        (void)hipGetLastError();
        return 0;
    }
    return count;
}

const char* x265hip_last_error(void) { return t_err; }

int x265hip_places(int n, const int* devices)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1)
    {
        (void)hipGetLastError();
        return set_error(X265HIP_ENODEV, "x265hip_places: no HIP device available; there is no CPU fallback");
    }
    if (n < 1 || n > 64 || !devices) return set_error(X265HIP_EINVAL, "x265hip_places: %d places", n);
    for (int i = 0; i < n; i++)
        if (devices[i] < 0 || devices[i] >= count)
            return set_error(X265HIP_EINVAL, "x265hip_places: place %d on device %d (have %d)", i, devices[i], count);
    if (g_nPlaces.load() > n) return set_error(X265HIP_EINVAL, "x265hip_places: places can be added, not removed");
    for (int i = 0; i < n; i++)
    {
        if (i < g_nPlaces.load() && g_places[i] != devices[i]) return set_error(X265HIP_EINVAL, "x265hip_places: place %d is on device %d already", i, g_places[i]);
        g_places[i] = devices[i];
    }
    g_nPlaces = n;
    return X265HIP_OK;
}

int x265hip_device_time(int clock, uint64_t* spans, uint64_t* nanoseconds, uint64_t* algorithmicBytes)
{
    if (clock >= 0 && clock < X265HIP_CLK_COUNT && algorithmicBytes) *algorithmicBytes = g_clkBytes[clock].load();
    if (clock < 0 || clock >= X265HIP_CLK_COUNT) return set_error(X265HIP_EINVAL, "x265hip_device_time: clock %d", clock);
    if (spans) *spans = g_clkSpans[clock].load();
    if (nanoseconds) *nanoseconds = g_clkNs[clock].load();
    return X265HIP_OK;
}
const char* x265hip_version(void) { return "x265hip 0.1 (gfx950)"; }

int x265hip_malloc(void** dptr, size_t bytes)
{
    XH_CHECK_DEV();
    if (!dptr) return set_error(X265HIP_EINVAL, "x265hip_malloc: null out pointer");
    return check_hip(hipMalloc(dptr, bytes ? bytes : 1), "hipMalloc");
}
int x265hip_free(void* dptr)
{
    XH_CHECK_DEV();
    return check_hip(device_free(dptr), "hipFree");
}
int x265hip_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream)
{
    XH_CHECK_DEV();
    if (!bytes) return X265HIP_OK;
    return check_hip(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, as_stream(stream)), "hipMemcpyAsync(h2d)");
}
int x265hip_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream)
{
    XH_CHECK_DEV();
    if (!bytes) return X265HIP_OK;
    int e = check_hip(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)), "hipMemcpyAsync(d2h)");
    if (e) return e;
    return check_hip(hipStreamSynchronize(as_stream(stream)), "hipStreamSynchronize");
}
int x265hip_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream)
{
    XH_CHECK_DEV();
    if (!bytes) return X265HIP_OK;
    return check_hip(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)), "hipMemcpyAsync(d2d)");
}
int x265hip_memset(void* dst, int value, size_t bytes, void* stream)
{
    XH_CHECK_DEV();
    if (!bytes) return X265HIP_OK;
    return check_hip(hipMemsetAsync(dst, value, bytes, as_stream(stream)), "hipMemsetAsync");
}
int x265hip_stream_create(void** stream)
{
    XH_CHECK_DEV();
    hipStream_t s;
    int e = check_hip(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate");
    if (!e) *stream = s;
    return e;
}
int x265hip_stream_destroy(void* stream)
{
    XH_CHECK_DEV();
    return check_hip(hipStreamDestroy(as_stream(stream)), "hipStreamDestroy");
}
int x265hip_stream_sync(void* stream)
{
    XH_CHECK_DEV();
    return check_hip(hipStreamSynchronize(as_stream(stream)), "hipStreamSynchronize");
}
int x265hip_event_create(void** ev)
{
    XH_CHECK_DEV();
    hipEvent_t e;
    int r = check_hip(hipEventCreate(&e), "hipEventCreate");
    if (!r) *ev = e;
    return r;
}
int x265hip_event_destroy(void* ev)
{
    XH_CHECK_DEV();
    return check_hip(hipEventDestroy(reinterpret_cast<hipEvent_t>(ev)), "hipEventDestroy");
}
int x265hip_event_record(void* ev, void* stream)
{
    XH_CHECK_DEV();
    return check_hip(hipEventRecord(reinterpret_cast<hipEvent_t>(ev), as_stream(stream)), "hipEventRecord");
}
int x265hip_event_elapsed_ms(void* start, void* stop, float* ms)
{
    XH_CHECK_DEV();
    int r = check_hip(hipEventSynchronize(reinterpret_cast<hipEvent_t>(stop)), "hipEventSynchronize");
    if (r) return r;
    return check_hip(hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop)),
                     "hipEventElapsedTime");
}

} // extern "C"
