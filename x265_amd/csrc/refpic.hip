// refpic.hip — reference-picture mirrors behind x265hip_refpic_* (include/x265hip.h): the "lookup" face of the boundary.
//
// x265's sub-pel work on a reference picture — every luma_hpp / luma_vpp / luma_hvpp call of MotionEstimate::subpelCompare
// (motion.cpp:1571-1600) and of Predict::predInterLumaPixel (predict.cpp:245-266) — is a per-pixel function of the reconstructed
// picture alone: value = f(picture, x, y, xFrac, yFrac).  A reference picture therefore needs it computed ONCE, not once per candidate
// of every PU of every frame that references it.  A refpic mirrors one reconstructed luma picture of the DPB on the device; as the
// encoder publishes finished CTU rows (FrameFilter::processPostRow, framefilter.cpp:654-664) a worker thread
//     uploads the new rows (copied into a page-locked staging plane first: the encoder's buffer is ordinary memory the encoder may free),
//     filters every phase row whose 8-tap support is final — all 15 fractional phases in one launch of the sub-pel plane kernel
//       (interp.hip; plane[yFrac * 4 + xFrac](x, y) == what the reference's filter returns for that pixel),
//     brings the 15 band slices back into page-locked host planes with ONE 2-D copy, and publishes `rowsReady`.
// The table slots then serve a filter call on a mirrored picture as a W x H block copy out of the right plane
// (x265_amd/host/x265_hip_refplanes.cpp); rows not yet published fall through to the C filter — same values either way, so the
// bitstream is identical whatever the timing.  The encoder never waits for this module.
#include "common.h"
#include "internal.h"
#include "refpic.h"
#include <chrono>
#include <cstring>
#include <map>
#include <utility>

namespace xh {

int build_subpel_rows(int depth, const void* refOrigin, int64_t stride, int x0, int x1, int y0, int y1, void* planesOrigin, int64_t planeElems, hipStream_t st);

static void process(const RefJob& j)
{
    if (j.kind)
    {
        sadsurf_job(j);
        return;
    }
    x265hip_refpic* rp = j.rp;
    if (j.epoch != rp->epoch.load() || rp->failed.load())
        return;
    if (hipSetDevice(rp->device) != hipSuccess) { rp->failed = 1; return; }
    const int B = rp->B;
    // buffer rows that are final: top margin + picture rows [0, rowsFinal); the whole buffer once the picture is complete
    const bool complete = j.rowsFinal >= rp->picH;
    const int finalRows = complete ? rp->marginY + rp->picH + rp->marginY : rp->marginY + j.rowsFinal;
    if (finalRows <= rp->uploaded)
        return;
    const size_t off = (size_t)rp->uploaded * rp->stride * B, bytes = (size_t)(finalRows - rp->uploaded) * rp->stride * B;
    memcpy(rp->hStage + off, rp->hostBase + off, bytes);                  // final rows: nobody writes them any more
    if (hipMemcpyAsync(rp->dPic + off, rp->hStage + off, bytes, hipMemcpyHostToDevice, rp->st) != hipSuccess) { rp->failed = 1; return; }
    rp->uploaded = finalRows;
    // phase rows whose support (rows y - 3 .. y + 4) is final; the outermost 4 rows of the buffer are never computed
    const int phaseEnd = finalRows - 4;
    if (phaseEnd > rp->phaseDone)
    {
        const int y0 = rp->phaseDone - rp->marginY, y1 = phaseEnd - rp->marginY;       // picture coordinates
        const char* org = rp->dPic + ((size_t)rp->marginY * rp->stride + rp->marginX) * B;
        char* porg = rp->dPlanes + ((size_t)rp->marginY * rp->stride + rp->marginX) * B;
        DevSpan span(X265HIP_CLK_PLANES, rp->st);
        if (build_subpel_rows(rp->depth, org, rp->stride, -rp->marginX + 4, rp->picW + rp->marginX - 4, y0, y1, porg, rp->planeElems, rp->st))
        { rp->failed = 1; return; }
        span.end();
        span.bytes = (uint64_t)(y1 - y0) * (uint64_t)(rp->picW + 2 * rp->marginX - 8) * 16 * B;
        // planes 1..15, rows [phaseDone, phaseEnd): one 2-D copy — "row" = the band of one plane, pitch = one plane
        const size_t bandOff = (size_t)rp->phaseDone * rp->stride * B, bandBytes = (size_t)(phaseEnd - rp->phaseDone) * rp->stride * B;
        const size_t pitch = (size_t)rp->planeElems * B;
        if (hipMemcpy2DAsync(rp->hPlanes + bandOff, pitch, rp->dPlanes + pitch + bandOff, pitch, bandBytes, 15, hipMemcpyDeviceToHost, rp->st) != hipSuccess)
        { rp->failed = 1; return; }
        if (hipStreamSynchronize(rp->st) != hipSuccess) { rp->failed = 1; return; }
        span.commit();
        rp->phaseDone = phaseEnd;
        if (j.epoch == rp->epoch.load())
            rp->rowsReady.store(phaseEnd - rp->marginY, std::memory_order_release);
    }
    else if (hipStreamSynchronize(rp->st) != hipSuccess)
        rp->failed = 1;
    // the SAD surfaces that follow this picture: after the planes, which the encoder needs first
    if (!rp->failed.load() && j.epoch == rp->epoch.load())
        sadsurf_rows_arrived(rp);
}

void RefWorker::run()
{
    for (;;)
    {
        RefJob j;
        {
            std::unique_lock<std::mutex> g(m);
            cv.wait(g, [this] { return stop || !q.empty(); });
            if (q.empty())
                return;
            j = q.front();
            q.pop_front();
        }
        if (j.kind == 1)
        {
            // a surface being attached: the first search of a frame against each of its reference pictures attaches one, microseconds apart — give the
            // siblings X265HIP_SADSURF_GATHER_US (default 100) to arrive and build them all in one launch (sadsurf.hip progress_multi); the attach jobs at the front of the queue only,
            // so that nothing overtakes a band
            static const int gatherUs = getenv("X265HIP_SADSURF_GATHER_US") ? atoi(getenv("X265HIP_SADSURF_GATHER_US")) : 100;
            std::vector<RefJob> batch(1, j);
            const auto t0 = std::chrono::steady_clock::now();
            for (;;)
            {
                {
                    std::lock_guard<std::mutex> g(m);
                    while (!q.empty() && q.front().kind == 1) { batch.push_back(q.front()); q.pop_front(); }
                    if (!q.empty() || stop) break;
                }
                if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(gatherUs)) break;
                __builtin_ia32_pause();
            }
            sadsurf_attach_batch(batch);
            bool idleNow = false;
            for (const RefJob& b : batch)
                if (b.rp && b.rp->pending.fetch_sub(1) == 1) idleNow = true;
            if (idleNow)
            {
                std::lock_guard<std::mutex> g(m);
                idle.notify_all();
            }
            continue;
        }
        process(j);
        if (j.rp && j.rp->pending.fetch_sub(1) == 1)
        {
            std::lock_guard<std::mutex> g(m);
            idle.notify_all();
        }
    }
}

} // namespace xh

using namespace xh;

extern "C" {

static x265hip_refpic* refpic_create(int place, int depth, int picW, int picH, int64_t stride, int marginX, int marginY, int bufRows, const void* hostBase);

// The buffers of destroyed mirrors, kept for the next mirror of the same size on the same device: page-locking 42 MB per mirror is the slow part of
// creating one (10-15 ms) and unlocking it the slow part of destroying one — an encoder that closes hands its mirrors to the next encoder of the
// process, and a process that ends leaves them to the runtime's own teardown.
namespace {
struct MirrorBufs { char* dPic; char* dPlanes; char* hPlanes; char* hStage; hipStream_t st; };
std::mutex g_bufLock;
std::multimap<std::pair<size_t, int>, MirrorBufs> g_bufPool;        // key: (bytes of one padded plane, device)
size_t g_bufPoolBytes = 0;                                           // what the pooled sets hold, device + page-locked
}

x265hip_refpic* x265hip_refpic_create(int depth, int picW, int picH, int64_t stride, int marginX, int marginY, int bufRows, const void* hostBase)
{
    if (ensure_device()) return nullptr;
    int dev = 0;
    (void)hipGetDevice(&dev);
    return refpic_create(place_of_device(dev), depth, picW, picH, stride, marginX, marginY, bufRows, hostBase);
}

x265hip_refpic* x265hip_refpic_create_at(int place, int depth, int picW, int picH, int64_t stride, int marginX, int marginY, int bufRows, const void* hostBase)
{
    const int dev = place >= 0 ? place_device(place) : -1;
    if (dev < 0)
    {
        set_error(X265HIP_EINVAL, "x265hip_refpic_create_at: no place %d (x265hip_places)", place);
        return nullptr;
    }
    // the mirror's memory and stream belong to the place's device; the calling thread's own device is restored
    int cur = 0;
    const bool had = hipGetDevice(&cur) == hipSuccess;
    if (hipSetDevice(dev) != hipSuccess) { set_error(X265HIP_EHIP, "x265hip_refpic_create_at: hipSetDevice(%d)", dev); return nullptr; }
    x265hip_refpic* rp = refpic_create(place, depth, picW, picH, stride, marginX, marginY, bufRows, hostBase);
    if (had) (void)hipSetDevice(cur);
    return rp;
}

static x265hip_refpic* refpic_create(int place, int depth, int picW, int picH, int64_t stride, int marginX, int marginY, int bufRows, const void* hostBase)
{
    if (!valid_depth(depth) || picW < 8 || picH < 8 || (picW & 3) || (marginX & 3) || marginX < 8 || marginY < 8 || stride < picW + 2 * marginX ||
        bufRows < picH + 2 * marginY || !hostBase)
    {
        set_error(X265HIP_EINVAL, "x265hip_refpic_create: depth %d pic %dx%d stride %lld margins %d,%d rows %d", depth, picW, picH, (long long)stride, marginX, marginY, bufRows);
        return nullptr;
    }
    x265hip_refpic* rp = new x265hip_refpic;
    rp->depth = depth; rp->B = depth == 8 ? 1 : 2;
    rp->picW = picW; rp->picH = picH; rp->marginX = marginX; rp->marginY = marginY; rp->bufRows = bufRows;
    rp->stride = stride; rp->planeElems = stride * (int64_t)bufRows;
    rp->hostBase = (const char*)hostBase;
    (void)hipGetDevice(&rp->device);
    rp->place = place;
    const size_t planeBytes = (size_t)rp->planeElems * rp->B;
    bool ok = false;
    {
        std::lock_guard<std::mutex> g(g_bufLock);
        auto it = g_bufPool.find({ planeBytes, rp->device });
        if (it != g_bufPool.end())
        {
            g_bufPoolBytes -= 33 * planeBytes;
            rp->dPic = it->second.dPic; rp->dPlanes = it->second.dPlanes; rp->hPlanes = it->second.hPlanes; rp->hStage = it->second.hStage; rp->st = it->second.st;
            g_bufPool.erase(it);
            ok = true;
        }
    }
    if (!ok)
        ok = hipStreamCreateWithFlags(&rp->st, hipStreamNonBlocking) == hipSuccess &&
             hipMalloc((void**)&rp->dPic, planeBytes) == hipSuccess && hipMalloc((void**)&rp->dPlanes, planeBytes * 16) == hipSuccess &&
             pinned_alloc((void**)&rp->hPlanes, planeBytes * 15) == hipSuccess &&
             pinned_alloc((void**)&rp->hStage, planeBytes) == hipSuccess;
    if (!ok)
    {
        set_error(X265HIP_ENOMEM, "x265hip_refpic_create: %zu bytes per plane", planeBytes);
        x265hip_refpic_destroy(rp);
        return nullptr;
    }
    RefWorker::worker(rp->place).start();
    return rp;
}

int x265hip_refpic_wait(x265hip_refpic* rp)
{
    if (!rp) return set_error(X265HIP_EINVAL, "x265hip_refpic_wait: null");
    RefWorker& w = RefWorker::worker(rp->place);
    std::unique_lock<std::mutex> g(w.m);
    w.idle.wait(g, [rp] { return rp->pending.load() == 0; });
    return rp->failed.load() ? set_error(X265HIP_EHIP, "x265hip_refpic: a device operation of the worker failed") : X265HIP_OK;
}

void x265hip_refpic_destroy(x265hip_refpic* rp)
{
    if (!rp) return;
    rp->epoch.fetch_add(1);
    if (rp->st) (void)x265hip_refpic_wait(rp);
    sadsurf_detach_all(rp);                                           // the worker is idle for rp: nobody else touches the list
    int cur = 0;
    const bool had = hipGetDevice(&cur) == hipSuccess;
    for (Replica* r : rp->replicas)
    {
        (void)hipSetDevice(r->device);
        if (r->st) { (void)hipStreamSynchronize(r->st); (void)hipStreamDestroy(r->st); }
        if (r->dPic) (void)device_free(r->dPic);
        if (r->dPlanes) (void)device_free(r->dPlanes);
        delete r;
    }
    rp->replicas.clear();
    (void)hipSetDevice(rp->device);
    if (rp->st && rp->dPic && rp->dPlanes && rp->hPlanes && rp->hStage)
    {
        // complete set: to the pool (the worker is idle for rp and its stream has been synchronised by the wait above)
        (void)hipStreamSynchronize(rp->st);
        const size_t planeBytes = (size_t)rp->planeElems * rp->B;
        std::vector<MirrorBufs> evicted;
        {
            std::lock_guard<std::mutex> g(g_bufLock);
            g_bufPool.insert({ { planeBytes, rp->device }, MirrorBufs{ rp->dPic, rp->dPlanes, rp->hPlanes, rp->hStage, rp->st } });
            g_bufPoolBytes += 33 * planeBytes;             // picture + 16 planes on the device, 15 planes + staging page-locked
            // the pool is for the NEXT mirror of the same size (it was never trimmed): beyond X265HIP_POOL_MB (default 4096) sets of other
            // sizes go first, then the oldest of this size
            static const size_t cap = (size_t)(getenv("X265HIP_POOL_MB") ? atoll(getenv("X265HIP_POOL_MB")) : 4096) << 20;
            while (g_bufPoolBytes > cap && g_bufPool.size() > 1)
            {
                auto victim = g_bufPool.begin();
                for (auto it = g_bufPool.begin(); it != g_bufPool.end(); ++it)
                    if (it->first.first != planeBytes) { victim = it; break; }
                evicted.push_back(victim->second);
                g_bufPoolBytes -= 33 * victim->first.first;
                g_bufPool.erase(victim);
            }
        }
        for (const MirrorBufs& b : evicted)                   // (allocated on the device of their key; hipFree and friends take any current device)
        {
            (void)pinned_free(b.hStage); (void)device_free(b.dPic); (void)device_free(b.dPlanes); (void)pinned_free(b.hPlanes);
            if (b.st) (void)hipStreamDestroy(b.st);
        }
    }
    else
    {
        if (rp->hStage) (void)pinned_free(rp->hStage);
        if (rp->dPic) (void)device_free(rp->dPic);
        if (rp->dPlanes) (void)device_free(rp->dPlanes);
        if (rp->hPlanes) (void)pinned_free(rp->hPlanes);
        if (rp->st) (void)hipStreamDestroy(rp->st);
    }
    delete rp;
    if (had) (void)hipSetDevice(cur);
}

int x265hip_refpic_reset(x265hip_refpic* rp)
{
    if (!rp) return set_error(X265HIP_EINVAL, "x265hip_refpic_reset: null");
    rp->rowsReady.store(-(1 << 30), std::memory_order_release);     // nothing is valid: readers fall through to the C filter
    rp->epoch.fetch_add(1);                                           // whatever is queued for the old picture is dropped
    int e = x265hip_refpic_wait(rp);                                  // the worker owns uploaded / phaseDone: take them over only when it is idle
    // a band the worker was finishing when the epoch moved may have published its rows after the store above (it checks the epoch before
    // it publishes, not atomically with it): nothing of the old picture may stay visible
    rp->rowsReady.store(-(1 << 30), std::memory_order_release);
    sadsurf_detach_all(rp);
    for (Replica* r : rp->replicas)
    {
        r->copied = 0;                                                // the replicas stay (same picture size), their rows are the old picture's
        r->phaseDone = 4;                                             // ... and so are their sub-pel planes
    }
    rp->uploaded = 0;
    rp->phaseDone = 4;
    return e;
}

int x265hip_refpic_rows_final(x265hip_refpic* rp, int rowsFinal)
{
    if (!rp || rowsFinal < 0) return set_error(X265HIP_EINVAL, "x265hip_refpic_rows_final: rows %d", rowsFinal);
    if (rp->failed.load()) return set_error(X265HIP_EHIP, "x265hip_refpic: a device operation of the worker failed");
    RefWorker::worker(rp->place).push(RefJob{ rp, rowsFinal > rp->picH ? rp->picH : rowsFinal, rp->epoch.load() });
    return X265HIP_OK;
}

const void* x265hip_refpic_plane(x265hip_refpic* rp, int phase)
{
    if (!rp || phase < 1 || phase > 15) return nullptr;
    return rp->hPlanes + (size_t)(phase - 1) * rp->planeElems * rp->B;
}

int x265hip_refpic_rows_ready(x265hip_refpic* rp) { return rp ? rp->rowsReady.load(std::memory_order_acquire) : -(1 << 30); }

const int* x265hip_refpic_rows_ready_ptr(x265hip_refpic* rp) { return rp ? reinterpret_cast<const int*>(&rp->rowsReady) : nullptr; }

} // extern "C"
