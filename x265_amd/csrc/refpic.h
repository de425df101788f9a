// refpic.h — the reference-picture mirror's state and its worker thread, shared by refpic.hip (sub-pel planes) and sadsurf.hip (SAD surfaces
// that follow a mirror's progress).  Not part of the C ABI.
#pragma once
#include "common.h"
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

struct x265hip_sadsurf;

namespace xh {
// A copy of a mirrored picture on another place (another GPU of the encoder): rows are pushed device to device (hipMemcpyPeerAsync, xGMI between
// the GPUs of a node) as the owner uploads them; the SAD surfaces of source pictures that live on that place are built from it, on its device.
// Round 5: a replica has sub-pel planes of its own, COMPUTED from the rows it has received (15 planes are not worth the fabric when the filters that make
// them run at HBM speed next to the copy): the sub-pel SATD tables of surfaces built from a replica come from them, so a frame whose source picture lives
// at another place than its reference is served like any other (round 4: integer-pel only).
struct Replica { int place = 0, device = 0; char* dPic = nullptr; char* dPlanes = nullptr; hipStream_t st = nullptr; int copied = 0, phaseDone = 4; };
int build_subpel_rows(int depth, const void* refOrigin, int64_t stride, int x0, int x1, int y0, int y1, void* planesOrigin, int64_t planeElems, hipStream_t st);   // refpic.hip
int place_device(int place);                 // runtime.hip: the HIP device of a place (x265hip_places), -1 if there is no such place
int place_of_device(int device);             // the anonymous place of objects created without one: -(device + 1)
}

struct x265hip_refpic
{
    int depth = 8, B = 1, picW = 0, picH = 0, marginX = 0, marginY = 0, bufRows = 0, device = 0;
    int place = -1;                           // x265hip_refpic_create_at; -(device + 1) for x265hip_refpic_create
    std::vector<xh::Replica*> replicas;       // worker only
    int64_t stride = 0, planeElems = 0;
    const char* hostBase = nullptr;          // the encoder's buffer (PicYuv::m_picBuf[0])
    char* hStage = nullptr;                   // page-locked staging copy of the rows on their way up
    char* dPic = nullptr;                     // device copy of the padded picture
    char* dPlanes = nullptr;                  // 16 planes (plane 0 unused), planeElems apart
    char* hPlanes = nullptr;                  // page-locked host planes 1..15 at (p - 1) * planeElems
    hipStream_t st = nullptr;
    // progress (buffer rows: 0 = first row of the top margin)
    int uploaded = 0;                         // rows [0, uploaded) of the buffer are on the device   (worker only)
    int ssDeferred = 0;                       // bands whose SAD-surface rows are waiting for company (sadsurf_rows_arrived; worker only)
    int phaseDone = 4;                        // phase rows [4, phaseDone) are in hPlanes             (worker only)
    std::atomic<int> rowsReady{ -(1 << 30) }; // published: phase rows of PICTURE rows [-(marginY - 4), rowsReady) are valid
    std::atomic<uint32_t> epoch{ 0 };         // bumped by reset(): queued work of an older picture is dropped
    std::atomic<int> pending{ 0 };            // queued + running jobs
    std::atomic<int> failed{ 0 };
    std::vector<x265hip_sadsurf*> surfaces;   // SAD surfaces attached to this picture (worker only; sadsurf.hip)
};

namespace xh {

// kind 0: rows [0, rowsFinal) of rp's picture are final.  kind 1 / 2: attach / release the SAD surface `ss` (sadsurf.hip); rp may be null for a
// release whose picture has gone
struct RefJob { x265hip_refpic* rp; int rowsFinal; uint32_t epoch; int kind = 0; x265hip_sadsurf* ss = nullptr; };

struct RefWorker
{
    std::mutex m;
    std::condition_variable cv, idle;
    std::deque<RefJob> q;
    std::thread th;
    bool started = false, stop = false;

    void start()
    {
        std::lock_guard<std::mutex> g(m);
        if (started) return;
        started = true;
        th = std::thread([this] { run(); });
        static std::once_flag once;
        std::call_once(once, [] { atexit([] { for (int p = 0; p < kWorkers; p++) if (workers()[p]) workers()[p]->shutdown(); }); });
    }
    void shutdown()
    {
        {
            std::lock_guard<std::mutex> g(m);
            if (!started || stop) return;
            stop = true;
        }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
    void push(const RefJob& j)
    {
        {
            std::lock_guard<std::mutex> g(m);
            if (stop)
                return;                 // the process is exiting (atexit joined the worker): nobody would run the job, and wait() must not block on it
            if (j.rp) j.rp->pending.fetch_add(1);
            q.push_back(j);
        }
        cv.notify_one();
    }
    // one worker per place (a GPU of the encoder): the bands, surfaces and replicas of the pictures that live there are driven by their own thread, so
    // the devices of one encoder do not queue behind one another; objects without a place share worker 0.  Leaked on purpose:
    // the workers live as long as the process.
    static constexpr int kWorkers = 16;
    static RefWorker** workers() { static RefWorker* w[kWorkers] = {}; return w; }
    static RefWorker& worker(int place)
    {
        static std::mutex lock;
        const int p = place >= 0 ? place % kWorkers : 0;
        std::lock_guard<std::mutex> g(lock);
        if (!workers()[p]) workers()[p] = new RefWorker;
        return *workers()[p];
    }
    void run();
};


// sadsurf.hip, called on the worker thread
void sadsurf_rows_arrived(x265hip_refpic* rp);            // after a band of rp has been uploaded (and its planes published)
void sadsurf_job(const RefJob& j);                        // kinds 1 and 2
void sadsurf_attach_batch(const std::vector<RefJob>& jobs);   // kind 1 jobs that were queued together
void sadsurf_detach_all(x265hip_refpic* rp);              // rp is being reset or destroyed (worker idle for rp)

} // namespace xh
