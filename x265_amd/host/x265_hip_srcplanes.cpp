// x265_hip_srcplanes.cpp — the fourth translation unit of the drop-in: source-picture energy planes and the psy_cost_pp slots they feed
// (include/x265hip.h, x265hip_source_energy).
//
// cu[].psy_cost_pp(source, sstride, recon, rstride) (reference source/common/pixel.cpp:726-757; RDCost::psyCost, encoder/rdcost.h:114) is the
// psycho-visual term of every RD cost at psy-rd > 0 (preset medium: 2.0).  Per 8x8 block it compares |energy(source) - energy(recon)| with
// energy(b) = sa8d_8x8(b, 0) - (sum(b) >> 2).  The source argument is always a block of the encoder's source-CU cache (Search / Analysis
// pass Mode::fencYuv) — a copy of the source picture — and the same block is asked about dozens of times while a CU's modes are compared.
// energy(source) is a function of the source picture and the position only, so the GPU computes it once per picture for every aligned 8x8
// (and 4x4) block of the three planes, and the slot computes only the reconstruction half.
//
// Knowing WHERE a source-cache block lies in the picture takes three pass-through seams (same link technique as the other seams):
//   PicYuv::copyFromPicture (common/picyuv.cpp:209)   a new picture enters a source buffer: its planes are stale from here on
//   Yuv::copyFromPicYuv     (common/yuv.cpp:100)      Analysis::compressCTU fills the CTU's source cache from the picture (analysis.cpp:154):
//                                                     cache buffer -> (picture, x, y); the picture's planes are built here if they are not yet
//   Yuv::copyPartToYuv      (common/yuv.cpp:155)      a sub-CU's cache is cut out of the CTU's (analysis.cpp:600, ...): child buffer -> position
// Exactness does not rest on that bookkeeping: before a plane value is used, the block handed to the slot is compared byte for byte with the
// picture at the claimed position; equal bytes have equal energy.  Anything that does not check out is computed by the C function.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

#define protected public
#define private public
#include "common.h"
#include "picyuv.h"
#include "primitives.h"
#include "yuv.h"
#undef protected
#undef private

#include "x265hip.h"
#include "x265_hip_debug.h"

namespace X265_NS {

const EncoderPrimitives& x265hip_c_table();          // x265_hip_primitives.cpp

extern void refCopyFromPicture(PicYuv* self, const x265_picture& pic, const x265_param& param, int padx, int pady)
    asm("_ZN4x2659PicYuvRef15copyFromPictureERK12x265_pictureRK10x265_paramii");
extern void refDestroy(PicYuv* self) asm("_ZN4x2659PicYuvRef7destroyEv");
void x265hip_refplanes_retire(const pixel* lo);     // x265_hip_refplanes.cpp
bool x265hip_sadplanes_wanted();                    // x265_hip_sadplanes.cpp: source pictures are kept on the device for the SAD surfaces
extern void refCopyFromPicYuv(Yuv* self, const PicYuv& srcPic, uint32_t cuAddr, uint32_t absPartIdx) asm("_ZN4x2656YuvRef14copyFromPicYuvERKNS_6PicYuvEjj");
extern void refCopyPartToYuv(const Yuv* self, Yuv& dstYuv, uint32_t absPartIdx) asm("_ZNK4x2656YuvRef13copyPartToYuvERS0_j");

namespace {

struct SrcPic                        // one source picture buffer and its energy planes
{
    std::atomic<const PicYuv*> pic;
    std::atomic<uint32_t> version;   // bumped whenever a new picture is copied into the buffer
    std::atomic<uint32_t> built;     // version the planes below belong to (0 = none)
    std::mutex lock;
    int32_t* e8[3];
    int32_t* e4[3];
    int bw[3], bh[3];                // 8x8 blocks per row / column of each plane
    x265hip_srcpic* dev;             // the luma plane on the device (x265_hip_sadplanes.cpp attaches SAD surfaces to it); version `built`
    int devW, devH;
};

const int kMaxPics = 128;
SrcPic g_pics[kMaxPics];
std::atomic<int> g_npics(0);
std::mutex g_lock;
int g_state = 0;
std::atomic<bool> g_noDeviceCopies(false);   // a source picture could not be copied to the device: no further attempts
EncoderPrimitives g_c;
struct alignas(64) Counter { std::atomic<uint64_t> v; };            // one cache line each: the slots run on every pool worker at once
Counter g_hit[64], g_miss[64];
std::atomic<int> g_shardNext(0);
__attribute__((tls_model("initial-exec"))) thread_local int t_shard = -1;
inline int shard() { if (t_shard < 0) t_shard = g_shardNext.fetch_add(1) & 63; return t_shard; }

// per thread: which source-cache buffers hold which part of which picture
struct CacheMap { const pixel* buf[3]; uint32_t size, csize; SrcPic* sp; uint32_t version; int x, y; };
const int kMaps = 8;
__attribute__((tls_model("initial-exec"))) thread_local CacheMap t_map[kMaps];
__attribute__((tls_model("initial-exec"))) thread_local int t_mapNext = 0;

void report()
{
    uint64_t h = 0, m = 0;
    for (int i = 0; i < 64; i++) { h += g_hit[i].v; m += g_miss[i].v; }
    fprintf(stderr, "x265hip: srcplanes: source half of %llu psy-cost calls served from GPU-built energy planes of %d source buffers, %llu computed on the host\n",
            (unsigned long long)h, g_npics.load(), (unsigned long long)m);
}

bool switched_on()
{
    const char* env = getenv("X265HIP_SRCPLANES");
    const char* all = getenv("X265HIP");
    const char* table = getenv("X265HIP_TABLE");
    return !((env && !strcmp(env, "0")) || (all && !strcmp(all, "0")) || (table && !strcmp(table, "percall")));
}

bool enabled()
{
    if (!g_state)
    {
        std::lock_guard<std::mutex> g(g_lock);
        if (!g_state)
        {
            const char* env = getenv("X265HIP_SRCPLANES");
            const char* all = getenv("X265HIP");
            const char* table = getenv("X265HIP_TABLE");
            if ((env && !strcmp(env, "0")) || (all && !strcmp(all, "0")) || (table && !strcmp(table, "percall")) || x265hip_device_count() < 1)
                g_state = -1;
            else
            {
                g_state = 1;
                if (getenv("X265HIP_VERBOSE"))
                    atexit(report);
            }
        }
    }
    return g_state > 0;
}

SrcPic* find_pic(const PicYuv* pic, bool create)
{
    const int n = g_npics.load(std::memory_order_acquire);
    for (int i = 0; i < n; i++)
        if (g_pics[i].pic.load(std::memory_order_relaxed) == pic)
            return &g_pics[i];
    if (!create)
        return NULL;
    std::lock_guard<std::mutex> g(g_lock);
    const int n2 = g_npics.load();
    for (int i = n; i < n2; i++)
        if (g_pics[i].pic.load() == pic)
            return &g_pics[i];
    for (int i = 0; i < n2; i++)
        if (!g_pics[i].pic.load())                     // the entry of a destroyed buffer (retire() below): its planes are reused or resized by build()
        {
            g_pics[i].pic.store(pic);
            return &g_pics[i];
        }
    if (n2 == kMaxPics)
        return NULL;
    SrcPic& s = g_pics[n2];
    s.pic = pic;
    s.version = 1;
    s.built = 0;
    for (int k = 0; k < 3; k++) { s.e8[k] = s.e4[k] = NULL; s.bw[k] = s.bh[k] = 0; }
    s.dev = NULL; s.devW = s.devH = 0;
    g_npics.store(n2 + 1, std::memory_order_release);
    return &s;
}

// the planes of the picture now in the buffer (worker thread: a picture enters the encoder a whole lookahead before its first CTU is analysed)
void build(SrcPic* sp, uint32_t v)
{
    std::lock_guard<std::mutex> g(sp->lock);           // PicYuv::destroy waits here for a build in progress
    const PicYuv* picp = sp->pic.load();
    if (!picp || sp->version.load() != v)
        return;                                        // the buffer is gone, or a newer picture is already on its way into it
    const PicYuv& pic = *picp;
    const int planes = pic.m_picCsp == X265_CSP_I400 ? 1 : 3;
    for (int k = 0; k < planes; k++)
    {
        const int w = k ? (int)(pic.m_picWidth >> pic.m_hChromaShift) : (int)pic.m_picWidth, h = k ? (int)(pic.m_picHeight >> pic.m_vChromaShift) : (int)pic.m_picHeight;
        const int bw = w >> 3, bh = h >> 3;
        if (bw < 1 || bh < 1)
            return;
        if (sp->bw[k] != bw || sp->bh[k] != bh)
        {
            free(sp->e8[k]);
            free(sp->e4[k]);
            sp->e8[k] = (int32_t*)malloc((size_t)bw * bh * 4);
            sp->e4[k] = (int32_t*)malloc((size_t)bw * bh * 16);
            sp->bw[k] = bw; sp->bh[k] = bh;
        }
        if (x265hip_source_energy(X265_DEPTH, pic.m_picOrg[k], k ? pic.m_strideC : pic.m_stride, w, h, sp->e8[k], sp->e4[k]))
        {
            // these planes are never marked built: the psy slots compute the source half themselves; no further attempts
            g_state = -1;
            x265hip_device_failure("srcplanes", "x265hip_source_energy");
            return;
        }
    }
    if (x265hip_sadplanes_wanted() && !g_noDeviceCopies.load(std::memory_order_relaxed))
    {
        if (sp->dev && (sp->devW != (int)pic.m_picWidth || sp->devH != (int)pic.m_picHeight))
        {
            x265hip_srcpic_destroy(sp->dev);
            sp->dev = NULL;
        }
        if (!sp->dev)
        {
            x265hip_debug_mark("create: device copy of a source picture");
            static std::atomic<int> nextPlace(0);
            const int places = x265hip_places_configured();
            sp->dev = places ? x265hip_srcpic_create_at(nextPlace.fetch_add(1) % places, X265_DEPTH, pic.m_picWidth, pic.m_picHeight)
                             : x265hip_srcpic_create(X265_DEPTH, pic.m_picWidth, pic.m_picHeight);
            x265hip_debug_mark("created: device copy of a source picture");
            sp->devW = pic.m_picWidth; sp->devH = pic.m_picHeight;
        }
        if (!sp->dev || x265hip_srcpic_upload(sp->dev, pic.m_picOrg[0], pic.m_stride))
        {
            // no device copy of this source picture (nor of later ones): no SAD surfaces can be attached to it, the searches use the C functions
            if (sp->dev) { x265hip_srcpic_destroy(sp->dev); sp->dev = NULL; }
            g_noDeviceCopies.store(true);
            x265hip_device_failure("srcplanes", "device copy of a source picture");
        }
    }
    if (sp->version.load() == v)                        // otherwise these planes are simply never used
        sp->built.store(v, std::memory_order_release);
}

struct Job { SrcPic* sp; uint32_t version; };
// the queue lives on the heap and is never destroyed: the worker is detached and may be waiting on it while the process runs its static
// destructors at exit (destroying a condition variable somebody waits on hangs)
struct Queue { std::mutex lock; std::condition_variable cv; std::deque<Job> jobs; bool workerUp = false; };
Queue& queue() { static Queue* q = new Queue; return *q; }

void worker()
{
    Queue& q = queue();
    for (;;)
    {
        Job j;
        {
            std::unique_lock<std::mutex> g(q.lock);
            q.cv.wait(g, [&q] { return !q.jobs.empty(); });
            j = q.jobs.front();
            q.jobs.pop_front();
        }
        build(j.sp, j.version);
    }
}

void enqueue(SrcPic* sp, uint32_t version)
{
    Queue& q = queue();
    {
        std::lock_guard<std::mutex> g(q.lock);
        if (!q.workerUp)
        {
            q.workerUp = true;
            std::thread(worker).detach();              // lives as long as the process; idle when no pictures arrive
        }
        q.jobs.push_back(Job{ sp, version });
    }
    q.cv.notify_one();
}

inline void remember(const Yuv* y, SrcPic* sp, uint32_t version, int px, int py)
{
    int slot = -1;
    for (int i = 0; i < kMaps; i++)
        if (t_map[i].buf[0] == y->m_buf[0]) { slot = i; break; }
    if (slot < 0) { slot = t_mapNext; t_mapNext = (t_mapNext + 1) % kMaps; }
    CacheMap& m = t_map[slot];
    m.buf[0] = y->m_buf[0]; m.buf[1] = y->m_buf[1]; m.buf[2] = y->m_buf[2];
    m.size = y->m_size; m.csize = y->m_csize;
    m.sp = sp; m.version = version; m.x = px; m.y = py;
}

// energy of the source block (N x N, N >= 8: sum over its 8x8 blocks is NOT what psyCost needs — it needs each 8x8 energy — so this returns a
// pointer-free accessor): locate `source` in this thread's source caches, verify the bytes against the picture, return the plane and position
struct Located { const int32_t* e8; const int32_t* e4; int bw; int bx8, by8; };

template <int N>
inline bool locate(const pixel* source, intptr_t sstride, Located& out)
{
    for (int i = 0; i < kMaps; i++)
    {
        const CacheMap& m = t_map[i];
        if (!m.buf[0])
            continue;
        for (int k = 0; k < 3; k++)
        {
            const uint32_t sz = k ? m.csize : m.size;
            if (!m.buf[k] || source < m.buf[k] || source >= m.buf[k] + (size_t)sz * sz || (intptr_t)sz != sstride)
                continue;
            SrcPic* sp = m.sp;
            if (sp->built.load(std::memory_order_acquire) != m.version || sp->version.load(std::memory_order_relaxed) != m.version)
                return false;
            const PicYuv* pic = sp->pic.load(std::memory_order_relaxed);
            const ptrdiff_t off = source - m.buf[k];
            const int oy = (int)(off / sz), ox = (int)(off - (ptrdiff_t)oy * sz);
            const int shx = k ? pic->m_hChromaShift : 0, shy = k ? pic->m_vChromaShift : 0;
            const int x = (m.x >> shx) + ox, y = (m.y >> shy) + oy;
            const int G = N >= 8 ? 8 : 4;
            if ((x | y) & (G - 1))
                return false;
            if (x + N > sp->bw[k] * 8 || y + N > sp->bh[k] * 8)
                return false;
            // equal bytes have equal energy: this comparison, not the bookkeeping above, is what makes the lookup exact
            const intptr_t ps = k ? pic->m_strideC : pic->m_stride;
            const pixel* p = pic->m_picOrg[k] + (intptr_t)y * ps + x;
            for (int r = 0; r < N; r++)
                if (memcmp(source + r * sstride, p + r * ps, N * sizeof(pixel)))
                    return false;
            out.e8 = sp->e8[k]; out.e4 = sp->e4[k]; out.bw = sp->bw[k]; out.bx8 = x >> 3; out.by8 = y >> 3;
            if (N < 8) { out.bx8 = x >> 2; out.by8 = y >> 2; }
            return true;
        }
    }
    return false;
}

static pixel s_zero[8];

template <int N, int CU> int psy_lookup(const pixel* source, intptr_t sstride, const pixel* recon, intptr_t rstride)
{
    Located L;
    if (!locate<N>(source, sstride, L))
    {
        g_miss[shard()].v.fetch_add(1, std::memory_order_relaxed);
        return g_c.cu[CU].psy_cost_pp(source, sstride, recon, rstride);
    }
    g_hit[shard()].v.fetch_add(1, std::memory_order_relaxed);
    if (N == 4)
    {
        const int src = L.e4[(size_t)L.by8 * (L.bw * 2) + L.bx8];
        const int rec = g_c.pu[LUMA_4x4].satd(recon, rstride, s_zero, 0) - (g_c.pu[LUMA_4x4].sad(recon, rstride, s_zero, 0) >> 2);     // pixel.cpp:752-753
        return abs(src - rec);
    }
    uint32_t tot = 0;
    for (int i = 0; i < N; i += 8)
        for (int j = 0; j < N; j += 8)
        {
            const int src = L.e8[(size_t)(L.by8 + (i >> 3)) * L.bw + L.bx8 + (j >> 3)];
            const pixel* r = recon + i * rstride + j;
            const int rec = g_c.cu[BLOCK_8x8].sa8d(r, rstride, s_zero, 0) - (g_c.pu[LUMA_8x8].sad(r, rstride, s_zero, 0) >> 2);          // pixel.cpp:743-744
            tot += abs(src - rec);
        }
    return (int)tot;
}

} // namespace

// x265_hip_sadplanes.cpp (MotionEstimate::setSourcePU): which source picture buffer, and where in it, does this thread's source cache `y` hold?
// *pic / *version name the picture (a buffer is reused for later pictures: the version tells them apart), (*x, *y) the luma position of the
// cache's first sample.  false: not a source cache this thread filled through the seams below.
bool x265hip_srcplanes_where(const Yuv& y, const PicYuv** pic, uint32_t* version, int* px, int* py)
{
    if (g_state <= 0)
        return false;
    for (int i = 0; i < kMaps; i++)
    {
        const CacheMap& m = t_map[i];
        if (m.buf[0] && m.buf[0] == y.m_buf[0] && m.size == y.m_size)
        {
            if (m.sp->version.load(std::memory_order_relaxed) != m.version)
                return false;
            *pic = m.sp->pic.load(std::memory_order_relaxed);
            *version = m.version;
            *px = m.x; *py = m.y;
            return *pic != NULL;
        }
    }
    return false;
}

// x265_hip_sadplanes.cpp: the device copy of source picture `pic` in its version `version` (NULL: not built, or the buffer has moved on)
x265hip_srcpic* x265hip_srcplanes_device(const PicYuv* pic, uint32_t version)
{
    SrcPic* sp = find_pic(pic, false);
    if (!sp || sp->built.load(std::memory_order_acquire) != version || sp->version.load(std::memory_order_relaxed) != version)
        return NULL;
    return sp->dev;
}

// is `version` still the picture in `pic`'s buffer?  (x265_hip_sadplanes.cpp retires the surfaces of pictures that have left)
bool x265hip_srcplanes_current(const PicYuv* pic, uint32_t version)
{
    SrcPic* sp = find_pic(pic, false);
    return sp && sp->version.load(std::memory_order_relaxed) == version;
}

// called by setupAssemblyPrimitives in the default table mode, after the C table is complete
void x265hip_install_psy_slots(EncoderPrimitives& p)
{
    // decided by the switches alone: whether a device exists is known a moment later (setupAssemblyPrimitives probes beside the encoder's own set-up),
    // and without one the wrappers find nothing to serve and call the C functions
    if (!switched_on())
        return;
    if (getenv("X265HIP_SRCPLANES_PSY") && !strcmp(getenv("X265HIP_SRCPLANES_PSY"), "0"))
        return;                                        // the seams and the planes stay, the psy slots keep the C functions (a diagnostic)
    // What the slots did before: the reference's C functions, from a table built for the purpose (x265hip_c_table, x265_hip_primitives.cpp), once.
    // Not a copy of `p`: x265_setup_primitives is not serialised between encoders opened at the same time (primitives.cpp:
    // `if (!primitives.pu[0].sad)`), so `p` may be half filled by another thread or already hold these very wrappers — a wrapper that
    // captured itself would call itself for ever (tests/test_encoder_lifetime.py, concurrent sessions).
    {
        static std::mutex once;
        static bool have = false;
        std::lock_guard<std::mutex> g(once);
        if (!have)
        {
            g_c = x265hip_c_table();
            have = true;
        }
    }
    // cu[BLOCK_4x4].sa8d is aliased to satd_4x4 only later (setupAliasPrimitives, primitives.cpp:139); the lookups call the pu[] slots directly
    p.cu[BLOCK_4x4].psy_cost_pp = psy_lookup<4, BLOCK_4x4>;
    p.cu[BLOCK_8x8].psy_cost_pp = psy_lookup<8, BLOCK_8x8>;
    p.cu[BLOCK_16x16].psy_cost_pp = psy_lookup<16, BLOCK_16x16>;
    p.cu[BLOCK_32x32].psy_cost_pp = psy_lookup<32, BLOCK_32x32>;
    p.cu[BLOCK_64x64].psy_cost_pp = psy_lookup<64, BLOCK_64x64>;
}

void PicYuv::copyFromPicture(const x265_picture& pic, const x265_param& param, int padx, int pady)
{
    SrcPic* sp = enabled() ? find_pic(this, true) : NULL;
    uint32_t v = 0;
    if (sp)
        v = sp->version.fetch_add(1) + 1;              // before the pixels change: nobody may trust the old planes from here on
    refCopyFromPicture(this, pic, param, padx, pady);
    if (sp)
        enqueue(sp, v);                                 // the picture is complete (padding included): its planes are built in the background
}

// the buffer goes away (encoder close; a process may open another encoder afterwards, with other picture sizes, and malloc may hand the same
// addresses out again): nothing may keep pointing into it — neither the source-picture entry nor a reference-picture mirror
void PicYuv::destroy()
{
    if (g_state > 0)
        if (SrcPic* sp = find_pic(this, false))
        {
            std::lock_guard<std::mutex> g(sp->lock);
            sp->version.fetch_add(1);                   // queued builds and remembered cache positions of this buffer are void
            sp->built.store(0);
            sp->pic.store(nullptr);
        }
    if (m_picBuf[0])
    {
        x265hip_debug_mark("PicYuv::destroy: retire");
        x265hip_refplanes_retire(m_picBuf[0]);
        x265hip_debug_mark("PicYuv::destroy: retired");
    }
    refDestroy(this);
}

void Yuv::copyFromPicYuv(const PicYuv& srcPic, uint32_t cuAddr, uint32_t absPartIdx)
{
    refCopyFromPicYuv(this, srcPic, cuAddr, absPartIdx);
    if (!enabled())
        return;
    SrcPic* sp = find_pic(&srcPic, false);             // only buffers that went through copyFromPicture are source pictures
    if (!sp)
        return;
    const ptrdiff_t off = srcPic.getLumaAddr(cuAddr, absPartIdx) - srcPic.m_picOrg[0];
    remember(this, sp, sp->version.load(), (int)(off % srcPic.m_stride), (int)(off / srcPic.m_stride));
}

void Yuv::copyPartToYuv(Yuv& dstYuv, uint32_t absPartIdx) const
{
    refCopyPartToYuv(this, dstYuv, absPartIdx);
    if (g_state <= 0)
        return;
    for (int i = 0; i < kMaps; i++)
        if (t_map[i].buf[0] == m_buf[0] && m_buf[0])
        {
            const CacheMap m = t_map[i];
            remember(&dstYuv, m.sp, m.version, m.x + g_zscanToPelX[absPartIdx], m.y + g_zscanToPelY[absPartIdx]);
            return;
        }
    // not a mapped source cache: prediction / reconstruction buffers use this function too.  A stale entry for dstYuv does no harm — the slot
    // verifies the bytes before it uses a plane value
}

} // namespace X265_NS
