// x265_hip_refplanes.cpp — the third translation unit of the drop-in: reference-picture mirrors and the lookup slots they feed
// (include/x265hip.h, x265hip_refpic_*).
//
// Every luma_hpp / luma_vpp / luma_hvpp call the encoder makes on a reconstructed reference picture — the sub-pel candidates of
// MotionEstimate::subpelCompare (reference source/encoder/motion.cpp:1571-1600) and Predict::predInterLumaPixel (common/predict.cpp:245-266)
// — asks for values that depend on the picture and the position only.  The GPU computes them once per reference picture (all 15 fractional
// planes, as CTU rows are published), and the table slot becomes a block copy out of the right plane.  Two seams:
//   * FrameFilter::processPostRow (encoder/framefilter.cpp:654-664), where x265 itself tells other frame encoders that a CTU row of
//     reconstructed pixels is final: the definition below runs the reference's own body, then hands the finished rows to the picture's
//     mirror (x265hip_refpic_rows_final, asynchronous);
//   * the three table slots per PU size, installed by x265hip_install_lookup_slots() from setupAssemblyPrimitives: a call whose source
//     pointer lies inside a mirrored picture and whose rows the mirror has published is served by memcpy; anything else — rows still on
//     their way, weighted reference copies, lowres planes, the TestBench's own buffers — goes to the C function the slot held before.
// Same values either way (tests/test_framepass.py::test_subpel_planes_match_reference_filters pins plane == filter), so the bitstream
// does not depend on how far the mirror has got.  The encoder never waits here.
//
// Linked like the lookahead seam: the reference's processPostRow is weakened in framefilter.o, and its original body stays reachable as
// FrameFilterRef::processPostRow from a second compile of framefilter.cpp (oracle/Makefile).
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unistd.h>

#define protected public
#define private public
#include "common.h"
#include "frame.h"
#include "framedata.h"
#include "picyuv.h"
#include "primitives.h"
#include "slice.h"
#include "framefilter.h"
#include "reference.h"
#undef protected
#undef private

#include "x265hip.h"
#include "x265_hip_debug.h"

namespace X265_NS {

const EncoderPrimitives& x265hip_c_table();          // x265_hip_primitives.cpp

extern void refProcessPostRow(FrameFilter* self, int row) asm("_ZN4x26514FrameFilterRef14processPostRowEi");
// the reference's MotionReference (reference.cpp compiled a second time as MotionReferenceRef): the weighted copies of reference pictures
extern int refMrInit(MotionReference* self, PicYuv* recPic, WeightParam* wp, const x265_param& p) asm("_ZN4x26518MotionReferenceRef4initEPNS_6PicYuvEPNS_11WeightParamERK10x265_param");
extern void refMrApplyWeight(MotionReference* self, uint32_t finishedRows, uint32_t maxNumRows, uint32_t maxNumRowsInSlice, uint32_t sliceId)
    asm("_ZN4x26518MotionReferenceRef11applyWeightEjjjj");
extern void refMrDestruct(MotionReference* self) asm("_ZN4x26518MotionReferenceRefD2Ev");
static_assert(sizeof("" "x265") == 5, "");

typedef void (*x265hip_filter_body)(const pixel* src, intptr_t srcStride, pixel* dst, intptr_t dstStride, int fx, int fy);
bool x265hip_sadplanes_subpel(const pixel* src, intptr_t srcStride, pixel* dst, intptr_t dstStride, int w, int h, int fx, int fy, x265hip_filter_body body);    // x265_hip_sadplanes.cpp

namespace {

struct Mirror
{
    const pixel* lo;            // PicYuv::m_picBuf[0]
    const pixel* hi;            // one past the buffer
    intptr_t stride;
    uint64_t recip;             // 2^40 / stride + 1
    int picW, picH, marginX, marginY;
    const pixel* plane[16];     // host planes, same layout as the buffer ([0] unused)
    const int* rowsReady;       // published by the worker (acquire)
    x265hip_refpic* rp;
    // writer side (frame filter threads)
    std::mutex lock;
    int poc;
    std::atomic<uint32_t> generation;   // bumped whenever the buffer starts a new picture (x265_hip_sadplanes.cpp keys its surfaces on it)
    bool tracking;              // the picture now in the buffer is a reference picture whose rows we publish
    uint64_t rowDone[4];        // CTU rows whose processPostRow has run (slices may finish out of order)
    int prefix;                 // CTU rows [0, prefix) are done
    bool dead;                  // a device call on this mirror failed: nothing is served from it any more (kill_mirror)
};

const int kMaxMirrors = 64;
Mirror g_mirror[kMaxMirrors];
// [lo, hi) of every entry once more, side by side: what a filter call scans to find its picture (the entries themselves are several cache lines each)
struct Range { const pixel* lo; const pixel* hi; };
Range g_range[kMaxMirrors];
uint64_t g_weightedMirrors = 0;  // under g_createLock
std::atomic<bool> g_noNewMirrors(false);   // a mirror could not be created: no further attempts (x265hip_device_failure said why)
std::atomic<int> g_count(0);
std::mutex g_createLock;
int g_state = 0;                 // 0 undecided, 1 on, -1 off
bool g_verify = false;           // X265HIP_VERIFY=1: served blocks are recomputed with the C filter and compared (debugging self-check)
EncoderPrimitives g_c;           // the slots' previous contents
struct alignas(64) Counter { std::atomic<uint64_t> v; };            // one cache line each: the slots run on every pool worker at once
Counter g_served[64], g_missed[64], g_foreign[64];
std::atomic<int> g_nextShard(0);
__attribute__((tls_model("initial-exec"))) thread_local int t_shard = -1;

inline int shard()
{
    if (t_shard < 0) t_shard = g_nextShard.fetch_add(1) & 63;
    return t_shard;
}
// the slots count in plain thread-local integers (an atomic add per filter call is felt at 20 million calls); a thread's share goes to the shared
// counters when the thread ends — the pool's workers end with their encoder, before the exit handlers print
struct ThreadCounts
{
    uint64_t served = 0, missed = 0, foreign = 0;
    ~ThreadCounts()
    {
        g_served[shard()].v.fetch_add(served, std::memory_order_relaxed);
        g_missed[shard()].v.fetch_add(missed, std::memory_order_relaxed);
        g_foreign[shard()].v.fetch_add(foreign, std::memory_order_relaxed);
    }
};
thread_local ThreadCounts t_counts;

void report()
{
    uint64_t s = 0, m = 0, f = 0;
    for (int i = 0; i < 64; i++) { s += g_served[i].v; m += g_missed[i].v; f += g_foreign[i].v; }
    fprintf(stderr, "x265hip: refplanes: %llu luma sub-pel filter calls served from GPU-built planes of %d mirrored pictures, %llu on mirrored pictures before "
                    "their rows arrived and %llu on other memory computed on the host\n", (unsigned long long)s, g_count.load(), (unsigned long long)m,
            (unsigned long long)f);
    if (g_weightedMirrors)
        fprintf(stderr, "x265hip: refplanes: %llu of the mirrored pictures are weighted copies of reference pictures (MotionReference::applyWeight)\n",
                (unsigned long long)g_weightedMirrors);
}

bool switched_on()
{
    const char* env = getenv("X265HIP_REFPLANES");
    const char* all = getenv("X265HIP");
    const char* table = getenv("X265HIP_TABLE");
    return !((env && !strcmp(env, "0")) || (all && !strcmp(all, "0")) || (table && !strcmp(table, "percall")));
}

bool enabled()
{
    if (!g_state)
    {
        std::lock_guard<std::mutex> g(g_createLock);
        if (!g_state)
        {
            const char* env = getenv("X265HIP_REFPLANES");
            const char* all = getenv("X265HIP");
            const char* table = getenv("X265HIP_TABLE");
            if ((env && !strcmp(env, "0")) || (all && !strcmp(all, "0")) || (table && !strcmp(table, "percall")) || x265hip_device_count() < 1)
                g_state = -1;
            else
            {
                g_state = 1;
                g_verify = getenv("X265HIP_VERIFY") != NULL;
                if (getenv("X265HIP_VERBOSE"))
                    atexit(report);
            }
        }
    }
    return g_state > 0;
}

inline const Mirror* find(const pixel* p)
{
    // the picture the thread's previous call lay in, first: the candidates of one sub-pel refinement and the blocks of one PU's compensation share it
    static __attribute__((tls_model("initial-exec"))) thread_local int t_last = 0;
    {
        const pixel* lo = __atomic_load_n(&g_range[t_last].lo, __ATOMIC_ACQUIRE);
        if (lo && p >= lo && p < g_range[t_last].hi)
            return &g_mirror[t_last];
    }
    const int n = g_count.load(std::memory_order_acquire);
    for (int i = 0; i < n; i++)
    {
        const pixel* lo = __atomic_load_n(&g_range[i].lo, __ATOMIC_ACQUIRE);      // NULL: a retired entry (or one being set up: lo is stored last)
        if (lo && p >= lo && p < g_range[i].hi)
        {
            t_last = i;
            return &g_mirror[i];
        }
    }
    return NULL;
}

// the mirror of the padded luma picture that starts at `lo` (a PicYuv's buffer, or a MotionReference's weighted copy of one: same geometry), created
// on first use; `create` false: only an existing one
Mirror* mirror_at(const pixel* lo, const PicYuv* pic, bool weighted, bool create = true)
{
    const int n = g_count.load(std::memory_order_acquire);
    for (int i = 0; i < n; i++)
        if (g_mirror[i].lo == lo)
            return &g_mirror[i];
    std::lock_guard<std::mutex> g(g_createLock);
    const int n2 = g_count.load();
    for (int i = n; i < n2; i++)
        if (g_mirror[i].lo == lo)
            return &g_mirror[i];
    if (!create || g_noNewMirrors.load(std::memory_order_relaxed))
        return NULL;
    int slot = n2;
    for (int i = 0; i < n2; i++)
        if (!g_mirror[i].lo && !g_mirror[i].rp)
        {
            slot = i;                              // the entry of a destroyed buffer (x265hip_refplanes_retire)
            break;
        }
    if (slot == kMaxMirrors)
        return NULL;
    Mirror& m = g_mirror[slot];
    const int maxCU = pic->m_param->maxCUSize;
    const int bufRows = (int)(((pic->m_picHeight + maxCU - 1) / maxCU) * maxCU + 2 * pic->m_lumaMarginY);       // picyuv.cpp:95-98
    x265hip_debug_mark("create: reference-picture mirror");
    const int places = x265hip_places_configured();
    m.rp = places ? x265hip_refpic_create_at(slot % places, X265_DEPTH, pic->m_picWidth, pic->m_picHeight, pic->m_stride, pic->m_lumaMarginX, pic->m_lumaMarginY, bufRows, lo)
                  : x265hip_refpic_create(X265_DEPTH, pic->m_picWidth, pic->m_picHeight, pic->m_stride, pic->m_lumaMarginX, pic->m_lumaMarginY, bufRows, lo);
    x265hip_debug_mark("created: reference-picture mirror");
    if (!m.rp)
    {
        // no mirror for this buffer (nor for any later one: the device is out of memory or gone): its filter calls are "on other memory" -> the C filters
        g_noNewMirrors.store(true);
        x265hip_device_failure("refplanes", "reference-picture mirror");
        return NULL;
    }
    m.hi = lo + (size_t)pic->m_stride * bufRows;
    g_range[slot].hi = m.hi;
    g_weightedMirrors += weighted;
    m.stride = pic->m_stride;
    m.recip = ((uint64_t)1 << 40) / (uint64_t)pic->m_stride + 1;
    m.picW = pic->m_picWidth; m.picH = pic->m_picHeight; m.marginX = pic->m_lumaMarginX; m.marginY = pic->m_lumaMarginY;
    m.plane[0] = NULL;
    for (int p = 1; p < 16; p++)
        m.plane[p] = (const pixel*)x265hip_refpic_plane(m.rp, p);
    m.rowsReady = x265hip_refpic_rows_ready_ptr(m.rp);
    m.poc = -1;
    m.dead = false;
    m.tracking = false;
    m.prefix = 0;
    memset(m.rowDone, 0, sizeof(m.rowDone));
    m.lo = lo;
    __atomic_store_n(&g_range[slot].lo, lo, __ATOMIC_RELEASE);       // last: find() matches an entry by [lo, hi)
    if (slot == n2)
        g_count.store(n2 + 1, std::memory_order_release);
    return &m;
}

Mirror* mirror_of(PicYuv* pic) { return mirror_at(pic->m_picBuf[0], pic, false); }

// a device call on this mirror failed: nothing of it is served any more (its "rows ready" count reads 0 for good), the C filters take over
const int g_noRows = -(1 << 30);      // the library's own "nothing is valid" value (refpic.hip)
void kill_mirror(Mirror* m, const char* what)
{
    m->rowsReady = &g_noRows;
    m->tracking = false;
    m->dead = true;
    x265hip_device_failure("refplanes", what);
}

// W x H block of phase plane `phase` at `src`, if `src` lies in a mirrored picture whose rows have arrived
template <int W, int H>
inline bool serve(const pixel* src, intptr_t srcStride, pixel* dst, intptr_t dstStride, int phase)
{
    const Mirror* m = find(src);
    if (!m)
    {
        t_counts.foreign++;
        return false;
    }
    const ptrdiff_t off = src - m->lo;
    // off / stride without a division: off < 2^26, stride < 2^14, recip = 2^40 / stride + 1 — the estimate's error is below 2^-14 of a row, less than
    // the distance of any off / stride from the next integer (1 / stride)
    const int by = (int)(((uint64_t)off * m->recip) >> 40), bx = (int)(off - (ptrdiff_t)by * m->stride);
    const int y = by - m->marginY, x = bx - m->marginX;
    // a block that lies entirely in the top margin needs the first band too (the margin's planes are built with it): at least one row must have arrived
    if (srcStride != m->stride || x < -(m->marginX - 4) || x + W > m->picW + m->marginX - 4 || y < -(m->marginY - 4) ||
        (y + H > 1 ? y + H : 1) > __atomic_load_n(m->rowsReady, __ATOMIC_ACQUIRE))
    {
        t_counts.missed++;
        return false;
    }
    const pixel* p = m->plane[phase] + off;
    for (int r = 0; r < H; r++)
        memcpy(dst + r * dstStride, p + r * m->stride, W * sizeof(pixel));
    t_counts.served++;
    return true;
}

// X265HIP_VERIFY=1: every served block is recomputed with the C filter and compared (self-check for debugging; off by default)
template <int W, int H>
void verify(const char* what, const pixel* s, pixel* d, intptr_t ds, const pixel* want, int cx, int cy)
{
    for (int r = 0; r < H; r++)
        if (memcmp(d + r * ds, want + r * W, W * sizeof(pixel)))
        {
            const Mirror* m = find(s);
            const ptrdiff_t off = s - m->lo;
            const int by = (int)(off / m->stride), bx = (int)(off - (ptrdiff_t)by * m->stride);
            fprintf(stderr, "x265hip: refplanes: VERIFY FAILED %s %dx%d phase (%d, %d) at x %d y %d row %d of poc %d (rowsReady %d, picH %d)\n", what, W, H, cx, cy,
                    bx - m->marginX, by - m->marginY, r, m->poc, *m->rowsReady, m->picH);
            abort();
        }
}
// the lookups proper: (cx, cy) = the fractional position; hpp uses cx, vpp cy
template <int W, int H, int PART> void hpp_body(const pixel* s, intptr_t ss, pixel* d, intptr_t ds, int c, int)
{
    if (!c || !serve<W, H>(s, ss, d, ds, c)) { g_c.pu[PART].luma_hpp(s, ss, d, ds, c); return; }
    if (g_verify) { pixel t[W * H]; g_c.pu[PART].luma_hpp(s, ss, t, W, c); verify<W, H>("hpp", s, d, ds, t, c, 0); }
}
template <int W, int H, int PART> void vpp_body(const pixel* s, intptr_t ss, pixel* d, intptr_t ds, int, int c)
{
    if (!c || !serve<W, H>(s, ss, d, ds, 4 * c)) { g_c.pu[PART].luma_vpp(s, ss, d, ds, c); return; }
    if (g_verify) { pixel t[W * H]; g_c.pu[PART].luma_vpp(s, ss, t, W, c); verify<W, H>("vpp", s, d, ds, t, 0, c); }
}
template <int W, int H, int PART> void hvpp_body(const pixel* s, intptr_t ss, pixel* d, intptr_t ds, int cx, int cy)
{
    if (!cx || !cy || !serve<W, H>(s, ss, d, ds, 4 * cy + cx)) { g_c.pu[PART].luma_hvpp(s, ss, d, ds, cx, cy); return; }
    if (g_verify) { pixel t[W * H]; g_c.pu[PART].luma_hvpp(s, ss, t, W, cx, cy); verify<W, H>("hvpp", s, d, ds, t, cx, cy); }
}
// the table slots: a square block of 16 and up may be the sub-pel candidate of a motion search whose SATD table holds the answer
// (x265hip_sadplanes_subpel: the block is then not produced at all unless somebody turns out to need it — it gets the body to do that)
template <int W, int H, int PART> void hpp_lookup(const pixel* s, intptr_t ss, pixel* d, intptr_t ds, int c)
{
    if (W == H && W >= 16 && x265hip_sadplanes_subpel(s, ss, d, ds, W, H, c, 0, hpp_body<W, H, PART>)) return;
    hpp_body<W, H, PART>(s, ss, d, ds, c, 0);
}
template <int W, int H, int PART> void vpp_lookup(const pixel* s, intptr_t ss, pixel* d, intptr_t ds, int c)
{
    if (W == H && W >= 16 && x265hip_sadplanes_subpel(s, ss, d, ds, W, H, 0, c, vpp_body<W, H, PART>)) return;
    vpp_body<W, H, PART>(s, ss, d, ds, 0, c);
}
template <int W, int H, int PART> void hvpp_lookup(const pixel* s, intptr_t ss, pixel* d, intptr_t ds, int cx, int cy)
{
    if (W == H && W >= 16 && x265hip_sadplanes_subpel(s, ss, d, ds, W, H, cx, cy, hvpp_body<W, H, PART>)) return;
    hvpp_body<W, H, PART>(s, ss, d, ds, cx, cy);
}

} // namespace

#define LOOKUP_PU(W, H) do { \
        p.pu[LUMA_ ## W ## x ## H].luma_hpp = hpp_lookup<W, H, LUMA_ ## W ## x ## H>; \
        p.pu[LUMA_ ## W ## x ## H].luma_vpp = vpp_lookup<W, H, LUMA_ ## W ## x ## H>; \
        p.pu[LUMA_ ## W ## x ## H].luma_hvpp = hvpp_lookup<W, H, LUMA_ ## W ## x ## H>; \
    } while (0)

// x265_hip_sadplanes.cpp: the device mirror of the reconstructed picture in `recon` and the generation of the picture it holds now
x265hip_refpic* x265hip_refplanes_device(const PicYuv* recon, uint32_t* generation)
{
    if (g_state <= 0)
        return NULL;
    const pixel* lo = recon->m_picBuf[0];
    const int n = g_count.load(std::memory_order_acquire);
    for (int i = 0; i < n; i++)
        if (__atomic_load_n(&g_mirror[i].lo, __ATOMIC_ACQUIRE) == lo)
        {
            *generation = g_mirror[i].generation.load(std::memory_order_acquire);
            return g_mirror[i].tracking ? g_mirror[i].rp : NULL;
        }
    return NULL;
}
bool x265hip_refplanes_current(const PicYuv* recon, uint32_t generation)
{
    uint32_t g = 0;
    return x265hip_refplanes_device(recon, &g) && g == generation;
}

// PicYuv::destroy (x265_hip_srcplanes.cpp): the buffer at `lo` is about to be freed.  Its mirror must stop answering for that address range
// before malloc can hand it to anybody else; nobody reads a picture that is being destroyed, so no reader is inside the entry.
void x265hip_refplanes_retire(const pixel* lo)
{
    if (g_state <= 0)
        return;
    std::lock_guard<std::mutex> g(g_createLock);
    const int n = g_count.load();
    for (int i = 0; i < n; i++)
        if (g_mirror[i].lo == lo)
        {
            Mirror& m = g_mirror[i];
            std::lock_guard<std::mutex> g2(m.lock);
            __atomic_store_n(&g_range[i].lo, (const pixel*)NULL, __ATOMIC_RELEASE);
            g_range[i].hi = NULL;
            m.lo = NULL;
            m.hi = NULL;
            m.rowsReady = NULL;
            m.tracking = false;
            m.poc = -1;
            m.generation.fetch_add(1, std::memory_order_release);
            x265hip_refpic_destroy(m.rp);           // waits for the worker's queued bands of this picture
            m.rp = NULL;
            return;
        }
}

// called by setupAssemblyPrimitives (x265_hip_primitives.cpp) in the default table mode
void x265hip_install_lookup_slots(EncoderPrimitives& p)
{
    // decided by the switches alone: whether a device exists is known a moment later (setupAssemblyPrimitives probes beside the encoder's own set-up),
    // and without one the wrappers find nothing to serve and call the C functions
    if (!switched_on())
        return;
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
    // X265HIP_REFPLANES_SMALL=0: blocks of 64 samples and fewer keep the C filters (a measurement switch: is a lookup of an 8x8 block — eight cold lines of a
    // page-locked plane — cheaper than filtering it out of the picture the search has just read?  profiles/r06_v1_refplanes_small_ab.txt: yes)
    const bool small = !(getenv("X265HIP_REFPLANES_SMALL") && !atoi(getenv("X265HIP_REFPLANES_SMALL")));
    if (small) { LOOKUP_PU(4, 4);   LOOKUP_PU(8, 8);   LOOKUP_PU(8, 4);   LOOKUP_PU(4, 8); }
    LOOKUP_PU(16, 16); LOOKUP_PU(32, 32); LOOKUP_PU(64, 64);
    LOOKUP_PU(16, 8);  LOOKUP_PU(8, 16);  LOOKUP_PU(32, 16); LOOKUP_PU(16, 32);
    LOOKUP_PU(64, 32); LOOKUP_PU(32, 64); LOOKUP_PU(16, 12); LOOKUP_PU(12, 16); LOOKUP_PU(16, 4);  LOOKUP_PU(4, 16);
    LOOKUP_PU(32, 24); LOOKUP_PU(24, 32); LOOKUP_PU(32, 8);  LOOKUP_PU(8, 32);  LOOKUP_PU(64, 48); LOOKUP_PU(48, 64);
    LOOKUP_PU(64, 16); LOOKUP_PU(16, 64);
}

void FrameFilter::processPostRow(int row)
{
    Mirror* m = NULL;
    // X265HIP_DEBUG_DELAY_US=n: sleep n microseconds here, seams on or off.  A diagnostic: the reference encoder's own output depends on thread
    // timing in a few corner configurations (tools/fuzz_encoder.py uses this to tell those from real mismatches)
    static const int delayUs = getenv("X265HIP_DEBUG_DELAY_US") ? atoi(getenv("X265HIP_DEBUG_DELAY_US")) : 0;
    if (delayUs > 0) usleep(delayUs);
    if (enabled() && m_frame && m_frame->m_reconPic && m_numRows <= 256)
    {
        // before the reference's body announces the row (m_reconRowFlag, framefilter.cpp:664): a buffer that starts a new picture must not
        // keep serving the old picture's planes
        m = mirror_of(m_frame->m_reconPic);
        if (m)
        {
            std::lock_guard<std::mutex> g(m->lock);
            if (m->poc != m_frame->m_poc)
            {
                if (!m->dead && x265hip_refpic_reset(m->rp))
                    kill_mirror(m, "x265hip_refpic_reset");
                m->poc = m_frame->m_poc;
                m->generation.fetch_add(1, std::memory_order_release);
                m->tracking = !m->dead && IS_REFERENCED(m_frame);           // unreferenced B pictures are never searched: nothing to build
                m->prefix = 0;
                memset(m->rowDone, 0, sizeof(m->rowDone));
            }
        }
    }
    // (before the body: its last statement lets the frame encoder go on to the next frame, framefilter.cpp:719-722, and m_frame changes under a reader)
    // X265HIP_DEBUG_TRACE=1: one line per finished CTU row — picture width, POC, slice type and QP, and a hash of the row's reconstructed luma and chroma; the
    // first line that differs between two runs names the first picture row whose reconstruction differs (seams on or off)
    static const bool trace = getenv("X265HIP_DEBUG_TRACE") != NULL;
    if (trace && m_frame && m_frame->m_reconPic && m_frame->m_encData)
    {
        const PicYuv* rec = m_frame->m_reconPic;
        const int cs = (int)m_param->maxCUSize, y0 = row * cs, y1 = X265_MIN((int)rec->m_picHeight, y0 + cs);
        uint64_t h = 1469598103934665603ull, hc = h;
        for (int y = y0; y < y1; y++)
            for (int x = 0; x < (int)rec->m_picWidth; x++) { h ^= rec->m_picOrg[0][(intptr_t)y * rec->m_stride + x]; h *= 1099511628211ull; }
        if (rec->m_picCsp != X265_CSP_I400)
            for (int k = 1; k < 3; k++)
                for (int y = y0 >> rec->m_vChromaShift; y < (y1 >> rec->m_vChromaShift); y++)
                    for (int x = 0; x < (int)(rec->m_picWidth >> rec->m_hChromaShift); x++) { hc ^= rec->m_picOrg[k][(intptr_t)y * rec->m_strideC + x]; hc *= 1099511628211ull; }
        const Slice* sl = m_frame->m_encData->m_slice;
        // ... and, per CTU of the row, of the decisions its analysis left in the picture's CU data (final once the row is analysed, unlike the pixels above,
        // which the in-loop filters of the next row still touch)
        const uint32_t wCtu = (rec->m_picWidth + cs - 1) / cs;
        for (uint32_t c = 0; c < wCtu; c++)
        {
            const CUData* ctu = m_frame->m_encData->getPicCTU(row * wCtu + c);
            const uint32_t np = ctu->m_numPartitions;
            // vectors count only where the list is used (the other entries are whatever the buffer held)
            auto hmv = [np, ctu](int list) { uint64_t x = 1469598103934665603ull; for (uint32_t i = 0; i < np; i++) if (ctu->m_predMode[i] != MODE_INTRA && ctu->m_refIdx[list][i] >= 0) { x ^= (uint32_t)ctu->m_mv[list][i].word; x *= 1099511628211ull; } return (unsigned)(x ^ (x >> 32)) & 0xffffff; };
            auto hb = [np](const void* p, size_t el) { uint64_t x = 1469598103934665603ull; const uint8_t* b = (const uint8_t*)p; for (size_t i = 0; i < np * el; i++) { x ^= b[i]; x *= 1099511628211ull; } return (unsigned)(x ^ (x >> 32)) & 0xffffff; };
            fprintf(stderr, "x265hip-trace: ctu w %d poc %d addr %u depth %06x pred %06x part %06x merge %06x skip %06x dir %06x ref0 %06x ref1 %06x mv0 %06x mv1 %06x mvp0 %06x mvp1 %06x tu %06x cbfY %06x cbfU %06x "
                    "cbfV %06x qp %06x intra %06x\n", (int)rec->m_picWidth, m_frame->m_poc, row * wCtu + c, hb(ctu->m_cuDepth, 1), hb(ctu->m_predMode, 1), hb(ctu->m_partSize, 1), hb(ctu->m_mergeFlag, 1),
                    hb(ctu->m_skipFlag[0], 1), hb(ctu->m_interDir, 1), hb(ctu->m_refIdx[0], 1), hb(ctu->m_refIdx[1], 1), hmv(0), hmv(1), hb(ctu->m_mvpIdx[0], 1),
                    hb(ctu->m_mvpIdx[1], 1), hb(ctu->m_tuDepth, 1), hb(ctu->m_cbf[0], 1), hb(ctu->m_cbf[1], 1), hb(ctu->m_cbf[2], 1), hb(ctu->m_qp, 1), hb(ctu->m_lumaIntraDir, 1));
        }
    }
    // the body's last statement counts the row as complete (framefilter.cpp:719-722): with the picture's last row the frame encoder goes on and
    // m_frame may already be the NEXT picture when the body returns — whatever is needed afterwards is read now
    const int pocOfRow = m_frame ? m_frame->m_poc : -1;
    refProcessPostRow(this, row);
    if (m)
    {
        std::lock_guard<std::mutex> g(m->lock);
        if (m->tracking && m->poc == pocOfRow)
        {
            m->rowDone[row >> 6] |= 1ull << (row & 63);
            int prefix = m->prefix;
            while (prefix < m_numRows && (m->rowDone[prefix >> 6] >> (prefix & 63) & 1))
                prefix++;
            if (prefix > m->prefix)
            {
                m->prefix = prefix;
                const int rows = prefix == m_numRows ? m->picH : prefix * (int)m_param->maxCUSize;
                if (x265hip_refpic_rows_final(m->rp, rows))
                    kill_mirror(m, "x265hip_refpic_rows_final");
            }
        }
    }
}

// ---- weighted reference pictures -------------------------------------------------------------------------------------------------------------
// With weighted prediction the motion search of a frame measures its candidates against a WEIGHTED copy of the reference picture
// (MotionReference::init points fpelPlane[0] at weightBuffer[0], reference.cpp:86-103; applyWeight fills it row by row as the reference picture's
// rows become available, :119-186), so the sub-pel filter calls of subpelCompare read that copy — memory no recon mirror covers.  The copy is a
// padded picture of the same geometry whose rows become final in order: it gets a mirror of its own, reset whenever the MotionReference is
// initialised for another frame, fed after every applyWeight; the lookup slots then serve it like any other picture.  One slice per picture only
// (with several, rows are weighted per slice).
int MotionReference::init(PicYuv* recPic, WeightParam* wp, const x265_param& p)
{
    const int r = refMrInit(this, recPic, wp, p);
    if (r || !enabled() || !weightBuffer[0])
        return r;
    const bool lumaWeighted = isWeighted && fpelPlane[0] != recPic->m_picOrg[0] && p.maxSlices == 1;
    Mirror* m = mirror_at(weightBuffer[0], recPic, true, lumaWeighted);
    if (m)
    {
        std::lock_guard<std::mutex> g(m->lock);
        if (!m->dead && x265hip_refpic_reset(m->rp))
            kill_mirror(m, "x265hip_refpic_reset");
        m->generation.fetch_add(1, std::memory_order_release);
        m->poc = -2;
        m->tracking = !m->dead && lumaWeighted;
        m->prefix = 0;
    }
    return r;
}

void MotionReference::applyWeight(uint32_t finishedRows, uint32_t maxNumRows, uint32_t maxNumRowsInSlice, uint32_t sliceId)
{
    refMrApplyWeight(this, finishedRows, maxNumRows, maxNumRowsInSlice, sliceId);
    if (g_state <= 0 || !isWeighted || sliceId || !weightBuffer[0] || fpelPlane[0] == reconPic->m_picOrg[0])
        return;
    Mirror* m = mirror_at(weightBuffer[0], reconPic, true, false);
    if (!m)
        return;
    std::lock_guard<std::mutex> g(m->lock);
    if (!m->tracking)
        return;
    // CTU rows [0, numSliceWeightedRows) are weighted; the call that reaches the last but one row does the last one with it, and the bottom margin
    const int done = (int)numSliceWeightedRows[0];
    const int rows = done >= (int)maxNumRows - 1 ? m->picH : done * (int)reconPic->m_param->maxCUSize;
    if (rows > m->prefix)
    {
        m->prefix = rows;
        if (x265hip_refpic_rows_final(m->rp, rows))
            kill_mirror(m, "x265hip_refpic_rows_final");
    }
}

MotionReference::~MotionReference()
{
    if (weightBuffer[0])
        x265hip_refplanes_retire(weightBuffer[0]);          // before the reference's body frees the buffer
    refMrDestruct(this);
}

} // namespace X265_NS
