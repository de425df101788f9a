// x265_hip_sadplanes.cpp — the fifth translation unit of the drop-in: the integer-pel SADs of the motion search served from GPU-built SAD
// surfaces (include/x265hip.h, x265hip_sadsurf_*).
//
// MotionEstimate::motionEstimate (reference source/encoder/motion.cpp:739-1569) measures its integer-pel candidates as
//     sad(fenc, FENC_STRIDE, fref + mx + my * stride, stride)                (macros :246-330; HEX :770-944, STAR :1132-1240, square refine :1430-1450)
// where fenc = fencPUYuv.m_buf[0] is a copy of the SOURCE picture's PU (setSourcePU, :194-222) and fref = ref->fpelPlane[0] + blockOffset a
// position in the finished reference picture (:752-756).  Which candidates a search visits depends on the decisions before it (the predictors
// come from the neighbours' vectors); what a candidate costs does not: SAD(source block at (x, y), reference block at (x + mx, y + my)) is a
// function of the two pictures, the position and the vector.  The GPU computes those values per (source picture, reference picture) pair for
// every aligned 8 / 16 / 32 / 64 block over a 16 x 16 window of vectors per block, placed around the block's own best match of an exhaustive
// +-32 search (x265hip_sadsurf, x265_amd/csrc/sadsurf.hip), as the reference picture's rows become final; a search then reads a candidate's
// SAD out of the table when the vector lies in the block's window and computes it with the C function otherwise.  Same value either way, so the
// bitstream does not depend on what has arrived or on where the windows lie.  Measured before it was built (every eligible
// call computed twice): the eligible calls are 9.8 % of the bound encoder's critical path at 1080p preset medium.
//
// Two seams on motion.o (same link technique as the other seams, oracle/Makefile):
//   MotionEstimate::setSourcePU     (analysis variant): notes, per MotionEstimate object of this thread, which source picture and position the
//                                   PU came from (x265hip_srcplanes_where) and compares the copied block with that picture byte for byte —
//                                   equal bytes have equal SADs: this comparison, not the bookkeeping, is what makes a lookup exact;
//   MotionEstimate::motionEstimate  runs the reference's own body with this object's sad / sad_x3 / sad_x4 pointers swapped for the lookups
//                                   while the (source picture, reference picture) pair has a surface whose rows cover this PU.
// Surfaces are attached on first use (the first search of a frame against a reference) and retired when either picture's buffer moves on.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#define protected public
#define private public
#include "common.h"
#include "frame.h"
#include "picyuv.h"
#include "primitives.h"
#include "yuv.h"
#include "lowres.h"
#include "mv.h"
#include "bitcost.h"
#include "motion.h"
#undef protected
#undef private

#include "x265hip.h"
#include "x265_hip_debug.h"

namespace X265_NS {

const EncoderPrimitives& x265hip_c_table();          // x265_hip_primitives.cpp
bool x265hip_srcplanes_where(const Yuv& y, const PicYuv** pic, uint32_t* version, int* px, int* py);     // x265_hip_srcplanes.cpp
x265hip_srcpic* x265hip_srcplanes_device(const PicYuv* pic, uint32_t version);
bool x265hip_srcplanes_current(const PicYuv* pic, uint32_t version);
x265hip_refpic* x265hip_refplanes_device(const PicYuv* recon, uint32_t* generation);                    // x265_hip_refplanes.cpp
bool x265hip_refplanes_current(const PicYuv* recon, uint32_t generation);

extern void refSetSourcePU(MotionEstimate* self, const Yuv& srcFencYuv, int ctuAddr, int cuPartIdx, int puPartIdx, int pwidth, int pheight, const int method,
                           const int refine, bool bChroma) asm("_ZN4x26517MotionEstimateRef11setSourcePUERKNS_3YuvEiiiiiiib");
#if X265_DEPTH == 8
extern int refMotionEstimate(MotionEstimate* self, ReferencePlanes* ref, const MV& mvmin, const MV& mvmax, const MV& qmvp, int numCandidates, const MV* mvc,
                             int merange, MV& outQMv, uint32_t maxSlices, pixel* srcReferencePlane)
    asm("_ZN4x26517MotionEstimateRef14motionEstimateEPNS_15ReferencePlanesERKNS_2MVES5_S5_iPS4_iRS3_jPh");
#else
extern int refMotionEstimate(MotionEstimate* self, ReferencePlanes* ref, const MV& mvmin, const MV& mvmax, const MV& qmvp, int numCandidates, const MV* mvc,
                             int merange, MV& outQMv, uint32_t maxSlices, pixel* srcReferencePlane)
    asm("_ZN4x26517MotionEstimateRef14motionEstimateEPNS_15ReferencePlanesERKNS_2MVES5_S5_iPS4_iRS3_jPt");
#endif
extern void refInitScales() asm("_ZN4x26517MotionEstimateRef10initScalesEv");

namespace {

const int WIN = X265HIP_SADSURF_WIN;

int g_state = 0;                 // 0 undecided, 1 on, -1 off
int g_exp = 0;                   // X265HIP_DEBUG_SADEXP=2: every eligible call is computed twice and nothing is looked up (the cost-doubling measurement)
int g_levels = 14;               // X265HIP_SADPLANES_LEVELS: bit l = blocks of 8 << l are looked up (8x8: only a third of the searches stay in the parent's window — off)
int g_time = 0;                  // X265HIP_DEBUG_SADTIME=1: cycles inside the reference's motionEstimate for the PUs a surface could serve, by block size;
                                 // =2: the same with the lookups switched off (the pair of runs measures what the lookups save)
std::atomic<uint64_t> g_cycles[4], g_timed[4];
bool g_missHist = false;         // X265HIP_DEBUG_SADMISS=1: how far outside their window do the misses lie, by block size (report at exit)
std::atomic<uint64_t> g_missBy[4][4];   // [level][0: within 8 of the window, 1: within 16, 2: within 32, 3: farther]
bool g_subpelHit = false;        // X265HIP_DEBUG_SUBPELHIT=1: how many served searches end within +-3 quarter-pels of the surface's own best vector (what a
                                 // table of sub-pel SATDs around that vector could serve at most), by block size (report at exit)
std::atomic<uint64_t> g_spHit[4], g_spAll[4];
std::atomic<uint64_t> g_spDist[4][5];
bool g_verify = false;           // X265HIP_VERIFY=1: every looked-up SAD is recomputed with the C function and compared (debugging self-check)
int g_range = 32;                // X265HIP_SADPLANES_RANGE: the exhaustive search that places the windows covers [-range, range)^2
EncoderPrimitives g_c;
std::mutex g_lock;

struct alignas(64) Counter { std::atomic<uint64_t> hit, miss, unserved; };
Counter g_count[64];
std::atomic<int> g_nextShard(0);
__attribute__((tls_model("initial-exec"))) thread_local int t_shard = -1;
inline int shard() { if (t_shard < 0) t_shard = g_nextShard.fetch_add(1) & 63; return t_shard; }

// what this thread's MotionEstimate objects hold (setSourcePU)
struct PuInfo { const MotionEstimate* me; const PicYuv* srcPic; uint32_t version; int x, y, w, h; };
const int kPu = 4;
__attribute__((tls_model("initial-exec"))) thread_local PuInfo t_pu[kPu];
__attribute__((tls_model("initial-exec"))) thread_local int t_puNext = 0;

// one (source picture, reference picture) pair
struct Pair
{
    const PicYuv* srcPic; uint32_t version;
    const PicYuv* recon; uint32_t generation;
    x265hip_sadsurf* ss;
    const x265hip_sadsurf_view* view;
};
std::vector<Pair*> g_pairs;          // under g_lock
uint64_t g_attached = 0, g_retired = 0;

// per thread: the pairs it met last (a frame encoder's workers keep asking for the same few)
const int kCache = 8;
__attribute__((tls_model("initial-exec"))) thread_local Pair t_cache[kCache];
__attribute__((tls_model("initial-exec"))) thread_local int t_cacheNext = 0;

// the lookup context of the motionEstimate call in progress on this thread
struct Ctx
{
    const pixel* fenc;           // fencPUYuv.m_buf[0]
    const pixel* winBase;        // reference position of window entry (0, 0)
    intptr_t stride;
    size_t span;                 // (WIN - 1) * stride + WIN: pointers at or beyond winBase + span are outside the window
    const void* tab;             // the block's WIN * WIN entries
    uint32_t hit, miss;
    int x, y, w, ox, oy;         // for X265HIP_VERIFY's message
};
__attribute__((tls_model("initial-exec"))) thread_local Ctx t_ctx;

// the sub-pel SATD context of the motionEstimate call in progress on this thread (round 4): MotionEstimate::subpelCompare (motion.cpp:1571-1600)
// measures a quarter-pel candidate as satd(fenc, luma_hpp / vpp / hvpp(fref + integer part)) — first a filter slot, then this object's satd.  While a
// context is active, the filter lookups of x265_hip_refplanes.cpp ask x265hip_sadplanes_subpel() first: a candidate within +-3 quarter-pels of the
// window's centre is in the block's table, so the filtered block is not even copied and the satd that follows returns the table's value.
struct SubCtx
{
    const uint32_t* tab;         // the block's 49 entries; NULL: no context
    const pixel* fenc;
    const pixel* centre;         // reference position of the centre vector c
    intptr_t stride;
    int n;                       // block size
    const pixel* dst;            // the buffer the last intercepted filter call was asked to fill (its satd comes next)
    int val, lastK;
    // what that filter call was, in case the comparison that follows is not the satd (a predictor measured with SAD, motion.cpp:778): then the block is
    // produced after all
    void (*body)(const pixel*, intptr_t, pixel*, intptr_t, int, int);
    const pixel* src; intptr_t dstStride; int fx, fy;
    uint32_t hit, miss;
};
__attribute__((tls_model("initial-exec"))) thread_local SubCtx t_sub;
int g_subpel = 1;                // X265HIP_SADPLANES_SUBPEL=0: integer-pel lookups only
int g_rect = 1;                  // X265HIP_SADPLANES_RECT=0: square PUs only
std::atomic<uint64_t> g_rectHit(0), g_rectMiss(0), g_rectSearches(0);
std::atomic<uint64_t> g_subHit(0), g_subMiss(0);
__attribute__((tls_model("initial-exec"))) thread_local uint64_t t_hit = 0, t_miss = 0, t_searches = 0;
// what a thread has not reported yet goes to the shared counters when the thread ends (the pool's workers end with the encoder)
struct FlushAtThreadExit
{
    ~FlushAtThreadExit()
    {
        if (t_hit | t_miss)
        {
            g_count[0].hit.fetch_add(t_hit, std::memory_order_relaxed);
            g_count[0].miss.fetch_add(t_miss, std::memory_order_relaxed);
            t_hit = t_miss = 0;
        }
    }
};
thread_local FlushAtThreadExit t_flushAtExit;

void report()
{
    uint64_t h = 0, m = 0, un = 0;
    for (int i = 0; i < 64; i++) { h += g_count[i].hit; m += g_count[i].miss; un += g_count[i].unserved; }
    uint64_t attached = 0, rows = 0, launches = 0, kernelNs = 0;
    x265hip_sadsurf_stats(&attached, &rows, &launches, &kernelNs);
    if (g_time)
        for (int l = 0; l < 4; l++)
            fprintf(stderr, "x265hip: sadplanes: block size %d: %llu searches, %.0f cycles each inside the reference's motionEstimate (%s)\n", 8 << l,
                    (unsigned long long)g_timed[l].load(), g_timed[l] ? (double)g_cycles[l].load() / g_timed[l].load() : 0.0, g_time == 2 ? "lookups off" : "lookups on");
    fprintf(stderr, "x265hip: sadplanes: %llu integer-pel SADs of the motion search served from GPU-built SAD surfaces (%llu surfaces, %llu CTU rows in %llu launches, %.3f ms of device time), %llu of the same "
                    "searches outside their block's window and %llu searches without a surface computed on the host\n", (unsigned long long)h,
            (unsigned long long)attached, (unsigned long long)rows, (unsigned long long)launches, kernelNs * 1e-6, (unsigned long long)m, (unsigned long long)un);
    fprintf(stderr, "x265hip: sadplanes: %llu sub-pel SATDs of the motion search (filter + satd) served from GPU-built tables around the windows' centres, %llu of the same "
                    "searches elsewhere computed on the host\n", (unsigned long long)g_subHit.load(), (unsigned long long)g_subMiss.load());
    if (g_rectSearches.load())
        fprintf(stderr, "x265hip: sadplanes: rectangular / asymmetric PUs: %llu searches, %llu integer-pel SADs served as sums of their squares' entries, %llu with a square's "
                        "window elsewhere computed on the host\n", (unsigned long long)g_rectSearches.load(), (unsigned long long)g_rectHit.load(), (unsigned long long)g_rectMiss.load());
    if (g_subpelHit)
        for (int l = 1; l < 4; l++)
            fprintf(stderr, "x265hip: sadplanes: block size %d: %llu of %llu served searches end within 3 quarter-pels of the surface's own best vector (%.1f %%)\n", 8 << l,
                    (unsigned long long)g_spHit[l].load(), (unsigned long long)g_spAll[l].load(), g_spAll[l] ? 100.0 * g_spHit[l].load() / g_spAll[l].load() : 0.0);
    if (g_subpelHit)
        for (int l = 1; l < 4; l++)
            fprintf(stderr, "x265hip: sadplanes: block size %d: searches ending within 3 / 7 / 11 / 19 quarter-pels of the centre and farther: %llu / %llu / %llu / %llu / %llu\n", 8 << l,
                    (unsigned long long)g_spDist[l][0].load(), (unsigned long long)g_spDist[l][1].load(), (unsigned long long)g_spDist[l][2].load(),
                    (unsigned long long)g_spDist[l][3].load(), (unsigned long long)g_spDist[l][4].load());
    if (g_missHist)
        for (int l = 0; l < 4; l++)
            fprintf(stderr, "x265hip: sadplanes: block size %d: misses within 8 / 16 / 32 vectors of the window and farther: %llu / %llu / %llu / %llu\n", 8 << l,
                    (unsigned long long)g_missBy[l][0].load(), (unsigned long long)g_missBy[l][1].load(), (unsigned long long)g_missBy[l][2].load(), (unsigned long long)g_missBy[l][3].load());
    const int places = x265hip_places_configured();
    if (places)
    {
        uint64_t replicas = 0, bands = 0, bytes = 0;
        x265hip_peer_stats(&replicas, &bands, &bytes);
        fprintf(stderr, "x265hip: places: %d (X265HIP_DEVICES=%s); %llu replicas of reference pictures at other places, %llu bands of reconstructed rows (%.1f MB) "
                        "pushed device to device\n", places, getenv("X265HIP_DEVICES"), (unsigned long long)replicas, (unsigned long long)bands, bytes * 1e-6);
    }
}

bool decide()
{
    std::lock_guard<std::mutex> g(g_lock);
    if (!g_state)
    {
        const char* env = getenv("X265HIP_SADPLANES");
        const char* all = getenv("X265HIP");
        const char* table = getenv("X265HIP_TABLE");
        const char* exp = getenv("X265HIP_DEBUG_SADEXP");
        g_exp = exp ? atoi(exp) : 0;
        g_time = getenv("X265HIP_DEBUG_SADTIME") ? atoi(getenv("X265HIP_DEBUG_SADTIME")) : 0;
        g_verify = getenv("X265HIP_VERIFY") != NULL;
        g_missHist = getenv("X265HIP_DEBUG_SADMISS") != NULL;
        g_subpelHit = getenv("X265HIP_DEBUG_SUBPELHIT") != NULL;
        if (getenv("X265HIP_SADPLANES_LEVELS")) g_levels = atoi(getenv("X265HIP_SADPLANES_LEVELS")) & 15;
        if (getenv("X265HIP_SADPLANES_RANGE")) g_range = atoi(getenv("X265HIP_SADPLANES_RANGE"));
        if (getenv("X265HIP_SADPLANES_SUBPEL")) g_subpel = atoi(getenv("X265HIP_SADPLANES_SUBPEL"));
        if (getenv("X265HIP_SADPLANES_RECT")) g_rect = atoi(getenv("X265HIP_SADPLANES_RECT"));
        if (g_range < 8) g_range = 8;
        if (g_range > 32) g_range = 32;
        g_range &= ~3;
        // 16-bit builds (Main10 / Main12): the device keeps a CTU's u32 surface in LDS up to a range of 16, and has no 8x8 level
        if (X265_DEPTH != 8)
        {
            if (g_range > 16) g_range = 16;
            g_levels &= 14;
        }
        if ((env && !strcmp(env, "0")) || (all && !strcmp(all, "0")) || (table && !strcmp(table, "percall")) || !g_levels ||
            x265hip_device_count() < 1)
            g_state = -1;
        else
        {
            g_c = x265hip_c_table();
            refInitScales();            // the reference body's own file-static table (motion.cpp:60, :120-160): its copy in the second object
            g_state = 1;
            if (getenv("X265HIP_VERBOSE"))
                atexit(report);
        }
    }
    return g_state > 0;
}

inline bool enabled() { return g_state ? g_state > 0 : decide(); }

// the pair's surface: this thread's cache, then the table; attached on first use.  NULL: no device copy of one of the pictures (yet)
const Pair* pair_of(const PicYuv* srcPic, uint32_t version, const PicYuv* recon, int lambda20)
{
    uint32_t generation = 0;
    x265hip_refpic* rp = x265hip_refplanes_device(recon, &generation);
    if (!rp)
        return NULL;
    for (int i = 0; i < kCache; i++)
    {
        const Pair& c = t_cache[i];
        if (c.srcPic == srcPic && c.version == version && c.recon == recon && c.generation == generation)
            return &c;
    }
    std::lock_guard<std::mutex> g(g_lock);
    Pair* found = NULL;
    for (size_t i = 0; i < g_pairs.size();)
    {
        Pair* p = g_pairs[i];
        if (p->srcPic == srcPic && p->version == version && p->recon == recon && p->generation == generation)
            found = p;
        else if (!x265hip_srcplanes_current(p->srcPic, p->version) || !x265hip_refplanes_current(p->recon, p->generation))
        {
            // one of the two buffers holds another picture by now: the frame this surface served is finished, nobody reads it any more
            x265hip_sadsurf_release(p->ss);
            delete p;
            g_pairs[i] = g_pairs.back();
            g_pairs.pop_back();
            g_retired++;
            continue;
        }
        i++;
    }
    if (!found)
    {
        x265hip_srcpic* sp = x265hip_srcplanes_device(srcPic, version);
        if (!sp)
            return NULL;                     // the source picture's upload has not finished: the next search asks again
        x265hip_sadsurf* ss = x265hip_sadsurf_attach_levels(sp, rp, g_range, lambda20, g_levels | 14 | (g_subpel ? 16 : 0));
        if (!ss)
        {
            // no table for this pair, nor for later ones (out of memory, or the device is gone): every search measures its candidates with the C functions
            g_state = -1;
            x265hip_device_failure("sadplanes", "x265hip_sadsurf_attach");
            return NULL;
        }
        found = new Pair{ srcPic, version, recon, generation, ss, x265hip_sadsurf_get_view(ss) };
        g_pairs.push_back(found);
        g_attached++;
    }
    Pair& c = t_cache[t_cacheNext];
    t_cacheNext = (t_cacheNext + 1) % kCache;
    c = *found;
    return &c;
}

// entry of `p` in the window of the search in progress, or -1
inline int locate(const Ctx& c, const pixel* p)
{
    const size_t d = (size_t)(p - c.winBase);        // below the window: wraps to a huge value
    if (d >= c.span)
        return -1;
    const unsigned dy = (unsigned)d / (unsigned)c.stride, dx = (unsigned)d - dy * (unsigned)c.stride;
    return dx < (unsigned)WIN ? (int)(dy * WIN + dx) : -1;
}

// X265HIP_DEBUG_SADMISS=1: a candidate outside the window — how far outside?
void count_miss(const Ctx& c, const pixel* p)
{
    const ptrdiff_t d = p - c.winBase;
    ptrdiff_t dy = d / c.stride, dx = d - dy * c.stride;
    if (dx > c.stride / 2) { dx -= c.stride; dy++; }
    if (dx < -c.stride / 2) { dx += c.stride; dy--; }
    const int ox = dx < 0 ? (int)-dx : dx >= WIN ? (int)(dx - WIN + 1) : 0, oy = dy < 0 ? (int)-dy : dy >= WIN ? (int)(dy - WIN + 1) : 0;
    const int out = ox > oy ? ox : oy, level = c.w == 8 ? 0 : c.w == 16 ? 1 : c.w == 32 ? 2 : 3;
    g_missBy[level][out <= 8 ? 0 : out <= 16 ? 1 : out <= 32 ? 2 : 3].fetch_add(1, std::memory_order_relaxed);
}

// X265HIP_VERIFY=1: the table entry against the C function
template <int PART> void verify_entry(const Ctx& c, const pixel* fenc, const pixel* ref, intptr_t rs, int k, int got)
{
    const int want = g_c.pu[PART].sad(fenc, FENC_STRIDE, ref, rs);
    if (want != got)
    {
        fprintf(stderr, "x265hip: sadplanes: VERIFY FAILED block %dx%d at (%d, %d), window origin (%d, %d), entry (%d, %d): table %d, sad() %d\n", c.w, c.w, c.x, c.y, c.ox, c.oy,
                k % WIN, k / WIN, got, want);
        abort();
    }
}

template <int PART, typename E> int sad_lookup(const pixel* fenc, intptr_t fs, const pixel* ref, intptr_t rs)
{
    if (t_sub.dst && ref == t_sub.dst)
    {
        // the filtered block a table-held candidate did not produce is wanted after all: a SAD comparison (motion.cpp:778, :805)
        SubCtx& sc = t_sub;
        sc.dst = NULL;
        if (!g_verify) sc.body(sc.src, sc.stride, const_cast<pixel*>(ref), sc.dstStride, sc.fx, sc.fy);
        return g_c.pu[PART].sad(fenc, fs, ref, rs);
    }
    Ctx& c = t_ctx;
    if (fenc == c.fenc && rs == c.stride)
    {
        const int k = locate(c, ref);
        if (k >= 0)
        {
            c.hit++;
            if (g_verify) verify_entry<PART>(c, fenc, ref, rs, k, (int)((const E*)c.tab)[k]);
            return (int)((const E*)c.tab)[k];
        }
        c.miss++;
        if (g_missHist) count_miss(c, ref);
    }
    return g_c.pu[PART].sad(fenc, fs, ref, rs);
}
template <int PART, typename E> void sad_x3_lookup(const pixel* fenc, const pixel* r0, const pixel* r1, const pixel* r2, intptr_t rs, int32_t* res)
{
    Ctx& c = t_ctx;
    if (fenc != c.fenc || rs != c.stride) { g_c.pu[PART].sad_x3(fenc, r0, r1, r2, rs, res); return; }
    const int k0 = locate(c, r0), k1 = locate(c, r1), k2 = locate(c, r2);
    if (k0 < 0 && k1 < 0 && k2 < 0)
    {
        c.miss += 3;
        if (g_missHist) { count_miss(c, r0); count_miss(c, r1); count_miss(c, r2); }
        g_c.pu[PART].sad_x3(fenc, r0, r1, r2, rs, res);
        return;
    }
    const E* t = (const E*)c.tab;
    // sad_x3 is three independent sad<lx, ly> (pixel.cpp:74-95): a candidate outside the window is measured on its own
    res[0] = k0 >= 0 ? (int32_t)t[k0] : g_c.pu[PART].sad(fenc, FENC_STRIDE, r0, rs);
    res[1] = k1 >= 0 ? (int32_t)t[k1] : g_c.pu[PART].sad(fenc, FENC_STRIDE, r1, rs);
    res[2] = k2 >= 0 ? (int32_t)t[k2] : g_c.pu[PART].sad(fenc, FENC_STRIDE, r2, rs);
    const int h = (k0 >= 0) + (k1 >= 0) + (k2 >= 0);
    c.hit += h; c.miss += 3 - h;
    if (g_missHist) { if (k0 < 0) count_miss(c, r0); if (k1 < 0) count_miss(c, r1); if (k2 < 0) count_miss(c, r2); }
    if (g_verify)
    {
        if (k0 >= 0) verify_entry<PART>(c, fenc, r0, rs, k0, res[0]);
        if (k1 >= 0) verify_entry<PART>(c, fenc, r1, rs, k1, res[1]);
        if (k2 >= 0) verify_entry<PART>(c, fenc, r2, rs, k2, res[2]);
    }
}
template <int PART, typename E> void sad_x4_lookup(const pixel* fenc, const pixel* r0, const pixel* r1, const pixel* r2, const pixel* r3, intptr_t rs, int32_t* res)
{
    Ctx& c = t_ctx;
    if (fenc != c.fenc || rs != c.stride) { g_c.pu[PART].sad_x4(fenc, r0, r1, r2, r3, rs, res); return; }
    const int k0 = locate(c, r0), k1 = locate(c, r1), k2 = locate(c, r2), k3 = locate(c, r3);
    if (k0 < 0 && k1 < 0 && k2 < 0 && k3 < 0)
    {
        c.miss += 4;
        if (g_missHist) { count_miss(c, r0); count_miss(c, r1); count_miss(c, r2); count_miss(c, r3); }
        g_c.pu[PART].sad_x4(fenc, r0, r1, r2, r3, rs, res);
        return;
    }
    const E* t = (const E*)c.tab;
    res[0] = k0 >= 0 ? (int32_t)t[k0] : g_c.pu[PART].sad(fenc, FENC_STRIDE, r0, rs);
    res[1] = k1 >= 0 ? (int32_t)t[k1] : g_c.pu[PART].sad(fenc, FENC_STRIDE, r1, rs);
    res[2] = k2 >= 0 ? (int32_t)t[k2] : g_c.pu[PART].sad(fenc, FENC_STRIDE, r2, rs);
    res[3] = k3 >= 0 ? (int32_t)t[k3] : g_c.pu[PART].sad(fenc, FENC_STRIDE, r3, rs);
    const int h = (k0 >= 0) + (k1 >= 0) + (k2 >= 0) + (k3 >= 0);
    c.hit += h; c.miss += 4 - h;
    if (g_missHist) { if (k0 < 0) count_miss(c, r0); if (k1 < 0) count_miss(c, r1); if (k2 < 0) count_miss(c, r2); if (k3 < 0) count_miss(c, r3); }
    if (g_verify)
    {
        if (k0 >= 0) verify_entry<PART>(c, fenc, r0, rs, k0, res[0]);
        if (k1 >= 0) verify_entry<PART>(c, fenc, r1, rs, k1, res[1]);
        if (k2 >= 0) verify_entry<PART>(c, fenc, r2, rs, k2, res[2]);
        if (k3 >= 0) verify_entry<PART>(c, fenc, r3, rs, k3, res[3]);
    }
}


// ---- rectangular and asymmetric PUs (round 4; preset slow and slower: --rect, --amp) -------------------------------------------------------------
// A W x H PU made of aligned 16x16 / 32x32 squares (32x16, 16x32, 64x32, 32x64, 64x16, 16x64, 64x48, 48x64): its SAD at a vector is the sum of its
// squares' SADs at that vector (pixel.cpp:40-55 sums over the block's pixels), and every square has a window of its own in the surface — placed around
// ITS best vector, so the lookup serves a candidate only when it lies in all of them (halves that move together: usually); otherwise the C function.
struct RectPart { const void* tab; const pixel* winBase; int entryBytes; };
struct RectCtx { const pixel* fenc; intptr_t stride; size_t span; int n; RectPart part[6]; uint32_t hit, miss; };
__attribute__((tls_model("initial-exec"))) thread_local RectCtx t_rect;

inline int rect_sum(const RectCtx& c, const pixel* p)
{
    int sum = 0;
    for (int i = 0; i < c.n; i++)
    {
        const RectPart& q = c.part[i];
        const size_t d = (size_t)(p - q.winBase);
        if (d >= c.span) return -1;
        const unsigned dy = (unsigned)d / (unsigned)c.stride, dx = (unsigned)d - dy * (unsigned)c.stride;
        if (dx >= (unsigned)WIN) return -1;
        const int k = (int)(dy * WIN + dx);
        sum += q.entryBytes == 2 ? (int)((const uint16_t*)q.tab)[k] : (int)((const uint32_t*)q.tab)[k];
    }
    return sum;
}
template <int PART> inline int rect_one(RectCtx& c, const pixel* fenc, const pixel* ref, intptr_t rs)
{
    const int v = rect_sum(c, ref);
    if (v < 0) { c.miss++; return g_c.pu[PART].sad(fenc, FENC_STRIDE, ref, rs); }
    c.hit++;
    if (g_verify)
    {
        const int want = g_c.pu[PART].sad(fenc, FENC_STRIDE, ref, rs);
        if (want != v) { fprintf(stderr, "x265hip: sadplanes: VERIFY FAILED rectangular PU (%d squares): tables %d, sad() %d\n", c.n, v, want); abort(); }
    }
    return v;
}
template <int PART> int sad_rect(const pixel* fenc, intptr_t fs, const pixel* ref, intptr_t rs)
{
    RectCtx& c = t_rect;
    if (fenc == c.fenc && rs == c.stride && fs == FENC_STRIDE) return rect_one<PART>(c, fenc, ref, rs);
    return g_c.pu[PART].sad(fenc, fs, ref, rs);
}
template <int PART> void sad_x3_rect(const pixel* fenc, const pixel* r0, const pixel* r1, const pixel* r2, intptr_t rs, int32_t* res)
{
    RectCtx& c = t_rect;
    if (fenc != c.fenc || rs != c.stride) { g_c.pu[PART].sad_x3(fenc, r0, r1, r2, rs, res); return; }
    res[0] = rect_one<PART>(c, fenc, r0, rs); res[1] = rect_one<PART>(c, fenc, r1, rs); res[2] = rect_one<PART>(c, fenc, r2, rs);
}
template <int PART> void sad_x4_rect(const pixel* fenc, const pixel* r0, const pixel* r1, const pixel* r2, const pixel* r3, intptr_t rs, int32_t* res)
{
    RectCtx& c = t_rect;
    if (fenc != c.fenc || rs != c.stride) { g_c.pu[PART].sad_x4(fenc, r0, r1, r2, r3, rs, res); return; }
    res[0] = rect_one<PART>(c, fenc, r0, rs); res[1] = rect_one<PART>(c, fenc, r1, rs); res[2] = rect_one<PART>(c, fenc, r2, rs); res[3] = rect_one<PART>(c, fenc, r3, rs);
}
template <int PART> inline void install_rect(MotionEstimate* me) { me->sad = sad_rect<PART>; me->sad_x3 = sad_x3_rect<PART>; me->sad_x4 = sad_x4_rect<PART>; }
inline bool install_rect_for(MotionEstimate* me, int w, int h)
{
    if (w == 32 && h == 16) install_rect<LUMA_32x16>(me); else if (w == 16 && h == 32) install_rect<LUMA_16x32>(me);
    else if (w == 64 && h == 32) install_rect<LUMA_64x32>(me); else if (w == 32 && h == 64) install_rect<LUMA_32x64>(me);
    else if (w == 64 && h == 16) install_rect<LUMA_64x16>(me); else if (w == 16 && h == 64) install_rect<LUMA_16x64>(me);
    else if (w == 64 && h == 48) install_rect<LUMA_64x48>(me); else if (w == 48 && h == 64) install_rect<LUMA_48x64>(me);
    else return false;
    return true;
}
inline bool rect_shape(int w, int h)
{
    return (w == 32 && h == 16) || (w == 16 && h == 32) || (w == 64 && h == 32) || (w == 32 && h == 64) || (w == 64 && h == 16) || (w == 16 && h == 64) ||
           (w == 64 && h == 48) || (w == 48 && h == 64);
}

// X265HIP_DEBUG_SADEXP=2: the measurement that preceded this file — every call a lookup could serve is computed twice
template <int PART> int sad_twice(const pixel* fenc, intptr_t fs, const pixel* ref, intptr_t rs)
{
    if (fenc == t_ctx.fenc && rs == t_ctx.stride) { volatile int sink = g_c.pu[PART].sad(fenc, fs, ref, rs); (void)sink; }
    return g_c.pu[PART].sad(fenc, fs, ref, rs);
}
template <int PART> void sad_x3_twice(const pixel* fenc, const pixel* r0, const pixel* r1, const pixel* r2, intptr_t rs, int32_t* res)
{
    if (fenc == t_ctx.fenc && rs == t_ctx.stride) { int32_t tmp[3]; g_c.pu[PART].sad_x3(fenc, r0, r1, r2, rs, tmp); volatile int sink = tmp[0] + tmp[1] + tmp[2]; (void)sink; }
    g_c.pu[PART].sad_x3(fenc, r0, r1, r2, rs, res);
}
template <int PART> void sad_x4_twice(const pixel* fenc, const pixel* r0, const pixel* r1, const pixel* r2, const pixel* r3, intptr_t rs, int32_t* res)
{
    if (fenc == t_ctx.fenc && rs == t_ctx.stride) { int32_t tmp[4]; g_c.pu[PART].sad_x4(fenc, r0, r1, r2, r3, rs, tmp); volatile int sink = tmp[0] + tmp[1] + tmp[2] + tmp[3]; (void)sink; }
    g_c.pu[PART].sad_x4(fenc, r0, r1, r2, r3, rs, res);
}

// entry of the quarter-pel vector (integer position p, fractions fx, fy) in the context's table, or -1
inline int sub_locate(const SubCtx& c, const pixel* p, intptr_t stride, int fx, int fy)
{
    if (stride != c.stride) return -1;
    const ptrdiff_t d = p - c.centre;
    // |dx| <= 1 here for anything of interest: split d = dy * stride + dx with dx in (-stride / 2, stride / 2]
    ptrdiff_t dy = (d + stride / 2) / stride;
    if (d + stride / 2 < 0) dy = -((-(d + stride / 2) + stride - 1) / stride);
    const ptrdiff_t dx = d - dy * stride;
    const long qx = 4 * (long)dx + fx, qy = 4 * (long)dy + fy;
    if (qx < -3 || qx > 3 || qy < -3 || qy > 3) return -1;
    return (int)((qy + 3) * 7 + qx + 3);
}

template <int PART> int satd_lookup(const pixel* fenc, intptr_t fs, const pixel* ref, intptr_t rs)
{
    SubCtx& c = t_sub;
    if (c.tab && fenc == c.fenc)
    {
        if (ref == c.dst)
        {
            // the block the filter lookup was asked for a moment ago (and did not copy)
            c.dst = NULL;
            c.hit++;
            if (g_verify)
            {
                const int want = g_c.pu[PART].satd(fenc, fs, ref, rs);
                if (want != c.val)
                {
                    fprintf(stderr, "x265hip: sadplanes: VERIFY FAILED sub-pel satd %dx%d: table %d, satd() %d (entry %d; the table:", c.n, c.n, c.val, want, c.lastK);
                    for (int i = 0; i < 49; i++) fprintf(stderr, " %u", c.tab[i]);
                    fprintf(stderr, ")\n");
                    abort();
                }
            }
            return c.val;
        }
        const int k = sub_locate(c, ref, rs, 0, 0);          // an integer candidate measured with satd (the search's best integer vector, motion.cpp:1514)
        if (k >= 0)
        {
            c.hit++;
            if (g_verify)
            {
                const int want = g_c.pu[PART].satd(fenc, fs, ref, rs);
                if (want != (int)c.tab[k]) { fprintf(stderr, "x265hip: sadplanes: VERIFY FAILED integer satd %dx%d: table %u, satd() %d\n", c.n, c.n, c.tab[k], want); abort(); }
            }
            return (int)c.tab[k];
        }
        c.miss++;
    }
    return g_c.pu[PART].satd(fenc, fs, ref, rs);
}

template <int PART> inline void install(MotionEstimate* me, int entryBytes)
{
    if (g_exp == 2) { me->sad = sad_twice<PART>; me->sad_x3 = sad_x3_twice<PART>; me->sad_x4 = sad_x4_twice<PART>; }
    else if (entryBytes == 2) { me->sad = sad_lookup<PART, uint16_t>; me->sad_x3 = sad_x3_lookup<PART, uint16_t>; me->sad_x4 = sad_x4_lookup<PART, uint16_t>; }
    else { me->sad = sad_lookup<PART, uint32_t>; me->sad_x3 = sad_x3_lookup<PART, uint32_t>; me->sad_x4 = sad_x4_lookup<PART, uint32_t>; }
}

} // namespace

// x265_hip_refplanes.cpp's filter lookups, before they copy anything: is this the sub-pel candidate of a search whose table holds its satd?  true: the
// value is remembered for the satd call that follows, `dst` is NOT filled (nobody else reads subpelCompare's buffer)
bool x265hip_sadplanes_subpel(const pixel* src, intptr_t srcStride, pixel* dst, intptr_t dstStride, int w, int h, int fx, int fy,
                              void (*body)(const pixel*, intptr_t, pixel*, intptr_t, int, int))
{
    SubCtx& c = t_sub;
    if (!c.tab || w != c.n || h != c.n)
        return false;
    // whatever an earlier candidate left behind is void now (a predictor measured with SAD goes through the filter slot too, motion.cpp:770, and
    // nobody consumes its value)
    c.dst = NULL;
    const int k = sub_locate(c, src, srcStride, fx, fy);
    if (k < 0)
        return false;
    c.dst = dst;
    c.val = (int)c.tab[k];
    c.lastK = k;
    c.body = body; c.src = src; c.dstStride = dstStride; c.fx = fx; c.fy = fy;
    return !g_verify;                // X265HIP_VERIFY: the block is filtered as usual and the satd that follows compares
}

// x265_hip_srcplanes.cpp keeps a device copy of every source picture only when this returns true
bool x265hip_sadplanes_wanted() { return enabled() && g_exp != 2; }

void MotionEstimate::setSourcePU(const Yuv& srcFencYuv, int _ctuAddr, int cuPartIdx, int puPartIdx, int pwidth, int pheight, const int method, const int refine,
                                 bool bChroma)
{
    refSetSourcePU(this, srcFencYuv, _ctuAddr, cuPartIdx, puPartIdx, pwidth, pheight, method, refine, bChroma);
    if (!enabled())
        return;
    int slot = -1;
    for (int i = 0; i < kPu; i++)
        if (t_pu[i].me == this) { slot = i; break; }
    if (slot < 0) { slot = t_puNext; t_puNext = (t_puNext + 1) % kPu; }
    PuInfo& u = t_pu[slot];
    u.me = this;
    u.srcPic = NULL;
    const PicYuv* pic; uint32_t version; int cx, cy;
    const bool rect = pwidth != pheight;
    if (rect ? !(g_rect && rect_shape(pwidth, pheight)) : (pwidth < 8 || !(g_levels >> (pwidth == 8 ? 0 : pwidth == 16 ? 1 : pwidth == 32 ? 2 : 3) & 1)))
        return;
    if (!x265hip_srcplanes_where(srcFencYuv, &pic, &version, &cx, &cy))
        return;
    // puPartIdx: the PU's offset inside the CU's source cache (the second PU of a 2NxN / Nx2N / AMP CU does not start at the CU's corner)
    const int x = cx + g_zscanToPelX[puPartIdx], y = cy + g_zscanToPelY[puPartIdx];
    if ((x | y) & ((rect ? 16 : pwidth) - 1) || x + pwidth > (int)pic->m_picWidth || y + pheight > (int)pic->m_picHeight)
        return;
    // equal bytes have equal SADs: this comparison, not the bookkeeping, is what makes a lookup exact
    const pixel* p = pic->m_picOrg[0] + (intptr_t)y * pic->m_stride + x;
    const pixel* f = fencPUYuv.m_buf[0];
    for (int r = 0; r < pheight; r++)
        if (memcmp(f + r * FENC_STRIDE, p + r * pic->m_stride, pwidth * sizeof(pixel)))
            return;
    u.srcPic = pic; u.version = version; u.x = x; u.y = y; u.w = pwidth; u.h = pheight;
}

int MotionEstimate::motionEstimate(ReferencePlanes* ref, const MV& mvmin, const MV& mvmax, const MV& qmvp, int numCandidates, const MV* mvc, int merange,
                                   MV& outQMv, uint32_t maxSlices, pixel* srcReferencePlane)
{
    // X265HIP_DEBUG_TRACE=2: one line per search — what went in (hashes of the source PU and of the reference block at the PU's position and around it,
    // the predictor, the candidates, the range) and what came out; sorted, the lines of two runs of the same encode are equal or name the first search that differs
    static const bool trace = getenv("X265HIP_DEBUG_TRACE") && atoi(getenv("X265HIP_DEBUG_TRACE")) >= 2;
    if (trace && ctuAddr >= 0 && !ref->isLowres)
    {
        auto hb = [](const pixel* p, intptr_t stride, int w, int h) { uint64_t x = 1469598103934665603ull; for (int r = 0; r < h; r++) for (int c = 0; c < w; c++) { x ^= p[r * stride + c]; x *= 1099511628211ull; } return (unsigned)(x ^ (x >> 32)) & 0xffffff; };
        const intptr_t off = ref->reconPic->m_cuOffsetY[ctuAddr] + ref->reconPic->m_buOffsetY[absPartIdx];
        static const uint8_t puH[25] = { 4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 12, 16, 4, 16, 24, 32, 8, 32, 48, 64, 16, 64 };     // primitives.h:41-55 (blockheight is never set, motion.cpp:178)
        const int blockheight = puH[partEnum];
        const unsigned hs = hb(fencPUYuv.m_buf[0], FENC_STRIDE, blockwidth, blockheight), hr = hb(ref->fpelPlane[0] + off, ref->lumaStride, blockwidth, blockheight);
        const unsigned hwin = hb(ref->fpelPlane[0] + off - 8 * ref->lumaStride - 8, ref->lumaStride, blockwidth + 16, blockheight + 16);
        uint64_t hc = 1469598103934665603ull;
        for (int i = 0; i < numCandidates; i++) { hc ^= (uint32_t)mvc[i].word; hc *= 1099511628211ull; }
        const int r = refMotionEstimate(this, ref, mvmin, mvmax, qmvp, numCandidates, mvc, merange, outQMv, maxSlices, srcReferencePlane);
        static std::atomic<uint64_t> seq(0);
        fprintf(stderr, "x265hip-trace: me seq %llu w %d ctu %d part %d pu %dx%d weighted %d src %06x ref %06x win %06x mvp %d,%d cand %d %06x range %d,%d..%d,%d -> %d,%d cost %d\n", (unsigned long long)seq.fetch_add(1), (int)ref->reconPic->m_picWidth, ctuAddr,
                absPartIdx, blockwidth, blockheight, (int)ref->isWeighted, hs, hr, hwin, qmvp.x, qmvp.y, numCandidates, (unsigned)(hc ^ (hc >> 32)) & 0xffffff, mvmin.x, mvmin.y, mvmax.x, mvmax.y, outQMv.x, outQMv.y, r);
        return r;
    }
    if (g_state <= 0 || ctuAddr < 0 || srcReferencePlane || ref->isWeighted || ref->isLowres)
        return refMotionEstimate(this, ref, mvmin, mvmax, qmvp, numCandidates, mvc, merange, outQMv, maxSlices, srcReferencePlane);
    const PuInfo* u = NULL;
    for (int i = 0; i < kPu; i++)
        if (t_pu[i].me == this && t_pu[i].srcPic) { u = &t_pu[i]; break; }
    if (!u)
        return refMotionEstimate(this, ref, mvmin, mvmax, qmvp, numCandidates, mvc, merange, outQMv, maxSlices, srcReferencePlane);
    if (u->w != u->h)
    {
        // ---- a rectangular / asymmetric PU: the sum of its aligned squares' entries (largest squares first: 32x32 where one fits on a 32-grid)
        if (g_exp == 2 || g_time == 2)
            return refMotionEstimate(this, ref, mvmin, mvmax, qmvp, numCandidates, mvc, merange, outQMv, maxSlices, srcReferencePlane);
        const int lambda20 = (int)m_cost[1024] - (int)m_cost[0];
        const Pair* pr = pair_of(u->srcPic, u->version, ref->reconPic, lambda20);
        const intptr_t off = ref->reconPic->m_cuOffsetY[ctuAddr] + ref->reconPic->m_buOffsetY[absPartIdx];
        if (!pr || off != (intptr_t)u->y * ref->lumaStride + u->x || ((u->y + u->h - 1) >> 6) >= __atomic_load_n(pr->view->ctuRowsReady, __ATOMIC_ACQUIRE))
        {
            g_count[shard()].unserved.fetch_add(1, std::memory_order_relaxed);
            return refMotionEstimate(this, ref, mvmin, mvmax, qmvp, numCandidates, mvc, merange, outQMv, maxSlices, srcReferencePlane);
        }
        RectCtx& rc = t_rect;
        rc.n = 0;
        rc.stride = ref->lumaStride;
        rc.span = (size_t)(WIN - 1) * rc.stride + WIN;
        const pixel* puRef = ref->fpelPlane[0] + off;
        bool ok = true;
        uint8_t done[4][4] = {};                       // 16x16 cells of the PU
        for (int cy16 = 0; cy16 < u->h / 16 && ok; cy16++)
            for (int cx16 = 0; cx16 < u->w / 16 && ok; cx16++)
            {
                if (done[cy16][cx16]) continue;
                const int bx = u->x + 16 * cx16, by = u->y + 16 * cy16;
                int lvl = 1;
                if (!((bx | by) & 31) && 16 * cx16 + 32 <= u->w && 16 * cy16 + 32 <= u->h)
                {
                    lvl = 2;
                    done[cy16][cx16 + 1] = done[cy16 + 1][cx16] = done[cy16 + 1][cx16 + 1] = 1;
                }
                const x265hip_sadsurf_level* lv = &pr->view->level[lvl];
                const int sh = 3 + lvl, qx = bx >> sh, qy = by >> sh, cr = by >> 6;
                if (!lv->origin || qx >= lv->blocksX || qy >= lv->blocksY || rc.n >= 6) { ok = false; break; }
                const size_t k = (size_t)(qy - (cr << (3 - lvl))) * lv->blocksX + qx;
                const int16_t* org = (const int16_t*)((const char*)lv->origin + (size_t)cr * pr->view->ctuRowPitch) + 2 * k;
                RectPart& q = rc.part[rc.n++];
                q.entryBytes = lv->entryBytes;
                q.tab = (const char*)lv->table + (size_t)cr * pr->view->ctuRowPitch + k * WIN * WIN * lv->entryBytes;
                // reference position of this square's window entry (0, 0), shifted back to the PU's corner: candidate pointers are PU pointers
                q.winBase = puRef + (intptr_t)org[1] * rc.stride + org[0];
                // (the square sits at (16 cx16, 16 cy16) inside the PU both in the source and in the reference, so the offsets cancel)
            }
        if (!ok || !rc.n)
            return refMotionEstimate(this, ref, mvmin, mvmax, qmvp, numCandidates, mvc, merange, outQMv, maxSlices, srcReferencePlane);
        rc.fenc = fencPUYuv.m_buf[0];
        rc.hit = rc.miss = 0;
        const pixelcmp_t s1 = sad; const pixelcmp_x3_t s3 = sad_x3; const pixelcmp_x4_t s4 = sad_x4;
        if (!install_rect_for(this, u->w, u->h))
            return refMotionEstimate(this, ref, mvmin, mvmax, qmvp, numCandidates, mvc, merange, outQMv, maxSlices, srcReferencePlane);
        const int r = refMotionEstimate(this, ref, mvmin, mvmax, qmvp, numCandidates, mvc, merange, outQMv, maxSlices, srcReferencePlane);
        sad = s1; sad_x3 = s3; sad_x4 = s4;
        rc.fenc = NULL;
        g_rectSearches.fetch_add(1, std::memory_order_relaxed);
        g_rectHit.fetch_add(rc.hit, std::memory_order_relaxed);
        g_rectMiss.fetch_add(rc.miss, std::memory_order_relaxed);
        return r;
    }
    Ctx& c = t_ctx;
    const int level = u->w == 8 ? 0 : u->w == 16 ? 1 : u->w == 32 ? 2 : 3;
    int entryBytes = 0;
    const uint32_t* subTab = NULL;
    if (g_exp != 2)
    {
        // 20 x lambda out of this search's own vector-cost table: m_cost[i] = lambda * (2 log2(i + 1) + 0.718) (bitcost.cpp:48-70), log2(1025) = 10.0
        const int lambda20 = (int)m_cost[1024] - (int)m_cost[0];
        const Pair* pr = pair_of(u->srcPic, u->version, ref->reconPic, lambda20);
        const x265hip_sadsurf_level* lv = pr ? &pr->view->level[level] : NULL;
        if (!lv || !lv->origin || (u->y >> 6) >= __atomic_load_n(pr->view->ctuRowsReady, __ATOMIC_ACQUIRE))
        {
            g_count[shard()].unserved.fetch_add(1, std::memory_order_relaxed);
            return refMotionEstimate(this, ref, mvmin, mvmax, qmvp, numCandidates, mvc, merange, outQMv, maxSlices, srcReferencePlane);
        }
        // the position the reference will search from (motion.cpp:752-756) must be the one the source block was verified at
        const intptr_t off = ref->reconPic->m_cuOffsetY[ctuAddr] + ref->reconPic->m_buOffsetY[absPartIdx];
        if (off != (intptr_t)u->y * ref->lumaStride + u->x || (u->x >> (3 + level)) >= lv->blocksX || (u->y >> (3 + level)) >= lv->blocksY)
            return refMotionEstimate(this, ref, mvmin, mvmax, qmvp, numCandidates, mvc, merange, outQMv, maxSlices, srcReferencePlane);
        // x265hip_sadsurf_level: results are laid out per row of 64 picture lines
        const int sh = 3 + level, by = u->y >> sh, cr = u->y >> 6;
        const size_t k = (size_t)(by - (cr << (3 - level))) * lv->blocksX + (u->x >> sh);
        const int16_t* org = (const int16_t*)((const char*)lv->origin + (size_t)cr * pr->view->ctuRowPitch) + 2 * k;
        const int ox = org[0], oy = org[1];
        if (ox == -32768)
        {
            // an 8x8 block without a window (its 16x16 parent does not lie inside the picture)
            g_count[shard()].unserved.fetch_add(1, std::memory_order_relaxed);
            return refMotionEstimate(this, ref, mvmin, mvmax, qmvp, numCandidates, mvc, merange, outQMv, maxSlices, srcReferencePlane);
        }
        entryBytes = lv->entryBytes;
        c.tab = (const char*)lv->table + (size_t)cr * pr->view->ctuRowPitch + k * WIN * WIN * entryBytes;
        if (lv->subpel && level >= 1)
            subTab = (const uint32_t*)((const char*)lv->subpel + (size_t)cr * pr->view->ctuRowPitch) + k * X265HIP_SADSURF_SUBPEL;
        c.stride = ref->lumaStride;
        c.winBase = ref->fpelPlane[0] + off + (intptr_t)oy * c.stride + ox;
        c.span = (size_t)(WIN - 1) * c.stride + WIN;
        c.x = u->x; c.y = u->y; c.w = u->w; c.ox = ox; c.oy = oy;
        // the block's entries were written by the device a moment ago: bring them in while the search sets itself up
        for (int i = 0; i < WIN * WIN * entryBytes; i += 64)
            __builtin_prefetch((const char*)c.tab + i);
    }
    else
        c.stride = ref->lumaStride;
    c.fenc = fencPUYuv.m_buf[0];
    c.hit = c.miss = 0;
    SubCtx& sc = t_sub;
    sc.tab = NULL;
    const pixelcmp_t sSatd = satd;
    if (g_exp != 2 && g_time != 2 && subTab)
    {
        sc.tab = subTab; sc.fenc = c.fenc; sc.stride = c.stride; sc.n = u->w;
        sc.centre = c.winBase + (intptr_t)(WIN / 2) * c.stride + WIN / 2;
        sc.dst = NULL; sc.hit = sc.miss = 0;
        switch (level)
        {
        case 1: satd = satd_lookup<LUMA_16x16>; break;
        case 2: satd = satd_lookup<LUMA_32x32>; break;
        default: satd = satd_lookup<LUMA_64x64>; break;
        }
    }
    const pixelcmp_t s1 = sad; const pixelcmp_x3_t s3 = sad_x3; const pixelcmp_x4_t s4 = sad_x4;
    if (g_time != 2)
        switch (level)
        {
        case 0: install<LUMA_8x8>(this, entryBytes); break;
        case 1: install<LUMA_16x16>(this, entryBytes); break;
        case 2: install<LUMA_32x32>(this, entryBytes); break;
        default: install<LUMA_64x64>(this, entryBytes); break;
        }
    const uint64_t t0 = g_time ? __builtin_ia32_rdtsc() : 0;
    const int r = refMotionEstimate(this, ref, mvmin, mvmax, qmvp, numCandidates, mvc, merange, outQMv, maxSlices, srcReferencePlane);
    if (g_time)
    {
        g_cycles[level].fetch_add(__builtin_ia32_rdtsc() - t0, std::memory_order_relaxed);
        g_timed[level].fetch_add(1, std::memory_order_relaxed);
    }
    sad = s1; sad_x3 = s3; sad_x4 = s4;
    satd = sSatd;
    if (sc.tab)
    {
        sc.tab = NULL;
        if (sc.hit | sc.miss) { g_subHit.fetch_add(sc.hit, std::memory_order_relaxed); g_subMiss.fetch_add(sc.miss, std::memory_order_relaxed); }
    }
    c.fenc = NULL;
    if (g_subpelHit && g_exp != 2)
    {
        const int dx = outQMv.x - 4 * (c.ox + WIN / 2), dy = outQMv.y - 4 * (c.oy + WIN / 2);
        g_spAll[level].fetch_add(1, std::memory_order_relaxed);
        if (dx >= -3 && dx <= 3 && dy >= -3 && dy <= 3) g_spHit[level].fetch_add(1, std::memory_order_relaxed);
        // (how far from the centre do the others end: what a table of +-7 / +-11 / +-19 quarter-pels would cover)
        const int far = (dx < 0 ? -dx : dx) > (dy < 0 ? -dy : dy) ? (dx < 0 ? -dx : dx) : (dy < 0 ? -dy : dy);
        g_spDist[level][far <= 3 ? 0 : far <= 7 ? 1 : far <= 11 ? 2 : far <= 19 ? 3 : 4].fetch_add(1, std::memory_order_relaxed);
    }
    // counters: per thread, flushed to the shared ones now and then (an atomic per search would be felt)
    t_hit += c.hit; t_miss += c.miss;
    (void)&t_flushAtExit;            // constructed on this thread's first search, destroyed (and flushed) when the thread ends
    if (((++t_searches) & 255) == 0)
    {
        Counter& k = g_count[shard()];
        k.hit.fetch_add(t_hit, std::memory_order_relaxed);
        k.miss.fetch_add(t_miss, std::memory_order_relaxed);
        t_hit = t_miss = 0;
    }
    return r;
}

} // namespace X265_NS
