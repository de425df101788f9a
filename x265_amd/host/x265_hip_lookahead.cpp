// x265_hip_lookahead.cpp — the second translation unit a maintainer adds to an x265 build: it binds the lookahead's own batching seam,
// CostEstimateGroup::finishBatch / estimateFrameCost (reference source/encoder/slicetype.cpp:3041-3048, :3115-3214), to the device-resident
// lookahead session of libx265hip.so (include/x265hip.h, x265hip_la_*).
//
// x265 already groups lookahead work there: slicetypeAnalyse queues every missing motion search and frame cost of the mini-GOP candidates
// (up to 512 estimates, :1942-2008) and then runs them over the thread pool, one estimateFrameCost per worker.  Here the whole queue is ONE
// call: the session holds each queued frame's half-resolution planes, intra costs and AQ factors in HBM, searches every (frame, reference)
// pair of the batch in one launch, and returns exactly what estimateCUCost leaves in the Lowres arrays (lowresMvs, lowresMvCosts,
// lowresCosts, rowSatds, costEst, costEstAq, intraMbs) — so everything downstream (scenecut, slicetypePath, cuTree, rate control) runs
// unchanged on identical numbers and the bitstream is byte-identical.
//
// How it is linked (oracle/Makefile): in a source tree a maintainer would add two `if (x265hip_lookahead(...)) return;`
// lines.  Against the read-only reference the same effect is obtained at link time: the two symbols are weakened in slicetype.o
// (objcopy --weaken-symbol) so the definitions below win, and the reference's original bodies stay reachable as CostEstimateGroupRef::*
// (slicetype.cpp compiled a second time with -DCostEstimateGroup=CostEstimateGroupRef, every other symbol of that object localised).  This
// file restates none of the reference's control flow: when the cache test says "not computed yet" it computes on the GPU and fills the
// Lowres arrays, then ALWAYS finishes through the reference's own estimateFrameCost, which finds its cache filled (:3121-3122).
//
// Without a usable device, or with X265HIP_LOOKAHEAD=0, or for configurations the device pass does not cover (HME, aq-motion), every call
// goes to the reference's original code.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <mutex>
#include <pthread.h>
#include <unistd.h>
#include <vector>

#define protected public
#define private public
#include "common.h"
#include "frame.h"
#include "lowres.h"
#include "slicetype.h"
#undef protected
#undef private

#include "x265hip.h"
#include "x265_hip_debug.h"

namespace X265_NS {

// the reference's original bodies (slicetype.cpp compiled as CostEstimateGroupRef; `this` is the first argument in the Itanium C++ ABI)
#define X265HIP_STR2(x) #x
#define X265HIP_STR(x) X265HIP_STR2(x)
#define X265HIP_NSLEN sizeof(X265HIP_STR(X265_NS)) - 1
// mangled: _ZN <len><ns> 20CostEstimateGroupRef ...  — the namespace is `x265` in every build of oracle/Makefile (checked below)
extern int64_t refEstimateFrameCost(CostEstimateGroup* self, LookaheadTLD& tld, int p0, int p1, int b, bool bIntraPenalty)
    asm("_ZN4x26520CostEstimateGroupRef17estimateFrameCostERNS_12LookaheadTLDEiiib");
extern void refFinishBatch(CostEstimateGroup* self) asm("_ZN4x26520CostEstimateGroupRef11finishBatchEv");
extern void refLookaheadDestroy(Lookahead* self) asm("_ZN4x26512LookaheadRef7destroyEv");
static_assert(sizeof(X265HIP_STR(X265_NS)) == sizeof("x265"), "the asm labels above assume -DX265_NS=x265");

namespace {

struct SlotEntry { const Lowres* lowres; int frameNum; uint64_t stamp; };

struct Session
{
    const Lookahead* owner = NULL;
    x265hip_la* la = NULL;
    std::vector<SlotEntry> slots;
    uint64_t stamp = 0;
    uint64_t uploads = 0;
    uint64_t waitNs = 0;          // wall time the lookahead thread spent inside compute() (descriptor build + device pass + write-back)
    uint64_t lastUse = 0;
    std::mutex m;                 // held across a compute(): the encoders of an ABR ladder have a session each and do not wait for one another
    int place = 0;
};

std::mutex g_lock;               // the session table (who owns which entry, eviction) and the totals below; order: g_lock, then a session's m
const int kMaxSessions = 4;        // encoders of one process that run at the same time (an ABR ladder); more than that take turns
Session g_sessions[kMaxSessions];
uint64_t g_useClock = 0;
int g_state = 0;                 // 0 = not decided, 1 = on, -1 = off
bool g_verbose = false, g_trace = false;
bool g_ahead = true;             // X265HIP_LOOKAHEAD_AHEAD=0: no searches ahead of the reference's requests (for A/B measurements)

// totals of the sessions that were closed with their encoders (Lookahead::destroy below)
uint64_t g_pastBatches = 0, g_pastEstimates = 0, g_pastSearches = 0, g_pastUploads = 0, g_pastWaitNs = 0;
uint64_t g_pastAhead[4] = { 0, 0, 0, 0 };      // x265hip_la_stats_ahead: launched ahead, used, search launches, pairs in them

void retire_session(Session& s)
{
    if (!s.la)
        return;
    uint64_t batches = 0, estimates = 0, searches = 0;
    x265hip_la_stats(s.la, &batches, &estimates, &searches);
    uint64_t ah[4] = { 0, 0, 0, 0 };
    x265hip_la_stats_ahead(s.la, &ah[0], &ah[1], &ah[2], &ah[3]);
    for (int i = 0; i < 4; i++) g_pastAhead[i] += ah[i];
    g_pastBatches += batches; g_pastEstimates += estimates; g_pastSearches += searches; g_pastUploads += s.uploads; g_pastWaitNs += s.waitNs;
    x265hip_la_destroy(s.la);
    s.la = NULL;
    s.owner = NULL;
    s.slots.clear();
    s.uploads = 0;
    s.waitNs = 0;
}

void report()
{
    uint64_t batches = g_pastBatches, estimates = g_pastEstimates, searches = g_pastSearches, uploads = g_pastUploads, waitNs = g_pastWaitNs;
    uint64_t ahead[4] = { g_pastAhead[0], g_pastAhead[1], g_pastAhead[2], g_pastAhead[3] };
    bool any = batches != 0;
    for (Session& s : g_sessions)
        if (s.la)
        {
            uint64_t b = 0, e = 0, m = 0, ah[4] = { 0, 0, 0, 0 };
            x265hip_la_stats(s.la, &b, &e, &m);
            x265hip_la_stats_ahead(s.la, &ah[0], &ah[1], &ah[2], &ah[3]);
            batches += b; estimates += e; searches += m; uploads += s.uploads; waitNs += s.waitNs;
            for (int i = 0; i < 4; i++) ahead[i] += ah[i];
            any = true;
        }
    if (!any)
        return;
    fprintf(stderr, "x265hip: lookahead: %llu frame-cost estimates (%llu motion-search passes over %llu lowres frames) served by the GPU in %llu batches, %.3f s inside the seam\n",
            (unsigned long long)estimates, (unsigned long long)searches, (unsigned long long)uploads, (unsigned long long)batches,
            waitNs * 1e-9);
    fprintf(stderr, "x265hip: lookahead: %llu searches launched ahead of their request, %llu of them used; %llu search launches of %.1f (frame, reference) pairs on average\n",
            (unsigned long long)ahead[0], (unsigned long long)ahead[1], (unsigned long long)ahead[2], ahead[2] ? (double)ahead[3] / ahead[2] : 0.0);
}

bool enabled()
{
    if (!g_state)
    {
        const char* env = getenv("X265HIP_LOOKAHEAD");
        const char* all = getenv("X265HIP");
        g_verbose = getenv("X265HIP_VERBOSE") != NULL;
        g_trace = getenv("X265HIP_DEBUG_TRACE") != NULL;
        const char* ahead = getenv("X265HIP_LOOKAHEAD_AHEAD");
        g_ahead = !(ahead && !strcmp(ahead, "0"));
        if ((env && !strcmp(env, "0")) || (all && !strcmp(all, "0")) || x265hip_device_count() < 1)
            g_state = -1;
        else
        {
            g_state = 1;
            if (g_verbose)
                atexit(report);
        }
    }
    return g_state > 0;
}

// a device call of the estimate path failed: thrown up to compute_guarded(), which hands the estimates back to the reference's own functions
struct DeviceFailure { const char* what; };
[[noreturn]] void die(const char* what) { throw DeviceFailure{ what }; }

bool covered(const Lookahead& l, const Lowres* fenc)
{
    const x265_param* p = l.m_param;
    return !p->bEnableHME && !p->bAQMotion && fenc->buffer[0] && (fenc->buffer[1] - fenc->buffer[0]) < (1 << 24) && p->bframes + 2 <= 18;
}

Session& session_for(const Lookahead& l, const Lowres* f)
{
    Session* pick = NULL;
    for (Session& c : g_sessions)
        if (c.la && c.owner == &l)
        {
            c.lastUse = ++g_useClock;
            return c;
        }
    for (Session& c : g_sessions)
        if (!c.la) { pick = &c; break; }
    if (!pick)
    {
        pick = &g_sessions[0];                  // more live encoders than sessions: the least recently used one gives way (correct, only slower)
        for (Session& c : g_sessions)
            if (c.lastUse < pick->lastUse) pick = &c;
        std::lock_guard<std::mutex> busy(pick->m);      // its own encoder may be inside a batch
        retire_session(*pick);
    }
    Session& s = *pick;
    s.lastUse = ++g_useClock;
    x265hip_la_config c;
    memset(&c, 0, sizeof(c));
    c.depth = X265_DEPTH;
    c.width = f->width;
    c.lines = f->lines;
    c.stride = f->lumaStride;
    c.planeElems = f->buffer[1] - f->buffer[0];
    c.padOffset = f->lowresPlane[0] - f->buffer[0];
    c.widthInCU = l.m_8x8Width;
    c.heightInCU = l.m_8x8Height;
    c.maxDist = l.m_param->bframes + 2;
    c.numSlots = l.m_param->lookaheadDepth + l.m_param->bframes + 10;
    x265hip_debug_mark("create: lookahead session");
    // several places (X265HIP_DEVICES): the sessions of the process take them in turn, the LAST place first — a lone encoder's lookahead then does not
    // sit on place 0 beside the first mirror, the first source picture and the first job server
    static int nextPlace = -1;                           // under g_lock (compute() holds it while it asks for the session)
    const int places = x265hip_places_configured();
    if (places > 1)
    {
        if (nextPlace < 0) nextPlace = places - 1;
        s.la = x265hip_la_create_at(nextPlace, &c);
        s.place = nextPlace;
        nextPlace = (nextPlace + places - 1) % places;
    }
    else
    {
        s.la = x265hip_la_create(&c);
        s.place = 0;
    }
    x265hip_debug_mark("created: lookahead session");
    if (!s.la)
        die("session");
    s.owner = &l;
    s.slots.assign(c.numSlots, SlotEntry{ NULL, 0, 0 });
    s.stamp = 0;
    return s;
}

int slot_of(Session& s, const Lookahead& l, const Lowres* f)
{
    int victim = -1;
    for (size_t i = 0; i < s.slots.size(); i++)
    {
        SlotEntry& e = s.slots[i];
        if (e.lowres == f && e.frameNum == f->frameNum)
        {
            e.stamp = s.stamp;
            return (int)i;
        }
        if (e.lowres == f || !e.lowres)
        {
            if (victim < 0 || s.slots[victim].lowres)
                victim = (int)i;            // the frame's old picture, or a free slot
        }
    }
    if (victim < 0)
    {
        for (size_t i = 0; i < s.slots.size(); i++)
            if (s.slots[i].stamp < s.stamp && (victim < 0 || s.slots[i].stamp < s.slots[victim].stamp))
                victim = (int)i;
        if (victim < 0)
            die("more frames in one batch than the session has slots");
    }
    const int32_t* invq = f->invQscaleFactor ? (l.m_param->rc.qgSize == 8 ? f->invQscaleFactor8x8 : f->invQscaleFactor) : NULL;
    if (x265hip_la_set_frame(s.la, victim, f->buffer[0], f->intraCost, invq))
        die("frame upload");
    s.slots[victim] = SlotEntry{ f, f->frameNum, s.stamp };
    s.uploads++;
    return victim;
}

struct Job { int p0, p1, b; };

// Searches the reference has not asked for yet but, by the shape of its own control flow, will: slicetypeAnalyse batches the list-0 searches of
// frames 2 .. numFrames-1 and the list-1 search at the SAME distance when that frame already exists (slicetype.cpp:1942-1968); every other search —
// list 1 of a frame whose partner arrived later, list 0 of the first and of the newest frame — is asked for one estimate at a time from
// slicetypePathCost / scenecut / slicetypeDecide (singleCost -> estimateFrameCost), which runs cooperative slices (:3141-3166).  Each of those is
// a function of the two frames, of the P / B flavour and of the slice geometry only (x265hip_la_search), so it can ride along with a launch that
// happens anyway and be picked up when the request comes.  Frames are taken from the group's own NULL-terminated array (slicetype.cpp:1408-1411).
const int kMaxAhead = 160;
void plan_ahead(Session& s, CostEstimateGroup& g, const Job* jobs, int n, const std::vector<x265hip_la_estimate>& est, std::vector<x265hip_la_search>& out)
{
    const Lookahead& l = g.m_lookahead;
    const x265_param* param = l.m_param;
    int lo = jobs[0].p0, hi = jobs[0].p1;
    for (int i = 0; i < n; i++)
    {
        lo = X265_MIN(lo, jobs[i].p0);
        hi = X265_MAX(hi, X265_MAX(jobs[i].p1, jobs[i].b));
    }
    const int bound = X265_LOOKAHEAD_MAX + X265_BFRAME_MAX + 2;          // the array has bound + 2 entries, zero-filled behind the last frame
    while (hi + 1 <= bound && g.m_frames[hi + 1])
        hi++;
    const int coopSlices = X265_MAX(1, l.m_numCoopSlices), coopRows = coopSlices > 1 ? l.m_numRowsPerSlice : l.m_8x8Height;
    auto real = [&](int b, int list, int dist) {
        for (int i = 0; i < n; i++)
            if (jobs[i].b == b && (list ? (est[i].search1 && est[i].dist1 == dist) : (est[i].search0 && est[i].dist0 == dist)))
                return true;
        return false;
    };
    for (int b = X265_MAX(lo, 1); b <= hi && (int)out.size() < kMaxAhead; b++)
    {
        Lowres* fenc = g.m_frames[b];
        if (!covered(l, fenc))
            return;
        int slotB = -1;
        for (int dist = 1; dist <= param->bframes + 1; dist++)
        {
            // list 1 (B flavour only): the partner frame exists now
            if (dist <= param->bframes && b + dist <= hi && fenc->lowresMvs[1][dist][0].x == 0x7FFF && !real(b, 1, dist))
            {
                if (slotB < 0) slotB = slot_of(s, l, fenc);
                if (!x265hip_la_has_ahead(s.la, slotB, 1, dist, 1, coopRows, coopSlices))
                {
                    x265hip_la_search a = { slotB, slot_of(s, l, g.m_frames[b + dist]), 1, dist, 1, -1, coopRows, coopSlices };
                    out.push_back(a);
                }
            }
            // list 0: with batched motion searches only the frames the batch leaves out (the first and the newest); both flavours otherwise
            const bool batched = l.m_bBatchMotionSearch && b != 1 && b != hi;
            if (b - dist >= lo && g.m_frames[b - dist] && !batched && fenc->lowresMvs[0][dist][0].x == 0x7FFF && !real(b, 0, dist))
            {
                if (slotB < 0) slotB = slot_of(s, l, fenc);
                const int slotR = slot_of(s, l, g.m_frames[b - dist]);
                int weightedId = -2;
                for (int bidir = 0; bidir <= (l.m_bBatchMotionSearch ? 0 : 1); bidir++)
                {
                    if (x265hip_la_has_ahead(s.la, slotB, 0, dist, bidir, coopRows, coopSlices))
                        continue;
                    if (weightedId == -2)
                    {
                        weightedId = -1;
                        if (param->bEnableWeightedPred)
                        {
                            x265hip_weight_param chosen;
                            int isWeighted = 0;
                            Lowres* ref0 = g.m_frames[b - dist];
                            if (x265hip_la_weights_analyse(s.la, slotB, slotR, fenc->wp_ssd[0], fenc->wp_sum[0], ref0->wp_ssd[0], ref0->wp_sum[0], &chosen, &isWeighted, &weightedId))
                                die("weights analysis (ahead)");
                        }
                    }
                    x265hip_la_search a = { slotB, slotR, 0, dist, bidir, weightedId, coopRows, coopSlices };
                    out.push_back(a);
                }
            }
        }
    }
}

// the missing results of `jobs` on the device, left in the Lowres arrays exactly as estimateCUCost leaves them (slicetype.cpp:3129-3207)
void compute(CostEstimateGroup& g, const Job* jobs, int n, bool coop)
{
    const Lookahead& l = g.m_lookahead;
    const x265_param* param = l.m_param;
    std::unique_lock<std::mutex> table(g_lock);
    const auto t0 = std::chrono::steady_clock::now();
    Session& s = session_for(l, g.m_frames[jobs[0].b]);
    std::lock_guard<std::mutex> guard(s.m);              // free: a session is used by its own encoder's lookahead only, and evictions hold g_lock
    table.unlock();
    s.stamp++;
    std::vector<x265hip_la_estimate> est(n);
    for (int i = 0; i < n; i++)
    {
        const Job& j = jobs[i];
        Lowres* fenc = g.m_frames[j.b];
        Lowres* ref0 = g.m_frames[j.p0];
        Lowres* ref1 = g.m_frames[j.p1];
        x265hip_la_estimate& e = est[i];
        memset(&e, 0, sizeof(e));
        e.b = slot_of(s, l, fenc);
        e.p0 = slot_of(s, l, ref0);
        e.p1 = j.p1 == j.b ? e.b : slot_of(s, l, ref1);
        e.dist0 = j.b - j.p0;
        e.dist1 = j.p1 - j.b;
        e.search0 = fenc->lowresMvs[0][e.dist0][0].x == 0x7FFF;
        e.search1 = j.p1 > j.b && fenc->lowresMvs[1][e.dist1][0].x == 0x7FFF;
        e.weightedId = -1;
        fenc->weightedRef[e.dist0].isWeighted = false;
        const int estRows = coop ? l.m_numRowsPerSlice : l.m_8x8Height, estSlices = coop ? l.m_numCoopSlices : 1;
        if (param->bEnableWeightedPred && e.search0 && !x265hip_la_has_ahead(s.la, e.b, 0, e.dist0, j.p1 != j.b, estRows, estSlices))
        {
            x265hip_weight_param chosen;
            int isWeighted = 0;
            if (x265hip_la_weights_analyse(s.la, e.b, e.p0, fenc->wp_ssd[0], fenc->wp_sum[0], ref0->wp_ssd[0], ref0->wp_sum[0], &chosen, &isWeighted, &e.weightedId))
                die("weights analysis");
            // the weighted planes only ever serve this estimate's list-0 search (slicetype.cpp:3222, :3269), which runs on the device:
            // the host-side ReferencePlanes stay unweighted.  weightedCostDelta = minscore / origscore is an integer division of a
            // smaller by a larger unsigned (:945): 0, the value Lowres::init left there.
        }
        // lists that are not searched must already be in the session (they are, unless another path produced them)
        for (int list = 0; list < (j.p1 > j.b ? 2 : 1); list++)
        {
            const int dist = list ? e.dist1 : e.dist0;
            if (!(list ? e.search1 : e.search0) && !x265hip_la_has_vectors(s.la, e.b, list, dist))
                if (x265hip_la_put_vectors(s.la, e.b, list, dist, (const int32_t*)fenc->lowresMvs[list][dist], fenc->lowresMvCosts[list][dist]))
                    die("vector upload");
        }
        e.mvs0 = (int32_t*)fenc->lowresMvs[0][e.dist0];
        e.mvCosts0 = fenc->lowresMvCosts[0][e.dist0];
        e.mvs1 = (int32_t*)fenc->lowresMvs[1][e.dist1];
        e.mvCosts1 = fenc->lowresMvCosts[1][e.dist1];
        e.lowresCosts = fenc->lowresCosts[e.dist0][e.dist1];
        e.rowSatds = fenc->rowSatds[e.dist0][e.dist1];
    }
    // batch mode never uses cooperative slices; a single estimate does when the pool allows and it has something to search or is a B estimate
    int rows = l.m_8x8Height, slices = 1;
    if (coop)
    {
        rows = l.m_numRowsPerSlice;
        slices = l.m_numCoopSlices;
    }
    std::vector<x265hip_la_search> ahead;
    if (g_ahead)
        plan_ahead(s, g, jobs, n, est, ahead);
    if (x265hip_la_estimate_batch_ahead(s.la, est.data(), n, rows, slices, ahead.data(), (int)ahead.size()))
        die("estimate batch");
    for (int i = 0; i < n; i++)
    {
        const Job& j = jobs[i];
        Lowres* fenc = g.m_frames[j.b];
        const x265hip_la_estimate& e = est[i];
        int64_t score = e.costEst;
        if (j.b != j.p1)
            score = score * 100 / (130 + param->bFrameBias);
        else
            fenc->intraMbs[e.dist0] += e.intraMbs;
        fenc->costEst[e.dist0][e.dist1] = score;
        fenc->costEstAq[e.dist0][e.dist1] = e.costEstAq;
        if (g_trace)
        {
            // X265HIP_DEBUG_TRACE=1: one line per estimate — what was asked and what came back (sums of the arrays), for diffing two runs
            auto sum32 = [](const int32_t* p, int n) { uint64_t h = 1469598103934665603ull; for (int k = 0; k < n; k++) { h ^= (uint32_t)p[k]; h *= 1099511628211ull; } return h; };
            const int ncu = l.m_8x8Width * l.m_8x8Height;
            fprintf(stderr, "x265hip-trace: w %d b %d p0 %d p1 %d search %d%d weighted %d coop %d -> cost %lld aq %lld intra %d mvs0 %016llx mvc0 %016llx mvs1 %016llx mvc1 %016llx lc %016llx rows %016llx intraMbsAcc %d in: intraCost %016llx planes %016llx ref %016llx\n",
                    fenc->width, fenc->frameNum, g.m_frames[j.p0]->frameNum, g.m_frames[j.p1]->frameNum, e.search0, e.search1, e.weightedId, (int)coop,
                    (long long)e.costEst, (long long)e.costEstAq, e.intraMbs, (unsigned long long)sum32((const int32_t*)fenc->lowresMvs[0][e.dist0], 2 * ncu),
                    (unsigned long long)sum32(fenc->lowresMvCosts[0][e.dist0], ncu),
                    (unsigned long long)(j.p1 > j.b ? sum32((const int32_t*)fenc->lowresMvs[1][e.dist1], 2 * ncu) : 0),
                    (unsigned long long)(j.p1 > j.b ? sum32(fenc->lowresMvCosts[1][e.dist1], ncu) : 0),
                    (unsigned long long)sum32((const int32_t*)fenc->lowresCosts[e.dist0][e.dist1], ncu / 2),
                    (unsigned long long)sum32(fenc->rowSatds[e.dist0][e.dist1], l.m_8x8Height), fenc->intraMbs[e.dist0],
                    (unsigned long long)sum32(fenc->intraCost, ncu),
                    (unsigned long long)sum32((const int32_t*)fenc->buffer[0], (int)((fenc->buffer[1] - fenc->buffer[0]) * sizeof(pixel) / 4)),
                    (unsigned long long)sum32((const int32_t*)g.m_frames[j.p0]->buffer[0], (int)((fenc->buffer[1] - fenc->buffer[0]) * sizeof(pixel) / 4)));
        }
    }
    s.waitNs += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
}

inline bool cached(const Lowres* fenc, int p0, int p1, int b)
{
    return fenc->costEst[b - p0][p1 - b] >= 0 && fenc->rowSatds[b - p0][p1 - b][0] != -1;
}

// compute(), or — when a device call fails on the way — nothing: the seam switches itself off, whatever the failed batch may have half-written
// into the Lowres arrays is marked "not searched / not estimated" again, and the reference's own estimateFrameCost computes it (SURVEY §8b "Errors")
void compute_guarded(CostEstimateGroup& g, const Job* jobs, int n, bool coop)
{
    struct Before { bool search0, search1; };
    std::vector<Before> before(n);
    for (int i = 0; i < n; i++)
    {
        const Lowres* fenc = g.m_frames[jobs[i].b];
        before[i].search0 = fenc->lowresMvs[0][jobs[i].b - jobs[i].p0][0].x == 0x7FFF;
        before[i].search1 = jobs[i].p1 > jobs[i].b && fenc->lowresMvs[1][jobs[i].p1 - jobs[i].b][0].x == 0x7FFF;
    }
    try
    {
        compute(g, jobs, n, coop);
    }
    catch (const DeviceFailure& f)
    {
        g_state = -1;
        for (int i = 0; i < n; i++)
        {
            Lowres* fenc = g.m_frames[jobs[i].b];
            const int d0 = jobs[i].b - jobs[i].p0, d1 = jobs[i].p1 - jobs[i].b;
            if (before[i].search0) fenc->lowresMvs[0][d0][0].x = 0x7FFF;
            if (before[i].search1) fenc->lowresMvs[1][d1][0].x = 0x7FFF;
            fenc->costEst[d0][d1] = -1;
            fenc->rowSatds[d0][d1][0] = -1;
        }
        x265hip_device_failure("lookahead", f.what);
    }
}

} // namespace

// The encoder is being closed: its Lowres objects and its Lookahead are about to be freed, and a later encoder of the same process may get the
// same addresses back from malloc — the session's slots are keyed by them, so the session must not outlive them (tests/support/two_encoders.cpp).
void Lookahead::destroy()
{
    x265hip_debug_mark("Lookahead::destroy (the encoder is being closed)");
    if (g_state > 0)
    {
        std::lock_guard<std::mutex> guard(g_lock);
        for (Session& c : g_sessions)
            if (c.owner == this)
            {
                std::lock_guard<std::mutex> busy(c.m);
                retire_session(c);
            }
    }
    refLookaheadDestroy(this);
    x265hip_debug_mark("Lookahead::destroy returns");
}

// X265HIP_DEBUG_DELAY_US (see x265_hip_refplanes.cpp): the same diagnostic sleep on the lookahead's side, seams on or off — the reference's output
// depends on how fast its lookahead is relative to its input in a few corner cases (clips shorter than the lookahead, tiny pictures)
static void debug_delay()
{
    // X265HIP_DEBUG_LA_DELAY_US: the same on this side only — how sensitive is the encode to the lookahead's speed?
    static const int delayUs = getenv("X265HIP_DEBUG_LA_DELAY_US") ? atoi(getenv("X265HIP_DEBUG_LA_DELAY_US")) :
                               getenv("X265HIP_DEBUG_DELAY_US") ? atoi(getenv("X265HIP_DEBUG_DELAY_US")) : 0;
    if (delayUs > 0)
        usleep(delayUs);
}

// X265HIP_DEBUG_LA_TIMELINE=file: one line per call of the two seams — enter / exit (ns since the first call), thread, what was asked, whether it was
// cached.  The gaps between the lines are the lookahead's own host work (pre-lookahead, path search, CU-tree): where a lookahead-bound encode spends
// its time (tools/la_timeline.py)
struct Timeline
{
    FILE* f = NULL;
    std::chrono::steady_clock::time_point t0;
    std::mutex m;
    Timeline()
    {
        const char* path = getenv("X265HIP_DEBUG_LA_TIMELINE");
        if (path) { f = fopen(path, "w"); t0 = std::chrono::steady_clock::now(); }
    }
    long long now() const { return (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
    void line(long long a, long long z, const char* what, int p0, int p1, int b, int cached, int n)
    {
        std::lock_guard<std::mutex> g(m);
        fprintf(f, "%lld %lld %lx %s %d %d %d %d %d\n", a, z, (unsigned long)pthread_self(), what, p0, p1, b, cached, n);
    }
};
static Timeline& timeline() { static Timeline* t = new Timeline; return *t; }

int64_t CostEstimateGroup::estimateFrameCost(LookaheadTLD& tld, int p0, int p1, int b, bool bIntraPenalty)
{
    debug_delay();
    if (timeline().f)
    {
        const long long a = timeline().now();
        const int wasCached = cached(m_frames[b], p0, p1, b);
        if (!wasCached && enabled() && covered(m_lookahead, m_frames[b]))
        {
            Lowres* fenc = m_frames[b];
            const bool search0 = fenc->lowresMvs[0][b - p0][0].x == 0x7FFF;
            const bool search1 = p1 > b && fenc->lowresMvs[1][p1 - b][0].x == 0x7FFF;
            const bool coop = !m_batchMode && m_lookahead.m_numCoopSlices > 1 && (p1 > b || search0 || search1);
            const Job j = { p0, p1, b };
            compute_guarded(*this, &j, 1, coop);
        }
        const int64_t r = refEstimateFrameCost(this, tld, p0, p1, b, bIntraPenalty);
        timeline().line(a, timeline().now(), m_batchMode ? "estB" : "est", p0, p1, m_frames[b]->frameNum, wasCached, 1);
        return r;
    }
    Lowres* fenc = m_frames[b];
    if (!cached(fenc, p0, p1, b) && enabled() && covered(m_lookahead, fenc))
    {
        const bool search0 = fenc->lowresMvs[0][b - p0][0].x == 0x7FFF;
        const bool search1 = p1 > b && fenc->lowresMvs[1][p1 - b][0].x == 0x7FFF;
        const bool coop = !m_batchMode && m_lookahead.m_numCoopSlices > 1 && (p1 > b || search0 || search1);
        const Job j = { p0, p1, b };
        compute_guarded(*this, &j, 1, coop);
    }
    return refEstimateFrameCost(this, tld, p0, p1, b, bIntraPenalty);
}

// (The pre-lookahead — PreLookaheadGroup::processTasks, slicetype.cpp:1380-1402: Lowres::init, adaptive quantisation, the intra estimate, one frame per bonded
// worker — stays the reference's: rounds 3-5 carried a pass-through seam here that only timed it.  x265hip_lowres_init / x265hip_lowres_intra_estimate exist and
// are pinned (tests/test_hip_parity.py), but the frame's lowres planes, intra costs and modes are read by host code in four places afterwards (AQ, weightp,
// cuTree, the slice-type decision), so they would have to come back over PCIe for work that is 1.0 % of the bound encoder's CPU time
// (profiles/r06_v1_cpu_profile_bound_encoder.txt: frame_init_lowres_core 0.34 %, the lookahead's intra predictors and satd below that) on threads that are not
// on a frame encoder's critical path.  Row f1's other half is host by decision; the seam is gone.)

void CostEstimateGroup::finishBatch()
{
    debug_delay();
    const long long tlA = timeline().f ? timeline().now() : 0;
    struct TlExit { long long a; int n; ~TlExit() { if (timeline().f) timeline().line(a, timeline().now(), "batch", 0, 0, 0, 0, n); } } tlExit{ tlA, m_jobTotal };
    if (m_jobTotal > 0 && enabled() && covered(m_lookahead, m_frames[m_estimates[0].b]))
    {
        std::vector<Job> jobs;
        jobs.reserve(m_jobTotal);
        for (int i = 0; i < m_jobTotal; i++)
        {
            const Estimate& e = m_estimates[i];
            if (!cached(m_frames[e.b], e.p0, e.p1, e.b))
                jobs.push_back(Job{ e.p0, e.p1, e.b });
        }
        if (!jobs.empty())
            compute_guarded(*this, jobs.data(), (int)jobs.size(), false);
        // every estimate of the queue is now cached; the reference's own loop (below) only reads the scores back
    }
    refFinishBatch(this);
}

} // namespace X265_NS
