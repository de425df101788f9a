// x265_hip_saostats.cpp — SAO statistics as jobs of the CU-job service (split from x265_hip_cuserve.cpp in round 6).
#include "x265_hip_cuserve.h"

namespace X265_NS {

using namespace cusvc;

// ---- SAO statistics as jobs of the same service (round 5; include/x265hip.h x265hip_saojob) --------------------------------------------------------------
// SAO::calcSaoStatsCTU (reference source/encoder/sao.cpp:735-917) is called per plane from rdoSaoUnitCu (:1293-1305) for every CTU, two columns behind the
// deblocking of the same row (framefilter.cpp:440-500): 5 classes x 3 planes of per-sample classification = ~50 us of CPU per CTU (6.8 % of the bound
// encoder's CPU time in round 4's profile).  Its result is a function of the deblocked CTU (with the row above and the column to the left), the source CTU and
// the rectangles the reference measures: the seam below computes the rectangles exactly as the reference does, hands the two blocks to the device — one job
// for all planes when luma is asked for (the chroma planes are measured while this thread runs the luma offsets' RDO) — and adds the sums and counts it gets
// back where the reference's primitives add theirs.  8-bit builds; X265HIP_SAOSTATS=0 switches it off; X265HIP_VERIFY recomputes with the reference's body.
namespace {

std::atomic<int> g_saoState(0);  // 0 undecided, 1 on, -1 off (written by whichever thread decides or sees the device fail)
bool g_saoParts = false;         // X265HIP_SAOSTATS_PARTS=4: the luma plane goes as two jobs (upper / lower half).  Measured: 3 jobs per CTU 33.2 fps, 4 jobs 32.6, SAO on the host 30.9
struct alignas(64) SaoCounters { std::atomic<uint64_t> jobs, planes, hostPlanes, waits, waitCycles, ahead; };
SaoCounters g_saoCount[16];
// One CTU's statistics are up to four PARTS, each a job of one block on a slot of its own — the upper and lower half of the luma CTU, Cb, Cr — so that four
// workgroups measure at the same time (a part is a dependent chain of ~10 us on the device; sums and counts of the halves add up).  When slots are short the
// CTU goes as fewer parts, down to one.
struct SaoPart { int plane, slot; uint32_t seq; Service* svc; };
struct SaoJob
{
    bool active;
    const SAO* sao; int addr;
    const void* encData; int poc;        // the picture the blocks were read from: a row's SAO object serves every frame its FrameEncoder codes, (sao, addr) alone
                                         // would let a set orphaned in an earlier frame (ParallelFilter::processTasks hops between pool threads) be adopted
    int nparts; SaoPart part[4];
    bool consumed[3];                    // per plane
    bool wanted[3];                      // planes this job carries
};
// two per thread: the CTU whose statistics are being asked for, and the NEXT CTU of the row, submitted ahead (see SAO::calcSaoStatsCTU below)
__attribute__((tls_model("initial-exec"))) thread_local SaoJob t_saoSet[2];
bool g_saoAhead = true;              // X265HIP_SAOSTATS_AHEAD=0: no CTU is submitted ahead of its request

void sao_report()
{
    uint64_t jobs = 0, planes = 0, host = 0, w = 0, wc = 0, ah = 0;
    for (int i = 0; i < 16; i++) { jobs += g_saoCount[i].jobs; planes += g_saoCount[i].planes; host += g_saoCount[i].hostPlanes; w += g_saoCount[i].waits; wc += g_saoCount[i].waitCycles; ah += g_saoCount[i].ahead; }
    fprintf(stderr, "x265hip: saostats: SAO statistics of %llu CTU planes (SAO::calcSaoStatsCTU: band + four edge classes) measured by the GPU in %llu jobs, %llu planes on the host; "
                    "%llu waits of %.0f cycles on average; %llu CTUs submitted one CTU ahead of their request\n", (unsigned long long)planes, (unsigned long long)jobs, (unsigned long long)host,
            (unsigned long long)w, w ? (double)wc / w : 0.0, (unsigned long long)ah);
}

bool sao_enabled()
{
    if (!g_saoState)
    {
        std::lock_guard<std::mutex> g(g_lock);
        if (!g_saoState)
        {
            const char* env = getenv("X265HIP_SAOSTATS");
            const char* all = getenv("X265HIP");
            const char* table = getenv("X265HIP_TABLE");
            if (getenv("X265HIP_SAOSTATS_PARTS")) g_saoParts = atoi(getenv("X265HIP_SAOSTATS_PARTS")) > 3;
            if (getenv("X265HIP_SAOSTATS_AHEAD")) g_saoAhead = atoi(getenv("X265HIP_SAOSTATS_AHEAD")) != 0;
            if (X265_DEPTH != 8 || (env && !strcmp(env, "0")) || (all && !strcmp(all, "0")) || (table && !strcmp(table, "percall")))
                g_saoState = -1;
            else
            {
                g_saoState = 1;
                if (getenv("X265HIP_VERBOSE")) atexit(sao_report);
            }
        }
    }
    return g_saoState > 0;
}

inline SaoCounters& sao_counters() { touch_shard(); return g_saoCount[t_shard & 15]; }

// the rectangles of one plane, exactly as sao.cpp:741-914 computes them (x265hip_saojob's plane block); false: a geometry the job does not carry
struct SaoPlane { int w, h, x0[5], y0[5], x1[5], y1[5]; const pixel* rec0; const pixel* fenc0; intptr_t stride; bool eo23; };
bool sao_rects(const SAO* sao, int addr, int plane, SaoPlane& out)
{
    const Frame* frame = sao->m_frame;
    const x265_param* param = sao->m_param;
    const Slice* slice = frame->m_encData->m_slice;
    const PicYuv* reconPic = frame->m_reconPic;
    const CUData* cu = frame->m_encData->getPicCTU(addr);
    out.fenc0 = frame->m_fencPic->getPlaneAddr(plane, addr);
    out.rec0 = reconPic->getPlaneAddr(plane, addr);
    out.stride = plane ? reconPic->m_strideC : reconPic->m_stride;
    if ((plane ? frame->m_fencPic->m_strideC : frame->m_fencPic->m_stride) != out.stride)
        return false;                                    // (the reference indexes both pictures with the reconstruction's stride, :786-806)
    uint32_t picWidth = param->sourceWidth, picHeight = param->sourceHeight;
    int ctuWidth = param->maxCUSize, ctuHeight = param->maxCUSize;
    uint32_t lpelx = cu->m_cuPelX, tpely = cu->m_cuPelY;
    const uint32_t bAboveUnavail = (!tpely) | cu->m_bFirstRowInSlice;
    if (plane)
    {
        picWidth >>= sao->m_hChromaShift; picHeight >>= sao->m_vChromaShift;
        ctuWidth >>= sao->m_hChromaShift; ctuHeight >>= sao->m_vChromaShift;
        lpelx >>= sao->m_hChromaShift; tpely >>= sao->m_vChromaShift;
    }
    const uint32_t rpelx = x265_min(lpelx + ctuWidth, picWidth), bpely = x265_min(tpely + ctuHeight, picHeight);
    ctuWidth = rpelx - lpelx; ctuHeight = bpely - tpely;
    if (cu->m_bLastRowInSlice)
        picHeight = bpely;
    if (ctuWidth < 1 || ctuHeight < 1 || ctuWidth > 64 || ctuHeight > 64)
        return false;
    const int po = plane ? 2 : 0;
    const bool nd = param->bSaoNonDeblocked != 0;
    const bool right = rpelx == picWidth, bottom = bpely == picHeight;
    int* x0 = out.x0; int* y0 = out.y0; int* x1 = out.x1; int* y1 = out.y1;
    // SAO_BO (:810-823): skipB 4 / skipR 5, non-deblocked 3 / 4
    { const int skipB = nd ? 3 : 4, skipR = nd ? 4 : 5;
      x0[0] = 0; y0[0] = 0; x1[0] = right ? ctuWidth : ctuWidth - skipR + po; y1[0] = bottom ? ctuHeight : ctuHeight - skipB + po; }
    // SAO_EO_0 (:826-839): skipB 4 / skipR 5, non-deblocked 3 / 5; the rows are NOT shortened at the picture's bottom
    { const int skipB = nd ? 3 : 4, skipR = 5;
      x0[1] = !lpelx; y0[1] = 0; x1[1] = right ? ctuWidth - 1 : ctuWidth - skipR + po; y1[1] = ctuHeight - skipB + po; }
    // SAO_EO_1 (:841-861): skipB 4, skipR 5 / non-deblocked 4
    { const int skipB = 4, skipR = nd ? 4 : 5;
      x0[2] = 0; y0[2] = bAboveUnavail; x1[2] = right ? ctuWidth : ctuWidth - skipR + po; y1[2] = bottom ? ctuHeight - 1 : ctuHeight - skipB + po; }
    // SAO_EO_2 / SAO_EO_3 (:865-914): skipB 4, skipR 5
    for (int c = 3; c < 5; c++)
    { const int skipB = 4, skipR = 5;
      x0[c] = !lpelx; y0[c] = bAboveUnavail; x1[c] = right ? ctuWidth - 1 : ctuWidth - skipR + po; y1[c] = bottom ? ctuHeight - 1 : ctuHeight - skipB + po; }
    out.eo23 = !param->bLimitSAO || ((slice->m_sliceType == P_SLICE && !cu->isSkipped(0)) || (slice->m_sliceType != B_SLICE));
    out.w = ctuWidth; out.h = ctuHeight;
    for (int c = 0; c < 5; c++)
        if (x1[c] <= x0[c] || y1[c] <= y0[c] || x1[c] < 0 || y1[c] < 0) { x0[c] = y0[c] = x1[c] = y1[c] = 0; }    // empty: the reference's loops measure nothing either
    return true;
}

void sao_drop(SaoJob& sj, bool deviceDone)
{
    if (sj.active && deviceDone)
        for (int k = 0; k < sj.nparts; k++) give_slot(sj.part[k].svc, sj.part[k].slot);
    sj.active = false;
}

// waits for one part of this thread's SAO job; false: the device did not deliver
bool sao_wait(SaoJob& sj, int k)
{
    const SaoPart& pt = sj.part[k];
    const uint32_t* ready = &pt.svc->mem[pt.slot].units[0].ready;
    if (__atomic_load_n(ready, __ATOMIC_ACQUIRE) == pt.seq) return true;
    const uint64_t t0 = __builtin_ia32_rdtsc();
    uint64_t spins = 0;
    int64_t waitedNs = 0, lastNs = -1;
    while (__atomic_load_n(ready, __ATOMIC_ACQUIRE) != pt.seq)
    {
        __builtin_ia32_pause();
        if (g_yieldAfter && spins >= (uint64_t)g_yieldAfter) sched_yield();
        if ((++spins & 255) == 0)
        {
            const int pk = x265hip_cuserve_poke(pt.svc->cs, pt.slot);
            timespec ts;
            clock_gettime(CLOCK_MONOTONIC, &ts);
            const int64_t nowNs = (int64_t)ts.tv_sec * 1000000000ll + ts.tv_nsec;
            if (pk == 0 && lastNs >= 0) waitedNs += nowNs - lastNs;
            lastNs = nowNs;
            if (pk < 0 || waitedNs > g_timeoutNs)
            {
                sj.active = false;                       // the slots are not given back: the device may still write into them
                g_saoState = -1;
                x265hip_device_failure("saostats", "an SAO statistics job did not come back");
                return false;
            }
        }
    }
    SaoCounters& c = sao_counters();
    c.waits.fetch_add(1, std::memory_order_relaxed);
    c.waitCycles.fetch_add(__builtin_ia32_rdtsc() - t0, std::memory_order_relaxed);
    return true;
}

// rows [ra, rb) of plane `pl` as a one-block job on a free slot; false: no slot / the device refused
bool sao_submit_part(SaoJob& sj, const SaoPlane& pl, int plane, int ra, int rb)
{
    Service* svc = NULL;
    const int slot = take_slot(&svc);
    if (slot < 0)
        return false;
    // the block: rows ra - 1 .. min(rb, h - 1) — one row above the first measured row and, unless the part ends with the plane, one below the last
    const int rowsBelow = rb < pl.h ? 1 : 0, hh = rb - ra + rowsBelow, w = pl.w;
    x265hip_saojob job;
    memset(&job, 0, sizeof(job));
    job.bitDepth = X265_DEPTH; job.planes = 1; job.eo23 = pl.eo23;
    job.plane[0].w = (uint16_t)w; job.plane[0].h = (uint16_t)hh;
    for (int c = 0; c < 5; c++)
    {
        int y0 = pl.y0[c] < ra ? ra : pl.y0[c], y1 = pl.y1[c] > rb ? rb : pl.y1[c];
        int x0 = pl.x0[c], x1 = pl.x1[c];
        if (y1 <= y0 || x1 <= x0) { x0 = x1 = 0; y0 = y1 = ra; }
        job.plane[0].x0[c] = (uint8_t)x0; job.plane[0].x1[c] = (uint8_t)x1; job.plane[0].y0[c] = (uint8_t)(y0 - ra); job.plane[0].y1[c] = (uint8_t)(y1 - ra);
    }
    static thread_local pixel staged[66 * 65 + 65 * 64];
    pixel* dst = staged;
    const pixel* r = pl.rec0 + (intptr_t)(ra - 1) * pl.stride - 1;
    for (int y = 0; y <= hh; y++, dst += w + 1) memcpy(dst, r + (intptr_t)y * pl.stride, (size_t)(w + 1) * sizeof(pixel));
    const pixel* f = pl.fenc0 + (intptr_t)ra * pl.stride;
    // (the source rows of the extra row below are carried but never measured: its samples lie outside every rectangle)
    for (int y = 0; y < hh; y++, dst += w) memcpy(dst, f + (intptr_t)y * pl.stride, (size_t)w * sizeof(pixel));
    memcpy(svc->mem[slot].pixels, staged, (size_t)(dst - staged) * sizeof(pixel));
    uint32_t seq = 0;
    if (x265hip_cuserve_submit_sao(svc->cs, slot, &job, &seq))
    {
        give_slot(svc, slot);
        g_saoState = -1;
        x265hip_device_failure("saostats", "x265hip_cuserve_submit_sao");
        return false;
    }
    SaoPart& pt = sj.part[sj.nparts++];
    pt.plane = plane; pt.slot = slot; pt.seq = seq; pt.svc = svc;
    sao_counters().jobs.fetch_add(1, std::memory_order_relaxed);
    return true;
}

// the parts for planes [first, first + n) of CTU `addr`; false: nothing was submitted
bool sao_submit(SaoJob& sj, SAO* sao, int addr, int first, int n)
{
    if (g_dead.load(std::memory_order_relaxed) || !service())
        return false;
    SaoPlane pl[3];
    for (int b = 0; b < n; b++)
        if (!sao_rects(sao, addr, first + b, pl[b]))
            return false;
    sj.nparts = 0;
    for (int p = 0; p < 3; p++) { sj.consumed[p] = false; sj.wanted[p] = false; }
    for (int b = 0; b < n && g_saoState > 0; b++)
    {
        const int plane = first + b, h = pl[b].h;
        // the luma plane in two halves when it is high enough to be worth a second workgroup
        const int mid = plane == 0 && g_saoParts && h >= 32 ? (h / 2 + 3) & ~3 : h;
        const int before = sj.nparts;
        bool ok = sao_submit_part(sj, pl[b], plane, 0, mid) && (mid == h || sao_submit_part(sj, pl[b], plane, mid, h));
        if (!ok)
        {
            // a plane is served whole or not at all: parts of it that did leave are waited out and dropped with the rest (below, by the caller's next call)
            if (sj.nparts > before || b > 0) break;
            return false;
        }
        sj.wanted[plane] = true;
    }
    // a plane whose second half found no slot: not wanted (its first half is still waited for before the slots go back)
    for (int p = 0; p < 3; p++)
    {
        int have = 0;
        for (int k = 0; k < sj.nparts; k++) have += sj.part[k].plane == p;
        if (sj.wanted[p] && !have) sj.wanted[p] = false;
    }
    if (!sj.nparts)
        return false;
    sj.active = true; sj.sao = sao; sj.addr = addr; sj.encData = sao->m_frame->m_encData; sj.poc = sao->m_frame->m_poc;
    return true;
}

// A pool thread that ends with sets still out (an ahead job whose CTU another thread served: up to four slots each) hands their slots back, like
// IntraThreadEnd below: with 2 x CPUs slots in all, one closed encoder could otherwise leave most of them taken for the rest of the process
struct SaoThreadEnd
{
    ~SaoThreadEnd()
    {
        std::lock_guard<std::mutex> g(g_lock);               // shutdown() closes the services under this lock
        if (g_dead.load(std::memory_order_relaxed))
            return;                                          // the services are closed (or failed): their slots are gone with them
        for (SaoJob& t : t_saoSet)
        {
            if (!t.active)
                continue;
            bool done = true;
            for (int k = 0; k < t.nparts && done; k++)
            {
                const SaoPart& pt = t.part[k];
                const uint32_t* ready = &pt.svc->mem[pt.slot].units[0].ready;
                for (int spins = 0; spins < 200000 && __atomic_load_n(ready, __ATOMIC_ACQUIRE) != pt.seq; spins++)     // a plane is ~15 us of device time
                    __builtin_ia32_pause();
                done = __atomic_load_n(ready, __ATOMIC_ACQUIRE) == pt.seq;
            }
            sao_drop(t, done);
        }
    }
};
thread_local SaoThreadEnd t_saoThreadEnd;

} // namespace

void SAO::calcSaoStatsCTU(int addr, int plane)
{
    if (g_saoState < 0 || !sao_enabled())
    {
        refCalcSaoStatsCTU(this, addr, plane);
        return;
    }
    // this thread's sets: the one of this CTU (submitted ahead during the previous CTU, or now), and one free for the next CTU.  A set of any other CTU
    // (its chroma planes were never asked for, or the row ended) is waited out and its slots go back
    const int numCuInWidth = m_numCuInWidth;
    const bool nextInRow = (addr + 1) % numCuInWidth != 0;
    SaoJob* cur = NULL;
    (void)&t_saoThreadEnd;                                   // (constructed on first use: registers the destructor with this thread)
    for (SaoJob& t : t_saoSet)
    {
        if (!t.active) continue;
        const bool thisPicture = t.sao == this && t.encData == (const void*)m_frame->m_encData && t.poc == m_frame->m_poc;
        if (thisPicture && t.addr == addr) { cur = &t; continue; }
        if (thisPicture && t.addr == addr + 1 && nextInRow) continue;
        bool done = true;
        for (int k = 0; k < t.nparts && done; k++) done = sao_wait(t, k);
        sao_drop(t, done);
    }
    const bool chroma = m_param->internalCsp != X265_CSP_I400 && m_frame->m_fencPic->m_picCsp != X265_CSP_I400;
    const SAOParam* sp = m_frame->m_encData->m_saoParam;
    // luma asked for: the chroma planes ride along when the reference is going to ask for them whatever the luma decision (no --limit-sao, :1299-1306)
    const int planesWithLuma = chroma && !m_param->bLimitSAO && sp && sp->bSaoFlag[1] && m_param->internalCsp == X265_CSP_I420 ? 3 : 1;
    if (!cur)
    {
        for (SaoJob& t : t_saoSet)
            if (!t.active) { cur = &t; break; }
        if (cur)
        {
            if (plane == 0)
                sao_submit(*cur, this, addr, 0, planesWithLuma);
            else if (plane == 1 && m_param->internalCsp == X265_CSP_I420)
                sao_submit(*cur, this, addr, 1, 2);
            if (!cur->active) cur = NULL;
        }
    }
    // The NEXT CTU of the row leaves now, while this thread decides this CTU's offsets.  Safe because of where the reference calls from
    // (FrameFilter::ParallelFilter::processTasks, framefilter.cpp:463-500): rdoSaoUnitCu(col - 2) runs after deblockCTU(col, EDGE_VER) and
    // deblockCTU(col - 1, EDGE_HOR) — CTU col - 1 is as deblocked as it will be when its own turn comes one column later (what deblockCTU(col + 1, VER) and
    // (col, HOR) still change lies in CTU col and beyond; what the next CTU ROW's horizontal edges change lies in the 3 bottom rows every class leaves out),
    // and the previous row's SAO is applied no further than column col - 3 before that turn (:495-499: processSaoCTU(col - 3) after this call), so the row
    // above CTU col - 1 and its two corner samples are still the deblocked ones.  X265HIP_VERIFY compares every plane served this way with the reference's
    // body at the reference's own time.
    if (plane == 0 && g_saoAhead && nextInRow && g_saoState > 0 && sp && sp->bSaoFlag[0])
    {
        SaoJob* nxt = NULL;
        bool have = false;
        for (SaoJob& t : t_saoSet)
        {
            if (t.active && t.sao == this && t.addr == addr + 1 && t.encData == (const void*)m_frame->m_encData && t.poc == m_frame->m_poc) have = true;
            else if (!t.active && &t != cur && !nxt) nxt = &t;
        }
        if (!have && nxt)
        {
            sao_submit(*nxt, this, addr + 1, 0, planesWithLuma);
            if (nxt->active) sao_counters().ahead.fetch_add(1, std::memory_order_relaxed);
        }
    }
    if (!cur)
    {
        sao_counters().hostPlanes.fetch_add(1, std::memory_order_relaxed);
        refCalcSaoStatsCTU(this, addr, plane);
        return;
    }
    SaoJob& sj = *cur;
    if (sj.active && g_saoState > 0 && sj.wanted[plane] && !sj.consumed[plane])
    {
        bool ok = true;
        for (int k = 0; k < sj.nparts && ok; k++)
            if (sj.part[k].plane == plane) ok = sao_wait(sj, k);
        if (ok)
        {
            static const int typeOf[5] = { SAO_BO, SAO_EO_0, SAO_EO_1, SAO_EO_2, SAO_EO_3 };
            int32_t st[160], ct[160];
            memset(st, 0, sizeof(st)); memset(ct, 0, sizeof(ct));
            for (int k = 0; k < sj.nparts; k++)
                if (sj.part[k].plane == plane)
                {
                    const int32_t* out = (const int32_t*)sj.part[k].svc->mem[sj.part[k].slot].levels;
                    for (int i = 0; i < 160; i++) { st[i] += out[i]; ct[i] += out[X265HIP_SAOJOB_STATS_ENTRIES + i]; }
                }
            if (g_verify)
            {
                PerClass keepC, keepO;
                memcpy(keepC, m_count[plane], sizeof(keepC)); memcpy(keepO, m_offsetOrg[plane], sizeof(keepO));
                refCalcSaoStatsCTU(this, addr, plane);
                for (int c = 0; c < 5; c++)
                    for (int k = 0; k < (c ? 5 : 32); k++)
                        if (m_count[plane][typeOf[c]][k] != keepC[typeOf[c]][k] + ct[c * 32 + k] || m_offsetOrg[plane][typeOf[c]][k] != keepO[typeOf[c]][k] + st[c * 32 + k])
                        {
                            fprintf(stderr, "x265hip: saostats: VERIFY FAILED CTU %d plane %d class %d bin %d: count %d + %d vs %d, sum %d + %d vs %d\n", addr, plane, c, k, keepC[typeOf[c]][k],
                                    ct[c * 32 + k], m_count[plane][typeOf[c]][k], keepO[typeOf[c]][k], st[c * 32 + k], m_offsetOrg[plane][typeOf[c]][k]);
                            abort();
                        }
            }
            else
                for (int c = 0; c < 5; c++)
                    for (int k = 0; k < (c ? 5 : 32); k++)
                    {
                        m_count[plane][typeOf[c]][k] += ct[c * 32 + k];
                        m_offsetOrg[plane][typeOf[c]][k] += st[c * 32 + k];
                    }
            sj.consumed[plane] = true;
            sao_counters().planes.fetch_add(1, std::memory_order_relaxed);
            bool all = true;
            for (int p = 0; p < 3; p++) all = all && (!sj.wanted[p] || sj.consumed[p]);
            if (all)
            {
                // (parts of planes that were dropped half-submitted are waited for like the rest)
                bool done = true;
                for (int k = 0; k < sj.nparts && done; k++) done = sao_wait(sj, k);
                sao_drop(sj, done);
            }
            return;
        }
    }
    sao_counters().hostPlanes.fetch_add(1, std::memory_order_relaxed);
    refCalcSaoStatsCTU(this, addr, plane);
}

} // namespace X265_NS
