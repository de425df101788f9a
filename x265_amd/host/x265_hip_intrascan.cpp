// x265_hip_intrascan.cpp — the intra mode scan of Search::checkIntraInInter as jobs of the CU-job service (split from x265_hip_cuserve.cpp in round 6;
// INTEGRATION.md §6l).
#include "x265_hip_cuserve.h"

namespace X265_NS {

using namespace cusvc;

// ---- the intra mode scan of Search::checkIntraInInter as a job ------------------------------------------------------------------------------------------
// In a P slice (or with --b-intra) every CU below 64x64 whose best inter mode has a residual is also tried as intra (analysis.cpp:1630-1663): 35 predictions
// and 35 sa8d calls per block, 5-6 % of the bound encoder's CPU time and all of it in the frames the others wait for (call chains of tools/prof/callers.py,
// profiles/r05_v1_sa8d_callers.txt).  The distortion half goes to the device as an x265hip_intrajob (include/x265hip.h): the two neighbour lines exactly as
// Predict::initAdiPattern leaves them in intraNeighbourBuf (so strong intra smoothing, constrained intra and unavailable neighbours are the host's business,
// not the device's) and the source block in, 35 costs out.  The reference's OWN body then runs — mode bits, costs, the comparison chain, fast-intra's subset —
// and only the table slots it calls for this block answer from the job: intra_pred[DC / planar], transpose and intra_pred_allangs into the Search object's
// scratch buffers do nothing, cu[].sa8d against those buffers returns the job's cost of the mode the call stands for (first call DC, second planar, then by
// offset into the all-angles buffer, search.cpp:1356-1390).  X265HIP_VERIFY: the slots do their work as well and every cost is compared.
//
// The job leaves AHEAD: the neighbours of a CU are final when its analysis starts (they belong to CUs coded before it; the sub-CU recursion writes inside
// the CU only), so Search::predInterSearch's seam — the 2Nx2N inter candidate, always before the intra try — submits it on entry, and checkIntraInInter adopts it if
// the lines and the source block it would send now compare equal to what was sent; a job nobody asks for is dropped at the next one.
std::atomic<int> g_intraState(0);    // X265HIP_INTRASCAN=0: off
bool g_intraAhead = true;            // X265HIP_INTRASCAN_AHEAD=0: the job leaves when checkIntraInInter is entered
int g_intraMinLog2 = 4;              // X265HIP_INTRASCAN_MIN=<log2>: smallest block handed over (an 8x8 scan is ~8 us of host code: less than a round trip)
bool g_intraAheadPredict = true;     // X265HIP_INTRASCAN_AHEAD=2: no prediction of whether the intra try will come, every candidate CU's job leaves
int g_intraSyncMinLog2 = 5;          // ... and the smallest one handed over when no job is ahead (the thread waits a whole round trip)
struct alignas(64) IntraCounters { std::atomic<uint64_t> jobs, served, ahead, aheadHit, dropped, waits, waitCycles, host, aheadBy[2]; };
IntraCounters g_intraCount[16];
constexpr int kIntraMaxSamples = 2 * (4 * 32 + 16) + 32 * 32;
struct IntraJob
{
    bool active;
    Service* svc; int slot; uint32_t seq;
    int log2n;
    pixel sent[kIntraMaxSamples];     // what the device was given: raw line, filtered line, source block
};
// inside refCheckIntraInInter of a served block: what the table slots answer from
struct IntraCtx { bool active, allangs, verify; const pixel* predBuf; const pixel* fencT; int n, lastMode; const int32_t* costs; };
__attribute__((tls_model("initial-exec"))) thread_local IntraJob t_intra;
__attribute__((tls_model("initial-exec"))) thread_local IntraCtx t_ictx;

inline IntraCounters& intra_counters() { touch_shard(); return g_intraCount[t_shard & 15]; }

void intra_report()
{
    uint64_t jobs = 0, served = 0, ah = 0, hit = 0, dr = 0, w = 0, wc = 0, host = 0, ab[2] = { 0, 0 };
    for (int i = 0; i < 16; i++)
    {
        for (int k = 0; k < 2; k++) ab[k] += g_intraCount[i].aheadBy[k];
        jobs += g_intraCount[i].jobs; served += g_intraCount[i].served; ah += g_intraCount[i].ahead; hit += g_intraCount[i].aheadHit; dr += g_intraCount[i].dropped;
        w += g_intraCount[i].waits; wc += g_intraCount[i].waitCycles; host += g_intraCount[i].host;
    }
    fprintf(stderr, "x265hip: intrascan: the 35-mode sa8d scans of %llu blocks >= %dx%d (Search::checkIntraInInter) measured by the GPU in %llu jobs, %llu scans on the host; %llu jobs "
                    "left ahead when predInterSearch was entered, %llu of them adopted, %llu never asked for; %llu waits of %.0f cycles on average\n",
            (unsigned long long)served, 1 << g_intraMinLog2, 1 << g_intraMinLog2, (unsigned long long)jobs, (unsigned long long)host, (unsigned long long)ah, (unsigned long long)hit,
            (unsigned long long)dr, (unsigned long long)w, w ? (double)wc / w : 0.0);
    fprintf(stderr, "x265hip: intrascan: %llu candidate CUs sent no job ahead because no sub-CU of theirs had chosen intra (analysis.cpp:1633: --limit-refs)\n", (unsigned long long)ab[0]);
}

bool intra_enabled()
{
    if (!g_intraState)
    {
        std::lock_guard<std::mutex> g(g_lock);
        if (!g_intraState)
        {
            const char* env = getenv("X265HIP_INTRASCAN");
            const char* all = getenv("X265HIP");
            const char* table = getenv("X265HIP_TABLE");
            if (getenv("X265HIP_INTRASCAN_AHEAD")) { g_intraAhead = atoi(getenv("X265HIP_INTRASCAN_AHEAD")) != 0; g_intraAheadPredict = atoi(getenv("X265HIP_INTRASCAN_AHEAD")) != 2; }
            if (getenv("X265HIP_INTRASCAN_MIN")) g_intraMinLog2 = x265_clip3(3, 5, atoi(getenv("X265HIP_INTRASCAN_MIN")));
            if (getenv("X265HIP_INTRASCAN_SYNC_MIN")) g_intraSyncMinLog2 = x265_clip3(3, 6, atoi(getenv("X265HIP_INTRASCAN_SYNC_MIN")));
            if ((env && !strcmp(env, "0")) || (all && !strcmp(all, "0")) || (table && !strcmp(table, "percall")))
                g_intraState = -1;
            else
            {
                g_intraState = 1;
                if (getenv("X265HIP_VERBOSE")) atexit(intra_report);
            }
        }
    }
    if (g_intraState > 0 && g_dead.load(std::memory_order_relaxed))
    {
        // the service the scans travel on has failed (a lost job of either kind takes the whole service down): said once, like every module that goes off
        bool first = false;
        {
            std::lock_guard<std::mutex> g(g_lock);
            if (g_intraState > 0) { g_intraState = -1; first = true; }
        }
        if (first) x265hip_device_failure("intrascan", "the CU-job service has failed");
    }
    return g_intraState > 0 && g_state > 0 && g_slots_installed;
}

// waits for this thread's intra job; false: the device did not deliver (the slot is kept: the device may still write into it)
bool intra_wait(IntraJob& ij)
{
    const uint32_t* ready = &ij.svc->mem[ij.slot].units[0].ready;
    if (__atomic_load_n(ready, __ATOMIC_ACQUIRE) == ij.seq) return true;
    const uint64_t t0 = __builtin_ia32_rdtsc();
    uint64_t spins = 0;
    int64_t waitedNs = 0, lastNs = -1;
    while (__atomic_load_n(ready, __ATOMIC_ACQUIRE) != ij.seq)
    {
        __builtin_ia32_pause();
        if ((++spins & 255) == 0)
        {
            const int pk = x265hip_cuserve_poke(ij.svc->cs, ij.slot);
            timespec ts;
            clock_gettime(CLOCK_MONOTONIC, &ts);
            const int64_t nowNs = (int64_t)ts.tv_sec * 1000000000ll + ts.tv_nsec;
            if (pk == 0 && lastNs >= 0) waitedNs += nowNs - lastNs;
            lastNs = nowNs;
            if (pk < 0 || waitedNs > g_timeoutNs)
            {
                ij.active = false;
                g_intraState = -1;
                x265hip_device_failure("intrascan", "an intra scan job did not come back");
                return false;
            }
        }
    }
    IntraCounters& c = intra_counters();
    c.waits.fetch_add(1, std::memory_order_relaxed);
    c.waitCycles.fetch_add(__builtin_ia32_rdtsc() - t0, std::memory_order_relaxed);
    return true;
}

// the job's scope ends (served, or never asked for): the slot goes back once the device has written what it is going to write.  Nobody waits for a job
// that was never asked for: its slot is parked (two per thread) and handed back when a later call finds its ready word set
struct IntraParked { Service* svc; int slot; uint32_t seq; int looks; };
__attribute__((tls_model("initial-exec"))) thread_local IntraParked t_parked[2];
inline bool intra_ready(Service* svc, int slot, uint32_t seq) { return __atomic_load_n(&svc->mem[slot].units[0].ready, __ATOMIC_ACQUIRE) == seq; }
void intra_sweep()
{
    for (IntraParked& z : t_parked)
    {
        if (!z.svc)
            continue;
        if (intra_ready(z.svc, z.slot, z.seq)) { give_slot(z.svc, z.slot); z.svc = NULL; }
        else if (++z.looks == 16 && x265hip_cuserve_poke(z.svc->cs, z.slot) < 0)
        {
            // a parked job that is still out after sixteen later jobs, on a service that reports a failure: it is not coming (the slot stays taken)
            z.svc = NULL;
            if (g_intraState > 0)
            {
                g_intraState = -1;
                x265hip_device_failure("intrascan", "an intra scan job did not come back");
            }
        }
    }
}
void intra_drop(IntraJob& ij)
{
    if (!ij.active)
        return;
    ij.active = false;
    if (intra_ready(ij.svc, ij.slot, ij.seq)) { give_slot(ij.svc, ij.slot); return; }
    intra_sweep();
    for (IntraParked& z : t_parked)
        if (!z.svc) { z.svc = ij.svc; z.slot = ij.slot; z.seq = ij.seq; z.looks = 0; return; }
    ij.active = true;                                       // both places taken by jobs still on the device: wait for this one after all
    if (intra_wait(ij))
        give_slot(ij.svc, ij.slot);
    ij.active = false;
}

// the block as checkIntraInInter sees it now: Predict::initIntraNeighbors + initAdiPattern exactly as search.cpp:1308-1310 calls them (they write
// intraNeighbourBuf[0] and, for 8 / 16 / 32, [1] from the reconstructed picture; the reference's body repeats the call), packed the way the job carries it
int intra_pack(Search* se, const CUData& cu, const CUGeom& cuGeom, const Yuv& fencYuv, pixel* dst)
{
    const int log2n = (int)cuGeom.log2CUSize, n = 1 << log2n, line = x265hipi_intrajob_line_samples(log2n);
    Predict::IntraNeighbors nb;
    se->initIntraNeighbors(cu, 0, 0, true, &nb);
    se->initAdiPattern(cu, cuGeom, 0, nb, ALL_IDX);
    memcpy(dst, se->intraNeighbourBuf[0], sizeof(pixel) * (4 * n + 1));
    memset(dst + 4 * n + 1, 0, sizeof(pixel) * 15);
    memcpy(dst + line, se->intraNeighbourBuf[1], sizeof(pixel) * (4 * n + 1));
    memset(dst + line + 4 * n + 1, 0, sizeof(pixel) * 15);
    pixel* f = dst + 2 * line;
    pack_rows(f, fencYuv.m_buf[0], fencYuv.m_size, n);
    return 2 * line + n * n;
}

// A thread that ends (x265's pool threads end with their encoder) hands back what it still holds — a job ahead nobody asked for, parked slots: in a process
// that opens and closes encoders for days, slots kept by dead threads would starve the service (the CU jobs would quietly stay on the host)
struct IntraThreadEnd
{
    ~IntraThreadEnd()
    {
        std::lock_guard<std::mutex> g(g_lock);               // shutdown() sets g_dead and closes the services (their pinned memory) under this lock
        if (g_dead.load(std::memory_order_relaxed))
            return;                                          // the services are closed (or failed): their slots are gone with them
        if (t_intra.active)
            intra_drop(t_intra);
        for (IntraParked& z : t_parked)
        {
            if (!z.svc)
                continue;
            for (int spins = 0; spins < 200000 && !intra_ready(z.svc, z.slot, z.seq); spins++)     // a scan is tens of microseconds of device time
                __builtin_ia32_pause();
            if (intra_ready(z.svc, z.slot, z.seq))
                give_slot(z.svc, z.slot);
            z.svc = NULL;
        }
    }
};
thread_local IntraThreadEnd t_intraThreadEnd;

bool intra_slots_in(const EncoderPrimitives& p, int log2n);
bool intra_submit(IntraJob& ij, const pixel* blob, int samples, int log2n)
{
    (void)&t_intraThreadEnd;                                 // (constructed on first use: registers the destructor with this thread)
    intra_sweep();
    if (!service())
        return false;
    Service* svc = NULL;
    const int slot = take_slot(&svc);
    if (slot < 0)
        return false;
    x265hip_intrajob job;
    job.bitDepth = X265_DEPTH; job.mark = X265HIP_INTRAJOB_MARK; job.log2Size = (uint32_t)log2n; job.reserved = 0;
    memcpy(svc->mem[slot].pixels, blob, sizeof(pixel) * samples);
    uint32_t seq = 0;
    if (x265hip_cuserve_submit_intra(svc->cs, slot, &job, &seq))
    {
        give_slot(svc, slot);
        g_intraState = -1;
        x265hip_device_failure("intrascan", "x265hip_cuserve_submit_intra");
        return false;
    }
    if (blob != ij.sent) memcpy(ij.sent, blob, sizeof(pixel) * samples);
    ij.svc = svc; ij.slot = slot; ij.seq = seq; ij.log2n = log2n; ij.active = true;
    intra_counters().jobs.fetch_add(1, std::memory_order_relaxed);
    return true;
}

void intra_unasked()
{
    IntraJob& ij = t_intra;
    if (ij.active)
    {
        intra_counters().dropped.fetch_add(1, std::memory_order_relaxed);
        intra_drop(ij);
    }
}

// called by Search::predInterSearch's seam for the 2Nx2N candidate: will this CU be tried as intra (analysis.cpp:1595)?  Then its scan leaves now.
void intra_ahead(Search* se, Mode& interMode, const CUGeom& cuGeom)
{
    IntraJob& ij = t_intra;
    if (ij.active)
    {
        intra_counters().dropped.fetch_add(1, std::memory_order_relaxed);
        intra_drop(ij);
    }
    const int log2n = (int)cuGeom.log2CUSize;
    const Slice* slice = interMode.cu.m_slice;
    // (only sizes whose table slots are installed: a job checkIntraInInter could never adopt would only hold a slot)
    if (!g_intraAhead || !intra_enabled() || log2n < g_intraMinLog2 || log2n > 5 || !intra_slots_in(primitives, log2n) || log2n == (int)g_log2Size[se->m_param->maxCUSize] ||
        !(slice->m_sliceType != B_SLICE || se->m_param->bIntraInBFrames) || se->m_param->rdLevel < 2 || se->m_param->rdLevel > 4 || se->m_param->bDistributeModeAnalysis ||
        se->m_param->analysisLoad || (se->m_param->bCTUInfo & 4))
        return;
    // With --limit-refs (preset medium and slower) the intra try needs `splitIntra` (analysis.cpp:1633): no sub-CU recursion for this CU, or a sub-CU whose
    // best mode is intra (:1182, :1353, :1369).  The recursion leaves its trace in the depth's split prediction — initSubCU to this CU (:1346), the sub-CUs'
    // data copied in quadrant by quadrant (:1370) — so the flag can be read back; a stale trace only costs a job nobody asks for, or a scan on the host.
    int likely = 1;
    if (se->m_param->limitReferences && g_intraAheadPredict)
    {
        const CUData& sp = static_cast<Analysis*>(se)->m_modeDepth[cuGeom.depth].pred[Analysis::PRED_SPLIT].cu;
        if (sp.m_cuAddr == interMode.cu.m_cuAddr && sp.m_absIdxInCTU == cuGeom.absPartIdx && sp.m_encData == interMode.cu.m_encData && cuGeom.log2CUSize > 3)
        {
            const uint32_t q = cuGeom.numPartitions >> 2;
            likely = 0;
            for (uint32_t k = 0; k < 4; k++)
                likely |= sp.m_predMode[k * q] == MODE_INTRA;
        }
    }
    if (!likely)
    {
        intra_counters().aheadBy[0].fetch_add(1, std::memory_order_relaxed);      // (not sent: counted to show what the prediction withholds)
        return;
    }
    const int samples = intra_pack(se, interMode.cu, cuGeom, *interMode.fencYuv, ij.sent);
    if (intra_submit(ij, ij.sent, samples, log2n))
    {
        intra_counters().ahead.fetch_add(1, std::memory_order_relaxed);
        intra_counters().aheadBy[1].fetch_add(1, std::memory_order_relaxed);
    }
}

// ---- the table slots the reference's body calls for the block.  Which mode a cu[].sa8d call stands for: the mode of the intra_pred[] call before it (every
// one of the 35 entries is a slot that notes its own index) — or, where the table carries intra_pred_allangs (x265_setup_primitives removes the C one,
// primitives.cpp; an assembly table has it), the offset into the all-angles buffer (search.cpp:1383-1387)
template <int CU>
int sa8d_slot(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb)
{
    IntraCtx& c = t_ictx;
    if (c.active && b >= c.predBuf && b < c.predBuf + 33 * c.n * c.n && (1 << (CU + 2)) == c.n)
    {
        const int mode = c.allangs ? 2 + (int)((b - c.predBuf) / (c.n * c.n)) : c.lastMode;
        const int v = c.costs[mode];
        if (c.verify)
        {
            const int want = g_prev.cu[CU].sa8d(a, sa, b, sb);
            if (want != v)
            {
                fprintf(stderr, "x265hip: VERIFY FAILED intra scan %dx%d mode %d: %d (job) vs %d\n", c.n, c.n, mode, v, want);
                abort();
            }
        }
        return v;
    }
    return g_prev.cu[CU].sa8d(a, sa, b, sb);
}
template <int CU, int MODE>
void intra_pred_slot(pixel* dst, intptr_t dstStride, const pixel* srcPix, int dirMode, int bFilter)
{
    IntraCtx& c = t_ictx;
    if (c.active && dst == c.predBuf && (1 << (CU + 2)) == c.n)
    {
        c.lastMode = MODE;
        c.allangs = false;
        if (!c.verify)
            return;
    }
    g_prev.cu[CU].intra_pred[MODE](dst, dstStride, srcPix, dirMode, bFilter);
}
template <int CU>
void allangs_slot(pixel* dst, pixel* refPix, pixel* filtPix, int bLuma)
{
    IntraCtx& c = t_ictx;
    if (c.active && dst == c.predBuf && (1 << (CU + 2)) == c.n)
    {
        c.allangs = true;
        if (!c.verify)
            return;
    }
    g_prev.cu[CU].intra_pred_allangs(dst, refPix, filtPix, bLuma);
}
template <int CU>
void transpose_slot(pixel* dst, const pixel* src, intptr_t stride)
{
    const IntraCtx& c = t_ictx;
    if (c.active && !c.verify && dst == c.fencT && (1 << (CU + 2)) == c.n)
        return;
    g_prev.cu[CU].transpose(dst, src, stride);
}
template <int CU, int M>
struct InstallPredSlots
{
    static void run(EncoderPrimitives& p) { p.cu[CU].intra_pred[M] = intra_pred_slot<CU, M>; InstallPredSlots<CU, M - 1>::run(p); }
};
template <int CU>
struct InstallPredSlots<CU, -1> { static void run(EncoderPrimitives&) {} };
template <int CU>
void install_intra_slots_for(EncoderPrimitives& p)
{
    p.cu[CU].sa8d = sa8d_slot<CU>;
    InstallPredSlots<CU, NUM_INTRA_MODE - 1>::run(p);
    if (g_prev.cu[CU].intra_pred_allangs)
    {
        p.cu[CU].intra_pred_allangs = allangs_slot<CU>;
        p.cu[CU].transpose = transpose_slot<CU>;
    }
}
void install_intra_slots(EncoderPrimitives& p)
{
    const char* env = getenv("X265HIP_INTRASCAN");
    if (env && !strcmp(env, "0"))
        return;
    install_intra_slots_for<BLOCK_16x16>(p);
    install_intra_slots_for<BLOCK_32x32>(p);
    // 8x8 blocks (X265HIP_INTRASCAN_MIN=3): their scans only ever leave AHEAD (an 8x8 scan is less host work than a round trip), and the answering slots sit on
    // cu[BLOCK_8x8].sa8d and its 35 predictors, which every other 8x8 caller then goes through as well
    const char* mn = getenv("X265HIP_INTRASCAN_MIN");
    if (mn && atoi(mn) <= 3)
        install_intra_slots_for<BLOCK_8x8>(p);
}
bool intra_slots_in(const EncoderPrimitives& p, int log2n)
{
    return log2n == 3 ? p.cu[BLOCK_8x8].sa8d == sa8d_slot<BLOCK_8x8>
         : log2n == 4 ? p.cu[BLOCK_16x16].sa8d == sa8d_slot<BLOCK_16x16> : log2n == 5 ? p.cu[BLOCK_32x32].sa8d == sa8d_slot<BLOCK_32x32> : false;
}

void Search::checkIntraInInter(Mode& intraMode, const CUGeom& cuGeom)
{
    IntraJob& ij = t_intra;
    const int log2n = (int)cuGeom.log2CUSize;
    if (intra_enabled() && log2n >= g_intraMinLog2 && log2n <= 5 && intra_slots_in(primitives, log2n) && !t_ictx.active)
    {
        pixel cur[kIntraMaxSamples];
        const int samples = intra_pack(this, intraMode.cu, cuGeom, *intraMode.fencYuv, cur);
        if (ij.active)
        {
            if (ij.log2n == log2n && !memcmp(cur, ij.sent, sizeof(pixel) * samples))
                intra_counters().aheadHit.fetch_add(1, std::memory_order_relaxed);
            else
            {
                intra_counters().dropped.fetch_add(1, std::memory_order_relaxed);
                intra_drop(ij);
            }
        }
        if (!ij.active && log2n >= g_intraSyncMinLog2)
            intra_submit(ij, cur, samples, log2n);
        if (ij.active && intra_wait(ij))
        {
            IntraCtx& c = t_ictx;
            c.active = true; c.allangs = false; c.predBuf = m_intraPredAngs; c.fencT = m_fencTransposed; c.n = 1 << log2n; c.lastMode = DC_IDX;
            c.costs = reinterpret_cast<const int32_t*>(ij.svc->mem[ij.slot].levels); c.verify = g_verify;
            if (g_time) { Timed t(13 + cuGeom.log2CUSize - 2); refCheckIntraInInter(this, intraMode, cuGeom); }
            else refCheckIntraInInter(this, intraMode, cuGeom);
            c.active = false;
            give_slot(ij.svc, ij.slot);
            ij.active = false;
            intra_counters().served.fetch_add(1, std::memory_order_relaxed);
            return;
        }
    }
    else if (ij.active)
    {
        intra_counters().dropped.fetch_add(1, std::memory_order_relaxed);
        intra_drop(ij);
    }
    intra_counters().host.fetch_add(1, std::memory_order_relaxed);
    if (g_time)
    {
        Timed t(13 + cuGeom.log2CUSize - 2);
        refCheckIntraInInter(this, intraMode, cuGeom);
        return;
    }
    refCheckIntraInInter(this, intraMode, cuGeom);
}

} // namespace X265_NS
