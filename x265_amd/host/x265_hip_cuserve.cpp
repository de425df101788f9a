// x265_hip_cuserve.cpp — the sixth translation unit of the drop-in: a CU's residual quad-tree arithmetic handed to the GPU as ONE job
// (include/x265hip.h, x265hip_cuserve_*).
//
// Search::encodeResAndCalcRdInterCU (reference source/encoder/search.cpp:2822-2975) prices the residual of an inter prediction with
// Search::estimateResidualQT (:3178-3560): per transform unit of the CU's quad-tree, for Y, Cb and Cr,
//     Quant::transformNxN   (common/quant.cpp:397-470)   residual -> cu[].dct -> quant -> signBitHidingHDQ -> levels, numSig
//     Entropy::codeCoeffNxN                                the levels' bits — CABAC state, stays on the host
//     Quant::invtransformNxN (quant.cpp:543-603)          levels -> dequant_normal -> cu[].idct -> reconstructed residual
//     cu[].add_ps, cu[].sse_pp, psy_cost_pp               reconstruction, distortion, psycho-visual energy
// and keeps the cheaper of "coded" and "cbf = 0".  Which alternative wins depends on the entropy coder's state; what each transform unit's
// levels, reconstructed residual and distortions ARE does not: they are functions of the CU's source block, its prediction, the QP and the
// transform size.  So the top-level call hands (source, prediction, QPs, the transform sizes the tree may try) to the device in one job
// (a mailbox slot in page-locked memory; x265_amd/csrc/cuserve.hip), and while the reference's own estimateResidualQT body runs on this
// thread, Quant::transformNxN / ::invtransformNxN called for a residual block of that CU copy the device's result instead of computing it.
// Same values either way (tests/test_cuserve.py pins the device against the oracle's restatement of those two functions, which is pinned
// against the reference's); anything the job does not cover (transform skip, transquant bypass, scaling lists, noise reduction, RDOQ,
// 4:2:2 / 4:4:4) runs the reference's functions as before.
//
// Beside the transforms the job answers the distortions of the same blocks — cu[].sse_pp(source, prediction) and cu[].sse_pp(source, reconstruction)
// of every unit (search.cpp:3269, :3295, and again at CU level :2872-2877) — and the time a thread spends waiting for the device goes into
// psy_cost_pp(source, prediction) of the unit it waits for, which the tree asks for next (:3272) and encodeResAndCalcRdInterCU asks for again
// (:2888): both are remembered for the scope of that one encodeResAndCalcRdInterCU call, keyed by the blocks' addresses in the mode's source /
// prediction / reconstruction buffers, which nobody writes inside that scope.
//
// Seams (same link technique as the other seams, oracle/Makefile): Search::estimateResidualQT, Search::checkIntraInInter on search.o;
// Quant::transformNxN, Quant::invtransformNxN on quant.o; Search::encodeResAndCalcRdInterCU (the scope) with the reference's body renamed in place.
#include "x265_hip_cuserve.h"

namespace X265_NS {

namespace cusvc {


int g_state = 0;                 // 0 undecided, 1 on, -1 off
int g_time = 0;                  // X265HIP_DEBUG_CUTIME=1: cycles inside the functions a job could replace, by block size (report at exit);
                                 // =2: the same with the jobs running (the pair of runs measures what the jobs save and what the waiting costs)
int g_minLog2 = 5;               // X265HIP_CUSERVE_MIN: smallest CU (log2) whose residual quad-tree becomes a job
int g_mode = 0;                  // X265HIP_CUSERVE_MODE: 0 resident server (mailbox), 1 one launch per job
int64_t g_timeoutNs = 10000000000ll;            // X265HIP_CUSERVE_TIMEOUT_MS
int g_yieldAfter = 0;                            // X265HIP_CUSERVE_YIELD
std::atomic<int> g_lateJobs(0);
// X265HIP_CUSERVE_INVERSE=0 / 1: at the RDOQ presets the inverse half of luma 32x32 units stays on the host / leaves as a job behind Quant::rdoQuant.  Measured on
// the MI355X box, 6 interleaved rounds each (profiles/r06_v1_configs2_4k_slow_star_ab.txt, ..._configs3_4k_main10_slower_ab.txt): BASELINE configs[2] (8 bit)
// 3.78 vs 3.76 fps, 91.1 vs 91.5 CPU seconds with the jobs (236 739 units served); configs[3] (Main10) 1.46 vs 1.48 fps, 78.4 vs 77.0 CPU seconds (342 252 units:
// six kilobytes through the BAR per job instead of four, and the tree's bit count of the levels is shorter than the round trip).  On for 8-bit builds, off above
int g_invJobs = X265_DEPTH == 8 ? 1 : 0;
int g_rdoqJobs = 1;              // X265HIP_CUSERVE_RDOQ=0: CUs quantised by Quant::rdoQuant are not handed over (round 4's behaviour).  On: measured on the MI355X box at
                                 // BASELINE configs[2] / configs[3] (profiles/r05_v1_configs*_ab.txt): +2 % / +6 % fps, -3 % / -6 % CPU seconds
int g_slots = 64;                // X265HIP_CUSERVE_SLOTS: jobs that can be in flight (default: twice the CPUs this process may use, 16..64)
bool g_verify = false;           // X265HIP_VERIFY=1: every served unit is recomputed by the reference's function and compared
bool g_require = false;          // X265HIP=require: a device failure is fatal instead of falling back
std::mutex g_lock;
Service g_svc[16];
std::atomic<int> g_nsvc(0);                  // services open (published last)
std::atomic<bool> g_dead(false); // the device failed once: every later CU is computed on the host

// X265HIP_DEBUG_CUTIME: [0..3] transformNxN by log2TrSize - 2, [4..7] invtransformNxN, [8..12] top-level estimateResidualQT by log2CUSize - 2,
// [13..17] checkIntraInInter by log2CUSize - 2; second index: 0 inside a top-level estimateResidualQT of this thread, 1 elsewhere
std::atomic<uint64_t> g_cycles[18][2], g_calls[18][2];
__attribute__((tls_model("initial-exec"))) thread_local int t_inRqt = 0;

struct alignas(64) Counters { std::atomic<uint64_t> jobs, fwd, inv, fwdMiss, invMiss, waitCycles, waits, skipped, dist, psyHit, psyAhead, psyCoded, deadSub, deadAdd, lateSub, lateAdd, siteWaits[6], siteCycles[6], lumaHist[24], spec, specHit, psySkip, specInter, specInterHit, invJobs, invDropped; };
Counters g_count[64];
std::atomic<int> g_nextShard(0);
__attribute__((tls_model("initial-exec"))) thread_local int t_shard = -1;
inline Counters& counters() { if (t_shard < 0) t_shard = g_nextShard.fetch_add(1) & 63; return g_count[t_shard]; }
void touch_shard() { counters(); }

// the job of the top-level estimateResidualQT in progress on this thread
struct Job
{
    bool active;
    const Quant* quant;                  // the Search object's quantiser: only its calls are looked at
    const int16_t* resi[3]; uint32_t resiStride[3];      // the CU's residual blocks (ShortYuv): what identifies a unit
    uint32_t log2CU;
    int sHi, sLo;
    const Mode* mode;                    // the mode whose residual this is (its cbf flags and final reconstruction are looked at after the tree)
    bool specInter;                      // left ahead of its scope from predInterSearch (not from the merge candidate's skip evaluation): counted apart
    bool inTree;                         // inside the top-level estimateResidualQT: transform units are looked up (afterwards only the remembered values serve)
    const Search* search;
    const pixel* fenc[3]; uint32_t fencStride[3];        // the mode's source and prediction blocks (Yuv): what identifies an sse / psy question
    const pixel* pred[3]; uint32_t predStride[3];
    bool psyRd;
    uint8_t invServed[X265HIP_CUJOB_MAX_UNITS];          // the unit's reconstructed residual came from the device (so its reconstruction is pred + that)
    uint8_t energyKnown[X265HIP_CUJOB_MAX_UNITS];        // psy_cost_pp(source, prediction) of the unit has been computed on this thread
    int32_t energy[X265HIP_CUJOB_MAX_UNITS];
    // work of the reference's body whose result no one reads once the job answers (X265HIP_CUSERVE_DIST >= 3): the CU's residual (read only by the
    // transformNxN calls the job serves) and the tree's reconstructions (read only by the sse_pp / psy-cost calls the job answers).  The call is
    // remembered, not run; whoever turns out to need the result after all (a unit the job does not serve, a failed device) runs it first.
    struct PendSub { bool pending; int16_t* dst; intptr_t ds; const pixel* a; const pixel* b; intptr_t sa, sb; pixel_sub_ps_t fn; } pendSub[3];
    struct PendAdd { bool pending; pixel* dst; intptr_t ds; const pixel* a; const int16_t* b; intptr_t sa, sb; pixel_add_ps_t fn; } pendAdd[X265HIP_CUJOB_MAX_UNITS];
    bool anyPendAdd, treeMine;
    // 0: submitted ahead of its encodeResAndCalcRdInterCU (from encodeResAndCalcRdSkipCU, see there): nothing is answered yet; 1: inside the scope
    int phase;
    const Yuv* fencYuv;                  // the source Yuv the pixels were taken from
    x265hip_cujob hdr;                   // what was submitted (an adopted job must be the job the scope would submit itself) ...
    alignas(64) pixel sent[X265HIP_CUJOB_PIXEL_BYTES / sizeof(pixel)];       // ... header AND pixels: a job's answers are a function of exactly these
    // coefficient-mode jobs (RDOQ presets): the inverse half of a luma 32x32 unit as a job of its own on a second slot (x265hip_cujob::coefMode ==
    // X265HIP_CUJOB_INVERSE), submitted when Quant::rdoQuant has made the unit's levels and collected by Quant::invtransformNxN (the tree codes the levels'
    // bits in between, search.cpp:3243-3262)
    struct InvAhead { bool active; int unit; Service* svc; int slot; uint32_t seq; const x265hip_cujob_unit* units; const int16_t* resi; int16_t sent[1024]; } inv;
    uint32_t seq;
    int slot;
    Service* svc;
    x265hip_cujob* job;
    const x265hip_cujob_unit* units;
    const int16_t* levels;
    const int16_t* resiOut;
};
__attribute__((tls_model("initial-exec"))) thread_local Job t_job;
__attribute__((tls_model("initial-exec"))) thread_local int t_inEncodeRes = 0;
EncoderPrimitives g_prev;            // the table as it was when the cuserve slots were installed (C functions + the psy lookups of x265_hip_srcplanes.cpp)
bool g_slots_installed = false;
bool g_specInter = true;             // X265HIP_CUSERVE_SPEC_INTER=0: the 2Nx2N inter candidate's job does NOT leave when predInterSearch returns.  Every such job is adopted, the host's
                                     // wait cycles fall by a fifth.  Round 5 (one-wave device chain, -O2 host) measured no fps for it and left it off; on round 6's final tree, at the
                                     // bench's thread arguments: +1.3 % / +1.9 % fps and -2 % CPU seconds over 2 x 8 interleaved rounds of 240 frames (profiles/r06_v2_spec_inter_ab.txt)
std::atomic<uint64_t> g_lumaWait[2][2][2];      // luma forward waits that spun: [CU 32 / 64][the job's first luma unit / a later one][count / cycles] (report)
bool g_spec = true;                  // X265HIP_CUSERVE_SPEC=0: no job is submitted ahead of its scope
int g_serveDist = 1;                 // X265HIP_CUSERVE_DIST=0: transforms only; 1: + the tree's distortions; 2: + the CU's final sse_pp / psy cost; 3 (default): + the body's sub_ps / add_ps calls nobody reads any more are not run
__attribute__((tls_model("initial-exec"))) thread_local int t_hint = -1;           // where this thread looks first

void report_time()
{
    static const char* const what[4] = { "Quant::transformNxN", "Quant::invtransformNxN", "Search::estimateResidualQT (top level, whole tree)", "Search::checkIntraInInter" };
    for (int k = 0; k < 18; k++)
    {
        const int fn = k < 4 ? 0 : k < 8 ? 1 : k < 13 ? 2 : 3, size = 4 << (k < 4 ? k : k < 8 ? k - 4 : k < 13 ? k - 8 : k - 13);
        for (int w = 0; w < 2; w++)
            if (g_calls[k][w].load())
                fprintf(stderr, "x265hip: cutime: %-52s %2dx%-2d %s: %9llu calls, %8.0f cycles each, %7.3f G cycles\n", what[fn], size, size,
                        fn >= 2 ? "" : w ? "(elsewhere)           " : "(in estimateResidualQT)", (unsigned long long)g_calls[k][w].load(),
                        (double)g_cycles[k][w].load() / g_calls[k][w].load(), g_cycles[k][w].load() * 1e-9);
    }
}

void report()
{
    uint64_t jobs = 0, fwd = 0, inv = 0, fm = 0, im = 0, wc = 0, w = 0, sk = 0, di = 0, ph = 0, pa = 0, pc = 0, dsb = 0, dad = 0, lsb = 0, lad = 0, spc = 0, sph = 0, pss = 0, spi = 0, sih = 0;
    for (int i = 0; i < 64; i++)
    {
        jobs += g_count[i].jobs; fwd += g_count[i].fwd; inv += g_count[i].inv; fm += g_count[i].fwdMiss; im += g_count[i].invMiss;
        wc += g_count[i].waitCycles; w += g_count[i].waits; sk += g_count[i].skipped; di += g_count[i].dist; ph += g_count[i].psyHit; pa += g_count[i].psyAhead; pc += g_count[i].psyCoded;
        dsb += g_count[i].deadSub; dad += g_count[i].deadAdd; lsb += g_count[i].lateSub; lad += g_count[i].lateAdd; spc += g_count[i].spec; sph += g_count[i].specHit; pss += g_count[i].psySkip; spi += g_count[i].specInter; sih += g_count[i].specInterHit;
    }
    uint64_t devJobs = 0, starts = 0, ns = 0;
    for (int k = 0; k < g_nsvc.load(); k++)
    {
        uint64_t a = 0, b = 0, c = 0;
        if (g_svc[k].cs && !x265hip_cuserve_stats(g_svc[k].cs, &a, &b, &c)) { devJobs += a; starts += b; ns += c; }
    }
    fprintf(stderr, "x265hip: cuserve: %llu CU residual quad-trees (CU >= %d) handed to the GPU as jobs (%s%s, %.3f ms of device time%s): %llu forward "
                    "transform+quant units and %llu inverse units served, %llu + %llu calls of those CUs computed on the host; %llu waits of %.0f cycles on average; "
                    "%llu CUs not submitted%s\n",
            (unsigned long long)jobs, 1 << g_minLog2, g_mode ? "one launch per job" : (std::string("resident server of ") + std::to_string(2 * g_slots) + " workgroups = two per slot").c_str(),
            g_nsvc.load() > 1 ? (std::string(" at each of ") + std::to_string(g_nsvc.load()) + " places").c_str() : "", ns * 1e-6,
            g_mode ? "" : (std::string(", ") + std::to_string(starts) + " server starts").c_str(), (unsigned long long)fwd, (unsigned long long)inv,
            (unsigned long long)fm, (unsigned long long)im, (unsigned long long)w, w ? (double)wc / w : 0.0, (unsigned long long)sk,
            g_dead.load() ? "; THE DEVICE FAILED during the run, the rest was computed on the host" : "");
    fprintf(stderr, "x265hip: cuserve: %llu sse_pp and %llu psy-cost (source, reconstruction) answers out of the jobs; %llu psy-cost (source, prediction) values computed while waiting "
                    "for the device, %llu psy-cost calls answered from values remembered within their encodeResAndCalcRdInterCU\n", (unsigned long long)di, (unsigned long long)pc,
            (unsigned long long)pa, (unsigned long long)ph);
    {
        static const char* const site[6] = { "luma forward half", "chroma forward half", "luma inverse half", "chroma inverse half", "distortion / psy answers", "end of the CU's scope" };
        char line[512];
        int n = 0;
        for (int k = 0; k < 6; k++)
        {
            uint64_t ws = 0, cy = 0;
            for (int i = 0; i < 64; i++) { ws += g_count[i].siteWaits[k]; cy += g_count[i].siteCycles[k]; }
            n += snprintf(line + n, sizeof(line) - n, "%s%s %llu x %.0f", k ? ", " : "", site[k], (unsigned long long)ws, ws ? (double)cy / ws : 0.0);
        }
        if (w) fprintf(stderr, "x265hip: cuserve: waits by what was waited for (count x cycles): %s\n", line);
        if (w)
        {
            // the same waits by length: floor(log2(cycles)) -> count (a waiter that loses its CPU meanwhile — more runnable threads than CPUs, the quota's throttling — shows up
            // in the top buckets: time it did not spend spinning)
            char hl[512];
            int hn = 0;
            for (int b = 8; b < 24; b++)
            {
                uint64_t cnt = 0;
                for (int i = 0; i < 64; i++) cnt += g_count[i].lumaHist[b];
                if (b == 8) for (int bb = 0; bb < 8; bb++) for (int i = 0; i < 64; i++) cnt += g_count[i].lumaHist[bb];
                hn += snprintf(hl + hn, sizeof(hl) - hn, " 2^%d:%llu", b, (unsigned long long)cnt);
            }
            fprintf(stderr, "x265hip: cuserve: luma forward waits by length (floor(log2(cycles)) : count):%s\n", hl);
        }
        if (w && (g_lumaWait[0][0][0] || g_lumaWait[1][0][0]))
            fprintf(stderr, "x265hip: cuserve: luma forward waits by CU size and unit (count x cycles): 32x32 CU %llu x %.0f; 64x64 CU first unit %llu x %.0f, later units %llu x %.0f\n",
                    (unsigned long long)(g_lumaWait[0][0][0] + g_lumaWait[0][1][0]),
                    (double)(g_lumaWait[0][0][1] + g_lumaWait[0][1][1]) / (double)(g_lumaWait[0][0][0] + g_lumaWait[0][1][0] ? g_lumaWait[0][0][0] + g_lumaWait[0][1][0] : 1),
                    (unsigned long long)g_lumaWait[1][0][0], (double)g_lumaWait[1][0][1] / (double)(g_lumaWait[1][0][0] ? g_lumaWait[1][0][0].load() : 1),
                    (unsigned long long)g_lumaWait[1][1][0], (double)g_lumaWait[1][1][1] / (double)(g_lumaWait[1][1][0] ? g_lumaWait[1][1][0].load() : 1));
    }
    if (spc)
        fprintf(stderr, "x265hip: cuserve: %llu jobs left ahead of their scope, when the merge candidate's skip evaluation started; %llu of them were the job their "
                        "encodeResAndCalcRdInterCU wanted, %llu psy-costs of the skip evaluation served the tree as well\n", (unsigned long long)spc, (unsigned long long)sph,
                (unsigned long long)pss);
    if (spi)
        fprintf(stderr, "x265hip: cuserve: %llu jobs left ahead of their scope when predInterSearch returned the 2Nx2N inter candidate's prediction; %llu of them were the job their "
                        "encodeResAndCalcRdInterCU wanted\n", (unsigned long long)spi, (unsigned long long)sih);
    {
        uint64_t ij = 0, idr = 0;
        for (int i = 0; i < 64; i++) { ij += g_count[i].invJobs; idr += g_count[i].invDropped; }
        if (ij)
            fprintf(stderr, "x265hip: cuserve: %llu inverse jobs (luma 32x32 units of coefficient-mode CUs: dequant -> MFMA idct -> sse / psy energy behind Quant::rdoQuant's levels) "
                            "left when the levels were made, %llu never collected\n", (unsigned long long)ij, (unsigned long long)idr);
    }
    if (dsb || dad)
        fprintf(stderr, "x265hip: cuserve: %llu sub_ps and %llu add_ps calls of those CUs put off because only the job's answers read their results (%llu + %llu run after all)\n",
                (unsigned long long)dsb, (unsigned long long)dad, (unsigned long long)lsb, (unsigned long long)lad);
}

bool decide()
{
    std::lock_guard<std::mutex> g(g_lock);
    if (!g_state)
    {
        g_time = getenv("X265HIP_DEBUG_CUTIME") ? atoi(getenv("X265HIP_DEBUG_CUTIME")) : 0;
        if (g_time)
            atexit(report_time);
        const char* env = getenv("X265HIP_CUSERVE");
        const char* all = getenv("X265HIP");
        const char* table = getenv("X265HIP_TABLE");
        g_require = all && !strcmp(all, "require");
        g_verify = getenv("X265HIP_VERIFY") != NULL;
        if (getenv("X265HIP_CUSERVE_MIN")) { const int v = atoi(getenv("X265HIP_CUSERVE_MIN")); g_minLog2 = v >= 64 ? 6 : v >= 32 ? 5 : 4; }
        if (getenv("X265HIP_CUSERVE_MODE")) g_mode = atoi(getenv("X265HIP_CUSERVE_MODE")) ? 1 : 0;
        if (getenv("X265HIP_CUSERVE_YIELD")) g_yieldAfter = atoi(getenv("X265HIP_CUSERVE_YIELD"));
        if (getenv("X265HIP_CUSERVE_RDOQ")) g_rdoqJobs = atoi(getenv("X265HIP_CUSERVE_RDOQ")) ? 1 : 0;
        if (getenv("X265HIP_CUSERVE_INVERSE")) g_invJobs = atoi(getenv("X265HIP_CUSERVE_INVERSE")) ? 1 : 0;
        if (getenv("X265HIP_CUSERVE_TIMEOUT_MS") && atoll(getenv("X265HIP_CUSERVE_TIMEOUT_MS")) > 0) g_timeoutNs = atoll(getenv("X265HIP_CUSERVE_TIMEOUT_MS")) * 1000000ll;
        if (getenv("X265HIP_CUSERVE_SLOTS")) g_slots = atoi(getenv("X265HIP_CUSERVE_SLOTS"));
        else
        {
            // every slot is a resident workgroup that keeps a CU from the kernels that want a CU's whole LDS (the SAD surfaces): no more of them than jobs can
            // be in flight — a job belongs to a running thread or to one preempted in the middle of its CU; CPUs = affinity mask capped by the cgroup's quota
            cpu_set_t set;
            int cpus = sched_getaffinity(0, sizeof(set), &set) == 0 ? CPU_COUNT(&set) : 64;
            if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r"))
            {
                long long quota = 0, period = 0;
                char q[32];
                if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") && period > 0 && (quota = atoll(q)) > 0 && (quota + period - 1) / period < cpus)
                    cpus = (int)((quota + period - 1) / period);
                fclose(f);
            }
            // (round 5: three per CPU was tried when the SAO statistics jobs arrived — a thread in the SAO decision holds up to six slots — and lost: 32.0 fps against 33.7 with two per CPU)
            g_slots = 2 * cpus < 16 ? 16 : 2 * cpus > 64 ? 64 : 2 * cpus;
        }
        if (getenv("X265HIP_CUSERVE_SPEC")) g_spec = atoi(getenv("X265HIP_CUSERVE_SPEC")) != 0;
        if (getenv("X265HIP_CUSERVE_SPEC_INTER")) g_specInter = atoi(getenv("X265HIP_CUSERVE_SPEC_INTER")) != 0;
        g_serveDist = getenv("X265HIP_CUSERVE_DIST") ? atoi(getenv("X265HIP_CUSERVE_DIST")) : 3;
        if (g_slots < 1) g_slots = 1;
        if (g_slots > 256) g_slots = 256;
        if ((env && !strcmp(env, "0")) || (all && !strcmp(all, "0")) || (table && !strcmp(table, "percall")) || g_time == 1)
            g_state = -1;
        else
            g_state = 1;
    }
    return g_state > 0;
}
struct DecideAtLoad { DecideAtLoad() { decide(); } } g_decideAtLoad;        // the Quant seams read g_time / g_state without asking

// a device failure: said once, fatal under X265HIP=require, otherwise every later CU is computed by the reference's functions (SURVEY §8b "Errors")
void device_failed(const char* what)
{
    if (!g_dead.exchange(true))
        fprintf(stderr, "x265hip: cuserve: %s: %s — CU jobs are OFF from here on, the host computes\n", what, x265hip_last_error());
    if (g_require)
        abort();
}

void shutdown()
{
    // runs at exit before the bindings' device-time report (registered later, so it runs earlier): closing the services hands their device time to the ledger
    if (getenv("X265HIP_VERBOSE")) report();
    std::lock_guard<std::mutex> g(g_lock);
    g_dead = true;
    const int n = g_nsvc.exchange(0);
    for (int k = 0; k < n; k++)
        if (g_svc[k].cs) { x265hip_cuserve_close(g_svc[k].cs); g_svc[k].cs = NULL; }
}

// opens the services on first use
bool service()
{
    if (g_nsvc.load(std::memory_order_acquire)) return true;
    std::lock_guard<std::mutex> g(g_lock);
    if (g_nsvc.load()) return true;
    if (g_dead.load()) return false;
    if (x265hip_device_count() < 1) { g_dead = true; return false; }       // said by setupAssemblyPrimitives already
    const int places = x265hip_places_configured();
    const int n = places > 16 ? 16 : places > 0 ? places : 1;
    for (int k = 0; k < n; k++)
    {
        Service& sv = g_svc[k];
        const int e = places ? x265hip_cuserve_open_at(k, g_slots, g_mode, &sv.cs) : x265hip_cuserve_open(g_slots, g_mode, &sv.cs);
        bool ok = !e;
        for (int s = 0; ok && s < g_slots; s++)
            ok = !x265hip_cuserve_slot(sv.cs, s, &sv.mem[s].job, &sv.mem[s].pixels, &sv.mem[s].units, &sv.mem[s].levels, &sv.mem[s].resi);
        if (!ok)
        {
            for (int j = 0; j <= k; j++)
                if (g_svc[j].cs) { x265hip_cuserve_close(g_svc[j].cs); g_svc[j].cs = NULL; }
            device_failed("x265hip_cuserve_open");
            return false;
        }
    }
    atexit(shutdown);
    g_nsvc.store(n, std::memory_order_release);
    return true;
}

// a free slot for this thread's next job (at the service this thread is attached to: threads spread over the places), or -1
int take_slot(Service** svc)
{
    static std::atomic<int> next(0);
    const int nsvc = g_nsvc.load(std::memory_order_relaxed);
    if (nsvc < 1) return -1;
    if (t_hint < 0) t_hint = next.fetch_add(1) % (g_slots * nsvc);
    Service& sv = g_svc[(t_hint / g_slots) % nsvc];
    const int h = t_hint % g_slots;
    for (int k = 0; k < g_slots; k++)
    {
        const int s = (h + k) % g_slots;
        std::atomic<uint64_t>& w = sv.busy[s >> 6];
        const uint64_t bit = 1ull << (s & 63);
        if (!(w.load(std::memory_order_relaxed) & bit) && !(w.fetch_or(bit, std::memory_order_acquire) & bit))
        {
            t_hint = (t_hint / g_slots) * g_slots + s;
            *svc = &sv;
            return s;
        }
    }
    return -1;
}

// unit of this thread's job a residual block belongs to, or -1
inline int locate(const Job& j, const int16_t* residual, uint32_t resiStride, uint32_t log2TrSize, int ttype, int* elemOff)
{
    const int s = ttype ? (int)log2TrSize + 1 : (int)log2TrSize;
    if (s > j.sHi || s < j.sLo || resiStride != j.resiStride[ttype]) return -1;
    const ptrdiff_t d = residual - j.resi[ttype];
    const int N = (1 << j.log2CU) >> (ttype ? 1 : 0), n = 1 << log2TrSize;
    if (d < 0 || d >= (ptrdiff_t)resiStride * N) return -1;
    const int y = (int)(d / resiStride), x = (int)(d % resiStride);
    if (x >= N || (x & (n - 1)) || (y & (n - 1))) return -1;
    *elemOff = x265hipi_cujob_elem_offset(j.job, j.sHi, s, ttype, x >> log2TrSize, y >> log2TrSize);
    return x265hipi_cujob_unit_index(j.job, j.sHi, s, ttype, x >> log2TrSize, y >> log2TrSize);
}

inline void flush_sub(Job& j)
{
    for (int p = 0; p < 3; p++)
        if (j.pendSub[p].pending)
        {
            Job::PendSub& s = j.pendSub[p];
            s.pending = false;
            s.fn(s.dst, s.ds, s.a, s.b, s.sa, s.sb);
            counters().lateSub.fetch_add(1, std::memory_order_relaxed);
        }
}
inline void flush_add(Job& j, int u)
{
    Job::PendAdd& a = j.pendAdd[u];
    if (a.pending)
    {
        a.pending = false;
        a.fn(a.dst, a.ds, a.a, a.b, a.sa, a.sb);
        counters().lateAdd.fetch_add(1, std::memory_order_relaxed);
    }
}
// the job stops answering (device failure): everything put off is run now, the reference's body goes on as if the job had never been
inline void abandon(Job& j)
{
    flush_sub(j);
    if (j.anyPendAdd)
        for (int u = 0; u < X265HIP_CUJOB_MAX_UNITS; u++) flush_add(j, u);
    j.anyPendAdd = false;
    j.active = false;
    j.inv.active = false;                // (a pending inverse job's slot is kept: the device may still write into it)
}

// waits for a ready word of this thread's job to take the job's ticket; false: the device did not deliver (the job is abandoned)
// site: 0 luma forward half (transformNxN), 1 chroma forward half, 2 luma inverse half (invtransformNxN), 3 chroma inverse half, 4 a distortion / psy answer,
// 5 the end of the job's scope (every unit's inverse half, so that the slot can go back)
inline bool wait_word(Job& j, const uint32_t* ready, int site)
{
    if (__atomic_load_n(ready, __ATOMIC_ACQUIRE) == j.seq) return true;
    const uint64_t t0 = __builtin_ia32_rdtsc();
    uint64_t spins = 0;
    int64_t waitedNs = 0, lastNs = -1;
    while (__atomic_load_n(ready, __ATOMIC_ACQUIRE) != j.seq)
    {
        __builtin_ia32_pause();
        // X265HIP_CUSERVE_YIELD=n: after n polls the waiter gives its CPU away between polls (x265 starts a pool thread per core it SEES, the box gives the
        // process 16 CPUs: a spinning waiter may be keeping a runnable row from running)
        if (g_yieldAfter && spins >= (uint64_t)g_yieldAfter) sched_yield();
        if ((++spins & 255) == 0)
        {
            // Not a latency, a failure — but only time the device could have used counts (wall clock; x265hip_cuserve_poke says 1 while the servers are
            // paused for somebody's hipFree or the server is still on its way onto the chip), the bound is generous (X265HIP_CUSERVE_TIMEOUT_MS, default 10 s),
            // and one late job costs that job only: the third one in an encode switches the offload off
            const int pk = x265hip_cuserve_poke(j.svc->cs, j.slot);
            timespec ts;
            clock_gettime(CLOCK_MONOTONIC, &ts);
            const int64_t nowNs = (int64_t)ts.tv_sec * 1000000000ll + ts.tv_nsec;
            if (pk == 0 && lastNs >= 0) waitedNs += nowNs - lastNs;
            lastNs = nowNs;
            if (pk < 0 || waitedNs > g_timeoutNs)
            {
                abandon(j);
                if (pk < 0 || g_lateJobs.fetch_add(1) + 1 >= 3)
                    device_failed("a job did not come back");
                else
                    fprintf(stderr, "x265hip: cuserve: a job did not come back within %lld ms; this CU is computed on the host\n", (long long)(g_timeoutNs / 1000000));
                return false;
            }
        }
    }
    Counters& c = counters();
    const uint64_t dt = __builtin_ia32_rdtsc() - t0;
    c.waitCycles.fetch_add(dt, std::memory_order_relaxed);
    c.waits.fetch_add(1, std::memory_order_relaxed);
    c.siteCycles[site].fetch_add(dt, std::memory_order_relaxed);
    c.siteWaits[site].fetch_add(1, std::memory_order_relaxed);
    if (!site) { int b = 63 - __builtin_clzll(dt | 1); c.lumaHist[b > 23 ? 23 : b].fetch_add(1, std::memory_order_relaxed); }
    return true;
}



} // namespace cusvc

using namespace cusvc;
namespace {

// the header of the job this CU's residual quad-tree is; false: not a CU the device serves
bool make_header(Search* se, const Mode& mode, uint32_t log2CUSize, const uint32_t depthRange[2], x265hip_cujob& hdr)
{
    const CUData& cu = mode.cu;
    const Quant& q = se->m_quant;
    const int csp = se->m_csp;
    const bool codeChroma = csp != X265_CSP_I400 && se->m_frame->m_fencPic->m_picCsp != X265_CSP_I400;
    // RDOQ (presets slow / slower): the quantiser is Quant::rdoQuant and stays on the host (its decisions read the entropy coder's state); the job carries
    // the transforms in front of it — coefficient mode (X265HIP_CUSERVE_RDOQ=0 switches it off)
    if (cu.m_tqBypass[0] || (q.m_rdoqLevel && !g_rdoqJobs) || (q.m_nr && q.m_nr->offset) || q.m_scalingList->m_bEnabled || (csp != X265_CSP_I420 && csp != X265_CSP_I400) ||
        (csp == X265_CSP_I420) != codeChroma)
        return false;
    memset(&hdr, 0, sizeof(hdr));
    hdr.log2CUSize = log2CUSize; hdr.log2TrMax = depthRange[1]; hdr.log2TrMin = depthRange[0];
    hdr.chroma = codeChroma; hdr.bitDepth = X265_DEPTH;
    hdr.quantOffset = cu.m_slice->m_sliceType == I_SLICE ? 171 : 85;
    hdr.signHide = cu.m_slice->m_pps->bSignHideEnabled;
    hdr.reserved = 0;
    for (int p = 0; p < 3; p++)
    {
        const QpParam& qp = q.m_qpParam[p];
        hdr.qpRem[p] = qp.rem; hdr.qpPer[p] = qp.per;
        hdr.quantScale[p] = q.m_scalingList->m_quantCoef[3][3 + p][qp.rem][0];          // flat: every entry of every size and list is s_quantScales[rem]
        hdr.dequantScale[p] = ScalingList::s_invQuantScales[qp.rem];
    }
    hdr.coefMode = q.m_rdoqLevel ? 1 : 0;
    hdr.sourceDct = q.m_rdoqLevel && q.m_psyRdoqScale ? 1 : 0;
    return true;
}

// the job of one CU: header + pixels into this thread's slot, submit
// ---- inverse jobs behind Quant::rdoQuant -------------------------------------------------------------------------------------------------------------------
// waits for the pending inverse job of this thread's job (its own slot, its own ticket); false: the device did not deliver
bool inv_wait(Job& j)
{
    Job::InvAhead& a = j.inv;
    const uint32_t* ready = &a.units[0].readyInv;
    if (__atomic_load_n(ready, __ATOMIC_ACQUIRE) == a.seq) return true;
    const uint64_t t0 = __builtin_ia32_rdtsc();
    uint64_t spins = 0;
    int64_t waitedNs = 0, lastNs = -1;
    while (__atomic_load_n(ready, __ATOMIC_ACQUIRE) != a.seq)
    {
        __builtin_ia32_pause();
        if ((++spins & 255) == 0)
        {
            const int pk = x265hip_cuserve_poke(a.svc->cs, a.slot);
            timespec ts;
            clock_gettime(CLOCK_MONOTONIC, &ts);
            const int64_t nowNs = (int64_t)ts.tv_sec * 1000000000ll + ts.tv_nsec;
            if (pk == 0 && lastNs >= 0) waitedNs += nowNs - lastNs;
            lastNs = nowNs;
            if (pk < 0 || waitedNs > g_timeoutNs)
            {
                a.active = false;                        // the slot is kept: the device may still write into it
                device_failed("an inverse job did not come back");
                return false;
            }
        }
    }
    Counters& c = counters();
    const uint64_t dt = __builtin_ia32_rdtsc() - t0;
    c.waitCycles.fetch_add(dt, std::memory_order_relaxed);
    c.waits.fetch_add(1, std::memory_order_relaxed);
    c.siteCycles[2].fetch_add(dt, std::memory_order_relaxed);
    c.siteWaits[2].fetch_add(1, std::memory_order_relaxed);
    return true;
}
// a pending inverse job nobody collected (the tree did not ask for the unit's inverse, or asked with other levels): its slot goes back when the device is done
void inv_drop(Job& j)
{
    if (!j.inv.active)
        return;
    if (inv_wait(j))
        give_slot(j.inv.svc, j.inv.slot);
    j.inv.active = false;
    counters().invDropped.fetch_add(1, std::memory_order_relaxed);
}
// the levels of luma unit `u` (32x32, at (x, y) of the CU) have just been made by Quant::rdoQuant: its inverse half leaves as a job of its own
void inv_submit(Job& j, int u, int x, int y, const coeff_t* coeff)
{
    if (!g_invJobs || g_dead.load(std::memory_order_relaxed))
        return;
    inv_drop(j);
    Service* svc = NULL;
    const int slot = take_slot(&svc);
    if (slot < 0)
        return;
    const int N = 1 << j.log2CU, planeElems = j.hdr.chroma ? N * N + N * N / 2 : N * N;
    // the unit's source block, its prediction (out of this thread's copy of what the CU job was given), then its levels: one front-to-back copy into the mailbox
    alignas(64) unsigned char staged[2 * 1024 * sizeof(pixel) + 2048];
    pixel* dst = reinterpret_cast<pixel*>(staged);
    pack_rows(dst, j.sent + (size_t)y * N + x, (uint32_t)N, 32);
    pack_rows(dst, j.sent + planeElems + (size_t)y * N + x, (uint32_t)N, 32);
    memcpy(dst, coeff, 2048);
    x265hip_cujob hdr = j.hdr;
    hdr.log2CUSize = 5; hdr.log2TrMax = 5; hdr.log2TrMin = 5; hdr.chroma = 0; hdr.coefMode = X265HIP_CUJOB_INVERSE; hdr.sourceDct = 0; hdr.reserved = 0;
    *svc->mem[slot].job = hdr;
    memcpy(svc->mem[slot].pixels, staged, sizeof(staged));
    uint32_t seq = 0;
    if (x265hip_cuserve_submit(svc->cs, slot, &seq))
    {
        give_slot(svc, slot);
        device_failed("x265hip_cuserve_submit (inverse job)");
        return;
    }
    Job::InvAhead& a = j.inv;
    a.unit = u; a.svc = svc; a.slot = slot; a.seq = seq; a.units = svc->mem[slot].units; a.resi = svc->mem[slot].resi;
    memcpy(a.sent, coeff, 2048);
    a.active = true;
    counters().invJobs.fetch_add(1, std::memory_order_relaxed);
}

bool submit(Search* se, Mode& mode, uint32_t log2CUSize, ShortYuv& resiYuv, const uint32_t depthRange[2])
{
    const Quant& q = se->m_quant;
    x265hip_cujob hdr;
    if (!make_header(se, mode, log2CUSize, depthRange, hdr))
        return false;
    const bool codeChroma = hdr.chroma != 0;
    int sHi, sLo;
    if (x265hipi_cujob_levels(&hdr, &sHi, &sLo) < 1 || !service())
        return false;
    Service* svc = NULL;
    const int slot = take_slot(&svc);
    if (slot < 0)
        return false;
    const SlotMem& mem = svc->mem[slot];
    const int N = 1 << log2CUSize;
    const Yuv* fenc = mode.fencYuv;
    const Yuv* pred = &mode.predYuv;
    *mem.job = hdr;
    Job& j = t_job;
    // packed in this thread's own memory first (what encodeResAndCalcRdInterCU compares a job submitted ahead with), then one front-to-back copy into the
    // mailbox — device memory behind a write-combining mapping likes that better than 16- and 32-byte rows anyway
    pixel* dst = j.sent;
    pack_rows(dst, fenc->m_buf[0], fenc->m_size, N);
    if (codeChroma) { pack_rows(dst, fenc->m_buf[1], fenc->m_csize, N / 2); pack_rows(dst, fenc->m_buf[2], fenc->m_csize, N / 2); }
    pack_rows(dst, pred->m_buf[0], pred->m_size, N);
    if (codeChroma) { pack_rows(dst, pred->m_buf[1], pred->m_csize, N / 2); pack_rows(dst, pred->m_buf[2], pred->m_csize, N / 2); }
    memcpy(mem.pixels, j.sent, (size_t)(dst - j.sent) * sizeof(pixel));
    if (x265hip_cuserve_submit(svc->cs, slot, &j.seq))
    {
        give_slot(svc, slot);
        device_failed("x265hip_cuserve_submit");
        return false;
    }
    j.quant = &q;
    j.search = se;
    j.mode = &mode;
    j.phase = 1;
    j.fencYuv = fenc;
    j.hdr = hdr;
    for (int p = 0; p < 3; p++)
    {
        j.resi[p] = resiYuv.m_buf[p]; j.resiStride[p] = p ? resiYuv.m_csize : resiYuv.m_size;
        j.fenc[p] = fenc->m_buf[p]; j.fencStride[p] = p ? fenc->m_csize : fenc->m_size;
        j.pred[p] = pred->m_buf[p]; j.predStride[p] = p ? pred->m_csize : pred->m_size;
    }
    if (!codeChroma) { j.resi[1] = j.resi[2] = NULL; j.fenc[1] = j.fenc[2] = j.pred[1] = j.pred[2] = NULL; }
    j.psyRd = se->m_rdCost.m_psyRd != 0;
    memset(j.invServed, 0, sizeof(j.invServed));
    memset(j.energyKnown, 0, sizeof(j.energyKnown));
    j.pendSub[0].pending = j.pendSub[1].pending = j.pendSub[2].pending = false;
    if (j.anyPendAdd) for (int u = 0; u < X265HIP_CUJOB_MAX_UNITS; u++) j.pendAdd[u].pending = false;
    j.anyPendAdd = false;
    j.treeMine = false;
    j.inTree = true;
    inv_drop(j);
    j.log2CU = log2CUSize; j.sHi = sHi; j.sLo = sLo; j.slot = slot; j.svc = svc;
    j.job = mem.job; j.units = mem.units; j.levels = mem.levels; j.resiOut = mem.resi;
    j.active = true;
    counters().jobs.fetch_add(1, std::memory_order_relaxed);
    return true;
}

// the job's scope ends: the slot goes back when the device has written everything it is going to write into it (an abandoned job: the device is
// dead, the slot is kept)
void end_job()
{
    Job& j = t_job;
    inv_drop(j);
    bool done = j.active;
    if (done && !j.treeMine)
        flush_sub(j);                  // the tree this job was made for never ran: whoever runs instead reads the residual
    j.pendSub[0].pending = j.pendSub[1].pending = j.pendSub[2].pending = false;
    if (done)
    {
        const int last = x265hipi_cujob_unit_index(j.job, j.sHi, j.sLo, j.resi[1] ? 2 : 0, (1 << (j.log2CU - j.sLo)) - 1, (1 << (j.log2CU - j.sLo)) - 1);
        for (int u = 0; u <= last && done; u++)
            done = wait_word(j, &j.units[u].readyInv, 5);
    }
    j.active = false;
    j.inTree = false;
    if (done) give_slot(j.svc, j.slot);
}

// ---- what the job knows about (source, other) blocks ---------------------------------------------------------------------------------------------
// the unit (or, for a block of the CU's own size above the largest transform, the units) a block of the mode's source Yuv covers
struct Where { int plane, x, y, n, s; };
inline bool where_in_source(const Job& j, const pixel* src, intptr_t stride, int n, Where& w)
{
    for (int p = 0; p < 3; p++)
    {
        if (!j.fenc[p] || (uint32_t)stride != j.fencStride[p]) continue;
        const ptrdiff_t d = src - j.fenc[p];
        const int N = (1 << j.log2CU) >> (p ? 1 : 0);
        if (d < 0 || d >= (ptrdiff_t)stride * N) continue;
        const int y = (int)(d / stride), x = (int)(d % stride);
        if (x >= N || (x & (n - 1)) || (y & (n - 1)) || x + n > N || y + n > N) return false;
        int lg = 0;
        while ((1 << lg) < n) lg++;
        w.plane = p; w.x = x; w.y = y; w.n = n; w.s = p ? lg + 1 : lg;
        return true;
    }
    return false;
}

// sse_pp(source block, other block) out of the job, or false
inline bool job_sse(Job& j, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb, int n, uint64_t& out)
{
    Where w;
    if (!where_in_source(j, a, sa, n, w)) return false;
    const bool vsPred = (uint32_t)sb == j.predStride[w.plane] && b == j.pred[w.plane] + (size_t)w.y * sb + w.x;
    if (vsPred)
    {
        // the unit itself, or — a block above the largest transform size (a 64x64 CU) — the sum over the units it is made of (sse is additive)
        const int s = w.s > j.sHi ? j.sHi : w.s;
        if (s < j.sLo) return false;
        const int k = 1 << (w.s - s), sh = w.plane ? s - 1 : s;
        uint64_t sum = 0;
        for (int ty = 0; ty < k; ty++)
            for (int tx = 0; tx < k; tx++)
            {
                const int u = x265hipi_cujob_unit_index(j.job, j.sHi, s, w.plane, (w.x >> sh) + tx, (w.y >> sh) + ty);
                if (!wait_word(j, &j.units[u].ready, 4)) return false;
                sum += j.units[u].zeroDist;
            }
        out = sum;
        return true;
    }
    if (w.s > j.sHi || w.s < j.sLo) return false;
    // against the tree's reconstruction of the unit (search.cpp:3293-3295: add_ps(recon, pred, inverse-transformed residual), then sse_pp(source, recon))
    const Yuv& rq = j.search->m_rqt[w.s - 2].reconQtYuv;
    const uint32_t rs = w.plane ? rq.m_csize : rq.m_size;
    if ((uint32_t)sb != rs || b != rq.m_buf[w.plane] + (size_t)w.y * rs + w.x) return false;
    const int sh = w.plane ? w.s - 1 : w.s;
    const int u = x265hipi_cujob_unit_index(j.job, j.sHi, w.s, w.plane, w.x >> sh, w.y >> sh);
    if (!j.invServed[u] || !wait_word(j, &j.units[u].readyInv, 4)) { flush_add(j, u); return false; }
    out = j.units[u].codedDist;
    return true;
}

// After the tree, encodeResAndCalcRdInterCU measures the CU's FINAL reconstruction (search.cpp:2940-2956: reconYuv = clip(pred + the residual the tree
// kept), then sse_pp and psyCost of the whole CU against it).  With one transform size in the job that reconstruction is, unit by unit, either the
// tree's coded reconstruction (the unit's cbf survived) or the prediction (cbf 0): the job and this thread know both answers.  `coded` / `zero` pick
// the per-unit values; false when anything is missing.
template <typename F> inline bool final_sum(Job& j, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb, int n, F unit_value, int64_t& out)
{
    Where w;
    // "one transform size in the job" must also mean "one transform size in the TREE": x265hipi_cujob_levels clamps the job's smallest size to 16, the tree
    // may go below it (--max-tu-size 16 --tu-inter-depth 3: depthRange [3, 4]) and a unit split further has neither the job's coded nor its zero answer
    if (g_serveDist < 2 || j.inTree || j.sHi != j.sLo || (int)j.job->log2TrMin < j.sLo || !where_in_source(j, a, sa, n, w) || w.x || w.y) return false;
    const int N = (1 << j.log2CU) >> (w.plane ? 1 : 0);
    if (n != N) return false;
    const Yuv& ry = j.mode->reconYuv;
    if ((uint32_t)sb != (w.plane ? ry.m_csize : ry.m_size) || b != ry.m_buf[w.plane]) return false;
    const int s = j.sHi, k = 1 << (j.log2CU - s), sh = w.plane ? s - 1 : s, tuDepth = (int)j.log2CU - s;
    const CUData& cu = j.mode->cu;
    int64_t sum = 0;
    for (int ty = 0; ty < k; ty++)
        for (int tx = 0; tx < k; tx++)
        {
            const int u = x265hipi_cujob_unit_index(j.job, j.sHi, s, w.plane, tx, ty);
            const uint32_t absPartIdx = g_rasterToZscan[((ty << s) >> 2) * 16 + ((tx << s) >> 2)];
            const bool cbf = (cu.m_cbf[w.plane][absPartIdx] >> tuDepth) & 1;
            int64_t v;
            if (!unit_value(u, cbf, v)) return false;
            sum += v;
        }
    (void)sh;
    out = sum;
    return true;
}
inline bool final_sse(Job& j, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb, int n, uint64_t& out)
{
    int64_t v;
    if (!final_sum(j, a, sa, b, sb, n, [&j](int u, bool cbf, int64_t& val) {
            if (cbf) { if (!j.invServed[u] || !wait_word(j, &j.units[u].readyInv, 4)) return false; val = (int64_t)j.units[u].codedDist; }
            else { if (!wait_word(j, &j.units[u].ready, 4)) return false; val = (int64_t)j.units[u].zeroDist; }
            return true; }, v))
        return false;
    out = (uint64_t)v;
    return true;
}
inline bool final_psy(Job& j, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb, int n, int& out)
{
    int64_t v;
    if (!final_sum(j, a, sa, b, sb, n, [&j](int u, bool cbf, int64_t& val) {
            if (cbf) { if (!j.invServed[u] || !wait_word(j, &j.units[u].readyInv, 4)) return false; val = (int64_t)j.units[u].codedEnergy; }
            else { if (!j.energyKnown[u]) return false; val = j.energy[u]; }
            return true; }, v))
        return false;
    out = (int)v;
    return true;
}

template <int CU, int N, bool CHROMA> sse_t sse_slot(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb)
{
    Job& j = t_job;

    if (j.active && j.phase)
    {
        uint64_t v;
        if (job_sse(j, a, sa, b, sb, N, v) || final_sse(j, a, sa, b, sb, N, v))
        {
            if (g_verify)
            {
                const sse_t want = CHROMA ? g_prev.chroma[X265_CSP_I420].cu[CU].sse_pp(a, sa, b, sb) : g_prev.cu[CU].sse_pp(a, sa, b, sb);
                if ((uint64_t)want != v) { fprintf(stderr, "x265hip: cuserve: VERIFY FAILED sse_pp %dx%d: %llu (job) vs %llu\n", N, N, (unsigned long long)v, (unsigned long long)want); abort(); }
            }
            counters().dist.fetch_add(1, std::memory_order_relaxed);
            return (sse_t)v;
        }
    }
    return CHROMA ? g_prev.chroma[X265_CSP_I420].cu[CU].sse_pp(a, sa, b, sb) : g_prev.cu[CU].sse_pp(a, sa, b, sb);
}

#if X265_DEPTH > 8
// Main10 / Main12: pixel == uint16_t and x265_setup_primitives aliases the luma cu[].sse_pp to cu[].sse_ss AFTER setupAssemblyPrimitives has run
// (setupAliasPrimitives, primitives.cpp:90-94) — a wrapper on sse_pp would be overwritten.  The slot the encoder really calls is sse_ss: wrapped here; a call
// that is not a question about this thread's job (int16 residual pairs included) goes to the function that was there.
template <int CU, int N> sse_t sse_ss_slot(const int16_t* a, intptr_t sa, const int16_t* b, intptr_t sb)
{
    Job& j = t_job;
    if (j.active && j.phase)
    {
        uint64_t v;
        if (job_sse(j, (const pixel*)a, sa, (const pixel*)b, sb, N, v) || final_sse(j, (const pixel*)a, sa, (const pixel*)b, sb, N, v))
        {
            if (g_verify && (uint64_t)g_prev.cu[CU].sse_ss(a, sa, b, sb) != v) { fprintf(stderr, "x265hip: cuserve: VERIFY FAILED sse_pp (as sse_ss) %dx%d\n", N, N); abort(); }
            counters().dist.fetch_add(1, std::memory_order_relaxed);
            return (sse_t)v;
        }
    }
    return g_prev.cu[CU].sse_ss(a, sa, b, sb);
}
#endif

// psy_cost_pp(source block, prediction block): remembered per unit within the job's scope; a block above the largest transform size is the sum of its
// units' values (psyCost_pp sums over 8x8 blocks, pixel.cpp:739-748)
template <int CU, int N> int psy_slot(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb)
{
    Job& j = t_job;
    Where w;
    int fin;
    if (j.active && !j.phase)
    {
        // inside the encodeResAndCalcRdSkipCU the job was submitted from: its psy-cost of (source, reconstruction = a copy of the prediction,
        // search.cpp:2786) is the value the residual quad-tree asks for again as (source, prediction) of the CU's only luma unit (search.cpp:3290)
        if (N == (1 << j.log2CU) && j.sHi == (int)j.log2CU && j.psyRd && a == j.fenc[0] && (uint32_t)sa == j.fencStride[0] && b == j.mode->reconYuv.m_buf[0] &&
            (uint32_t)sb == j.mode->reconYuv.m_size)
        {
            const int v = g_prev.cu[CU].psy_cost_pp(a, sa, b, sb);
            const int u = x265hipi_cujob_unit_index(j.job, j.sHi, j.sHi, 0, 0, 0);
            j.energy[u] = v; j.energyKnown[u] = 2;
            return v;
        }
        return g_prev.cu[CU].psy_cost_pp(a, sa, b, sb);
    }
    if (j.active && N >= 8 && final_psy(j, a, sa, b, sb, N, fin))
    {
        if (g_verify && g_prev.cu[CU].psy_cost_pp(a, sa, b, sb) != fin) { fprintf(stderr, "x265hip: cuserve: VERIFY FAILED psy_cost_pp(source, final reconstruction) %dx%d\n", N, N); abort(); }
        counters().psyCoded.fetch_add(1, std::memory_order_relaxed);
        return fin;
    }
    if (j.active && N >= 8 && where_in_source(j, a, sa, N, w) && w.s <= j.sHi && w.s >= j.sLo)
    {
        // against the tree's reconstruction of a unit whose residual came from the device (search.cpp:3299, :3373): the job measured it
        const Yuv& rq = j.search->m_rqt[w.s - 2].reconQtYuv;
        const uint32_t rs = w.plane ? rq.m_csize : rq.m_size;
        if ((uint32_t)sb == rs && b == rq.m_buf[w.plane] + (size_t)w.y * rs + w.x)
        {
            const int sh = w.plane ? w.s - 1 : w.s;
            const int u = x265hipi_cujob_unit_index(j.job, j.sHi, w.s, w.plane, w.x >> sh, w.y >> sh);
            if (j.invServed[u] && wait_word(j, &j.units[u].readyInv, 4))
            {
                const int v = (int)j.units[u].codedEnergy;
                if (g_verify && g_prev.cu[CU].psy_cost_pp(a, sa, b, sb) != v) { fprintf(stderr, "x265hip: cuserve: VERIFY FAILED psy_cost_pp(source, reconstruction) %dx%d\n", N, N); abort(); }
                counters().psyCoded.fetch_add(1, std::memory_order_relaxed);
                return v;
            }
            flush_add(j, u);            // the reconstruction is read on the host after all
        }
    }
    if (j.active && N >= 8 && where_in_source(j, a, sa, N, w) && (uint32_t)sb == j.predStride[w.plane] && b == j.pred[w.plane] + (size_t)w.y * sb + w.x)
    {
        const int s = w.s > j.sHi ? j.sHi : w.s;
        if (s >= j.sLo)
        {
            const int k = 1 << (w.s - s), sh = w.plane ? s - 1 : s;
            bool all = true;
            int64_t sum = 0;
            for (int ty = 0; ty < k && all; ty++)
                for (int tx = 0; tx < k && all; tx++)
                {
                    const int u = x265hipi_cujob_unit_index(j.job, j.sHi, s, w.plane, (w.x >> sh) + tx, (w.y >> sh) + ty);
                    all = j.energyKnown[u];
                    sum += j.energy[u];
                }
            if (all)
            {
                if (g_verify && g_prev.cu[CU].psy_cost_pp(a, sa, b, sb) != (int)sum) { fprintf(stderr, "x265hip: cuserve: VERIFY FAILED psy_cost_pp %dx%d\n", N, N); abort(); }
                counters().psyHit.fetch_add(1, std::memory_order_relaxed);
                return (int)sum;
            }
            const int v = g_prev.cu[CU].psy_cost_pp(a, sa, b, sb);
            if (k == 1)
            {
                const int u = x265hipi_cujob_unit_index(j.job, j.sHi, s, w.plane, w.x >> sh, w.y >> sh);
                j.energy[u] = v; j.energyKnown[u] = 1;
            }
            return v;
        }
    }
    return g_prev.cu[CU].psy_cost_pp(a, sa, b, sb);
}


// the CU's residual = source - prediction (encodeResAndCalcRdInterCU, search.cpp:2838 through ShortYuv::subtract): its only readers are the tree's
// transformNxN calls, which the job serves from its own copy of source and prediction.  Put off (flush_sub) instead of run.
template <int CU, int N, bool CHROMA> void sub_ps_slot(int16_t* dst, intptr_t ds, const pixel* a, const pixel* b, intptr_t sa, intptr_t sb)
{
    Job& j = t_job;
    const pixel_sub_ps_t fn = CHROMA ? g_prev.chroma[X265_CSP_I420].cu[CU].sub_ps : g_prev.cu[CU].sub_ps;
    if (j.active && !j.inTree && !j.treeMine && t_inEncodeRes && g_serveDist >= 3 && !g_verify && (N << (CHROMA ? 1 : 0)) == (1 << j.log2CU))
        for (int p = CHROMA ? 1 : 0; p < (CHROMA ? 3 : 1); p++)
            if (dst == j.resi[p] && (uint32_t)ds == j.resiStride[p] && a == j.fenc[p] && (uint32_t)sa == j.fencStride[p] && b == j.pred[p] && (uint32_t)sb == j.predStride[p])
            {
                j.pendSub[p] = Job::PendSub{ true, dst, ds, a, b, sa, sb, fn };
                counters().deadSub.fetch_add(1, std::memory_order_relaxed);
                return;
            }
    fn(dst, ds, a, b, sa, sb);
}

// a unit's reconstruction in the tree = prediction + the residual the device sent (search.cpp:3314 luma, :3433 chroma): its only readers are the
// sse_pp and psy-cost calls right after, which the job answers.  Put off (flush_add) instead of run.
template <int CU, int N, int AL> void add_ps_slot(pixel* dst, intptr_t ds, const pixel* a, const int16_t* b, intptr_t sa, intptr_t sb)
{
    Job& j = t_job;
    // ... which holds only while the table really carries the answering slots (see x265hip_install_cuserve_slots about two set-ups at once)
    bool answered = primitives.cu[CU].psy_cost_pp == psy_slot<CU, N> && primitives.cu[CU].sse_pp == sse_slot<CU, N, false>;
#if X265_DEPTH > 8
    answered = primitives.cu[CU].psy_cost_pp == psy_slot<CU, N> &&
               (primitives.cu[CU].sse_pp == sse_slot<CU, N, false> || primitives.cu[CU].sse_pp == (pixel_sse_t)sse_ss_slot<CU, N>);
#endif
    if (answered && j.active && j.inTree && g_serveDist >= 3 && !g_verify && !j.search->m_rdCost.m_ssimRd)
        for (int p = 0; p < 3; p++)
        {
            if (!j.pred[p] || (uint32_t)sa != j.predStride[p]) continue;
            const ptrdiff_t d = a - j.pred[p];
            const int W = (1 << j.log2CU) >> (p ? 1 : 0);
            if (d < 0 || d >= (ptrdiff_t)sa * W) continue;
            const int y = (int)(d / sa), x = (int)(d % sa);
            if (x >= W || (x & (N - 1)) || (y & (N - 1))) break;
            int lg = 0;
            while ((1 << lg) < N) lg++;
            const int s = p ? lg + 1 : lg;
            if (s > j.sHi || s < j.sLo) break;
            const Yuv& rq = j.search->m_rqt[s - 2].reconQtYuv;
            const ShortYuv& rs = j.search->m_rqt[s - 2].resiQtYuv;
            const uint32_t rqs = p ? rq.m_csize : rq.m_size, rss = p ? rs.m_csize : rs.m_size;
            if ((uint32_t)ds != rqs || dst != rq.m_buf[p] + (size_t)y * rqs + x || (uint32_t)sb != rss || b != rs.m_buf[p] + (size_t)y * rss + x) break;
            const int u = x265hipi_cujob_unit_index(j.job, j.sHi, s, p, x >> lg, y >> lg);
            if (!j.invServed[u]) break;
            j.pendAdd[u] = Job::PendAdd{ true, dst, ds, a, b, sa, sb, g_prev.cu[CU].add_ps[AL] };
            j.anyPendAdd = true;
            counters().deadAdd.fetch_add(1, std::memory_order_relaxed);
            return;
        }
    g_prev.cu[CU].add_ps[AL](dst, ds, a, b, sa, sb);
}

// while this thread waits for unit u's forward half: the psy-cost of (source, prediction) of the same unit, which the tree asks for right after
// (search.cpp:3272 luma, :3345 chroma) and encodeResAndCalcRdInterCU once more (:2888)
inline void psy_ahead(Job& j, int u, int plane, int x, int y, int n)
{
    if (!j.psyRd || !g_slots_installed || j.energyKnown[u] || n < 8) return;
    const pixel* a = j.fenc[plane] + (size_t)y * j.fencStride[plane] + x;
    const pixel* b = j.pred[plane] + (size_t)y * j.predStride[plane] + x;
    const int cu = n == 8 ? BLOCK_8x8 : n == 16 ? BLOCK_16x16 : BLOCK_32x32;
    j.energy[u] = g_prev.cu[cu].psy_cost_pp(a, j.fencStride[plane], b, j.predStride[plane]);
    j.energyKnown[u] = 1;
    counters().psyAhead.fetch_add(1, std::memory_order_relaxed);
}

} // namespace

// called by setupAssemblyPrimitives in the default table mode, after the psy lookups of x265_hip_srcplanes.cpp are in the table
// X265HIP_DEBUG_CUTIME: the transform primitives themselves, timed (what of Quant::transformNxN / ::invtransformNxN is transform and what is the quantiser —
// with RDOQ the quantiser is Quant::rdoQuant and stays on the host, so this is the share a job could take over there)
dct_t g_timedDct[4]; idct_t g_timedIdct[4];
std::atomic<uint64_t> g_primCycles[8][2], g_primCalls[8][2];
template <int S> void dct_timed(const int16_t* src, int16_t* dst, intptr_t stride)
{
    const uint64_t t0 = __builtin_ia32_rdtsc();
    g_timedDct[S](src, dst, stride);
    g_primCycles[S][!t_inRqt].fetch_add(__builtin_ia32_rdtsc() - t0, std::memory_order_relaxed);
    g_primCalls[S][!t_inRqt].fetch_add(1, std::memory_order_relaxed);
}
template <int S> void idct_timed(const int16_t* src, int16_t* dst, intptr_t stride)
{
    const uint64_t t0 = __builtin_ia32_rdtsc();
    g_timedIdct[S](src, dst, stride);
    g_primCycles[4 + S][!t_inRqt].fetch_add(__builtin_ia32_rdtsc() - t0, std::memory_order_relaxed);
    g_primCalls[4 + S][!t_inRqt].fetch_add(1, std::memory_order_relaxed);
}
void report_prim_time()
{
    for (int k = 0; k < 8; k++)
        for (int w = 0; w < 2; w++)
            if (g_primCalls[k][w].load())
                fprintf(stderr, "x265hip: cutime: %-52s %2dx%-2d %s: %9llu calls, %8.0f cycles each, %7.3f G cycles\n", k < 4 ? "cu[].dct (the primitive alone)" : "cu[].idct (the primitive alone)",
                        4 << (k & 3), 4 << (k & 3), w ? "(elsewhere)           " : "(in estimateResidualQT)", (unsigned long long)g_primCalls[k][w].load(),
                        (double)g_primCycles[k][w].load() / g_primCalls[k][w].load(), g_primCycles[k][w].load() * 1e-9);
}

// X265HIP_DEBUG_SCANHIT=1: how much of scanPosLast (dct.cpp:757-788; called by Entropy::codeCoeffNxN, entropy.cpp:1856) runs on levels a CU job has just
// delivered — the call right after a served Quant::transformNxN, same coefficient buffer — i.e. what a job that also shipped the scan's per-group arrays could
// take over (a measurement for the next step, not a product path)
scanPosLast_t g_scanPrev;
std::atomic<uint64_t> g_scanCycles[2], g_scanCalls[2];
__attribute__((tls_model("initial-exec"))) thread_local const coeff_t* t_lastServedCoeff = NULL;
int scan_counted(const uint16_t* scan, const coeff_t* coeff, uint16_t* coeffSign, uint16_t* coeffFlag, uint8_t* coeffNum, int numSig, const uint16_t* scanCG4x4, const int trSize)
{
    const int k = coeff == t_lastServedCoeff ? 0 : 1;
    const uint64_t t0 = __builtin_ia32_rdtsc();
    const int r = g_scanPrev(scan, coeff, coeffSign, coeffFlag, coeffNum, numSig, scanCG4x4, trSize);
    g_scanCycles[k].fetch_add(__builtin_ia32_rdtsc() - t0, std::memory_order_relaxed);
    g_scanCalls[k].fetch_add(1, std::memory_order_relaxed);
    return r;
}
void report_scanhit()
{
    fprintf(stderr, "x265hip: scanhit: scanPosLast on levels a CU job has just delivered: %llu calls, %.3f G cycles; elsewhere: %llu calls, %.3f G cycles\n",
            (unsigned long long)g_scanCalls[0].load(), g_scanCycles[0].load() * 1e-9, (unsigned long long)g_scanCalls[1].load(), g_scanCycles[1].load() * 1e-9);
}


void x265hip_install_cuserve_slots(EncoderPrimitives& p)
{
    decide();
    if (getenv("X265HIP_DEBUG_SCANHIT"))
    {
        static std::mutex onceS;
        std::lock_guard<std::mutex> g(onceS);
        if (!g_scanPrev) { g_scanPrev = p.scanPosLast; atexit(report_scanhit); }
        p.scanPosLast = scan_counted;
    }
    if (g_time)
    {
        static std::mutex onceT;
        std::lock_guard<std::mutex> g(onceT);
        if (!g_timedDct[0])
        {
            for (int k = 0; k < 4; k++) { g_timedDct[k] = p.cu[k].dct; g_timedIdct[k] = p.cu[k].idct; }
            atexit(report_prim_time);
        }
        p.cu[0].dct = dct_timed<0>; p.cu[1].dct = dct_timed<1>; p.cu[2].dct = dct_timed<2>; p.cu[3].dct = dct_timed<3>;
        p.cu[0].idct = idct_timed<0>; p.cu[1].idct = idct_timed<1>; p.cu[2].idct = idct_timed<2>; p.cu[3].idct = idct_timed<3>;
    }
    if (g_state <= 0 || !g_serveDist)
        return;
    // x265_setup_primitives is not serialised: two encoders opened at the same moment both find the table empty and both set it up, the second one's
    // setupCPrimitives writing C functions over slots the first has already wrapped (tests/test_encoder_lifetime.py, parallel sessions: one run in ten ended
    // with add_ps wrapped and sse_pp not).  So the wrappers are written on EVERY call — the table ends up wrapped whoever comes last — while "what was
    // there" is taken once, from the first call (later calls may already see wrappers in the table).
    static std::mutex once;
    std::lock_guard<std::mutex> g(once);
    if (!g_slots_installed)
    {
        g_prev = p;
        g_verify = getenv("X265HIP_VERIFY") != NULL;
    }
    p.cu[BLOCK_8x8].sse_pp = sse_slot<BLOCK_8x8, 8, false>;
    p.cu[BLOCK_16x16].sse_pp = sse_slot<BLOCK_16x16, 16, false>;
    p.cu[BLOCK_32x32].sse_pp = sse_slot<BLOCK_32x32, 32, false>;
    p.cu[BLOCK_64x64].sse_pp = sse_slot<BLOCK_64x64, 64, false>;
#if X265_DEPTH > 8
    p.cu[BLOCK_8x8].sse_ss = sse_ss_slot<BLOCK_8x8, 8>;
    p.cu[BLOCK_16x16].sse_ss = sse_ss_slot<BLOCK_16x16, 16>;
    p.cu[BLOCK_32x32].sse_ss = sse_ss_slot<BLOCK_32x32, 32>;
    p.cu[BLOCK_64x64].sse_ss = sse_ss_slot<BLOCK_64x64, 64>;
#endif
    // 4:2:0: chroma[].cu[i] is the chroma block of luma CU i (primitives.h:80-90)
    p.chroma[X265_CSP_I420].cu[BLOCK_16x16].sse_pp = sse_slot<BLOCK_16x16, 8, true>;
    p.chroma[X265_CSP_I420].cu[BLOCK_32x32].sse_pp = sse_slot<BLOCK_32x32, 16, true>;
    p.chroma[X265_CSP_I420].cu[BLOCK_64x64].sse_pp = sse_slot<BLOCK_64x64, 32, true>;
    p.cu[BLOCK_8x8].psy_cost_pp = psy_slot<BLOCK_8x8, 8>;
    p.cu[BLOCK_16x16].psy_cost_pp = psy_slot<BLOCK_16x16, 16>;
    p.cu[BLOCK_32x32].psy_cost_pp = psy_slot<BLOCK_32x32, 32>;
    p.cu[BLOCK_64x64].psy_cost_pp = psy_slot<BLOCK_64x64, 64>;
    const int dead = getenv("X265HIP_CUSERVE_DEAD") ? atoi(getenv("X265HIP_CUSERVE_DEAD")) : 3;
    if (g_serveDist >= 3 && (dead & 1))
    {
        p.cu[BLOCK_32x32].sub_ps = sub_ps_slot<BLOCK_32x32, 32, false>;
        p.cu[BLOCK_64x64].sub_ps = sub_ps_slot<BLOCK_64x64, 64, false>;
        p.chroma[X265_CSP_I420].cu[BLOCK_32x32].sub_ps = sub_ps_slot<BLOCK_32x32, 16, true>;
        p.chroma[X265_CSP_I420].cu[BLOCK_64x64].sub_ps = sub_ps_slot<BLOCK_64x64, 32, true>;
    }
    if (g_serveDist >= 3 && (dead & 2))
    {
        p.cu[BLOCK_8x8].add_ps[0] = add_ps_slot<BLOCK_8x8, 8, 0>;     p.cu[BLOCK_8x8].add_ps[1] = add_ps_slot<BLOCK_8x8, 8, 1>;
        p.cu[BLOCK_16x16].add_ps[0] = add_ps_slot<BLOCK_16x16, 16, 0>; p.cu[BLOCK_16x16].add_ps[1] = add_ps_slot<BLOCK_16x16, 16, 1>;
        p.cu[BLOCK_32x32].add_ps[0] = add_ps_slot<BLOCK_32x32, 32, 0>; p.cu[BLOCK_32x32].add_ps[1] = add_ps_slot<BLOCK_32x32, 32, 1>;
    }
    install_intra_slots(p);
    g_slots_installed = true;
}

void Search::estimateResidualQT(Mode& mode, const CUGeom& cuGeom, uint32_t absPartIdx, uint32_t tuDepth, ShortYuv& resiYuv, Cost& outCosts, const uint32_t depthRange[2],
                                int32_t splitMore)
{
    // only the top-level call arrives here (the reference body recurses into its own copy).  Normally the job is already on its way: it was submitted
    // when encodeResAndCalcRdInterCU was entered (below), before the host even computed the residual
    Job& j = t_job;
    bool mine = j.active && j.phase && !tuDepth && !absPartIdx && j.resi[0] == resiYuv.m_buf[0] && (int)j.job->log2TrMax == (int)depthRange[1] &&
                (int)j.job->log2TrMin == (int)depthRange[0];
    if (j.active && !mine)
        end_job();                      // a tree this job was not made for (never seen): nothing of it is used
    if (!mine && g_state > 0 && !g_dead.load(std::memory_order_relaxed) && (int)cuGeom.log2CUSize >= g_minLog2 && !tuDepth && !absPartIdx)
    {
        mine = submit(this, mode, cuGeom.log2CUSize, resiYuv, depthRange);
        if (!mine) counters().skipped.fetch_add(1, std::memory_order_relaxed);
    }
    if (mine) { j.inTree = true; j.treeMine = true; }
    if (g_time)
    {
        Timed t(8 + cuGeom.log2CUSize - 2);
        t_inRqt++;
        refEstimateResidualQT(this, mode, cuGeom, absPartIdx, tuDepth, resiYuv, outCosts, depthRange, splitMore);
        t_inRqt--;
    }
    else
        refEstimateResidualQT(this, mode, cuGeom, absPartIdx, tuDepth, resiYuv, outCosts, depthRange, splitMore);
    if (mine)
    {
        j.inTree = false;
        if (!t_inEncodeRes)
            end_job();                 // not expected (estimateResidualQT has one caller), but then nothing is remembered beyond the tree
    }
}

// The scope of a job's remembered values: one call of encodeResAndCalcRdInterCU (reference search.cpp:2822-2975).  Inside it nobody writes the mode's
// source or prediction block, nor — after the tree — the tree's reconstruction buffers.  The reference's body runs under its renamed symbol.
// is the job submitted ahead (phase 0) the job this scope would submit?  Same search object, source Yuv, residual buffer and header, and a prediction with
// the same samples as the one the job took (the merge path hands encodeResAndCalcRdInterCU a COPY of it, analysis.cpp:2864): then it becomes this scope's job
bool adopt(Search* se, Mode& mode, const CUGeom& cuGeom)
{
    Job& j = t_job;
    uint32_t range[2];
    mode.cu.getInterTUQtDepthRange(range, 0);
    x265hip_cujob hdr;
    if (j.search != se || j.fencYuv != mode.fencYuv || j.log2CU != cuGeom.log2CUSize || j.resi[0] != se->m_rqt[cuGeom.depth].tmpResiYuv.m_buf[0] ||
        !make_header(se, mode, cuGeom.log2CUSize, range, hdr) || memcmp(&hdr, &j.hdr, sizeof(hdr)))
        return false;
    const Yuv& pred = mode.predYuv;
    const Yuv& fenc = *mode.fencYuv;
    const int N = 1 << j.log2CU;
    // the samples the job was given against the samples this scope would give it: source first, prediction second, plane by plane (the layout of submit).
    // Buffer addresses say nothing — the mode's Yuv buffers are reused from CU to CU (an analysis-load encode evaluates skips that no
    // encodeResAndCalcRdInterCU follows, analysis.cpp:2547: the job then waits for the NEXT CU's call)
    const pixel* sent = j.sent;
    for (int half = 0; half < 2; half++)
        for (int p = 0; p < (j.pred[1] ? 3 : 1); p++)
        {
            const Yuv& y = half ? pred : fenc;
            const int n = p ? N / 2 : N;
            const uint32_t st = p ? y.m_csize : y.m_size;
            for (int r = 0; r < n; r++, sent += n)
                if (memcmp(y.m_buf[p] + (size_t)r * st, sent, sizeof(pixel) * n))
                    return false;
        }
    for (int p = 0; p < (j.pred[1] ? 3 : 1); p++) { j.pred[p] = pred.m_buf[p]; j.predStride[p] = p ? pred.m_csize : pred.m_size; }
    j.mode = &mode;
    j.phase = 1;
    // an energy remembered from the skip evaluation is the (source, prediction) energy now
    for (int u = 0; u < X265HIP_CUJOB_MAX_UNITS; u++) if (j.energyKnown[u] == 2) { j.energyKnown[u] = 1; counters().psySkip.fetch_add(1, std::memory_order_relaxed); }
    return true;
}

// Merge candidates at rd levels 1..4 (Analysis::checkMerge2Nx2N_rd0_4, analysis.cpp:2846-2866): the best candidate is evaluated as a skip —
// encodeResAndCalcRdSkipCU, a few microseconds of distortions, psy-cost and bits — and then ALWAYS with its residual, encodeResAndCalcRdInterCU on a copy
// of the same prediction.  The prediction is final when the skip evaluation starts: the job leaves there, and the device works on it while this thread
// evaluates the skip.  Nothing is answered from the job before its own scope starts (phase 0); a job whose scope never comes is dropped at the next
// call of either function.


void Search::encodeResAndCalcRdSkipCU(Mode& interMode)
{
    intra_unasked();                     // an intra scan submitted ahead for a CU whose intra try never came (below)
    Job& j = t_job;
    if (j.active && !t_inEncodeRes)
        end_job();
    const uint32_t log2CU = interMode.cu.m_log2CUSize[0];
    if (g_spec && g_state > 0 && !g_dead.load(std::memory_order_relaxed) && (int)log2CU >= g_minLog2 && !j.active && !t_inEncodeRes && m_param->rdLevel >= 1 &&
        m_param->rdLevel <= 4 && !m_param->bLossless && !m_param->interRefine && !m_param->bDynamicRefine && !m_param->analysisLoad)       // (a loaded analysis
        // evaluates skips that nothing follows, analysis.cpp:2547: a job sent from there would only be waited for and dropped)
    {
        uint32_t range[2];
        interMode.cu.getInterTUQtDepthRange(range, 0);
        if (submit(this, interMode, log2CU, m_rqt[interMode.cu.m_cuDepth[0]].tmpResiYuv, range))
        {
            j.inTree = false;
            j.phase = 0;
            j.specInter = false;
            counters().spec.fetch_add(1, std::memory_order_relaxed);
        }
    }
    refEncodeResAndCalcRdSkipCU(this, interMode);
}

// The 2Nx2N inter candidate at rd levels 3 and 4 without rectangular partitions (Analysis::compressInterCU_rd0_4, analysis.cpp:1421-1611): its prediction
// is final when predInterSearch returns (luma and chroma compensated, search.cpp:2181-2560), and it is ALWAYS evaluated with its residual (:1609-1611,
// bestInter = the only inter candidate) — after checkInter_rd0_4's sa8d of the three planes and, in a B slice, after checkBidir2Nx2N has built and
// measured the bidirectional prediction.  The job leaves here and the device works on it while this thread does that; the same adoption rule as for the
// merge candidate (sample for sample the job's blocks, or it is dropped).
void Search::predInterSearch(Mode& interMode, const CUGeom& cuGeom, bool bChromaMC, uint32_t refMasks[2])
{
    // the intra scan of the same CU, should it be tried as intra (below): its inputs are final already — the job leaves before the motion search, not after it
    if (interMode.cu.m_partSize[0] == SIZE_2Nx2N)
        intra_ahead(this, interMode, cuGeom);
    refPredInterSearch(this, interMode, cuGeom, bChromaMC, refMasks);
    Job& j = t_job;
    if (g_specInter && g_spec && g_state > 0 && !g_dead.load(std::memory_order_relaxed) && (int)cuGeom.log2CUSize >= g_minLog2 && !t_inEncodeRes &&
        interMode.cu.m_partSize[0] == SIZE_2Nx2N && (bChromaMC || m_csp == X265_CSP_I400) && m_param->rdLevel >= 3 && m_param->rdLevel <= 4 &&
        !m_param->bEnableRectInter && !m_param->bEnableAMP && !m_param->bDistributeModeAnalysis && !m_param->bLossless && !m_param->interRefine &&
        !m_param->bDynamicRefine && !m_param->analysisLoad && !m_param->analysisSave)
    {
        if (j.active)
            end_job();
        uint32_t range[2];
        interMode.cu.getInterTUQtDepthRange(range, 0);
        if (submit(this, interMode, cuGeom.log2CUSize, m_rqt[cuGeom.depth].tmpResiYuv, range))
        {
            j.inTree = false;
            j.phase = 0;
            j.specInter = true;
            counters().specInter.fetch_add(1, std::memory_order_relaxed);
        }
    }
}

void Search::encodeResAndCalcRdInterCU(Mode& interMode, const CUGeom& cuGeom)
{
    t_inEncodeRes++;
    if (t_job.active && !t_job.phase)
    {
        if (adopt(this, interMode, cuGeom)) (t_job.specInter ? counters().specInterHit : counters().specHit).fetch_add(1, std::memory_order_relaxed);
        else end_job();
    }
    if (g_state > 0 && !g_dead.load(std::memory_order_relaxed) && (int)cuGeom.log2CUSize >= g_minLog2 && !t_job.active)
    {
        // the job leaves now: the device fetches source and prediction while this thread still subtracts them (search.cpp:2838) and sets the tree up
        uint32_t range[2];
        interMode.cu.getInterTUQtDepthRange(range, 0);
        if (!submit(this, interMode, cuGeom.log2CUSize, m_rqt[cuGeom.depth].tmpResiYuv, range))
            counters().skipped.fetch_add(1, std::memory_order_relaxed);
        else
            t_job.inTree = false;
    }
    refEncodeResAndCalcRdInterCU(this, interMode, cuGeom);
    t_inEncodeRes--;
    if (t_job.active)
        end_job();
}


uint32_t Quant::transformNxN(const CUData& cu, const pixel* fenc, uint32_t fencStride, const int16_t* residual, uint32_t resiStride, coeff_t* coeff, uint32_t log2TrSize,
                             TextType ttype, uint32_t absPartIdx, bool useTransformSkip)
{
    Job& j = t_job;
    if (j.active && j.inTree && j.quant == this && !useTransformSkip)
    {
        int eo = 0;
        const int u = locate(j, residual, resiStride, log2TrSize, (int)ttype, &eo);
        if (u >= 0)
        {
            const uint64_t t0 = g_time ? __builtin_ia32_rdtsc() : 0;
            if (__atomic_load_n(&j.units[u].ready, __ATOMIC_ACQUIRE) != j.seq)
            {
                // not there yet: do something the tree needs next anyway instead of spinning — this unit's psy-cost, then (a luma unit: chroma follows
                // it in the tree, search.cpp:3314-3345) the psy-costs of the chroma units under it
                const ptrdiff_t d = residual - j.resi[ttype];
                const int x = (int)(d % resiStride), y = (int)(d / resiStride), n = 1 << log2TrSize;
                psy_ahead(j, u, (int)ttype, x, y, n);
                if (ttype == TEXT_LUMA && j.resi[1] && n >= 16)
                    for (int p = 1; p <= 2 && __atomic_load_n(&j.units[u].ready, __ATOMIC_ACQUIRE) != j.seq; p++)
                        psy_ahead(j, x265hipi_cujob_unit_index(j.job, j.sHi, (int)log2TrSize, p, x >> log2TrSize, y >> log2TrSize), p, x >> 1, y >> 1, n >> 1);
            }
            if (j.hdr.coefMode)
            {
                // RDOQ: the device has transformed (residual -> m_resiDctCoeff, and for psy-rdoq source -> m_fencDctCoeff: quant.cpp:432, :436-442); the
                // quantiser is the reference's own, called as Quant::transformNxN calls it (:454-455)
                const bool usePsy = m_psyRdoqScale && ttype == TEXT_LUMA;
                // (the source block's transform is a work item of its own on the device and releases `readyInv`)
                if (!wait_word(j, &j.units[u].ready, ttype == TEXT_LUMA ? 0 : 1) || (usePsy && !wait_word(j, &j.units[u].readyInv, 0)))
                    goto host;
                const int n2 = 1 << (2 * log2TrSize);
                memcpy(m_resiDctCoeff, j.levels + eo, sizeof(int16_t) * n2);
                if (usePsy)
                    memcpy(m_fencDctCoeff, j.resiOut + eo, sizeof(int16_t) * n2);
                if (g_verify)
                {
                    int16_t gotR[1024], gotF[1024];
                    memcpy(gotR, m_resiDctCoeff, sizeof(int16_t) * n2);
                    if (usePsy) memcpy(gotF, m_fencDctCoeff, sizeof(int16_t) * n2);
                    coeff_t want[1024];
                    const uint32_t ns = refTransformNxN(this, cu, fenc, fencStride, residual, resiStride, want, log2TrSize, ttype, absPartIdx, useTransformSkip);
                    if (memcmp(gotR, m_resiDctCoeff, sizeof(int16_t) * n2) || (usePsy && memcmp(gotF, m_fencDctCoeff, sizeof(int16_t) * n2)))
                    {
                        fprintf(stderr, "x265hip: cuserve: VERIFY FAILED transformNxN (RDOQ) %dx%d plane %d: the device's transform coefficients differ from cu[].dct's\n",
                                1 << log2TrSize, 1 << log2TrSize, (int)ttype);
                        abort();
                    }
                    const uint32_t numSigV = (this->*rdoQuant_func[log2TrSize - 2])(cu, coeff, ttype, absPartIdx, usePsy);
                    if (ns != numSigV || memcmp(want, coeff, sizeof(coeff_t) * n2)) { fprintf(stderr, "x265hip: cuserve: VERIFY FAILED rdoQuant is not a function of its inputs?\n"); abort(); }
                    counters().fwd.fetch_add(1, std::memory_order_relaxed);
                    if (numSigV && ttype == TEXT_LUMA && log2TrSize == 5)
                    {
                        const ptrdiff_t dI = residual - j.resi[0];
                        inv_submit(j, u, (int)(dI % resiStride), (int)(dI / resiStride), coeff);
                    }
                    return numSigV;
                }
                const uint32_t numSigQ = (this->*rdoQuant_func[log2TrSize - 2])(cu, coeff, ttype, absPartIdx, usePsy);
                counters().fwd.fetch_add(1, std::memory_order_relaxed);
                // the levels exist now; the tree codes their bits (search.cpp:3243-3262) and then asks for the unit's inverse: that half leaves here
                if (numSigQ && ttype == TEXT_LUMA && log2TrSize == 5)
                {
                    const ptrdiff_t dI = residual - j.resi[0];
                    inv_submit(j, u, (int)(dI % resiStride), (int)(dI / resiStride), coeff);
                }
                if (g_time)
                {
                    g_cycles[log2TrSize - 2][0].fetch_add(__builtin_ia32_rdtsc() - t0, std::memory_order_relaxed);
                    g_calls[log2TrSize - 2][0].fetch_add(1, std::memory_order_relaxed);
                }
                return numSigQ;
            }
            const bool lumaSpin = ttype == TEXT_LUMA && __atomic_load_n(&j.units[u].ready, __ATOMIC_ACQUIRE) != j.seq;
            const uint64_t w0 = lumaSpin ? __builtin_ia32_rdtsc() : 0;
            if (wait_word(j, &j.units[u].ready, ttype == TEXT_LUMA ? 0 : 1))
            {
                if (lumaSpin)
                {
                    std::atomic<uint64_t>* c = g_lumaWait[j.hdr.log2CUSize >= 6][u != 0];
                    c[0].fetch_add(1, std::memory_order_relaxed);
                    c[1].fetch_add(__builtin_ia32_rdtsc() - w0, std::memory_order_relaxed);
                }
                const int n2 = 1 << (2 * log2TrSize);
                memcpy(coeff, j.levels + eo, sizeof(coeff_t) * n2);
                const uint32_t numSig = j.units[u].numSig;
                if (g_verify)
                {
                    coeff_t want[1024];
                    const uint32_t ns = refTransformNxN(this, cu, fenc, fencStride, residual, resiStride, want, log2TrSize, ttype, absPartIdx, useTransformSkip);
                    if (ns != numSig || memcmp(want, coeff, sizeof(coeff_t) * n2))
                    {
                        fprintf(stderr, "x265hip: cuserve: VERIFY FAILED transformNxN %dx%d plane %d: numSig %u (device) vs %u\n", 1 << log2TrSize, 1 << log2TrSize, (int)ttype,
                                numSig, ns);
                        abort();
                    }
                }
                counters().fwd.fetch_add(1, std::memory_order_relaxed);
                t_lastServedCoeff = coeff;
                if (g_time)
                {
                    g_cycles[log2TrSize - 2][0].fetch_add(__builtin_ia32_rdtsc() - t0, std::memory_order_relaxed);
                    g_calls[log2TrSize - 2][0].fetch_add(1, std::memory_order_relaxed);
                }
                return numSig;
            }
        }
        else
            counters().fwdMiss.fetch_add(1, std::memory_order_relaxed);
    }
host:
    if (j.active && j.quant == this)
        flush_sub(j);                   // this call reads the CU's residual on the host after all
    if (g_time)
    {
        Timed t(log2TrSize - 2);
        return refTransformNxN(this, cu, fenc, fencStride, residual, resiStride, coeff, log2TrSize, ttype, absPartIdx, useTransformSkip);
    }
    return refTransformNxN(this, cu, fenc, fencStride, residual, resiStride, coeff, log2TrSize, ttype, absPartIdx, useTransformSkip);
}

void Quant::invtransformNxN(const CUData& cu, int16_t* residual, uint32_t resiStride, const coeff_t* coeff, uint32_t log2TrSize, TextType ttype, bool bIntra,
                            bool useTransformSkip, uint32_t numSig)
{
    Job& j = t_job;
    if (j.active && j.inTree && j.quant == this && j.inv.active)
    {
        // coefficient mode: the inverse job submitted behind rdoQuant — served only if these ARE the levels it was given (equal levels have equal inverses)
        Job::InvAhead& a = j.inv;
        if (!useTransformSkip && !bIntra && ttype == TEXT_LUMA && log2TrSize == 5 && !memcmp(coeff, a.sent, 2048) && inv_wait(j))
        {
            const uint64_t t0 = g_time ? __builtin_ia32_rdtsc() : 0;
            for (int y = 0; y < 32; y++)
                memcpy(residual + (size_t)y * resiStride, a.resi + y * 32, sizeof(int16_t) * 32);
            if (g_verify)
            {
                int16_t want[1024];
                refInvtransformNxN(this, cu, want, 32, coeff, log2TrSize, ttype, bIntra, useTransformSkip, numSig);
                if (memcmp(want, a.resi, sizeof(want)) || a.units[0].numSig != numSig)
                {
                    fprintf(stderr, "x265hip: cuserve: VERIFY FAILED invtransformNxN (inverse job) 32x32 numSig %u (device counted %u)\n", numSig, a.units[0].numSig);
                    abort();
                }
            }
            // the unit's measurements behind the inverse (sse_pp and psy energy of the reconstruction) go where the answering slots look for them
            x265hip_cujob_unit& un = const_cast<x265hip_cujob_unit&>(j.units[a.unit]);
            un.codedDist = a.units[0].codedDist;
            un.codedEnergy = a.units[0].codedEnergy;
            j.invServed[a.unit] = 1;
            give_slot(a.svc, a.slot);
            a.active = false;
            counters().inv.fetch_add(1, std::memory_order_relaxed);
            if (g_time)
            {
                g_cycles[4 + log2TrSize - 2][0].fetch_add(__builtin_ia32_rdtsc() - t0, std::memory_order_relaxed);
                g_calls[4 + log2TrSize - 2][0].fetch_add(1, std::memory_order_relaxed);
            }
            return;
        }
        if (a.active) inv_drop(j);
    }
    if (j.active && j.inTree && j.quant == this && !useTransformSkip && !bIntra && !j.hdr.coefMode)
    {
        // which unit?  the one of this size and plane whose levels these are: equal levels have equal inverse transforms, so the comparison — not
        // any bookkeeping — is what makes the copy exact.  The tree asks for a unit's inverse right after its forward transform: look there first.
        const int s = ttype ? (int)log2TrSize + 1 : (int)log2TrSize;
        if (s <= j.sHi && s >= j.sLo && (ttype == TEXT_LUMA || j.resi[ttype]))
        {
            const uint64_t t0 = g_time ? __builtin_ia32_rdtsc() : 0;
            const int per = 1 << (j.log2CU - s), n = 1 << log2TrSize, n2 = n * n;
            const int first = x265hipi_cujob_unit_index(j.job, j.sHi, s, (int)ttype, 0, 0), eo0 = x265hipi_cujob_elem_offset(j.job, j.sHi, s, (int)ttype, 0, 0);
            for (int t = 0; t < per * per; t++)
            {
                const x265hip_cujob_unit& un = j.units[first + t];
                if (__atomic_load_n(&un.ready, __ATOMIC_ACQUIRE) != j.seq || un.numSig != numSig || memcmp(coeff, j.levels + eo0 + t * n2, sizeof(coeff_t) * n2))
                    continue;
                if (!wait_word(j, &un.readyInv, ttype == TEXT_LUMA ? 2 : 3))
                    break;
                const int16_t* src = j.resiOut + eo0 + t * n2;
                for (int y = 0; y < n; y++)
                    memcpy(residual + (size_t)y * resiStride, src + y * n, sizeof(int16_t) * n);
                if (g_verify)
                {
                    int16_t want[1024];
                    refInvtransformNxN(this, cu, want, n, coeff, log2TrSize, ttype, bIntra, useTransformSkip, numSig);
                    if (memcmp(want, src, sizeof(int16_t) * n2))
                    {
                        fprintf(stderr, "x265hip: cuserve: VERIFY FAILED invtransformNxN %dx%d plane %d numSig %u\n", n, n, (int)ttype, numSig);
                        abort();
                    }
                }
                j.invServed[first + t] = 1;
                counters().inv.fetch_add(1, std::memory_order_relaxed);
                if (g_time)
                {
                    g_cycles[4 + log2TrSize - 2][0].fetch_add(__builtin_ia32_rdtsc() - t0, std::memory_order_relaxed);
                    g_calls[4 + log2TrSize - 2][0].fetch_add(1, std::memory_order_relaxed);
                }
                return;
            }
            counters().invMiss.fetch_add(1, std::memory_order_relaxed);
        }
    }
    if (g_time)
    {
        Timed t(4 + log2TrSize - 2);
        refInvtransformNxN(this, cu, residual, resiStride, coeff, log2TrSize, ttype, bIntra, useTransformSkip, numSig);
        return;
    }
    refInvtransformNxN(this, cu, residual, resiStride, coeff, log2TrSize, ttype, bIntra, useTransformSkip, numSig);
}


} // namespace X265_NS
