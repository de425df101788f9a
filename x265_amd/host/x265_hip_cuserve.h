// x265_hip_cuserve.h — what the three translation units of the job-service seams share (not an installed header: internal to x265_amd/host):
//   x265_hip_cuserve.cpp    the services and their slots, CU residual quad-tree jobs (Search::estimateResidualQT / encodeResAndCalcRd*CU / predInterSearch, Quant::*)
//   x265_hip_saostats.cpp   SAO statistics jobs (SAO::calcSaoStatsCTU)
//   x265_hip_intrascan.cpp  intra mode scan jobs (Search::checkIntraInInter and the table slots it answers through)
#pragma once
#include <atomic>
#include <ctime>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <utility>
#include <string>

#define protected public
#define private public
#include "common.h"
#include "frame.h"
#include "framedata.h"
#include "picyuv.h"
#include "primitives.h"
#include "yuv.h"
#include "shortyuv.h"
#include "cudata.h"
#include "quant.h"
#include "scalinglist.h"
#include "search.h"
#include "analysis.h"
#include "sao.h"
#undef protected
#undef private

#include <sched.h>
#include "x265hip.h"
#include "x265_hip_debug.h"

namespace X265_NS {


const EncoderPrimitives& x265hip_c_table();          // x265_hip_primitives.cpp

extern void refEstimateResidualQT(Search* self, Mode& mode, const CUGeom& cuGeom, uint32_t absPartIdx, uint32_t tuDepth, ShortYuv& resiYuv, Search::Cost& outCosts,
                                  const uint32_t depthRange[2], int32_t splitMore)
    asm("_ZN4x2659SearchRef18estimateResidualQTERNS_4ModeERKNS_6CUGeomEjjRNS_8ShortYuvERNS0_4CostEPKji");
extern void refCheckIntraInInter(Search* self, Mode& intraMode, const CUGeom& cuGeom) asm("_ZN4x2659SearchRef17checkIntraInInterERNS_4ModeERKNS_6CUGeomE");
extern void refEncodeResAndCalcRdInterCU(Search* self, Mode& interMode, const CUGeom& cuGeom) asm("_ZN4x2656Search29encodeResAndCalcRdInterCUBodyERNS_4ModeERKNS_6CUGeomE");
extern void refEncodeResAndCalcRdSkipCU(Search* self, Mode& interMode) asm("_ZN4x2656Search28encodeResAndCalcRdSkipCUBodyERNS_4ModeE");
extern void refPredInterSearch(Search* self, Mode& interMode, const CUGeom& cuGeom, bool bChromaMC, uint32_t refMasks[2])
    asm("_ZN4x2656Search19predInterSearchBodyERNS_4ModeERKNS_6CUGeomEbPj");
#if X265_DEPTH == 8
extern uint32_t refTransformNxN(Quant* self, const CUData& cu, const pixel* fenc, uint32_t fencStride, const int16_t* residual, uint32_t resiStride, coeff_t* coeff,
                                uint32_t log2TrSize, TextType ttype, uint32_t absPartIdx, bool useTransformSkip)
    asm("_ZN4x2658QuantRef12transformNxNERKNS_6CUDataEPKhjPKsjPsjNS_8TextTypeEjb");
#else
extern uint32_t refTransformNxN(Quant* self, const CUData& cu, const pixel* fenc, uint32_t fencStride, const int16_t* residual, uint32_t resiStride, coeff_t* coeff,
                                uint32_t log2TrSize, TextType ttype, uint32_t absPartIdx, bool useTransformSkip)
    asm("_ZN4x2658QuantRef12transformNxNERKNS_6CUDataEPKtjPKsjPsjNS_8TextTypeEjb");
#endif
extern void refInvtransformNxN(Quant* self, const CUData& cu, int16_t* residual, uint32_t resiStride, const coeff_t* coeff, uint32_t log2TrSize, TextType ttype,
                               bool bIntra, bool useTransformSkip, uint32_t numSig)
    asm("_ZN4x2658QuantRef15invtransformNxNERKNS_6CUDataEPsjPKsjNS_8TextTypeEbbj");

extern void refCalcSaoStatsCTU(SAO* self, int addr, int plane) asm("_ZN4x2656SAORef15calcSaoStatsCTUEii");


namespace cusvc {

struct SlotMem { x265hip_cujob* job; void* pixels; const x265hip_cujob_unit* units; const int16_t* levels; const int16_t* resi; };
// one job service per place (X265HIP_DEVICES; one on the calling thread's device when no places are configured): every GPU of the encoder serves CU jobs
struct Service
{
    x265hip_cuserve* cs;
    SlotMem mem[256];
    std::atomic<uint64_t> busy[4];           // bit s of word s / 64: slot s holds a job of some thread (a slot is taken per job, not per thread: x265 starts
                                             // one pool thread per core it sees, far more than ever run at once under a CPU quota)
};
extern int g_state;                          // 0 undecided, 1 on, -1 off (the CU jobs; the SAO and intra modules keep their own)
extern int g_time;                           // X265HIP_DEBUG_CUTIME
extern int64_t g_timeoutNs;                  // X265HIP_CUSERVE_TIMEOUT_MS
extern int g_yieldAfter;                     // X265HIP_CUSERVE_YIELD
extern bool g_verify;                        // X265HIP_VERIFY=1
extern std::mutex g_lock;                    // decisions, service open / shutdown
extern std::atomic<bool> g_dead;             // the device failed once (or the services are closed): nothing is handed over any more
extern EncoderPrimitives g_prev;             // the table as it was when the cuserve slots were installed
extern bool g_slots_installed;
extern std::atomic<uint64_t> g_cycles[18][2], g_calls[18][2];
extern __attribute__((tls_model("initial-exec"))) thread_local int t_inRqt;
extern __attribute__((tls_model("initial-exec"))) thread_local int t_shard;
void touch_shard();                          // gives the calling thread its counter shard (t_shard >= 0 afterwards)
bool service();                              // opens the services on first use; false: no device
int take_slot(Service** svc);                // a free slot for this thread's next job, or -1
inline void give_slot(Service* sv, int s) { sv->busy[s >> 6].fetch_and(~(1ull << (s & 63)), std::memory_order_release); }
template <typename T> inline void pack_rows(T*& dst, const T* src, uint32_t stride, int n)
{
    if ((int)stride == n) { memcpy(dst, src, sizeof(T) * n * n); dst += n * n; return; }
    for (int y = 0; y < n; y++, dst += n) memcpy(dst, src + (size_t)y * stride, sizeof(T) * n);
}
// X265HIP_DEBUG_CUTIME: cycles of one call into the table of report_time()
struct Timed
{
    int k; uint64_t t0;
    Timed(int slot) : k(slot), t0(__builtin_ia32_rdtsc()) {}
    ~Timed()
    {
        const int w = k < 8 ? !t_inRqt : 0;
        g_cycles[k][w].fetch_add(__builtin_ia32_rdtsc() - t0, std::memory_order_relaxed);
        g_calls[k][w].fetch_add(1, std::memory_order_relaxed);
    }
};

} // namespace cusvc

// x265_hip_intrascan.cpp, called by the CU-job seams
void intra_ahead(Search* se, Mode& interMode, const CUGeom& cuGeom);
void intra_unasked();
void install_intra_slots(EncoderPrimitives& p);

} // namespace X265_NS
