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
// Seams (same link technique as the other seams, oracle/Makefile): Search::estimateResidualQT, Search::checkIntraInInter on search.o;
// Quant::transformNxN, Quant::invtransformNxN on quant.o.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

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
#include "search.h"
#undef protected
#undef private

#include "x265hip.h"
#include "x265_hip_debug.h"

namespace X265_NS {

const EncoderPrimitives& x265hip_c_table();          // x265_hip_primitives.cpp

extern void refEstimateResidualQT(Search* self, Mode& mode, const CUGeom& cuGeom, uint32_t absPartIdx, uint32_t tuDepth, ShortYuv& resiYuv, Search::Cost& outCosts,
                                  const uint32_t depthRange[2], int32_t splitMore)
    asm("_ZN4x2659SearchRef18estimateResidualQTERNS_4ModeERKNS_6CUGeomEjjRNS_8ShortYuvERNS0_4CostEPKji");
extern void refCheckIntraInInter(Search* self, Mode& intraMode, const CUGeom& cuGeom) asm("_ZN4x2659SearchRef17checkIntraInInterERNS_4ModeERKNS_6CUGeomE");
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

namespace {

int g_state = 0;                 // 0 undecided, 1 on, -1 off
int g_time = 0;                  // X265HIP_DEBUG_CUTIME=1: cycles inside the functions a job could replace, by block size (report at exit)
std::mutex g_lock;

// X265HIP_DEBUG_CUTIME: [0..3] transformNxN by log2TrSize - 2, [4..7] invtransformNxN, [8..12] top-level estimateResidualQT by log2CUSize - 2,
// [13..17] checkIntraInInter by log2CUSize - 2; second index: 0 inside a top-level estimateResidualQT of this thread, 1 elsewhere
std::atomic<uint64_t> g_cycles[18][2], g_calls[18][2];
__attribute__((tls_model("initial-exec"))) thread_local int t_inRqt = 0;

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

bool decide()
{
    std::lock_guard<std::mutex> g(g_lock);
    if (!g_state)
    {
        g_time = getenv("X265HIP_DEBUG_CUTIME") ? atoi(getenv("X265HIP_DEBUG_CUTIME")) : 0;
        if (g_time)
            atexit(report_time);
        g_state = -1;
    }
    return g_state > 0;
}
inline bool enabled() { return g_state ? g_state > 0 : decide(); }
struct DecideAtLoad { DecideAtLoad() { decide(); } } g_decideAtLoad;        // the Quant seams read g_time without asking

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

} // namespace

void Search::estimateResidualQT(Mode& mode, const CUGeom& cuGeom, uint32_t absPartIdx, uint32_t tuDepth, ShortYuv& resiYuv, Cost& outCosts, const uint32_t depthRange[2],
                                int32_t splitMore)
{
    enabled();
    if (g_time)
    {
        Timed t(8 + cuGeom.log2CUSize - 2);
        t_inRqt++;
        refEstimateResidualQT(this, mode, cuGeom, absPartIdx, tuDepth, resiYuv, outCosts, depthRange, splitMore);
        t_inRqt--;
        return;
    }
    refEstimateResidualQT(this, mode, cuGeom, absPartIdx, tuDepth, resiYuv, outCosts, depthRange, splitMore);
}

void Search::checkIntraInInter(Mode& intraMode, const CUGeom& cuGeom)
{
    enabled();
    if (g_time)
    {
        Timed t(13 + cuGeom.log2CUSize - 2);
        refCheckIntraInInter(this, intraMode, cuGeom);
        return;
    }
    refCheckIntraInInter(this, intraMode, cuGeom);
}

uint32_t Quant::transformNxN(const CUData& cu, const pixel* fenc, uint32_t fencStride, const int16_t* residual, uint32_t resiStride, coeff_t* coeff, uint32_t log2TrSize,
                             TextType ttype, uint32_t absPartIdx, bool useTransformSkip)
{
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
    if (g_time)
    {
        Timed t(4 + log2TrSize - 2);
        refInvtransformNxN(this, cu, residual, resiStride, coeff, log2TrSize, ttype, bIntra, useTransformSkip, numSig);
        return;
    }
    refInvtransformNxN(this, cu, residual, resiStride, coeff, log2TrSize, ttype, bIntra, useTransformSkip, numSig);
}

} // namespace X265_NS
