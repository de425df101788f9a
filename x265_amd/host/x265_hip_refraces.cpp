// x265_hip_refraces.cpp — the seventh translation unit of the drop-in: a place where the reference's output depends on what else the PROCESS has
// done, closed for binaries that carry the bindings.  Not GPU work; it had to be told apart from a binding bug, and it
// is the reason "encoders running concurrently in one process intermittently emit a different bitstream" (round 4's review) — a process with the
// bindings allocates and frees large blocks (lookahead sessions, page-locked staging, mirrors), the reference alone does not.
//
// Analysis::m_refineLevel is read uninitialised.  compressInterCU_rd0_4 / _rd5_6 test `m_param->bEnableEarlySkip || m_refineLevel == 2`
//     (reference source/encoder/analysis.cpp:1314, :2019); the member is assigned only in recodeCU (:2435-2437, analysis refinement) and the
//     constructor (:73-83) leaves it out.  The Analysis objects live in `new ThreadLocalData[numTLD]` (frameencoder.cpp:298, 2.4 MB for four workers):
//     straight from mmap — zero pages — in a process that has never freed a large block, recycled heap memory otherwise.  With early skip off (presets
//     slow and slower) a stale 2 at that offset makes every CU whose best mode so far is a skip stop evaluating modes: another bitstream.  Shown with
//     the reference's objects alone in tests/test_reference_races.py (an allocator shim writes 2 into exactly those four bytes: 9 942 -> 9 839 bytes,
//     first difference at byte 7 333 — the very file the review's stress runs produced), found by bisecting which allocation, then which bytes of it,
//     had to be scrubbed for the mismatch to disappear.  Closed by defining the constructor here — the reference's assignments plus the
//     members it forgets, all zero: what a fresh mapping holds, i.e. what the reference alone computes with.
#define protected public
#define private public
#include "common.h"
#include "primitives.h"
#include "analysis.h"
#undef protected
#undef private

namespace X265_NS {

// the reference's constructor body (analysis.cpp:73-83) and the plain members it leaves out
Analysis::Analysis()
{
    m_reuseInterDataCTU = NULL;
    m_reuseRef = NULL;
    m_bHD = false;
    m_modeFlag[0] = false;
    m_modeFlag[1] = false;
    m_checkMergeAndSkipOnly[0] = false;
    m_checkMergeAndSkipOnly[1] = false;
    m_evaluateInter = 0;
    // not in the reference: read before any assignment at analysis.cpp:1314 / :2019 (m_refineLevel); the others are assigned before use today and
    // are zeroed for the same reason — the value the reference alone has always seen there
    m_refineLevel = 0;
    m_bTryLossless = false;
    m_bChromaSa8d = false;
    m_reuseDepth = NULL; m_reuseModes = NULL; m_reusePartSize = NULL; m_reuseMergeFlag = NULL;
    m_reuseMv[0] = m_reuseMv[1] = NULL;
    m_reuseMvpIdx[0] = m_reuseMvpIdx[1] = NULL;
    m_splitRefIdx[0] = m_splitRefIdx[1] = m_splitRefIdx[2] = m_splitRefIdx[3] = 0;
    cacheCost = NULL;
    m_additionalCtuInfo = NULL;
    m_prevCtuInfoChange = NULL;
}

} // namespace X265_NS
