// x265_hip_sadplanes.cpp — the fifth translation unit of the drop-in: integer-pel SAD served from GPU-built SAD surfaces
// (include/x265hip.h, x265hip_sadsurf_*).
//
// MotionEstimate::motionEstimate (reference source/encoder/motion.cpp:739-1569) measures its integer-pel candidates as
//     sad(fenc, FENC_STRIDE, fref + mx + my * stride, stride)                                  (:246-330 macros, :770-944 HEX, :1132-1240 STAR)
// where fenc = fencPUYuv.m_buf[0] is a copy of the SOURCE picture's PU (setSourcePU, :194-222) and fref = ref->fpelPlane[0] + blockOffset
// a position in the finished reference picture (:752-756).  Which candidates the search visits depends on the decisions before it; what a
// candidate costs does not: SAD(source block at (x, y), reference block at (x + mx, y + my)) is a function of the two pictures and the position.
// ...
//
// Two seams on motion.o (same link technique as the other seams, oracle/Makefile):
//   MotionEstimate::setSourcePU     (analysis variant): remembers, per MotionEstimate object of this thread, which source picture the PU came
//                                   from (x265hip_srcplanes_where) and verifies the copied block against that picture byte for byte;
//   MotionEstimate::motionEstimate  runs the reference's own body with this object's sad / sad_x3 / sad_x4 pointers swapped for lookups
//                                   when the (source picture, reference picture) pair has a surface that covers this PU.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

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

namespace X265_NS {

const EncoderPrimitives& x265hip_c_table();          // x265_hip_primitives.cpp
bool x265hip_srcplanes_where(const Yuv& y, const PicYuv** pic, uint32_t* version, int* px, int* py);     // x265_hip_srcplanes.cpp

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

int g_state = 0;                 // 0 undecided, 1 on, -1 off
int g_exp = 0;                   // X265HIP_DEBUG_SADEXP: 1 = statistics of the eligible calls, 2 = eligible calls are computed twice (cost doubling)
EncoderPrimitives g_c;
std::mutex g_lock;

struct alignas(64) Counter { std::atomic<uint64_t> v[8]; };
// [size class 0..3 = 8, 16, 32, 64][bucket]: |mv - centre| <= 8, 12, 16, 24, 32, more; [6] = mv == 0 outside 16; [7] = calls
Counter g_hist[4][64], g_histOwn[4][64];     // centre = the CTU's first vector / the vector this very search ends on
std::atomic<int> g_nextShard(0);
__attribute__((tls_model("initial-exec"))) thread_local int t_shard = -1;
inline int shard() { if (t_shard < 0) t_shard = g_nextShard.fetch_add(1) & 63; return t_shard; }

// what this thread's MotionEstimate objects hold (setSourcePU)
struct PuInfo { const MotionEstimate* me; const PicYuv* srcPic; uint32_t version; int x, y, w, h; };
const int kPu = 4;
__attribute__((tls_model("initial-exec"))) thread_local PuInfo t_pu[kPu];
__attribute__((tls_model("initial-exec"))) thread_local int t_puNext = 0;

// the lookup context of the motionEstimate call in progress on this thread
struct Ctx
{
    const pixel* fenc;           // fencPUYuv.m_buf[0]
    const pixel* fref;           // reference position of mv (0, 0)
    intptr_t stride;
    int sizeClass;
    int cx, cy;                  // window centre (experiment: the first vector found for this CTU and reference)
    bool haveCentre;
};
__attribute__((tls_model("initial-exec"))) thread_local Ctx t_ctx;

// experiment: the centre a device search would pick per (CTU, reference) is approximated by the first motionEstimate result in that CTU
struct CentreMemo { const PicYuv* ref; int ctu; int cx, cy; };
__attribute__((tls_model("initial-exec"))) thread_local CentreMemo t_centre[8];
__attribute__((tls_model("initial-exec"))) thread_local int t_centreNext = 0;

void report()
{
    static const char* names[4] = { "8x8", "16x16", "32x32", "64x64" };
    for (int c = 0; c < 8; c++)
    {
        uint64_t b[8] = { 0 };
        for (int s = 0; s < 64; s++)
            for (int k = 0; k < 8; k++)
                b[k] += (c < 4 ? g_hist[c][s] : g_histOwn[c - 4][s]).v[k];
        if (!b[7])
            continue;
        fprintf(stderr, c < 4 ? "x265hip: sadplanes: experiment %s: %llu eligible candidate SADs; distance from the CTU's first vector <=8: %.1f%% <=12: %.1f%% <=16: %.1f%% <=24: %.1f%% <=32: %.1f%% "
                        "more: %.1f%% (of which mv 0: %.1f%%)\n" : "x265hip: sadplanes: experiment %s: %llu eligible candidate SADs; distance from this search's own result <=8: %.1f%% <=12: %.1f%% <=16: %.1f%% <=24: %.1f%% <=32: %.1f%% "
                        "more: %.1f%% (of which mv 0: %.1f%%)\n", names[c & 3], (unsigned long long)b[7], 100.0 * b[0] / b[7], 100.0 * (b[0] + b[1]) / b[7],
                100.0 * (b[0] + b[1] + b[2]) / b[7], 100.0 * (b[0] + b[1] + b[2] + b[3]) / b[7], 100.0 * (b[0] + b[1] + b[2] + b[3] + b[4]) / b[7], 100.0 * b[5] / b[7],
                100.0 * b[6] / b[7]);
    }
}

bool enabled()
{
    if (!g_state)
    {
        std::lock_guard<std::mutex> g(g_lock);
        if (!g_state)
        {
            const char* env = getenv("X265HIP_SADPLANES");
            const char* all = getenv("X265HIP");
            const char* table = getenv("X265HIP_TABLE");
            const char* exp = getenv("X265HIP_DEBUG_SADEXP");
            g_exp = exp ? atoi(exp) : 0;
            if ((env && !strcmp(env, "0")) || (all && !strcmp(all, "0")) || (table && !strcmp(table, "percall")) || x265hip_device_count() < 1)
                g_state = -1;
            else
            {
                g_c = x265hip_c_table();
                refInitScales();            // the reference body's own file-static table (motion.cpp:60, :120-160): its copy in the second object
                g_state = 1;
                if (g_exp == 1)
                    atexit(report);
            }
        }
    }
    return g_state > 0;
}

__attribute__((tls_model("initial-exec"))) thread_local int16_t t_cand[1024][2];
__attribute__((tls_model("initial-exec"))) thread_local int t_ncand = 0;

inline void bucket(Counter& h, int mx, int my, int cx, int cy)
{
    const int dx = abs(mx - cx), dy = abs(my - cy), d = dx > dy ? dx : dy;
    const int k = d <= 8 ? 0 : d <= 12 ? 1 : d <= 16 ? 2 : d <= 24 ? 3 : d <= 32 ? 4 : 5;
    h.v[k].fetch_add(1, std::memory_order_relaxed);
    h.v[7].fetch_add(1, std::memory_order_relaxed);
    if (d > 16 && !mx && !my)
        h.v[6].fetch_add(1, std::memory_order_relaxed);
}

inline void account(int mx, int my)
{
    const Ctx& c = t_ctx;
    if (g_exp != 1)
        return;
    if (t_ncand < 1024) { t_cand[t_ncand][0] = (int16_t)mx; t_cand[t_ncand][1] = (int16_t)my; t_ncand++; }
    if (c.haveCentre)
        bucket(g_hist[c.sizeClass][shard()], mx, my, c.cx, c.cy);
}

inline bool decode(const pixel* p, int& mx, int& my)
{
    const Ctx& c = t_ctx;
    const ptrdiff_t d = p - c.fref + 128 * c.stride + 128;          // candidates within +-128 of the block position
    if (d < 0)
        return false;
    my = (int)(d / c.stride);
    mx = (int)(d - my * c.stride);
    if (my > 256 || mx > 256)
        return false;
    mx -= 128; my -= 128;
    return true;
}

template <int PART> int sad_exp(const pixel* fenc, intptr_t fs, const pixel* ref, intptr_t rs)
{
    int mx, my;
    if (fenc == t_ctx.fenc && rs == t_ctx.stride && decode(ref, mx, my))
    {
        account(mx, my);
        if (g_exp == 2)
        {
            volatile int sink = g_c.pu[PART].sad(fenc, fs, ref, rs);
            (void)sink;
        }
    }
    return g_c.pu[PART].sad(fenc, fs, ref, rs);
}
template <int PART> void sad_x3_exp(const pixel* fenc, const pixel* r0, const pixel* r1, const pixel* r2, intptr_t rs, int32_t* res)
{
    int mx, my;
    if (fenc == t_ctx.fenc && rs == t_ctx.stride && decode(r0, mx, my))
    {
        account(mx, my);
        if (decode(r1, mx, my)) account(mx, my);
        if (decode(r2, mx, my)) account(mx, my);
        if (g_exp == 2)
        {
            int32_t tmp[3];
            g_c.pu[PART].sad_x3(fenc, r0, r1, r2, rs, tmp);
            volatile int sink = tmp[0] + tmp[1] + tmp[2];
            (void)sink;
        }
    }
    g_c.pu[PART].sad_x3(fenc, r0, r1, r2, rs, res);
}
template <int PART> void sad_x4_exp(const pixel* fenc, const pixel* r0, const pixel* r1, const pixel* r2, const pixel* r3, intptr_t rs, int32_t* res)
{
    int mx, my;
    if (fenc == t_ctx.fenc && rs == t_ctx.stride && decode(r0, mx, my))
    {
        account(mx, my);
        if (decode(r1, mx, my)) account(mx, my);
        if (decode(r2, mx, my)) account(mx, my);
        if (decode(r3, mx, my)) account(mx, my);
        if (g_exp == 2)
        {
            int32_t tmp[4];
            g_c.pu[PART].sad_x4(fenc, r0, r1, r2, r3, rs, tmp);
            volatile int sink = tmp[0] + tmp[1] + tmp[2] + tmp[3];
            (void)sink;
        }
    }
    g_c.pu[PART].sad_x4(fenc, r0, r1, r2, r3, rs, res);
}

} // namespace

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
    if (pwidth != pheight || pwidth < 8 || !x265hip_srcplanes_where(srcFencYuv, &pic, &version, &cx, &cy))
        return;
    const int x = cx + g_zscanToPelX[puPartIdx], y = cy + g_zscanToPelY[puPartIdx];
    if ((x | y) & (pwidth - 1) || x + pwidth > (int)pic->m_picWidth || y + pheight > (int)pic->m_picHeight)
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
    if (g_state <= 0 || ctuAddr < 0 || srcReferencePlane || ref->isWeighted || ref->isLowres)
        return refMotionEstimate(this, ref, mvmin, mvmax, qmvp, numCandidates, mvc, merange, outQMv, maxSlices, srcReferencePlane);
    const PuInfo* u = NULL;
    for (int i = 0; i < kPu; i++)
        if (t_pu[i].me == this && t_pu[i].srcPic) { u = &t_pu[i]; break; }
    if (!u || !g_exp)
        return refMotionEstimate(this, ref, mvmin, mvmax, qmvp, numCandidates, mvc, merange, outQMv, maxSlices, srcReferencePlane);
    // experiment modes
    Ctx& c = t_ctx;
    c.fenc = fencPUYuv.m_buf[0];
    c.fref = ref->fpelPlane[0] + ref->reconPic->m_cuOffsetY[ctuAddr] + ref->reconPic->m_buOffsetY[absPartIdx];
    c.stride = ref->lumaStride;
    c.sizeClass = u->w == 8 ? 0 : u->w == 16 ? 1 : u->w == 32 ? 2 : 3;
    c.haveCentre = false;
    t_ncand = 0;
    CentreMemo* memo = NULL;
    for (int i = 0; i < 8; i++)
        if (t_centre[i].ref == ref->reconPic && t_centre[i].ctu == ctuAddr) { memo = &t_centre[i]; break; }
    if (memo) { c.cx = memo->cx; c.cy = memo->cy; c.haveCentre = true; }
    const pixelcmp_t s1 = sad; const pixelcmp_x3_t s3 = sad_x3; const pixelcmp_x4_t s4 = sad_x4;
    switch (c.sizeClass)
    {
    case 0: sad = sad_exp<LUMA_8x8>; sad_x3 = sad_x3_exp<LUMA_8x8>; sad_x4 = sad_x4_exp<LUMA_8x8>; break;
    case 1: sad = sad_exp<LUMA_16x16>; sad_x3 = sad_x3_exp<LUMA_16x16>; sad_x4 = sad_x4_exp<LUMA_16x16>; break;
    case 2: sad = sad_exp<LUMA_32x32>; sad_x3 = sad_x3_exp<LUMA_32x32>; sad_x4 = sad_x4_exp<LUMA_32x32>; break;
    default: sad = sad_exp<LUMA_64x64>; sad_x3 = sad_x3_exp<LUMA_64x64>; sad_x4 = sad_x4_exp<LUMA_64x64>; break;
    }
    const int r = refMotionEstimate(this, ref, mvmin, mvmax, qmvp, numCandidates, mvc, merange, outQMv, maxSlices, srcReferencePlane);
    sad = s1; sad_x3 = s3; sad_x4 = s4;
    c.fenc = NULL;
    for (int i = 0; i < t_ncand; i++)
        bucket(g_histOwn[c.sizeClass][shard()], t_cand[i][0], t_cand[i][1], outQMv.x >> 2, outQMv.y >> 2);
    if (!memo)
    {
        CentreMemo& m = t_centre[t_centreNext];
        t_centreNext = (t_centreNext + 1) & 7;
        m.ref = ref->reconPic; m.ctu = ctuAddr; m.cx = outQMv.x >> 2; m.cy = outQMv.y >> 2;
    }
    return r;
}

} // namespace X265_NS
