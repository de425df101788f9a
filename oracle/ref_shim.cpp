/* ref_shim.cpp — TEST INFRASTRUCTURE ONLY.
 *
 * A thin extern "C" window onto the REAL x265 reference, linked together with the reference's own objects into
 * oracle/_ref/libx265ref{8,10}.so by oracle/Makefile.  Nothing here re-implements anything: every entry point
 * dispatches through the reference's C primitive table (setupCPrimitives + setupAliasPrimitives,
 * common/primitives.cpp:63-73, :88-209) or drives the reference's MotionEstimate / BitCost classes
 * (encoder/motion.cpp, encoder/bitcost.cpp).  It is how oracle/x265_oracle.c gets pinned to the reference.
 *
 * `part` = LumaPU enum (common/primitives.h:41-55), `cu` = LumaCU enum (:57-65, log2(size)-2).
 * Pixel pointers are uint8_t* in the 8-bit build and uint16_t* in the 10-bit build (common/common.h:126-142).
 */
#include "common.h"
#include "primitives.h"
#include "bitcost.h"
#include "motion.h"
#include "lowres.h"
#include "yuv.h"
#include "x265.h"
#include "picyuv.h"
#include "slicetype.h"
#include "predict.h"
#include "shortyuv.h"
#include "frame.h"
#include "framedata.h"
#include "deblock.h"
#include "constants.h"
#include "contexts.h"
#include "cudata.h"
#include "scalinglist.h"
#include "entropy.h"
#include "quant.h"

using namespace X265_NS;

namespace {
EncoderPrimitives& T()
{
    static bool done = false;
    if (!done)
    {
        /* fill the GLOBAL table too: MotionEstimate and the hv filters read it */
        setupCPrimitives(primitives);
        setupAliasPrimitives(primitives);
        MotionEstimate::initScales();      /* as Encoder::create does (encoder.cpp:123): the UMH search reads sizeScale[] */
        done = true;
    }
    return primitives;
}

struct CostProbe : public BitCost
{
    const uint16_t* table() const { return m_cost; }
};
}

extern "C" {

int ref_depth() { return X265_DEPTH; }
int ref_sizeof_pixel() { return (int)sizeof(pixel); }

/* ---- pixel compare ---- */
int ref_sad(int part, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb) { return T().pu[part].sad(a, sa, b, sb); }
void ref_sad_x3(int part, const pixel* f, const pixel* r0, const pixel* r1, const pixel* r2, intptr_t rs, int32_t* res) { T().pu[part].sad_x3(f, r0, r1, r2, rs, res); }
void ref_sad_x4(int part, const pixel* f, const pixel* r0, const pixel* r1, const pixel* r2, const pixel* r3, intptr_t rs, int32_t* res) { T().pu[part].sad_x4(f, r0, r1, r2, r3, rs, res); }
int ref_satd(int part, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb) { return T().pu[part].satd(a, sa, b, sb); }
int ref_sa8d(int cu, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb) { return T().cu[cu].sa8d(a, sa, b, sb); }
int ref_chroma_satd(int part, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb)
{
    pixelcmp_t f = T().chroma[X265_CSP_I420].pu[part].satd;
    return f ? f(a, sa, b, sb) : -1;
}
int ref_chroma_sa8d(int cu, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb)
{
    pixelcmp_t f = T().chroma[X265_CSP_I420].cu[cu].sa8d;
    return f ? f(a, sa, b, sb) : -1;
}
uint64_t ref_sse_pp(int cu, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb) { return T().cu[cu].sse_pp(a, sa, b, sb); }
uint64_t ref_sse_ss(int cu, const int16_t* a, intptr_t sa, const int16_t* b, intptr_t sb) { return T().cu[cu].sse_ss(a, sa, b, sb); }
uint64_t ref_ssd_s(int cu, const int16_t* a, intptr_t sa) { return T().cu[cu].ssd_s[NONALIGNED](a, sa); }
int ref_psy_cost_pp(int cu, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb) { return T().cu[cu].psy_cost_pp(a, sa, b, sb); }
uint64_t ref_var(int cu, const pixel* a, intptr_t sa) { return T().cu[cu].var(a, sa); }
void ref_weight_pp(const pixel* s, pixel* d, intptr_t stride, int w, int h, int w0, int round, int shift, int offset) { T().weight_pp(s, d, stride, w, h, w0, round, shift, offset); }
void ref_weight_sp(const int16_t* s, pixel* d, intptr_t ss, intptr_t ds, int w, int h, int w0, int round, int shift, int offset) { T().weight_sp(s, d, ss, ds, w, h, w0, round, shift, offset); }
void ref_scale1d_128to64(pixel* d, const pixel* s) { T().scale1D_128to64[NONALIGNED](d, s); }
void ref_scale2d_64to32(pixel* d, const pixel* s, intptr_t stride) { T().scale2D_64to32(d, s, stride); }
void ref_transpose(int cu, pixel* d, const pixel* s, intptr_t stride) { T().cu[cu].transpose(d, s, stride); }

/* ---- block arithmetic / copies ---- */
void ref_sub_ps(int cu, int16_t* d, intptr_t ds, const pixel* a, const pixel* b, intptr_t sa, intptr_t sb) { T().cu[cu].sub_ps(d, ds, a, b, sa, sb); }
void ref_add_ps(int cu, pixel* d, intptr_t ds, const pixel* a, const int16_t* r, intptr_t sa, intptr_t sr) { T().cu[cu].add_ps[NONALIGNED](d, ds, a, r, sa, sr); }
void ref_calcresidual(int cu, const pixel* f, const pixel* p, int16_t* r, intptr_t s) { T().cu[cu].calcresidual[NONALIGNED](f, p, r, s); }
void ref_addAvg(int part, const int16_t* a, const int16_t* b, pixel* d, intptr_t sa, intptr_t sb, intptr_t ds) { T().pu[part].addAvg[NONALIGNED](a, b, d, sa, sb, ds); }
void ref_pixelavg_pp(int part, pixel* d, intptr_t ds, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb) { T().pu[part].pixelavg_pp[NONALIGNED](d, ds, a, sa, b, sb, 32); }
void ref_copy_pp(int part, pixel* d, intptr_t ds, const pixel* s, intptr_t ss) { T().pu[part].copy_pp(d, ds, s, ss); }
void ref_copy_sp(int cu, pixel* d, intptr_t ds, const int16_t* s, intptr_t ss) { T().cu[cu].copy_sp(d, ds, s, ss); }
void ref_copy_ps(int cu, int16_t* d, intptr_t ds, const pixel* s, intptr_t ss) { T().cu[cu].copy_ps(d, ds, s, ss); }
void ref_copy_ss(int cu, int16_t* d, intptr_t ds, const int16_t* s, intptr_t ss) { T().cu[cu].copy_ss(d, ds, s, ss); }
void ref_blockfill_s(int cu, int16_t* d, intptr_t ds, int16_t v) { T().cu[cu].blockfill_s[NONALIGNED](d, ds, v); }
void ref_cpy2Dto1D_shl(int cu, int16_t* d, const int16_t* s, intptr_t ss, int sh) { T().cu[cu].cpy2Dto1D_shl(d, s, ss, sh); }
void ref_cpy2Dto1D_shr(int cu, int16_t* d, const int16_t* s, intptr_t ss, int sh) { T().cu[cu].cpy2Dto1D_shr(d, s, ss, sh); }
void ref_cpy1Dto2D_shl(int cu, int16_t* d, const int16_t* s, intptr_t ds, int sh) { T().cu[cu].cpy1Dto2D_shl[NONALIGNED](d, s, ds, sh); }
void ref_cpy1Dto2D_shr(int cu, int16_t* d, const int16_t* s, intptr_t ds, int sh) { T().cu[cu].cpy1Dto2D_shr(d, s, ds, sh); }
uint32_t ref_copy_cnt(int cu, int16_t* c, const int16_t* r, intptr_t rs) { return T().cu[cu].copy_cnt(c, r, rs); }
int ref_count_nonzero(int cu, const int16_t* q) { return T().cu[cu].count_nonzero(q); }

/* ---- transforms ---- */
void ref_dct(int cu, const int16_t* s, int16_t* d, intptr_t ss) { T().cu[cu].dct(s, d, ss); }
void ref_idct(int cu, const int16_t* s, int16_t* d, intptr_t ds) { T().cu[cu].idct(s, d, ds); }
void ref_dst4(const int16_t* s, int16_t* d, intptr_t ss) { T().dst4x4(s, d, ss); }
void ref_idst4(const int16_t* s, int16_t* d, intptr_t ds) { T().idst4x4(s, d, ds); }
uint32_t ref_quant(const int16_t* c, const int32_t* qc, int32_t* du, int16_t* q, int qBits, int add, int n) { return T().quant(c, qc, du, q, qBits, add, n); }
uint32_t ref_nquant(const int16_t* c, const int32_t* qc, int16_t* q, int qBits, int add, int n) { return T().nquant(c, qc, q, qBits, add, n); }
void ref_dequant_normal(const int16_t* q, int16_t* c, int n, int scale, int shift) { T().dequant_normal(q, c, n, scale, shift); }
void ref_dequant_scaling(const int16_t* q, const int32_t* dq, int16_t* c, int n, int per, int shift) { T().dequant_scaling(q, dq, c, n, per, shift); }
void ref_denoise_dct(int16_t* c, uint32_t* rs, const uint16_t* off, int n) { T().denoiseDct(c, rs, off, n); }
void ref_nonpsy_rdoquant(int cu, int16_t* r, int64_t* cu_, int64_t* tu, int64_t* tr, uint32_t pos) { T().cu[cu].nonPsyRdoQuant(r, cu_, tu, tr, pos); }
void ref_psy_rdoquant(int cu, int16_t* r, int16_t* f, int64_t* cu_, int64_t* tu, int64_t* tr, int64_t* ps, uint32_t pos) { T().cu[cu].psyRdoQuant(r, f, cu_, tu, tr, ps, pos); }
void ref_psy_rdoquant_1p(int cu, int16_t* r, int64_t* cu_, int64_t* tu, int64_t* tr, uint32_t pos) { T().cu[cu].psyRdoQuant_1p(r, cu_, tu, tr, pos); }
void ref_psy_rdoquant_2p(int cu, int16_t* r, int16_t* f, int64_t* cu_, int64_t* tu, int64_t* tr, int64_t* ps, uint32_t pos) { T().cu[cu].psyRdoQuant_2p(r, f, cu_, tu, tr, ps, pos); }

/* ---- interpolation: chroma != 0 selects the 4:2:0 4-tap table entry of the same LumaPU index ---- */
void ref_interp_hpp(int chroma, int part, const pixel* s, intptr_t ss, pixel* d, intptr_t ds, int idx)
{ (chroma ? T().chroma[X265_CSP_I420].pu[part].filter_hpp : T().pu[part].luma_hpp)(s, ss, d, ds, idx); }
void ref_interp_hps(int chroma, int part, const pixel* s, intptr_t ss, int16_t* d, intptr_t ds, int idx, int ext)
{ (chroma ? T().chroma[X265_CSP_I420].pu[part].filter_hps : T().pu[part].luma_hps)(s, ss, d, ds, idx, ext); }
void ref_interp_vpp(int chroma, int part, const pixel* s, intptr_t ss, pixel* d, intptr_t ds, int idx)
{ (chroma ? T().chroma[X265_CSP_I420].pu[part].filter_vpp : T().pu[part].luma_vpp)(s, ss, d, ds, idx); }
void ref_interp_vps(int chroma, int part, const pixel* s, intptr_t ss, int16_t* d, intptr_t ds, int idx)
{ (chroma ? T().chroma[X265_CSP_I420].pu[part].filter_vps : T().pu[part].luma_vps)(s, ss, d, ds, idx); }
void ref_interp_vsp(int chroma, int part, const int16_t* s, intptr_t ss, pixel* d, intptr_t ds, int idx)
{ (chroma ? T().chroma[X265_CSP_I420].pu[part].filter_vsp : T().pu[part].luma_vsp)(s, ss, d, ds, idx); }
void ref_interp_vss(int chroma, int part, const int16_t* s, intptr_t ss, int16_t* d, intptr_t ds, int idx)
{ (chroma ? T().chroma[X265_CSP_I420].pu[part].filter_vss : T().pu[part].luma_vss)(s, ss, d, ds, idx); }
void ref_interp_hvpp(int part, const pixel* s, intptr_t ss, pixel* d, intptr_t ds, int ix, int iy) { T().pu[part].luma_hvpp(s, ss, d, ds, ix, iy); }
void ref_p2s(int chroma, int part, const pixel* s, intptr_t ss, int16_t* d, intptr_t ds)
{ (chroma ? T().chroma[X265_CSP_I420].pu[part].p2s[NONALIGNED] : T().pu[part].convert_p2s[NONALIGNED])(s, ss, d, ds); }

/* ---- tables ---- */
const int16_t* ref_dct_matrix(int log2n) { return log2n == 2 ? &g_t4[0][0] : log2n == 3 ? &g_t8[0][0] : log2n == 4 ? &g_t16[0][0] : &g_t32[0][0]; }
const int16_t* ref_luma_filter(int i) { return g_lumaFilter[i]; }
const int16_t* ref_chroma_filter(int i) { return g_chromaFilter[i]; }
int ref_partition_from_sizes(int w, int h) { return partitionFromSizes(w, h); }
/* BitCost::setQP (encoder/bitcost.cpp:32): copies the lambda-scaled MVD cost row, out[i + 65536], i in [-65536, 65536] */
void ref_mvcost_table(int qp, uint16_t* out)
{
    CostProbe p;
    p.setQP(qp);
    memcpy(out, p.table() - 2 * 32768, (4 * 32768 + 1) * sizeof(uint16_t));
}

/* ---- the real MotionEstimate::motionEstimate (encoder/motion.cpp:739), lookahead-style entry (:167): the source
 * PU is read from `fencPlane` at the same (bx,by)/stride as the reference plane. Returns bcost; outQMv = qpel MV. */
int ref_motion_estimate(pixel* refPlane, pixel* fencPlane, intptr_t stride, int bx, int by, int w, int h,
                        const int32_t* mvmin, const int32_t* mvmax, const int32_t* qmvp, int numCand, const int32_t* mvc,
                        int merange, int method, int subme, int qp, int32_t* outQMv)
{
    T();
    MotionEstimate me;
    me.init(X265_CSP_I400);
    me.setQP(qp);
    me.setSourcePU(fencPlane, stride, bx + (intptr_t)by * stride, w, h, method, method, method, subme);
    ReferencePlanes ref;
    ref.fpelPlane[0] = refPlane;
    ref.lumaStride = stride;
    ref.isLowres = false;
    ref.isHMELowres = false;
    MV cands[16];
    for (int i = 0; i < numCand && i < 16; i++)
        cands[i] = MV(mvc[2 * i], mvc[2 * i + 1]);
    MV out(0, 0);
    int cost = me.motionEstimate(&ref, MV(mvmin[0], mvmin[1]), MV(mvmax[0], mvmax[1]), MV(qmvp[0], qmvp[1]),
                                 numCand, cands, merange, out, 1, 0);
    outQMv[0] = out.x;
    outQMv[1] = out.y;
    return cost;
}

/* ---- the real window-sum planes of --me sea: FrameFilter::computeMEIntegral (encoder/framefilter.cpp:684-830) needs a whole FrameEncoder; this
 * drives the reference's own integral_inith / integral_initv primitives over all rows of a padded picture in the order that function does
 * (row y: horizontal running sums accumulated onto row y + 1 of every plane, then the vertical difference h rows up).  planes = 12 buffers of
 * planeElems uint32 (zeroed by the caller), laid out like the picture buffer (origin at padY * stride + padX).  rows = maxHeight + 2 * padY. */
void ref_integral_planes(pixel* picOrg, intptr_t stride, int maxHeight, int padX, int padY, uint32_t* planes, int64_t planeElems)
{
    EncoderPrimitives& p = T();
    static const int win[12][2] = { {32,32},{32,24},{32,8},{24,32},{16,16},{16,12},{16,4},{12,16},{8,32},{8,8},{4,16},{4,4} };
    auto idx = [](int v) { return v == 4 ? INTEGRAL_4 : v == 8 ? INTEGRAL_8 : v == 12 ? INTEGRAL_12 : v == 16 ? INTEGRAL_16 : v == 24 ? INTEGRAL_24 : INTEGRAL_32; };
    for (int y = -padY; y < maxHeight + padY - 1; y++)
    {
        pixel* pix = picOrg + (intptr_t)y * stride - padX;
        for (int k = 0; k < 12; k++)
        {
            uint32_t* org = planes + (int64_t)k * planeElems + (intptr_t)padY * stride + padX;
            uint32_t* sum = org + (intptr_t)(y + 1) * stride - padX;
            p.integral_inith[idx(win[k][0])](sum, pix, stride);
            if (y >= win[k][1] - padY)
                p.integral_initv[idx(win[k][1])](sum - (intptr_t)win[k][1] * stride, stride);
        }
    }
}

/* the real MotionEstimate::motionEstimate with X265_SEA (motion.cpp:1242-1395); `planes` as ref_integral_planes left them */
int ref_motion_estimate_sea(pixel* refPlane, pixel* fencPlane, intptr_t stride, uint32_t* planes, int64_t planeElems, int padX, int padY, int bx, int by, int w, int h,
                            const int32_t* mvmin, const int32_t* mvmax, const int32_t* qmvp, int numCand, const int32_t* mvc,
                            int merange, int subme, int qp, int32_t* outQMv)
{
    T();
    MotionEstimate me;
    me.init(X265_CSP_I400);
    me.setQP(qp);
    me.setSourcePU(fencPlane, stride, bx + (intptr_t)by * stride, w, h, X265_SEA, X265_SEA, X265_SEA, subme);
    for (int k = 0; k < INTEGRAL_PLANE_NUM; k++)
        me.integral[k] = planes + (int64_t)k * planeElems + (intptr_t)padY * stride + padX + bx + (intptr_t)by * stride;     /* search.cpp: planes moved to the PU */
    ReferencePlanes ref;
    ref.fpelPlane[0] = refPlane;
    ref.lumaStride = stride;
    ref.isLowres = false;
    ref.isHMELowres = false;
    MV cands[16];
    for (int i = 0; i < numCand && i < 16; i++)
        cands[i] = MV(mvc[2 * i], mvc[2 * i + 1]);
    MV out(0, 0);
    int cost = me.motionEstimate(&ref, MV(mvmin[0], mvmin[1]), MV(mvmax[0], mvmax[1]), MV(qmvp[0], qmvp[1]), numCand, cands, merange, out, 1, 0);
    outQMv[0] = out.x;
    outQMv[1] = out.y;
    return cost;
}

/* ---- intra prediction primitives (common/intrapred.cpp via the C table), cu = log2(size) - 2 ---- */
void ref_intra_filter(int cu, const pixel* nb, pixel* out) { T().cu[cu].intra_filter(nb, out); }
void ref_intra_pred(int cu, int mode, pixel* dst, intptr_t ds, const pixel* nb, int bFilter) { T().cu[cu].intra_pred[mode](dst, ds, nb, mode, bFilter); }
void ref_intra_allangs(int cu, pixel* dest, pixel* nb, pixel* nbf, int bLuma) { T().cu[cu].intra_pred_allangs(dest, nb, nbf, bLuma); }
int ref_intra_filter_flags(int mode) { return g_intraFilterFlags[mode]; }
void ref_frame_init_lowres(const pixel* src, pixel* d0, pixel* dh, pixel* dv, pixel* dc, intptr_t ss, intptr_t ds, int w, int h)
{ T().frameInitLowres(src, d0, dh, dv, dc, ss, ds, w, h); }

/* ---- the real Lowres::create/init (common/lowres.cpp:50,259) + LookaheadTLD::lowresIntraEstimate
 * (encoder/slicetype.cpp:696) on a caller-owned source plane with extended margins.  AQ off (invQscaleFactor NULL).
 * geom[0..3] = lumaStride, width, lines, planesize; planes receives the four padded hpel planes (4*planesize).
 * Returns costEst[0][0], or -1 if the buffers are too small (cap = elements available in `planes`). */
int ref_lowres_intra_estimate(pixel* picOrg, intptr_t stride, int w, int h, int marginX, int marginY,
                              int64_t* geom, pixel* planes, int64_t cap, int32_t* intraCost, uint8_t* intraMode, int32_t* rowSatd)
{
    T();
    x265_param* param = x265_param_alloc();
    x265_param_default(param);
    param->sourceWidth = w;
    param->sourceHeight = h;
    param->rc.aqMode = 0;
    param->rc.hevcAq = 0;
    param->bAQMotion = 0;
    param->bEnableHME = 0;
    PicYuv pic;
    pic.m_picWidth = w;
    pic.m_picHeight = h;
    pic.m_lumaMarginX = marginX;
    pic.m_lumaMarginY = marginY;
    pic.m_stride = stride;
    pic.m_picOrg[0] = picOrg;
    pic.m_param = param;
    Lowres lr;
    memset((void*)&lr, 0, sizeof(lr));
    int ret = -1;
    if (lr.create(param, &pic, param->rc.qgSize))
    {
        size_t planesize = lr.lumaStride * (lr.lines + 2 * marginY);
        geom[0] = lr.lumaStride; geom[1] = lr.width; geom[2] = lr.lines; geom[3] = (int64_t)planesize;
        if ((int64_t)(4 * planesize) <= cap)
        {
            lr.init(&pic, 0);
            LookaheadTLD tld;
            tld.init(lr.maxBlocksInRow, lr.maxBlocksInCol, lr.maxBlocksInRow * lr.maxBlocksInCol);
            tld.lowresIntraEstimate(lr, param->rc.qgSize);
            memcpy(planes, lr.buffer[0], 4 * planesize * sizeof(pixel));
            int ncu = lr.maxBlocksInRow * lr.maxBlocksInCol;
            memcpy(intraCost, lr.intraCost, ncu * sizeof(int32_t));
            memcpy(intraMode, lr.intraMode, ncu);
            memcpy(rowSatd, lr.rowSatds[0][0], lr.maxBlocksInCol * sizeof(int32_t));
            ret = lr.costEst[0][0];
        }
    }
    lr.destroy();
    pic.m_picOrg[0] = NULL;
    pic.m_param = NULL;
    x265_param_free(param);
    return ret;
}

/* ---- the real lookahead P-frame cost: Lowres::create/init of two pictures, LookaheadTLD::lowresIntraEstimate of the
 * second, then CostEstimateGroup::estimateCUCost (encoder/slicetype.cpp:3218) over every 8x8 block of frame 1 with frame 0
 * as its reference.  numSlices == 1 goes through the reference's own singleCost() -> estimateFrameCost() serial loop
 * (:3021, :3170-3179); numSlices > 1 repeats the cooperative-slice row loop of processTasks (:3092-3104, which needs a
 * thread pool to be reached) around the reference's estimateCUCost, and adds the slice sums as :3152-3158 does.
 * AQ, weighted prediction and HME are off.  Outputs are frame 1's lowresMvs[0][1], lowresMvCosts[0][1], lowresCosts[1][0],
 * rowSatds[1][0], intraMbs[1], intraCost.  Returns costEst[1][0]. */
namespace {
struct CostProbeGroup : public CostEstimateGroup
{
    CostProbeGroup(Lookahead& l, Lowres** f) : CostEstimateGroup(l, f) {}
    void cu(LookaheadTLD& tld, int x, int y, bool doSearch[2], bool lastRow, int slice) { estimateCUCost(tld, x, y, 0, 1, 1, doSearch, lastRow, slice, 0); }
};
}

static int64_t lookahead_cost_p_impl(pixel* pic0, pixel* pic1, intptr_t stride, int w, int h, int marginX, int marginY,
                                     int numRowsPerSlice, int numSlices,
                                     int32_t* mvs, int32_t* mvCosts, uint16_t* lowresCosts, int32_t* rowSatds, int32_t* intraMbs, int32_t* intraCost,
                                     const uint64_t* wpStats, int* isWeighted)
{
    T();
    x265_param* param = x265_param_alloc();
    x265_param_default(param);
    param->sourceWidth = w;
    param->sourceHeight = h;
    param->rc.aqMode = 0;
    param->rc.hevcAq = 0;
    param->bAQMotion = 0;
    param->bEnableHME = 0;
    param->bEnableWeightedPred = wpStats ? 1 : 0;      /* estimateFrameCost then runs weightsAnalyse and searches list 0 on the weighted planes (:3136-3138, :3222) */
    param->bEnableWeightedBiPred = 0;
    param->lookaheadSlices = 0;
    PicYuv pics[2];
    Lowres lr[2];
    pixel* org[2] = { pic0, pic1 };
    int64_t ret = -1;
    bool ok = true;
    for (int i = 0; i < 2; i++)
    {
        pics[i].m_picWidth = w;
        pics[i].m_picHeight = h;
        pics[i].m_lumaMarginX = marginX;
        pics[i].m_lumaMarginY = marginY;
        pics[i].m_stride = stride;
        pics[i].m_picOrg[0] = org[i];
        pics[i].m_param = param;
        memset((void*)&lr[i], 0, sizeof(Lowres));
        ok = ok && lr[i].create(param, &pics[i], param->rc.qgSize);
    }
    if (ok)
    {
        lr[0].init(&pics[0], 0);
        lr[1].init(&pics[1], 1);
        Lookahead la(param, NULL);
        la.create();
        LookaheadTLD& tld = la.m_tld[0];
        tld.lowresIntraEstimate(lr[1], param->rc.qgSize);
        if (wpStats)
        {
            lr[1].wp_ssd[0] = wpStats[0]; lr[1].wp_sum[0] = wpStats[1];
            lr[0].wp_ssd[0] = wpStats[2]; lr[0].wp_sum[0] = wpStats[3];
        }
        Lowres* frames[2] = { &lr[0], &lr[1] };
        CostProbeGroup g(la, frames);
        Lowres* fenc = frames[1];
        const int W = la.m_8x8Width, H = la.m_8x8Height, ncu = W * H;
        if (numSlices <= 1)
            ret = g.singleCost(0, 1, 1, false);
        else
        {
            bool doSearch[2] = { true, false };
            fenc->weightedRef[1].isWeighted = false;
            fenc->costEst[1][0] = 0;
            fenc->costEstAq[1][0] = 0;
            memset(g.m_slice, 0, sizeof(g.m_slice));
            for (int sl = 0; sl < numSlices; sl++)
            {
                int firstY = numRowsPerSlice * sl;
                int lastY = sl == numSlices - 1 ? H - 1 : numRowsPerSlice * (sl + 1) - 1;
                bool lastRow = true;
                for (int cuY = lastY; cuY >= firstY; cuY--)
                {
                    fenc->rowSatds[1][0][cuY] = 0;
                    for (int cuX = W - 1; cuX >= 0; cuX--)
                        g.cu(tld, cuX, cuY, doSearch, lastRow, sl);
                    lastRow = false;
                }
            }
            for (int sl = 0; sl < numSlices; sl++)
            {
                fenc->costEst[1][0] += g.m_slice[sl].costEst;
                fenc->intraMbs[1] += g.m_slice[sl].intraMbs;
            }
            ret = fenc->costEst[1][0];
        }
        for (int i = 0; i < ncu; i++)
        {
            mvs[2 * i] = fenc->lowresMvs[0][1][i].x;
            mvs[2 * i + 1] = fenc->lowresMvs[0][1][i].y;
        }
        memcpy(mvCosts, fenc->lowresMvCosts[0][1], ncu * sizeof(int32_t));
        memcpy(lowresCosts, fenc->lowresCosts[1][0], ncu * sizeof(uint16_t));
        memcpy(rowSatds, fenc->rowSatds[1][0], H * sizeof(int32_t));
        memcpy(intraCost, fenc->intraCost, ncu * sizeof(int32_t));
        *intraMbs = fenc->intraMbs[1];
        if (isWeighted) *isWeighted = fenc->weightedRef[1].isWeighted ? 1 : 0;
        la.destroy();
    }
    for (int i = 0; i < 2; i++)
    {
        lr[i].destroy();
        pics[i].m_picOrg[0] = NULL;
        pics[i].m_param = NULL;
    }
    x265_param_free(param);
    return ret;
}

int64_t ref_lookahead_cost_p(pixel* pic0, pixel* pic1, intptr_t stride, int w, int h, int marginX, int marginY,
                             int numRowsPerSlice, int numSlices,
                             int32_t* mvs, int32_t* mvCosts, uint16_t* lowresCosts, int32_t* rowSatds, int32_t* intraMbs, int32_t* intraCost)
{
    return lookahead_cost_p_impl(pic0, pic1, stride, w, h, marginX, marginY, numRowsPerSlice, numSlices, mvs, mvCosts, lowresCosts, rowSatds, intraMbs,
                                 intraCost, NULL, NULL);
}

/* the same with --weightp on (serial path only): wpStats = { fenc wp_ssd, fenc wp_sum, ref wp_ssd, ref wp_sum } */
int64_t ref_lookahead_cost_p_weightp(pixel* pic0, pixel* pic1, intptr_t stride, int w, int h, int marginX, int marginY, const uint64_t* wpStats,
                                     int32_t* mvs, int32_t* mvCosts, uint16_t* lowresCosts, int32_t* rowSatds, int32_t* intraMbs, int32_t* intraCost,
                                     int* isWeighted)
{
    return lookahead_cost_p_impl(pic0, pic1, stride, w, h, marginX, marginY, 1 << 20, 1, mvs, mvCosts, lowresCosts, rowSatds, intraMbs, intraCost, wpStats,
                                 isWeighted);
}

/* ---- the real MotionEstimate::motionEstimate with the chroma SATD term of subpelCompare (encoder/motion.cpp:1601-1660), the
 * way Search::predInterSearch runs it on a 4:2:0 picture.  The search, subpelCompare and every primitive are the reference's;
 * only the PU set-up is done here (what setSourcePU(const Yuv&, ...) does at motion.cpp:196-224, without needing a CTU-sized
 * Yuv and the encoder's z-scan tables): luma through the lookahead-style setSourcePU, chroma source copied into the PU cache,
 * bChromaSATD = subme > 2 && chromaSatd (:212), and a one-entry offset table so ReferencePlanes::getCbAddr / getLumaAddr
 * resolve to the PU position. */
namespace {
struct ChromaME : public MotionEstimate
{
    void arm(int subme, const pixel* fencCb, const pixel* fencCr, intptr_t fencStrideC, int w, int h)
    {
        chromaSatd = primitives.chroma[X265_CSP_I420].pu[partEnum].satd;
        bChromaSATD = subme > 2 && chromaSatd;
        for (int y = 0; y < h / 2; y++)
        {
            memcpy(fencPUYuv.m_buf[1] + y * fencPUYuv.m_csize, fencCb + y * fencStrideC, (w / 2) * sizeof(pixel));
            memcpy(fencPUYuv.m_buf[2] + y * fencPUYuv.m_csize, fencCr + y * fencStrideC, (w / 2) * sizeof(pixel));
        }
        ctuAddr = 0;
        absPartIdx = 0;
    }
};
}

int ref_motion_estimate_chroma(pixel* refY, pixel* refCb, pixel* refCr, pixel* fencY, pixel* fencCb, pixel* fencCr, intptr_t stride, intptr_t strideC,
                               int bx, int by, int w, int h, const int32_t* mvmin, const int32_t* mvmax, const int32_t* qmvp,
                               int numCand, const int32_t* mvc, int merange, int method, int subme, int qp, int32_t* outQMv)
{
    T();
    ChromaME me;
    me.init(X265_CSP_I420);
    me.setQP(qp);
    me.setSourcePU(fencY, stride, bx + (intptr_t)by * stride, w, h, method, method, method, subme);
    me.arm(subme, fencCb + (bx >> 1) + (intptr_t)(by >> 1) * strideC, fencCr + (bx >> 1) + (intptr_t)(by >> 1) * strideC, strideC, w, h);
    PicYuv recon;
    intptr_t zero = 0, offY = bx + (intptr_t)by * stride, offC = (bx >> 1) + (intptr_t)(by >> 1) * strideC;
    recon.m_cuOffsetY = &zero;
    recon.m_cuOffsetC = &zero;
    recon.m_buOffsetY = &offY;
    recon.m_buOffsetC = &offC;
    recon.m_picOrg[0] = refY;
    recon.m_picOrg[1] = refCb;
    recon.m_picOrg[2] = refCr;
    recon.m_stride = stride;
    recon.m_strideC = strideC;
    ReferencePlanes ref;
    ref.fpelPlane[0] = refY;
    ref.fpelPlane[1] = refCb;
    ref.fpelPlane[2] = refCr;
    ref.reconPic = &recon;
    ref.lumaStride = stride;
    ref.chromaStride = strideC;
    MV cands[16];
    for (int i = 0; i < numCand && i < 16; i++)
        cands[i] = MV(mvc[2 * i], mvc[2 * i + 1]);
    MV out(0, 0);
    int cost = me.motionEstimate(&ref, MV(mvmin[0], mvmin[1]), MV(mvmax[0], mvmax[1]), MV(qmvp[0], qmvp[1]),
                                 numCand, cands, merange, out, 1, 0);
    outQMv[0] = out.x;
    outQMv[1] = out.y;
    recon.m_picOrg[0] = recon.m_picOrg[1] = recon.m_picOrg[2] = NULL;
    recon.m_cuOffsetY = recon.m_cuOffsetC = recon.m_buOffsetY = recon.m_buOffsetC = NULL;
    return cost;
}

/* ---- the real lookahead B-frame cost: three pictures p0 = 0, b = 1, p1 = 2; CostEstimateGroup::singleCost(0, 2, 1)
 * (numSlices == 1) or the cooperative-slice row loop around estimateCUCost as in ref_lookahead_cost_p.  prefillL0 != 0 first
 * runs the P estimate (0, 1, 1) so list 0 is already searched (bDoSearch[0] false, slicetype.cpp:3126) and its stored costs are
 * reused.  Outputs are frame 1's lowresMvs / lowresMvCosts of both lists at distance 1, lowresCosts[1][1], rowSatds[1][1].
 * Returns costEst[1][1], i.e. the raw sum scaled by 100 / (130 + bFrameBias) (:3183-3186). */
int64_t ref_lookahead_cost_b(pixel* pic0, pixel* pic1, pixel* pic2, intptr_t stride, int w, int h, int marginX, int marginY,
                             int numRowsPerSlice, int numSlices, int prefillL0,
                             int32_t* mvs0, int32_t* mvCosts0, int32_t* mvs1, int32_t* mvCosts1, uint16_t* lowresCosts, int32_t* rowSatds)
{
    T();
    x265_param* param = x265_param_alloc();
    x265_param_default(param);
    param->sourceWidth = w;
    param->sourceHeight = h;
    param->rc.aqMode = 0;
    param->rc.hevcAq = 0;
    param->bAQMotion = 0;
    param->bEnableHME = 0;
    param->bEnableWeightedPred = 0;
    param->bEnableWeightedBiPred = 0;
    param->lookaheadSlices = 0;
    PicYuv pics[3];
    Lowres lr[3];
    pixel* org[3] = { pic0, pic1, pic2 };
    int64_t ret = -1;
    bool ok = true;
    for (int i = 0; i < 3; i++)
    {
        pics[i].m_picWidth = w;
        pics[i].m_picHeight = h;
        pics[i].m_lumaMarginX = marginX;
        pics[i].m_lumaMarginY = marginY;
        pics[i].m_stride = stride;
        pics[i].m_picOrg[0] = org[i];
        pics[i].m_param = param;
        memset((void*)&lr[i], 0, sizeof(Lowres));
        ok = ok && lr[i].create(param, &pics[i], param->rc.qgSize);
    }
    if (ok)
    {
        for (int i = 0; i < 3; i++)
            lr[i].init(&pics[i], i);
        Lookahead la(param, NULL);
        la.create();
        LookaheadTLD& tld = la.m_tld[0];
        tld.lowresIntraEstimate(lr[1], param->rc.qgSize);
        Lowres* frames[3] = { &lr[0], &lr[1], &lr[2] };
        Lowres* fenc = frames[1];
        const int W = la.m_8x8Width, H = la.m_8x8Height, ncu = W * H;
        if (prefillL0)
        {
            CostProbeGroup g0(la, frames);
            g0.singleCost(0, 1, 1, false);
        }
        if (numSlices <= 1)
        {
            CostProbeGroup g(la, frames);
            ret = g.singleCost(0, 2, 1, false);
        }
        else
        {
            struct BGroup : public CostEstimateGroup
            {
                BGroup(Lookahead& l, Lowres** f) : CostEstimateGroup(l, f) {}
                void cu(LookaheadTLD& t, int x, int y, bool ds[2], bool lastRow, int slice) { estimateCUCost(t, x, y, 0, 2, 1, ds, lastRow, slice, 0); }
            } g(la, frames);
            bool doSearch[2] = { fenc->lowresMvs[0][1][0].x == 0x7FFF, fenc->lowresMvs[1][1][0].x == 0x7FFF };
            fenc->weightedRef[1].isWeighted = false;
            fenc->costEst[1][1] = 0;
            fenc->costEstAq[1][1] = 0;
            memset(g.m_slice, 0, sizeof(g.m_slice));
            for (int sl = 0; sl < numSlices; sl++)
            {
                int firstY = numRowsPerSlice * sl;
                int lastY = sl == numSlices - 1 ? H - 1 : numRowsPerSlice * (sl + 1) - 1;
                bool lastRow = true;
                for (int cuY = lastY; cuY >= firstY; cuY--)
                {
                    fenc->rowSatds[1][1][cuY] = 0;
                    for (int cuX = W - 1; cuX >= 0; cuX--)
                        g.cu(tld, cuX, cuY, doSearch, lastRow, sl);
                    lastRow = false;
                }
            }
            int64_t score = 0;
            for (int sl = 0; sl < numSlices; sl++)
                score += g.m_slice[sl].costEst;
            ret = score * 100 / (130 + param->bFrameBias);
        }
        for (int i = 0; i < ncu; i++)
        {
            mvs0[2 * i] = fenc->lowresMvs[0][1][i].x; mvs0[2 * i + 1] = fenc->lowresMvs[0][1][i].y;
            mvs1[2 * i] = fenc->lowresMvs[1][1][i].x; mvs1[2 * i + 1] = fenc->lowresMvs[1][1][i].y;
        }
        memcpy(mvCosts0, fenc->lowresMvCosts[0][1], ncu * sizeof(int32_t));
        memcpy(mvCosts1, fenc->lowresMvCosts[1][1], ncu * sizeof(int32_t));
        memcpy(lowresCosts, fenc->lowresCosts[1][1], ncu * sizeof(uint16_t));
        memcpy(rowSatds, fenc->rowSatds[1][1], H * sizeof(int32_t));
        la.destroy();
    }
    for (int i = 0; i < 3; i++)
    {
        lr[i].destroy();
        pics[i].m_picOrg[0] = NULL;
        pics[i].m_param = NULL;
    }
    x265_param_free(param);
    return ret;
}

/* ---- the real bi-predictive motion compensation pieces: Predict::predInterLumaShort / predInterChromaShort (common/predict.cpp:268, :364)
 * for both references and Yuv::addAvg (common/yuv.cpp:189), i.e. what Predict::motionCompensation does for an unweighted bi-predicted PU
 * (:131-199).  One-entry offset tables place the PU at (bx, by) of caller-owned planes, as in ref_motion_estimate_chroma. */
void ref_pred_inter_bi(pixel* r0y, pixel* r0cb, pixel* r0cr, pixel* r1y, pixel* r1cb, pixel* r1cr, intptr_t stride, intptr_t strideC,
                       int bx, int by, int w, int h, const int32_t* mv0, const int32_t* mv1, pixel* dstY, pixel* dstCb, pixel* dstCr)
{
    T();
    Predict pr;
    pr.allocBuffers(X265_CSP_I420);
    ShortYuv s[2];
    Yuv out;
    s[0].create(MAX_CU_SIZE, X265_CSP_I420);
    s[1].create(MAX_CU_SIZE, X265_CSP_I420);
    out.create(MAX_CU_SIZE, X265_CSP_I420);
    intptr_t zero = 0, offY = bx + (intptr_t)by * stride, offC = (bx >> 1) + (intptr_t)(by >> 1) * strideC;
    pixel* planes[2][3] = { { r0y, r0cb, r0cr }, { r1y, r1cb, r1cr } };
    const int32_t* mvs[2] = { mv0, mv1 };
    /* PredictionUnit only has the (CUData, CUGeom, puIdx) constructor; it is five plain fields, filled here directly */
    alignas(PredictionUnit) unsigned char puRaw[sizeof(PredictionUnit)];
    PredictionUnit& pu = *reinterpret_cast<PredictionUnit*>(puRaw);
    pu.ctuAddr = 0; pu.cuAbsPartIdx = 0; pu.puAbsPartIdx = 0; pu.width = w; pu.height = h;
    for (int l = 0; l < 2; l++)
    {
        PicYuv ref;
        ref.m_cuOffsetY = &zero; ref.m_cuOffsetC = &zero; ref.m_buOffsetY = &offY; ref.m_buOffsetC = &offC;
        ref.m_picOrg[0] = planes[l][0]; ref.m_picOrg[1] = planes[l][1]; ref.m_picOrg[2] = planes[l][2];
        ref.m_stride = stride; ref.m_strideC = strideC;
        MV mv(mvs[l][0], mvs[l][1]);
        pr.predInterLumaShort(pu, s[l], ref, mv);
        pr.predInterChromaShort(pu, s[l], ref, mv);
        ref.m_picOrg[0] = ref.m_picOrg[1] = ref.m_picOrg[2] = NULL;
        ref.m_cuOffsetY = ref.m_cuOffsetC = ref.m_buOffsetY = ref.m_buOffsetC = NULL;
    }
    out.addAvg(s[0], s[1], 0, w, h, true, true);
    for (int y = 0; y < h; y++)
        memcpy(dstY + y * w, out.m_buf[0] + y * out.m_size, w * sizeof(pixel));
    for (int y = 0; y < h / 2; y++)
    {
        memcpy(dstCb + y * (w / 2), out.m_buf[1] + y * out.m_csize, (w / 2) * sizeof(pixel));
        memcpy(dstCr + y * (w / 2), out.m_buf[2] + y * out.m_csize, (w / 2) * sizeof(pixel));
    }
    s[0].destroy(); s[1].destroy(); out.destroy();
}

/* ---- the real Predict::motionCompensation (common/predict.cpp:77-266), every branch: a one-PU CUData / Slice / PPS around caller-owned
 * planes.  r1y == NULL: uni-prediction (sliceP: a P slice; else a B slice whose only reference sits in list `uniList`); wp0 / wp1: three
 * {inputWeight, inputOffset, log2WeightDenom, wtPresent} per list or NULL = weighted prediction off in the PPS.  dst* = h x w blocks. */
void ref_motion_compensation(pixel* r0y, pixel* r0cb, pixel* r0cr, pixel* r1y, pixel* r1cb, pixel* r1cr, intptr_t stride, intptr_t strideC,
                             int bx, int by, int w, int h, const int32_t* mv0, const int32_t* mv1, const int32_t* wp0, const int32_t* wp1,
                             int sliceP, int uniList, pixel* dstY, pixel* dstCb, pixel* dstCr)
{
    T();
    Predict pr;
    pr.allocBuffers(X265_CSP_I420);
    Yuv out;
    out.create(MAX_CU_SIZE, X265_CSP_I420);
    x265_param param;
    memset(&param, 0, sizeof(param));
    param.maxCUSize = 64;
    FrameData enc;
    enc.m_param = &param;
    alignas(SPS) unsigned char spsRaw[sizeof(SPS)];
    alignas(PPS) unsigned char ppsRaw[sizeof(PPS)];
    memset(spsRaw, 0, sizeof(spsRaw));
    memset(ppsRaw, 0, sizeof(ppsRaw));
    SPS& sps = *reinterpret_cast<SPS*>(spsRaw);
    PPS& pps = *reinterpret_cast<PPS*>(ppsRaw);
    sps.picWidthInLumaSamples = 1 << 16;                  /* clipMv (cudata.cpp:1915) leaves the test vectors alone */
    sps.picHeightInLumaSamples = 1 << 16;
    const bool bi = r1y != NULL;
    pps.bUseWeightPred = wp0 != NULL;
    pps.bUseWeightedBiPred = bi ? (wp0 != NULL && wp1 != NULL) : (wp0 != NULL);
    Slice slice;
    slice.m_sps = &sps;
    slice.m_pps = &pps;
    slice.m_sliceType = sliceP ? P_SLICE : B_SLICE;
    slice.m_numRefIdx[0] = slice.m_numRefIdx[1] = 1;
    intptr_t zero = 0, offY = bx + (intptr_t)by * stride, offC = (bx >> 1) + (intptr_t)(by >> 1) * strideC;
    PicYuv pics[2];
    pixel* planes[2][3] = { { r0y, r0cb, r0cr }, { r1y, r1cb, r1cr } };
    const int32_t* mvIn[2] = { mv0, mv1 };
    const int32_t* wpIn[2] = { wp0, wp1 };
    int8_t refIdx[2][1] = { { -1 }, { -1 } };
    MV mvs[2][1];
    CUData cu;
    cu.m_slice = &slice;
    cu.m_encData = &enc;
    for (int k = 0; k < (bi ? 2 : 1); k++)
    {
        const int list = bi ? k : (sliceP ? 0 : uniList);  /* input k goes to this list */
        PicYuv& ref = pics[list];
        ref.m_cuOffsetY = &zero; ref.m_cuOffsetC = &zero; ref.m_buOffsetY = &offY; ref.m_buOffsetC = &offC;
        ref.m_picOrg[0] = planes[k][0]; ref.m_picOrg[1] = planes[k][1]; ref.m_picOrg[2] = planes[k][2];
        ref.m_stride = stride; ref.m_strideC = strideC;
        slice.m_refReconPicList[list][0] = &ref;
        refIdx[list][0] = 0;
        mvs[list][0] = MV(mvIn[k][0], mvIn[k][1]);
        if (wpIn[k])
            for (int pl = 0; pl < 3; pl++)
            {
                WeightParam& p = slice.m_weightPredTable[list][0][pl];
                p.inputWeight = wpIn[k][4 * pl]; p.inputOffset = wpIn[k][4 * pl + 1];
                p.log2WeightDenom = (uint32_t)wpIn[k][4 * pl + 2]; p.wtPresent = wpIn[k][4 * pl + 3];
            }
    }
    cu.m_refIdx[0] = refIdx[0]; cu.m_refIdx[1] = refIdx[1];
    cu.m_mv[0] = mvs[0]; cu.m_mv[1] = mvs[1];
    alignas(PredictionUnit) unsigned char puRaw[sizeof(PredictionUnit)];
    PredictionUnit& pu = *reinterpret_cast<PredictionUnit*>(puRaw);
    pu.ctuAddr = 0; pu.cuAbsPartIdx = 0; pu.puAbsPartIdx = 0; pu.width = w; pu.height = h;
    pr.motionCompensation(cu, pu, out, true, true);
    for (int y = 0; y < h; y++)
        memcpy(dstY + y * w, out.m_buf[0] + y * out.m_size, w * sizeof(pixel));
    for (int y = 0; y < h / 2; y++)
    {
        memcpy(dstCb + y * (w / 2), out.m_buf[1] + y * out.m_csize, (w / 2) * sizeof(pixel));
        memcpy(dstCr + y * (w / 2), out.m_buf[2] + y * out.m_csize, (w / 2) * sizeof(pixel));
    }
    for (int l = 0; l < 2; l++)
    {
        pics[l].m_picOrg[0] = pics[l].m_picOrg[1] = pics[l].m_picOrg[2] = NULL;
        pics[l].m_cuOffsetY = pics[l].m_cuOffsetC = pics[l].m_buOffsetY = pics[l].m_buOffsetC = NULL;
    }
    out.destroy();
}

/* ---- coefficient-scan cost primitives (common/dct.cpp:757-1006 via the C table) and the tables they walk (constants.cpp) ---- */
void ref_scan_order(int type, int sizeIdx, uint16_t* out) { memcpy(out, g_scanOrder[type][sizeIdx], (size_t)(16 << (2 * sizeIdx)) * sizeof(uint16_t)); }
void ref_scan4x4(int type, uint16_t* out) { memcpy(out, g_scan4x4[type], 16 * sizeof(uint16_t)); }
void ref_entropy_state_bits(uint32_t* out) { memcpy(out, PFX(entropyStateBits), 128 * sizeof(uint32_t)); }
int ref_scanPosLast(const uint16_t* scan, const coeff_t* coeff, uint16_t* sign, uint16_t* flag, uint8_t* num, int numSig, const uint16_t* cg4x4, int trSize)
{
    return T().scanPosLast(scan, coeff, sign, flag, num, numSig, cg4x4, trSize);
}
uint32_t ref_findPosFirstLast(const int16_t* c, intptr_t trSize, const uint16_t* scanTbl) { return T().findPosFirstLast(c, trSize, scanTbl); }
uint32_t ref_costCoeffNxN(const uint16_t* scan, const coeff_t* coeff, intptr_t trSize, uint16_t* absCoeff, const uint8_t* tabSigCtx, uint32_t mask,
                          uint8_t* baseCtx, int offset, int scanPosSigOff, int subPosBase)
{
    return T().costCoeffNxN(scan, coeff, trSize, absCoeff, tabSigCtx, mask, baseCtx, offset, scanPosSigOff, subPosBase);
}
uint32_t ref_costCoeffRemain(uint16_t* absCoeff, int numNonZero, int idx) { return T().costCoeffRemain(absCoeff, numNonZero, idx); }
uint32_t ref_costC1C2Flag(uint16_t* absCoeff, intptr_t n, uint8_t* ctx, intptr_t ctxOffset) { return T().costC1C2Flag(absCoeff, n, ctx, ctxOffset); }

/* ---- in-loop filter primitives through the C table (common/loopfilter.cpp, encoder/sao.cpp:1762-1925) ---- */
void ref_pelFilterLumaStrong(int dir, pixel* src, intptr_t srcStep, intptr_t offset, int32_t tcP, int32_t tcQ) { T().pelFilterLumaStrong[dir](src, srcStep, offset, tcP, tcQ); }
void ref_pelFilterChroma(int dir, pixel* src, intptr_t srcStep, intptr_t offset, int32_t tc, int32_t maskP, int32_t maskQ) { T().pelFilterChroma[dir](src, srcStep, offset, tc, maskP, maskQ); }
void ref_saoSign(int8_t* dst, const pixel* a, const pixel* b, int endX) { T().sign(dst, a, b, endX); }
void ref_saoCuOrgE0(pixel* rec, int8_t* offsetEo, int width, int8_t* signLeft, intptr_t stride) { T().saoCuOrgE0(rec, offsetEo, width, signLeft, stride); }
void ref_saoCuOrgE1(pixel* rec, int8_t* upBuff1, int8_t* offsetEo, intptr_t stride, int width, int rows)
{
    if (rows == 2) T().saoCuOrgE1_2Rows(rec, upBuff1, offsetEo, stride, width);
    else T().saoCuOrgE1(rec, upBuff1, offsetEo, stride, width);
}
void ref_saoCuOrgE2(pixel* rec, int8_t* bufft, int8_t* buff1, int8_t* offsetEo, int width, intptr_t stride) { T().saoCuOrgE2[width > 16](rec, bufft, buff1, offsetEo, width, stride); }
void ref_saoCuOrgE3(pixel* rec, int8_t* upBuff1, int8_t* offsetEo, intptr_t stride, int startX, int endX) { T().saoCuOrgE3[(endX - startX) > 16](rec, upBuff1, offsetEo, stride, startX, endX); }
void ref_saoCuOrgB0(pixel* rec, const int8_t* offset, int w, int h, intptr_t stride) { T().saoCuOrgB0(rec, offset, w, h, stride); }
void ref_saoCuStatsBO(const int16_t* diff, const pixel* rec, intptr_t stride, int endX, int endY, int32_t* stats, int32_t* count) { T().saoCuStatsBO(diff, rec, stride, endX, endY, stats, count); }
void ref_saoCuStatsE0(const int16_t* diff, const pixel* rec, intptr_t stride, int endX, int endY, int32_t* stats, int32_t* count) { T().saoCuStatsE0(diff, rec, stride, endX, endY, stats, count); }
void ref_saoCuStatsE1(const int16_t* diff, const pixel* rec, intptr_t stride, int8_t* up, int endX, int endY, int32_t* stats, int32_t* count) { T().saoCuStatsE1(diff, rec, stride, up, endX, endY, stats, count); }
void ref_saoCuStatsE2(const int16_t* diff, const pixel* rec, intptr_t stride, int8_t* up, int8_t* upt, int endX, int endY, int32_t* stats, int32_t* count) { T().saoCuStatsE2(diff, rec, stride, up, upt, endX, endY, stats, count); }
void ref_saoCuStatsE3(const int16_t* diff, const pixel* rec, intptr_t stride, int8_t* up, int endX, int endY, int32_t* stats, int32_t* count) { T().saoCuStatsE3(diff, rec, stride, up, endX, endY, stats, count); }

/* ---- the real Deblock::edgeFilterLuma / edgeFilterChroma (common/deblock.cpp:317-513) on one 64x64 CTU whose inner edges are filtered:
 * a CUData that is its own picture (one CTU), block strengths / QPs / lossless flags given per 4x4 unit in RASTER order (16 x 16).
 * y / cb / cr point at the CTU's top-left sample inside caller-owned planes.  edge: in 4-sample units from the CTU's left / top (1..15). */
struct DeblockProbe : public Deblock
{
    static void luma(const CUData* cu, int dir, int edge, const uint8_t* bs) { edgeFilterLuma(cu, 0, 0, dir, edge, bs); }
    static void chroma(const CUData* cu, int dir, int edge, const uint8_t* bs) { edgeFilterChroma(cu, 0, 0, dir, edge, bs); }
};
void ref_deblock_ctu_edge(pixel* y, pixel* cb, pixel* cr, intptr_t stride, intptr_t strideC, int dir, int edge, const uint8_t* bsRaster,
                          const int8_t* qpRaster, const uint8_t* bypassRaster, int betaOffsetDiv2, int tcOffsetDiv2, int cbQpOffset, int crQpOffset,
                          int doLuma, int doChroma)
{
    T();
    x265_param param;
    memset(&param, 0, sizeof(param));
    param.maxCUSize = 64;
    alignas(SPS) unsigned char spsRaw[sizeof(SPS)];
    alignas(PPS) unsigned char ppsRaw[sizeof(PPS)];
    memset(spsRaw, 0, sizeof(spsRaw));
    memset(ppsRaw, 0, sizeof(ppsRaw));
    SPS& sps = *reinterpret_cast<SPS*>(spsRaw);
    PPS& pps = *reinterpret_cast<PPS*>(ppsRaw);
    sps.numPartInCUSize = 16;
    sps.numPartitions = 256;
    pps.deblockingFilterBetaOffsetDiv2 = betaOffsetDiv2;
    pps.deblockingFilterTcOffsetDiv2 = tcOffsetDiv2;
    pps.chromaQpOffset[0] = cbQpOffset;
    pps.chromaQpOffset[1] = crQpOffset;
    pps.bTransquantBypassEnabled = bypassRaster != NULL;
    Slice slice;
    slice.m_sps = &sps;
    slice.m_pps = &pps;
    /* offsets of the 256 units inside the CTU, z-order -> sample offset (PicYuv::createOffsets does the same from the sps) */
    intptr_t zero = 0, buY[256], buC[256];
    uint8_t bs[256], bypass[256];
    int8_t qp[256];
    for (int r = 0; r < 256; r++)
    {
        const uint32_t z = g_rasterToZscan[r];
        const int ux = r & 15, uy = r >> 4;
        buY[z] = (intptr_t)uy * 4 * stride + ux * 4;
        buC[z] = (intptr_t)uy * 2 * strideC + ux * 2;
        bs[z] = bsRaster[r];
        qp[z] = qpRaster[r];
        bypass[z] = bypassRaster ? bypassRaster[r] : 0;
    }
    PicYuv pic;
    pic.m_cuOffsetY = &zero; pic.m_cuOffsetC = &zero; pic.m_buOffsetY = buY; pic.m_buOffsetC = buC;
    pic.m_picOrg[0] = y; pic.m_picOrg[1] = cb; pic.m_picOrg[2] = cr;
    pic.m_stride = stride; pic.m_strideC = strideC;
    FrameData enc;
    enc.m_param = &param;
    enc.m_reconPic = &pic;
    CUData cu;
    enc.m_picCTU = &cu;
    cu.m_encData = &enc;
    cu.m_slice = &slice;
    cu.m_cuAddr = 0;
    cu.m_absIdxInCTU = 0;
    cu.m_chromaFormat = X265_CSP_I420;
    cu.m_hChromaShift = cu.m_vChromaShift = 1;
    cu.m_qp = qp;
    cu.m_tqBypass = bypass;
    cu.s_numPartInCUSize = 16;
    if (doLuma) DeblockProbe::luma(&cu, dir, edge, bs);
    if (doChroma) DeblockProbe::chroma(&cu, dir, edge, bs);
    pic.m_picOrg[0] = pic.m_picOrg[1] = pic.m_picOrg[2] = NULL;
    pic.m_cuOffsetY = pic.m_cuOffsetC = pic.m_buOffsetY = pic.m_buOffsetC = NULL;
    enc.m_picCTU = NULL;
}

/* ---- the real LookaheadTLD::weightsAnalyse (encoder/slicetype.cpp:860-960; weightCostLuma :807-840) on real Lowres objects: pic0 is the
 * reference (frame 0), pic1 the frame being weighted against it (frame 1); wp_ssd / wp_sum as AQ would have left them.  Outputs: the four
 * weighted lowres buffers (planes: 4 x planesize elements, only written when the function decides to weight) and weightedCostDelta.
 * Returns weightedRef[1].isWeighted, or -1. */
int ref_weights_analyse(pixel* pic0, pixel* pic1, intptr_t stride, int w, int h, int marginX, int marginY, uint64_t fencSsd, uint64_t fencSum,
                        uint64_t refSsd, uint64_t refSum, pixel* planes, int64_t planesCap, double* costDelta)
{
    T();
    x265_param* param = x265_param_alloc();
    x265_param_default(param);
    param->sourceWidth = w;
    param->sourceHeight = h;
    param->rc.aqMode = 0;
    param->rc.hevcAq = 0;
    param->bAQMotion = 0;
    param->bEnableHME = 0;
    param->lookaheadSlices = 0;
    PicYuv pics[2];
    Lowres lr[2];
    pixel* org[2] = { pic0, pic1 };
    int ret = -1;
    bool ok = true;
    for (int i = 0; i < 2; i++)
    {
        pics[i].m_picWidth = w;
        pics[i].m_picHeight = h;
        pics[i].m_lumaMarginX = marginX;
        pics[i].m_lumaMarginY = marginY;
        pics[i].m_stride = stride;
        pics[i].m_picOrg[0] = org[i];
        pics[i].m_param = param;
        memset((void*)&lr[i], 0, sizeof(Lowres));
        ok = ok && lr[i].create(param, &pics[i], param->rc.qgSize);
    }
    if (ok)
    {
        lr[0].init(&pics[0], 0);
        lr[1].init(&pics[1], 1);
        Lookahead la(param, NULL);
        la.create();
        LookaheadTLD& tld = la.m_tld[0];
        tld.lowresIntraEstimate(lr[1], param->rc.qgSize);
        lr[1].wp_ssd[0] = fencSsd; lr[1].wp_sum[0] = fencSum;
        lr[0].wp_ssd[0] = refSsd; lr[0].wp_sum[0] = refSum;
        lr[1].weightedRef[1].isWeighted = false;
        lr[1].weightedCostDelta[1] = -1.0;
        tld.weightsAnalyse(lr[1], lr[0]);
        ret = lr[1].weightedRef[1].isWeighted ? 1 : 0;
        *costDelta = lr[1].weightedCostDelta[1];
        const intptr_t planesize = lr[1].buffer[1] - lr[1].buffer[0];
        if (ret == 1 && 4 * planesize <= planesCap)
            for (int i = 0; i < 4; i++)
                memcpy(planes + i * planesize, tld.wbuffer[i], planesize * sizeof(pixel));
        la.destroy();
    }
    for (int i = 0; i < 2; i++)
    {
        lr[i].destroy();
        pics[i].m_picOrg[0] = NULL;
        pics[i].m_param = NULL;
    }
    x265_param_free(param);
    return ret;
}

/* ---- the real LookaheadTLD::calcAdaptiveQuantFrame (encoder/slicetype.cpp:444-700) on a Frame whose source picture is the caller's planes:
 * per-block AQ offsets (aqMode 0..3; 4 needs the edge pictures and is not driven), invQscaleFactor, and the frame statistics weightsAnalyse
 * reads (wp_sum / wp_ssd, all three planes).  Returns the number of AQ blocks, or -1. */
int ref_aq_frame(pixel* y, pixel* cb, pixel* cr, intptr_t stride, intptr_t strideC, int w, int h, int marginX, int marginY, int qgSize, int aqMode,
                 double aqStrength, int weightp, double* qpAqOffset, int32_t* invQscaleFactor, int32_t* invQscaleFactor8x8, uint64_t* wpStats)
{
    T();
    x265_param* param = x265_param_alloc();
    x265_param_default(param);
    param->sourceWidth = w;
    param->sourceHeight = h;
    param->internalCsp = X265_CSP_I420;
    param->rc.aqMode = aqMode;
    param->rc.aqStrength = aqStrength;
    param->rc.qgSize = qgSize;
    param->rc.hevcAq = 0;
    param->rc.cuTree = 0;
    param->bAQMotion = 0;
    param->bEnableHME = 0;
    param->bEnableWeightedPred = weightp;
    param->bEnableWeightedBiPred = 0;
    param->bHDR10Opt = 0;
    param->bDynamicRefine = 0;
    param->bEnableFades = 0;
    param->lookaheadSlices = 0;
    int ret = -1;
    {
        PicYuv pic;
        pic.m_picWidth = w; pic.m_picHeight = h;
        pic.m_lumaMarginX = marginX; pic.m_lumaMarginY = marginY;
        pic.m_chromaMarginX = marginX / 2; pic.m_chromaMarginY = marginY / 2;
        pic.m_stride = stride; pic.m_strideC = strideC;
        pic.m_picOrg[0] = y; pic.m_picOrg[1] = cb; pic.m_picOrg[2] = cr;
        pic.m_picCsp = X265_CSP_I420;
        pic.m_hChromaShift = pic.m_vChromaShift = 1;
        pic.m_param = param;
        Frame* frame = new Frame;
        frame->m_param = param;
        frame->m_fencPic = &pic;
        if (frame->m_lowres.create(param, &pic, qgSize))
        {
            Lookahead la(param, NULL);
            la.create();
            LookaheadTLD& tld = la.m_tld[0];
            tld.calcAdaptiveQuantFrame(frame, param);
            Lowres& lr = frame->m_lowres;
            const int blocks = qgSize == 8 ? lr.maxBlocksInRowFullRes * lr.maxBlocksInColFullRes : lr.maxBlocksInRow * lr.maxBlocksInCol;
            ret = blocks;
            if (lr.qpAqOffset)
            {
                memcpy(qpAqOffset, lr.qpAqOffset, blocks * sizeof(double));
                memcpy(invQscaleFactor, lr.invQscaleFactor, blocks * sizeof(int));
                if (qgSize == 8 && lr.invQscaleFactor8x8)
                    memcpy(invQscaleFactor8x8, lr.invQscaleFactor8x8, lr.maxBlocksInRow * lr.maxBlocksInCol * sizeof(int));
            }
            for (int i = 0; i < 3; i++) { wpStats[i] = lr.wp_sum[i]; wpStats[3 + i] = lr.wp_ssd[i]; }
            la.destroy();
        }
        frame->m_lowres.destroy();
        frame->m_fencPic = NULL;
        delete frame;
        pic.m_picOrg[0] = pic.m_picOrg[1] = pic.m_picOrg[2] = NULL;
        pic.m_param = NULL;
    }
    x265_param_free(param);
    return ret;
}

/* ---- the real Lookahead::estimateCUPropagate (+ cuTreeFinish when vbv is set; encoder/slicetype.cpp:2641-2750, :2889-2937) on three real
 * Lowres objects (frames 0, 1, 2 = p0, b, p1; isP: p1 = b = 1).  Frame b's per-8x8 inputs are given; the reference frames' propagateCost
 * arrays are read and updated.  pic: any padded picture of the right size (Lowres::create wants one).  Returns the 8x8 block count. */
namespace {
struct LookaheadProbe : public Lookahead
{
    LookaheadProbe(x265_param* p) : Lookahead(p, NULL) {}
    void propagate(Lowres** f, double dur, int p0, int p1, int b, int referenced) { estimateCUPropagate(f, dur, p0, p1, b, referenced); }
    void finish(Lowres* f, double dur, int ref0Distance) { cuTreeFinish(f, dur, ref0Distance); }
};
}
int ref_cutree_propagate(pixel* pic, intptr_t stride, int w, int h, int marginX, int marginY, int qgSize, int fpsNum, int fpsDenom, double averageDuration,
                         int isP, int referenced, int weightedBiPred, const uint16_t* propagateIn, const int32_t* intraCost, const uint16_t* lowresCosts,
                         const int32_t* invQscale, const int32_t* mvs0, const int32_t* mvs1, uint16_t* refCosts0, uint16_t* refCosts1,
                         int doFinish, double qCompress, const double* qpAqOffset, double* qpCuTreeOffset, int aqBlocks)
{
    T();
    x265_param* param = x265_param_alloc();
    x265_param_default(param);
    param->sourceWidth = w;
    param->sourceHeight = h;
    param->rc.aqMode = 2;
    param->rc.qgSize = qgSize;
    param->rc.hevcAq = 0;
    param->rc.cuTree = 1;
    param->rc.qCompress = qCompress;
    param->bAQMotion = 0;
    param->bEnableHME = 0;
    param->bEnableWeightedBiPred = weightedBiPred;
    param->fpsNum = fpsNum;
    param->fpsDenom = fpsDenom;
    param->lookaheadSlices = 0;
    PicYuv pics[3];
    Lowres lr[3];
    int ret = -1;
    bool ok = true;
    for (int i = 0; i < 3; i++)
    {
        pics[i].m_picWidth = w; pics[i].m_picHeight = h;
        pics[i].m_lumaMarginX = marginX; pics[i].m_lumaMarginY = marginY;
        pics[i].m_stride = stride;
        pics[i].m_picOrg[0] = pic;
        pics[i].m_param = param;
        memset((void*)&lr[i], 0, sizeof(Lowres));
        ok = ok && lr[i].create(param, &pics[i], qgSize);
    }
    if (ok)
    {
        LookaheadProbe la(param);
        la.create();
        const int ncu = la.m_8x8Width * la.m_8x8Height;
        const int p0 = 0, b = 1, p1 = isP ? 1 : 2;
        Lowres* frames[3] = { &lr[0], &lr[1], &lr[2] };
        Lowres& fb = lr[b];
        memcpy(fb.propagateCost, propagateIn, ncu * sizeof(uint16_t));
        memcpy(fb.intraCost, intraCost, ncu * sizeof(int32_t));
        memcpy(fb.lowresCosts[b - p0][p1 - b], lowresCosts, ncu * sizeof(uint16_t));
        if (qgSize == 8) memcpy(fb.invQscaleFactor8x8, invQscale, ncu * sizeof(int));
        else memcpy(fb.invQscaleFactor, invQscale, ncu * sizeof(int));
        for (int i = 0; i < ncu; i++)
        {
            fb.lowresMvs[0][b - p0][i] = MV(mvs0[2 * i], mvs0[2 * i + 1]);
            if (!isP) fb.lowresMvs[1][p1 - b][i] = MV(mvs1[2 * i], mvs1[2 * i + 1]);
        }
        memcpy(lr[p0].propagateCost, refCosts0, ncu * sizeof(uint16_t));
        if (!isP) memcpy(lr[p1].propagateCost, refCosts1, ncu * sizeof(uint16_t));
        la.propagate(frames, averageDuration, p0, p1, b, referenced);
        memcpy(refCosts0, lr[p0].propagateCost, ncu * sizeof(uint16_t));
        if (!isP) memcpy(refCosts1, lr[p1].propagateCost, ncu * sizeof(uint16_t));
        if (doFinish)
        {
            memcpy(fb.qpAqOffset, qpAqOffset, aqBlocks * sizeof(double));
            memcpy(fb.qpCuTreeOffset, qpAqOffset, aqBlocks * sizeof(double));
            fb.weightedCostDelta[0] = 0;
            la.finish(&fb, averageDuration, isP ? 1 : 0);
            memcpy(qpCuTreeOffset, fb.qpCuTreeOffset, aqBlocks * sizeof(double));
        }
        ret = ncu;
        la.destroy();
    }
    for (int i = 0; i < 3; i++)
    {
        lr[i].destroy();
        pics[i].m_picOrg[0] = NULL;
        pics[i].m_param = NULL;
    }
    x265_param_free(param);
    return ret;
}


/* ---- the real Quant::transformNxN / ::invtransformNxN (common/quant.cpp:397-470, :543-603) on an inter unit: a Quant object with flat
 * quantiser matrices (ScalingList::setupQuantMatrices with m_bEnabled = false), RDOQ and noise reduction off, a one-partition inter CUData
 * whose slice is I (sliceI: rounding offset 171) or P (85) and whose PPS enables sign hiding or not.  qp: the QpParam value of the plane
 * (Quant::m_qpParam[ttype].setQpParam argument, i.e. qp + QP_BD_OFFSET).  resiDct (may be NULL) receives Quant::m_resiDctCoeff. */
namespace {
struct QuantProbe : public Quant                 /* the fields are protected */
{
    void plain() { m_rdoqLevel = 0; m_nr = NULL; }
    void setQp(int ttype, int qp) { m_qpParam[ttype].setQpParam(qp); }
    const int16_t* resiDct() const { return m_resiDctCoeff; }
};
struct QuantRig
{
    ScalingList sl;
    Entropy entropy;
    QuantProbe q;
    bool ok;
    QuantRig()
    {
        T();
        ok = sl.init();
        sl.m_bEnabled = false;
        sl.m_bDataPresent = false;
        sl.setupQuantMatrices(X265_CSP_I420);
        ok = ok && q.init(0.0, sl, entropy);
        q.plain();
    }
};
QuantRig& rig() { static QuantRig r; return r; }
struct InterCU
{
    CUData cu;
    Slice slice;
    unsigned char ppsRaw[sizeof(PPS)];
    uint8_t tqBypass[256], predMode[256];
    InterCU(int sliceI, int signHide)
    {
        memset(ppsRaw, 0, sizeof(ppsRaw));
        PPS& pps = *reinterpret_cast<PPS*>(ppsRaw);
        pps.bSignHideEnabled = !!signHide;
        slice.m_pps = &pps;
        slice.m_sliceType = sliceI ? I_SLICE : P_SLICE;
        memset(tqBypass, 0, sizeof(tqBypass));
        memset(predMode, MODE_INTER, sizeof(predMode));
        cu.m_slice = &slice;
        cu.m_tqBypass = tqBypass;
        cu.m_predMode = predMode;
        cu.m_chromaFormat = X265_CSP_I420;
        cu.m_hChromaShift = cu.m_vChromaShift = 1;
    }
};
}
uint32_t ref_transform_nxn(const int16_t* residual, uint32_t resiStride, int16_t* coeff, int16_t* resiDct, int log2TrSize, int ttype, int qp, int sliceI,
                           int signHide)
{
    QuantRig& r = rig();
    if (!r.ok) return 0xffffffffu;
    InterCU c(sliceI, signHide);
    r.q.setQp(ttype, qp);
    pixel fenc[4] = { 0, 0, 0, 0 };                 /* only read for psy-rdoq, which is off */
    const uint32_t numSig = r.q.transformNxN(c.cu, fenc, 2, residual, resiStride, coeff, (uint32_t)log2TrSize, (TextType)ttype, 0, false);
    if (resiDct)
        memcpy(resiDct, r.q.resiDct(), sizeof(int16_t) << (2 * log2TrSize));
    return numSig;
}
void ref_invtransform_nxn(int16_t* residual, uint32_t resiStride, const int16_t* coeff, int log2TrSize, int ttype, int qp, uint32_t numSig)
{
    QuantRig& r = rig();
    if (!r.ok) return;
    InterCU c(0, 0);
    r.q.setQp(ttype, qp);
    r.q.invtransformNxN(c.cu, residual, resiStride, coeff, (uint32_t)log2TrSize, (TextType)ttype, false, false, numSig);
}
/* the QpParam fields and flat matrix entries the job header carries, as the reference computes them */
void ref_qp_param(int qp, int32_t* out) /* rem, per, quantScale, dequantScale */
{
    QuantRig& r = rig();
    QpParam p;
    p.setQpParam(qp);
    out[0] = p.rem; out[1] = p.per;
    out[2] = r.sl.m_quantCoef[3][3][p.rem][0];
    out[3] = ScalingList::s_invQuantScales[p.rem];
}
} // extern "C"
