// framepass.hip — host-side pipeline: one P-frame of the hot path as a fixed sequence of batched launches on one
// stream (include/x265hip.h "frame pass").  Pure host C++ over the C ABI of this same library; also hosts
// BitCost::setQP (bitcost.cpp:32-60) because the MVD cost table is float host math in the reference too.
#include "common.h"
#include "internal.h"
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

struct x265hip_framepass
{
    int width, height, depth, qp, merange, method, subme;
    int nLevel[4];                 // PUs per CU size 64, 32, 16, 8
    int32_t* puXY[4];              // device
    int32_t* parent[4];            // device: index into the level above, -1 when that CU is not inside the picture
    int32_t *qmvp[4], *mvmin[4], *mvmax[4], *mv[4], *cost[4], *sa8d[4];
    int32_t* cuOff[4];             // y*stride+x offsets for sa8d launches are stride dependent -> rebuilt per run when strides change
    int64_t cuOffStrideS, cuOffStrideP;
    int32_t* cuOffP[4];
    int nTu[2];                    // TU size 32, 8
    int32_t* tuXY[2];              // device (x, y)
    int32_t *tuOffF[2], *tuOffP[2], *tuOffR[2];
    int64_t tuStrideF, tuStrideP, tuStrideR;
    int16_t* level[2];
    uint32_t* numSig[2];
    uint64_t* dist[2];
    uint16_t* mvcost;              // device, 4*32768+1 entries
    int32_t* quantCoeff[2];        // flat scaling: quantScales[qp % 6] (scalinglist.cpp:129)
    std::vector<int32_t> hCuXY[4], hTuXY[2];
    // chroma (4:2:0) TU classes: 16x16 under every 32x32 luma TU, 4x4 under every 8x8 luma TU; one set of outputs per plane
    int32_t *ctuOffF[2], *ctuOffP[2], *ctuOffR[2];       // [class][2n]: Cb TUs then Cr TUs, Cr addressed from the Cb base pointer
    int64_t cStrideF, cStrideP, cStrideR, cDeltaF, cDeltaP, cDeltaR;
    int16_t* clevel[2][2];         // [plane Cb/Cr][class]
    uint32_t* cnumSig[2][2];
    uint64_t* cdist[2][2];
    int32_t* cquantCoeff[2];
    std::vector<int32_t> hCtuXY[2];
    // B pass: list-1 search state and the second reference's planes, allocated on the first B run
    int32_t *qmvp1[4], *mvmin1[4], *mvmax1[4], *mv1[4], *cost1[4];
    void* planes1;
    void* planes;                  // 16 sub-pel planes of the current reference (device), sized on first run
    int64_t planeElems, planeStride;
    int planeMarginX, planeMarginY;
    bool profile;                  // record a HIP event at every stage boundary of run()
    hipEvent_t ev[12];
    // hipGraph cache of x265hip_framepass_run_yuv: the pass is a fixed sequence of 15 launches whose only variables are the picture
    // pointers, so each distinct pointer set is captured once (after a plain run has done the allocations) and replayed as ONE launch
    struct GraphKey { const void* p[12]; int64_t s[8]; int mx, my; };
    struct GraphEntry { GraphKey key; hipGraphExec_t exec; };
    std::vector<GraphEntry> graphs;
    GraphKey lastPlain;            // strides / margins of the last plain (uncaptured) run
    int plainRuns;
};

namespace xh {

static const int kCuSize[4] = { 64, 32, 16, 8 };
static const int kTuSize[2] = { 32, 8 };
static const int kCTuSize[2] = { 16, 4 };
// g_chromaScale (reference: common/constants.cpp:346-350), 4:2:0 chroma QP mapping
static const uint8_t kChromaScale[70] = {
    0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31, 32, 33, 33, 34, 34, 35,
    35, 36, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51, 51, 51, 51, 51, 51, 51, 51, 51, 51, 51, 51, 51 };
static const int kMvHalf = 2 * 32768;

template <typename T>
static int dev_upload(T** d, const std::vector<T>& h)
{
    int e = check_hip(hipMalloc((void**)d, (h.size() ? h.size() : 1) * sizeof(T)), "hipMalloc(framepass)");
    if (e) return e;
    if (h.size())
        e = check_hip(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy(framepass)");
    return e;
}
template <typename T>
static int dev_alloc(T** d, size_t n) { return check_hip(hipMalloc((void**)d, (n ? n : 1) * sizeof(T)), "hipMalloc(framepass)"); }

static int upload_offsets(int32_t** d, const std::vector<int32_t>& xy, int64_t stride)
{
    std::vector<int32_t> off(xy.size() / 2);
    for (size_t i = 0; i < off.size(); i++)
        off[i] = (int32_t)((int64_t)xy[2 * i + 1] * stride + xy[2 * i]);
    if (*d) (void)device_free(*d);
    *d = nullptr;
    return dev_upload(d, off);
}

} // namespace xh

using namespace xh;

#define FP_TRY(x) do { int e_ = (x); if (e_) return e_; } while (0)

extern "C" {

int x265hip_mvcost_table(int qp, int depth, uint16_t* table, int half)
{
    if (qp < 0 || qp > 69 || !valid_depth(depth) || !table || half < 1)
        return set_error(X265HIP_EINVAL, "mvcost_table: qp %d depth %d half %d", qp, depth, half);
    // x265_lambda_tab[qp] = 2^(qp/6 - 2) * 2^(depth - 8)... stored as 4-decimal literals (constants.cpp:34-152)
    const double e = (double)qp / 6.0 - 2.0 + (double)(depth - 8);
    const double lambda = std::floor(std::pow(2.0, e) * 10000.0 + 0.5) / 10000.0;
    // CalculateLogs (bitcost.cpp:108-125): float table, log() evaluated in double on a float argument
    const float log2_2 = (float)(2.0f / std::log((double)2.0f));
    for (int i = 0; i <= half; i++)
    {
        const float bits = i ? (float)(std::log((double)(float)(i + 1)) * log2_2 + 1.718f) : 0.718f;
        const double v = bits * lambda + 0.5f;
        const double cap = (double)((1 << 15) - 1);
        const uint16_t c = (uint16_t)(v < cap ? v : cap);
        table[half + i] = table[half - i] = c;
    }
    return X265HIP_OK;
}

int x265hip_framepass_destroy(x265hip_framepass* fp);

int x265hip_framepass_create(int width, int height, int depth, int qp, int merange, int searchMethod, int subme,
                             x265hip_framepass** out)
{
    XH_CHECK_DEV();
    if (!out || width < 8 || height < 8 || (width & 7) || (height & 7) || !valid_depth(depth) || qp < 0 || qp > 51 ||
        merange < 1 || merange > 256 || subme < 0 || subme > 7 || (searchMethod != 0 && searchMethod != 1 && searchMethod != 2 && searchMethod != 3 && searchMethod != 5))
        return set_error(X265HIP_EINVAL, "framepass_create: %dx%d depth %d qp %d merange %d me %d subme %d", width, height, depth, qp,
                         merange, searchMethod, subme);
    x265hip_framepass* fp = new x265hip_framepass();
    // an error below releases everything allocated so far
#define FP_TRYC(x) do { int e_ = (x); if (e_) { x265hip_framepass_destroy(fp); return e_; } } while (0)
    fp->width = width; fp->height = height; fp->depth = depth; fp->qp = qp;
    fp->merange = merange; fp->method = searchMethod; fp->subme = subme;
    fp->cuOffStrideS = fp->cuOffStrideP = fp->tuStrideF = fp->tuStrideP = fp->tuStrideR = -1;
    fp->profile = false;
    fp->planes = nullptr;
    fp->planeElems = fp->planeStride = 0;
    fp->planeMarginX = fp->planeMarginY = 0;
    for (int i = 0; i < 12; i++)
        FP_TRYC(check_hip(hipEventCreate(&fp->ev[i]), "hipEventCreate(framepass)"));
    for (int l = 0; l < 4; l++)
    {
        const int sz = kCuSize[l];
        std::vector<int32_t> par;
        for (int y = 0; y + sz <= height; y += sz)
            for (int x = 0; x + sz <= width; x += sz)
            {
                fp->hCuXY[l].push_back(x);
                fp->hCuXY[l].push_back(y);
                int p = -1;
                if (l > 0)
                {
                    const int ps = kCuSize[l - 1], pw = width / ps, ph = height / ps;     // parents fully inside
                    const int px = x / ps, py = y / ps;
                    if (px < pw && py < ph)
                        p = py * pw + px;
                }
                par.push_back(p);
            }
        const int n = (int)par.size();
        fp->nLevel[l] = n;
        FP_TRYC(dev_upload(&fp->puXY[l], fp->hCuXY[l]));
        FP_TRYC(dev_upload(&fp->parent[l], par));
        FP_TRYC(dev_alloc(&fp->qmvp[l], 2 * (size_t)n));
        FP_TRYC(dev_alloc(&fp->mvmin[l], 2 * (size_t)n));
        FP_TRYC(dev_alloc(&fp->mvmax[l], 2 * (size_t)n));
        FP_TRYC(dev_alloc(&fp->mv[l], 2 * (size_t)n));
        FP_TRYC(dev_alloc(&fp->cost[l], (size_t)n));
        FP_TRYC(dev_alloc(&fp->sa8d[l], (size_t)n));
        fp->cuOff[l] = fp->cuOffP[l] = nullptr;
    }
    // TUs: 32x32 over the 32-aligned area, 8x8 over the remaining right / bottom strips
    const int w32 = width & ~31, h32 = height & ~31;
    for (int y = 0; y < h32; y += 32)
        for (int x = 0; x < w32; x += 32) { fp->hTuXY[0].push_back(x); fp->hTuXY[0].push_back(y); }
    for (int y = 0; y < height; y += 8)
        for (int x = 0; x < width; x += 8)
            if (x >= w32 || y >= h32) { fp->hTuXY[1].push_back(x); fp->hTuXY[1].push_back(y); }
    static const int quantScales[6] = { 26214, 23302, 20560, 18396, 16384, 14564 };      // scalinglist.cpp:129 s_quantScales
    for (int t = 0; t < 2; t++)
    {
        const int n = (int)fp->hTuXY[t].size() / 2, nc = kTuSize[t] * kTuSize[t];
        fp->nTu[t] = n;
        FP_TRYC(dev_upload(&fp->tuXY[t], fp->hTuXY[t]));
        FP_TRYC(dev_alloc(&fp->level[t], (size_t)n * nc));
        FP_TRYC(dev_alloc(&fp->numSig[t], (size_t)n));
        FP_TRYC(dev_alloc(&fp->dist[t], (size_t)n));
        std::vector<int32_t> qc(nc, quantScales[(qp + 6 * (depth - 8)) % 6]);       // QpParam: qp + QP_BD_OFFSET (quant.cpp:224)
        FP_TRYC(dev_upload(&fp->quantCoeff[t], qc));
        fp->tuOffF[t] = fp->tuOffP[t] = fp->tuOffR[t] = nullptr;
    }
    // chroma TUs mirror the luma TU lists at half resolution; QpParam of chroma: Quant::setChromaQP (quant.cpp:233-243)
    {
        const int bd = 6 * (depth - 8);
        int qpc = qp < -bd ? -bd : (qp > 57 ? 57 : qp);
        if (qpc >= 30) qpc = kChromaScale[qpc];
        qpc += bd;
        for (int t = 0; t < 2; t++)
        {
            for (size_t i = 0; i < fp->hTuXY[t].size(); i++)
                fp->hCtuXY[t].push_back(fp->hTuXY[t][i] >> 1);
            const int n = fp->nTu[t], nc = kCTuSize[t] * kCTuSize[t];
            std::vector<int32_t> qc(nc, quantScales[qpc % 6]);
            FP_TRYC(dev_upload(&fp->cquantCoeff[t], qc));
            // Cb and Cr outputs are one allocation each ([2n]: Cb then Cr) so both planes go through ONE chain launch
            FP_TRYC(dev_alloc(&fp->clevel[0][t], (size_t)2 * n * nc));
            FP_TRYC(dev_alloc(&fp->cnumSig[0][t], (size_t)2 * n));
            FP_TRYC(dev_alloc(&fp->cdist[0][t], (size_t)2 * n));
            fp->clevel[1][t] = fp->clevel[0][t] + (size_t)n * nc;
            fp->cnumSig[1][t] = fp->cnumSig[0][t] + n;
            fp->cdist[1][t] = fp->cdist[0][t] + n;
            fp->ctuOffF[t] = fp->ctuOffP[t] = fp->ctuOffR[t] = nullptr;
        }
        fp->cStrideF = fp->cStrideP = fp->cStrideR = -1;
        fp->cDeltaF = fp->cDeltaP = fp->cDeltaR = 0;
    }
    std::vector<uint16_t> tab(2 * kMvHalf + 1);
    FP_TRYC(x265hip_mvcost_table(qp, depth, tab.data(), kMvHalf));
    FP_TRYC(dev_upload(&fp->mvcost, tab));
    *out = fp;
#undef FP_TRYC
    return X265HIP_OK;
}

int x265hip_framepass_destroy(x265hip_framepass* fp)
{
    if (!fp) return X265HIP_OK;
    for (int l = 0; l < 4; l++)
    {
        void* ptrs[] = { fp->puXY[l], fp->parent[l], fp->qmvp[l], fp->mvmin[l], fp->mvmax[l], fp->mv[l], fp->cost[l], fp->sa8d[l], fp->cuOff[l], fp->cuOffP[l] };
        for (void* p : ptrs) if (p) (void)device_free(p);
    }
    for (int t = 0; t < 2; t++)
    {
        void* ptrs[] = { fp->tuXY[t], fp->tuOffF[t], fp->tuOffP[t], fp->tuOffR[t], fp->level[t], fp->numSig[t], fp->dist[t], fp->quantCoeff[t] };
        for (void* p : ptrs) if (p) (void)device_free(p);
    }
    for (int t = 0; t < 2; t++)
    {
        void* ptrs[] = { fp->ctuOffF[t], fp->ctuOffP[t], fp->ctuOffR[t], fp->cquantCoeff[t], fp->clevel[0][t], fp->cnumSig[0][t], fp->cdist[0][t] };
        for (void* p : ptrs) if (p) (void)device_free(p);
    }
    if (fp->mvcost) (void)device_free(fp->mvcost);
    for (auto& g : fp->graphs) (void)hipGraphExecDestroy(g.exec);
    fp->graphs.clear();
    if (fp->planes) (void)device_free(fp->planes);
    if (fp->planes1) (void)device_free(fp->planes1);
    for (int l = 0; l < 4; l++)
    {
        void* ptrs[] = { fp->qmvp1[l], fp->mvmin1[l], fp->mvmax1[l], fp->mv1[l], fp->cost1[l] };
        for (void* p : ptrs) if (p) (void)device_free(p);
    }
    for (int i = 0; i < 12; i++) (void)hipEventDestroy(fp->ev[i]);
    delete fp;
    return X265HIP_OK;
}

} // extern "C"

struct ChromaArgs { const void *srcCb, *srcCr, *refCb, *refCr; void *predCb, *predCr, *recCb, *recCr; int64_t sS, sR, sP, sRec; };
struct BArgs { const void *ref1, *ref1Cb, *ref1Cr; };       // second reference of a B pass (same strides and margins as the first)

// cached graphs reference the offset tables / plane buffers of the geometry they were captured with
static void drop_graphs(x265hip_framepass* fp)
{
    for (auto& g : fp->graphs) (void)hipGraphExecDestroy(g.exec);
    fp->graphs.clear();
    fp->plainRuns = 0;
}

static int framepass_run_impl(x265hip_framepass* fp, const void* src, int64_t strideS, const void* ref, int64_t strideR,
                              void* pred, int64_t strideP, void* recon, int64_t strideRec, int marginX, int marginY, const ChromaArgs* ca,
                              void* stream, const BArgs* ba = nullptr)
{
    XH_CHECK_DEV();
    if (!fp || !src || !ref || !pred || !recon)
        return set_error(X265HIP_EINVAL, "framepass_run: null argument");
    const int depth = fp->depth;
    // offset tables depend on the caller's strides: (re)build them when the strides change (first run, normally once)
    if (fp->cuOffStrideS != strideS || fp->cuOffStrideP != strideP)
    {
        drop_graphs(fp);
        FP_TRY(check_hip(hipStreamSynchronize(as_stream(stream)), "framepass sync"));
        for (int l = 0; l < 4; l++)
        {
            FP_TRY(upload_offsets(&fp->cuOff[l], fp->hCuXY[l], strideS));
            FP_TRY(upload_offsets(&fp->cuOffP[l], fp->hCuXY[l], strideP));
        }
        fp->cuOffStrideS = strideS; fp->cuOffStrideP = strideP;
    }
    if (fp->tuStrideF != strideS || fp->tuStrideP != strideP || fp->tuStrideR != strideRec)
    {
        drop_graphs(fp);
        FP_TRY(check_hip(hipStreamSynchronize(as_stream(stream)), "framepass sync"));
        for (int t = 0; t < 2; t++)
        {
            FP_TRY(upload_offsets(&fp->tuOffF[t], fp->hTuXY[t], strideS));
            FP_TRY(upload_offsets(&fp->tuOffP[t], fp->hTuXY[t], strideP));
            FP_TRY(upload_offsets(&fp->tuOffR[t], fp->hTuXY[t], strideRec));
        }
        fp->tuStrideF = strideS; fp->tuStrideP = strideP; fp->tuStrideR = strideRec;
    }
    // sub-pel planes of this reference (16 x padded picture, e.g. 43 MB at 1080p 8-bit): allocated once per geometry
    if (!fp->planes || fp->planeStride != strideR || fp->planeMarginX != marginX || fp->planeMarginY != marginY)
    {
        drop_graphs(fp);
        FP_TRY(check_hip(hipStreamSynchronize(as_stream(stream)), "framepass sync"));
        if (fp->planes) (void)device_free(fp->planes);
        fp->planes = nullptr;
        fp->planeStride = strideR; fp->planeMarginX = marginX; fp->planeMarginY = marginY;
        fp->planeElems = strideR * (int64_t)(fp->height + 2 * marginY);
        const size_t B = depth == 8 ? 1 : 2;
        FP_TRY(check_hip(hipMalloc(&fp->planes, (size_t)fp->planeElems * 16 * B), "hipMalloc(subpel planes)"));
        FP_TRY(check_hip(hipMemsetAsync(fp->planes, 0, (size_t)fp->planeElems * 16 * B, as_stream(stream)), "hipMemset(subpel planes)"));
        if (fp->planes1) (void)device_free(fp->planes1);
        fp->planes1 = nullptr;
    }
    if (ba && !fp->planes1)
    {
        FP_TRY(check_hip(hipStreamSynchronize(as_stream(stream)), "framepass sync"));
        const size_t B = depth == 8 ? 1 : 2;
        FP_TRY(check_hip(hipMalloc(&fp->planes1, (size_t)fp->planeElems * 16 * B), "hipMalloc(subpel planes, list 1)"));
        FP_TRY(check_hip(hipMemsetAsync(fp->planes1, 0, (size_t)fp->planeElems * 16 * B, as_stream(stream)), "hipMemset(subpel planes)"));
        for (int l = 0; l < 4; l++)
            if (!fp->mv1[l])
            {
                const size_t n = (size_t)fp->nLevel[l];
                FP_TRY(dev_alloc(&fp->qmvp1[l], 2 * n)); FP_TRY(dev_alloc(&fp->mvmin1[l], 2 * n)); FP_TRY(dev_alloc(&fp->mvmax1[l], 2 * n));
                FP_TRY(dev_alloc(&fp->mv1[l], 2 * n)); FP_TRY(dev_alloc(&fp->cost1[l], n));
            }
    }
    const size_t Bp = depth == 8 ? 1 : 2;
    void* planesOrigin = (char*)fp->planes + ((int64_t)marginY * strideR + marginX) * Bp;
#define FP_MARK(i) do { if (fp->profile) FP_TRY(check_hip(hipEventRecord(fp->ev[i], as_stream(stream)), "hipEventRecord")); } while (0)
    // 0. the 16 quarter-pel planes of this reference
    FP_MARK(0);
    FP_TRY(x265hip_build_subpel_planes(depth, ref, strideR, fp->width, fp->height, marginX, marginY, planesOrigin, fp->planeElems, stream));
    void* planes1Origin = ba ? (char*)fp->planes1 + ((int64_t)marginY * strideR + marginX) * Bp : nullptr;
    if (ba)
        FP_TRY(x265hip_build_subpel_planes(depth, ba->ref1, strideR, fp->width, fp->height, marginX, marginY, planes1Origin, fp->planeElems, stream));
    // 1. top-down motion search
    int predDone = 0;
    for (int l = 0; l < 4; l++)
    {
        const int n = fp->nLevel[l], sz = kCuSize[l];
        FP_MARK(1 + l);
        if (!n) continue;
        // per list (one for a P pass, two for a B pass): setSearchRange + motionEstimate, children around their own list's parent vector
        for (int list = 0; list < (ba ? 2 : 1); list++)
        {
            int32_t** MV = list ? fp->mv1 : fp->mv; int32_t** COST = list ? fp->cost1 : fp->cost;
            int32_t** QMVP = list ? fp->qmvp1 : fp->qmvp; int32_t** MVMIN = list ? fp->mvmin1 : fp->mvmin; int32_t** MVMAX = list ? fp->mvmax1 : fp->mvmax;
            const void* REFY = list ? ba->ref1 : ref;
            const void* REFCB = list ? ba->ref1Cb : (ca ? ca->refCb : nullptr);
            const void* REFCR = list ? ba->ref1Cr : (ca ? ca->refCr : nullptr);
            void* PLANES = list ? planes1Origin : planesOrigin;
            auto search = [&]() -> int {
                // setSearchRange + motionEstimate in one launch (searchrange.h); qmvp / mvmin / mvmax arrays are still produced
                DeriveRange dr{};
                dr.enable = 1;
                dr.mvSrc = l ? MV[l - 1] : nullptr;
                dr.srcIdx = fp->parent[l];
                dr.picW = fp->width; dr.picH = fp->height; dr.maxCUSize = 64;
                dr.refLagPixels = fp->height;                     // -F1: m_refLagPixels = sourceHeight (search.cpp:92)
                dr.qmvpO = QMVP[l]; dr.mvminO = MVMIN[l]; dr.mvmaxO = MVMAX[l];
                if (l == 3 && !ba)
                {
                    // P pass: the 8x8 level writes the luma prediction itself (row-team kernel epilogue); predDone stays 0 on the other kernels
                    dr.predOut = pred; dr.predStride = strideP; dr.predDone = &predDone;
                }
                if (ca && fp->subme > 2)
                {
                    // 4:2:0 picture at subme > 2: every sub-pel comparison carries the chroma SATD term (motion.cpp:212, :1601).  8 / 16 / 32 PUs:
                    // the plane-based row-team kernel with an in-register 4-tap chroma path; 64x64: the generic kernel, ranges from their own launch.
                    const ChromaPlanes cpl{ ca->srcCb, ca->srcCr, ca->sS, REFCB, REFCR, ca->sR, 1 };
                    int rc = X265HIP_OK;
                    static const bool genericChroma = getenv("X265HIP_CHROMA_ME_GENERIC") != nullptr;
                    if (!genericChroma && motion_estimate_fused_chroma(depth, sz, src, strideS, strideR, PLANES, fp->planeElems, cpl, fp->puXY[l], dr, fp->merange,
                                                                       fp->method, fp->subme, fp->mvcost + kMvHalf, n, MV[l], COST[l], as_stream(stream), &rc))
                    {
                        FP_TRY(rc);
                        return X265HIP_OK;
                    }
                    FP_TRY(x265hip_set_search_range_batch(fp->width, fp->height, 64, fp->merange, fp->height, fp->puXY[l], dr.mvSrc, dr.srcIdx, n,
                                                          QMVP[l], MVMIN[l], MVMAX[l], stream));
                    FP_TRY(x265hip_motion_estimate_chroma_batch(depth, sz, sz, src, strideS, ca->srcCb, ca->srcCr, ca->sS, REFY, strideR, REFCB, REFCR,
                                                                ca->sR, fp->puXY[l], MVMIN[l], MVMAX[l], QMVP[l], 0, nullptr, fp->merange,
                                                                fp->method, fp->subme, fp->mvcost + kMvHalf, kMvHalf, n, MV[l], COST[l], stream));
                    return X265HIP_OK;
                }
                FP_TRY(motion_estimate_fused(depth, sz, src, strideS, REFY, strideR, PLANES, fp->planeElems, fp->puXY[l], dr, fp->merange,
                                             fp->method, fp->subme, fp->mvcost + kMvHalf, n, MV[l], COST[l], as_stream(stream)));
    
                return X265HIP_OK;
            };
            FP_TRY(search());
        }
    }
    FP_MARK(5);
    // 2. prediction from the 8x8 vectors
    if (ba)
    {
        // B pass: bi-predictive motion compensation from the two lists' 8x8 vectors, luma and chroma in one launch (bipred.hip)
        const x265hip_yuv r0{ (void*)ref, (void*)ca->refCb, (void*)ca->refCr, strideR, ca->sR }, r1{ (void*)ba->ref1, (void*)ba->ref1Cb, (void*)ba->ref1Cr, strideR, ca->sR };
        const x265hip_yuv pd{ pred, ca->predCb, ca->predCr, strideP, ca->sP };
        FP_TRY(x265hip_pred_inter_bi_batch(depth, 8, 8, &r0, &r1, &pd, fp->puXY[3], fp->mv[3], fp->mv1[3], fp->nLevel[3], stream));
    }
    else if (!predDone)
        FP_TRY(pred_from_planes(depth, 8, planesOrigin, fp->planeElems, strideR, pred, strideP, fp->puXY[3], fp->mv[3], fp->nLevel[3], as_stream(stream)));
    // 3. chroma prediction (4:2:0, when the caller passed Cb / Cr planes): predInterChromaPixel from the 8x8 vectors (the B pass has
    //    already predicted chroma in its bi-predictive launch)
    if (ca)
    {
        const int64_t B = depth == 8 ? 1 : 2;
        const int64_t dF = ((const char*)ca->srcCr - (const char*)ca->srcCb) / B, dP = ((const char*)ca->predCr - (const char*)ca->predCb) / B;
        const int64_t dR = ((const char*)ca->recCr - (const char*)ca->recCb) / B;
        const int64_t lim = 0x7fffffffLL - (int64_t)(fp->height / 2 + 8) * (ca->sS > ca->sRec ? ca->sS : ca->sRec);
        if (dF > lim || dF < -lim || dP > lim || dP < -lim || dR > lim || dR < -lim)
            return set_error(X265HIP_EINVAL, "framepass_run_yuv: Cb and Cr planes are more than 2^31 elements apart");
        if (fp->cStrideF != ca->sS || fp->cStrideP != ca->sP || fp->cStrideR != ca->sRec || fp->cDeltaF != dF || fp->cDeltaP != dP || fp->cDeltaR != dR)
        {
            drop_graphs(fp);
            FP_TRY(check_hip(hipStreamSynchronize(as_stream(stream)), "framepass sync"));
            for (int t = 0; t < 2; t++)
            {
                const int64_t strides[3] = { ca->sS, ca->sP, ca->sRec }, deltas[3] = { dF, dP, dR };
                int32_t** dst[3] = { &fp->ctuOffF[t], &fp->ctuOffP[t], &fp->ctuOffR[t] };
                for (int k = 0; k < 3; k++)
                {
                    const size_t n = fp->hCtuXY[t].size() / 2;
                    std::vector<int32_t> off(2 * n);
                    for (size_t i = 0; i < n; i++)
                    {
                        const int64_t o = (int64_t)fp->hCtuXY[t][2 * i + 1] * strides[k] + fp->hCtuXY[t][2 * i];
                        off[i] = (int32_t)o;
                        off[n + i] = (int32_t)(o + deltas[k]);
                    }
                    if (*dst[k]) (void)device_free(*dst[k]);
                    *dst[k] = nullptr;
                    FP_TRY(dev_upload(dst[k], off));
                }
            }
            fp->cStrideF = ca->sS; fp->cStrideP = ca->sP; fp->cStrideR = ca->sRec;
            fp->cDeltaF = dF; fp->cDeltaP = dP; fp->cDeltaR = dR;
        }
        if (!ba)
            FP_TRY(x265hip_pred_inter_chroma_batch(depth, 8, 8, ca->refCb, ca->refCr, ca->sR, ca->predCb, ca->predCr, ca->sP, fp->puXY[3], fp->mv[3],
                                                   fp->nLevel[3], stream));
    }
    FP_MARK(6);
    // 4. residual chains: luma 32x32 TUs on the 32-aligned area + 8x8 TUs on the rest, chroma 16x16 + 4x4 (Cb and Cr together: Cr TUs are
    //    addressed from the Cb base pointers through the offset tables above) with the chroma QpParam — ONE launch (frame.hip)
    const int qp = fp->qp + 6 * (depth - 8);                                            // QpParam.qp = slice qp + QP_BD_OFFSET (quant.cpp:224)
    static const int invQuantScales[6] = { 40, 45, 51, 57, 64, 72 };                     // scalinglist.cpp:130
    {
        ChainJob jobs[4];
        int nj = 0;
        for (int t = 0; t < 2; t++)
        {
            if (!fp->nTu[t]) continue;
            const int log2n = kTuSize[t] == 32 ? 5 : 3;
            const int transformShift = 15 - depth - log2n;                               // MAX_TR_DYNAMIC_RANGE - X265_DEPTH - log2TrSize (quant.cpp:408)
            const int qBits = 14 + qp / 6 + transformShift;                              // QUANT_SHIFT + per + transformShift (quant.cpp:465)
            jobs[nj++] = ChainJob{ kTuSize[t], src, strideS, pred, strideP, recon, strideRec, fp->tuOffF[t], fp->tuOffP[t], fp->tuOffR[t], fp->quantCoeff[t],
                                   qBits, 85 << (qBits - 9),                             // inter rounding (quant.cpp:466)
                                   invQuantScales[qp % 6] << (qp / 6),                   // quant.cpp:567
                                   20 - 14 - transformShift,                             // QUANT_IQUANT_SHIFT - QUANT_SHIFT - transformShift (quant.cpp:552)
                                   fp->level[t], fp->numSig[t], fp->dist[t], fp->nTu[t] };
        }
        if (ca)
        {
            const int bd = 6 * (depth - 8);
            int qpc = fp->qp < -bd ? -bd : (fp->qp > 57 ? 57 : fp->qp);
            if (qpc >= 30) qpc = kChromaScale[qpc];
            qpc += bd;
            for (int t = 0; t < 2; t++)
            {
                if (!fp->nTu[t]) continue;
                const int log2n = kCTuSize[t] == 16 ? 4 : 2;
                const int transformShift = 15 - depth - log2n;
                const int qBits = 14 + qpc / 6 + transformShift;
                jobs[nj++] = ChainJob{ kCTuSize[t], ca->srcCb, ca->sS, ca->predCb, ca->sP, ca->recCb, ca->sRec, fp->ctuOffF[t], fp->ctuOffP[t], fp->ctuOffR[t],
                                       fp->cquantCoeff[t], qBits, 85 << (qBits - 9), invQuantScales[qpc % 6] << (qpc / 6), 20 - 14 - transformShift,
                                       fp->clevel[0][t], fp->cnumSig[0][t], fp->cdist[0][t], 2 * fp->nTu[t] };
            }
        }
        static const bool separate = getenv("X265HIP_CHAIN_SEPARATE") != nullptr;       // A/B switch: one launch per chain, as in round 1
        if (!separate)
            FP_TRY(residual_chain_multi(depth, jobs, nj, as_stream(stream)));
        else
            for (int k = 0; k < nj; k++)
            {
                const ChainJob& j = jobs[k];
                FP_TRY(x265hip_residual_chain_batch(j.size, depth, j.fenc, j.sF, j.pred, j.sP, j.recon, j.sR, j.offF, j.offP, j.offR, j.quantCoeff, j.qBits, j.add,
                                                    j.dqScale, j.dqShift, j.level, j.numSig, j.dist, j.n, stream));
            }
    }
    FP_MARK(7);
    FP_MARK(8);
    // 5. mode costs
    // 4. mode costs
    {
        static const bool perLevel = getenv("X265HIP_SA8D_LEVELS") != nullptr;       // the first version: one Hadamard pass per CU size
        if (perLevel)
        {
            Sa8dLevel lv[4];
            for (int l = 0; l < 4; l++)
                lv[l] = Sa8dLevel{ fp->cuOff[l], fp->cuOffP[l], fp->sa8d[l], fp->nLevel[l], kCuSize[l] };
            FP_TRY(sa8d_levels(depth, src, strideS, pred, strideP, lv, 4, as_stream(stream)));
        }
        else
            FP_TRY(sa8d_pyramid(depth, src, strideS, pred, strideP, fp->width, fp->height, fp->sa8d, as_stream(stream)));
    }
    FP_MARK(9);
    FP_MARK(10);
    // 6. the reconstructed picture becomes a reference
    {
        void* pics[3] = { recon, ca ? ca->recCb : nullptr, ca ? ca->recCr : nullptr };
        const int64_t strides[3] = { strideRec, ca ? ca->sRec : 0, ca ? ca->sRec : 0 };
        const int ws[3] = { fp->width, fp->width / 2, fp->width / 2 }, hs[3] = { fp->height, fp->height / 2, fp->height / 2 };
        const int mxs[3] = { marginX, marginX / 2, marginX / 2 }, mys[3] = { marginY, marginY / 2, marginY / 2 };
        FP_TRY(extend_border_planes(depth, ca ? 3 : 1, pics, strides, ws, hs, mxs, mys, as_stream(stream)));
    }
    FP_MARK(11);
#undef FP_MARK
    return X265HIP_OK;
}

extern "C" {

int x265hip_framepass_run(x265hip_framepass* fp, const void* src, int64_t strideS, const void* ref, int64_t strideR,
                          void* pred, int64_t strideP, void* recon, int64_t strideRec, int marginX, int marginY, void* stream)
{
    return framepass_run_impl(fp, src, strideS, ref, strideR, pred, strideP, recon, strideRec, marginX, marginY, nullptr, stream);
}

int x265hip_framepass_run_yuv(x265hip_framepass* fp, const x265hip_yuv* src, const x265hip_yuv* ref, const x265hip_yuv* pred,
                              const x265hip_yuv* recon, int marginX, int marginY, void* stream)
{
    if (!fp || !src || !ref || !pred || !recon || !src->cb || !src->cr || !ref->cb || !ref->cr || !pred->cb || !pred->cr || !recon->cb || !recon->cr)
        return set_error(X265HIP_EINVAL, "framepass_run_yuv: null plane");
    if ((marginX & 1) || (marginY & 1))
        return set_error(X265HIP_EINVAL, "framepass_run_yuv: margins must be even (4:2:0)");
    ChromaArgs ca = { src->cb, src->cr, ref->cb, ref->cr, pred->cb, pred->cr, recon->cb, recon->cr, src->strideC, ref->strideC, pred->strideC, recon->strideC };
    auto plain = [&]() {
        return framepass_run_impl(fp, src->y, src->strideY, ref->y, ref->strideY, pred->y, pred->strideY, recon->y, recon->strideY, marginX, marginY, &ca, stream);
    };
    // opt-in (X265HIP_GRAPH=1): measured on MI355X at F = 3 the replay cuts the host time to enqueue a step from 0.25 to 0.06 ms but costs 2 % of
    // throughput (7.72 k vs 7.87 k frames/s: every graph node is its own barrier-separated dispatch), and the pass is GPU-bound, not launch-bound
    static const bool useGraph = getenv("X265HIP_GRAPH") != nullptr;
    if (!useGraph || fp->profile || !stream)            // the legacy NULL stream cannot be captured; profiling records events between the stages
    {
        int rc = plain();
        if (!rc) { fp->plainRuns++; }
        return rc;
    }
    x265hip_framepass::GraphKey key{};
    const x265hip_yuv* pics[4] = { src, ref, pred, recon };
    for (int i = 0; i < 4; i++)
    {
        key.p[3 * i] = pics[i]->y; key.p[3 * i + 1] = pics[i]->cb; key.p[3 * i + 2] = pics[i]->cr;
        key.s[2 * i] = pics[i]->strideY; key.s[2 * i + 1] = pics[i]->strideC;
    }
    key.mx = marginX; key.my = marginY;
    auto same_geom = [](const x265hip_framepass::GraphKey& a, const x265hip_framepass::GraphKey& b) {
        return !memcmp(a.s, b.s, sizeof(a.s)) && a.mx == b.mx && a.my == b.my;
    };
    if (fp->plainRuns == 0 || !same_geom(key, fp->lastPlain))
    {
        // first run with these strides: it (re)allocates and uploads tables, which cannot be captured; cached graphs hold the old buffers
        for (auto& g : fp->graphs) (void)hipGraphExecDestroy(g.exec);
        fp->graphs.clear();
        int rc = plain();
        if (!rc) { fp->plainRuns++; fp->lastPlain = key; }
        return rc;
    }
    for (auto& g : fp->graphs)
        if (!memcmp(g.key.p, key.p, sizeof(key.p)))
            return check_hip(hipGraphLaunch(g.exec, as_stream(stream)), "hipGraphLaunch(framepass)");
    if (fp->graphs.size() >= 32)
        return plain();                                  // more distinct pictures than the cache is meant for: stay on plain launches
    if (hipStreamBeginCapture(as_stream(stream), hipStreamCaptureModeThreadLocal) != hipSuccess)
    {
        (void)hipGetLastError();
        return plain();
    }
    const int rc = plain();
    hipGraph_t graph = nullptr;
    const hipError_t ce = hipStreamEndCapture(as_stream(stream), &graph);
    if (rc || ce != hipSuccess || !graph)
    {
        // the capture is over either way; whatever went wrong inside it (an allocation or a synchronize is illegal while capturing), the
        // pass itself is still owed: run it plainly
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        const int rp = plain();
        if (!rp) { fp->plainRuns++; fp->lastPlain = key; }
        return rp;
    }
    hipGraphExec_t exec = nullptr;
    const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ie != hipSuccess || !exec)
    {
        (void)hipGetLastError();
        return plain();
    }
    fp->graphs.push_back(x265hip_framepass::GraphEntry{ key, exec });
    return check_hip(hipGraphLaunch(exec, as_stream(stream)), "hipGraphLaunch(framepass)");
}

int x265hip_framepass_run_yuv_b(x265hip_framepass* fp, const x265hip_yuv* src, const x265hip_yuv* ref0, const x265hip_yuv* ref1, const x265hip_yuv* pred,
                                const x265hip_yuv* recon, int marginX, int marginY, void* stream)
{
    if (!fp || !src || !ref0 || !ref1 || !pred || !recon || !src->cb || !src->cr || !ref0->cb || !ref0->cr || !ref1->y || !ref1->cb || !ref1->cr || !pred->cb ||
        !pred->cr || !recon->cb || !recon->cr)
        return set_error(X265HIP_EINVAL, "framepass_run_yuv_b: null plane");
    if ((marginX & 1) || (marginY & 1))
        return set_error(X265HIP_EINVAL, "framepass_run_yuv_b: margins must be even (4:2:0)");
    if (ref0->strideY != ref1->strideY || ref0->strideC != ref1->strideC)
        return set_error(X265HIP_EINVAL, "framepass_run_yuv_b: the two references must share their strides");
    ChromaArgs ca = { src->cb, src->cr, ref0->cb, ref0->cr, pred->cb, pred->cr, recon->cb, recon->cr, src->strideC, ref0->strideC, pred->strideC, recon->strideC };
    BArgs ba = { ref1->y, ref1->cb, ref1->cr };
    return framepass_run_impl(fp, src->y, src->strideY, ref0->y, ref0->strideY, pred->y, pred->strideY, recon->y, recon->strideY, marginX, marginY, &ca, stream, &ba);
}

int x265hip_framepass_set_profiling(x265hip_framepass* fp, int enable)
{
    if (!fp) return set_error(X265HIP_EINVAL, "framepass_set_profiling: null handle");
    fp->profile = enable != 0;
    return X265HIP_OK;
}

int x265hip_framepass_stage_ms(x265hip_framepass* fp, float* ms11)
{
    if (!fp || !ms11 || !fp->profile)
        return set_error(X265HIP_EINVAL, "framepass_stage_ms: profiling is off");
    FP_TRY(check_hip(hipEventSynchronize(fp->ev[11]), "hipEventSynchronize"));
    for (int i = 0; i < 11; i++)
        FP_TRY(check_hip(hipEventElapsedTime(&ms11[i], fp->ev[i], fp->ev[i + 1]), "hipEventElapsedTime"));
    return X265HIP_OK;
}

int x265hip_framepass_output(x265hip_framepass* fp, int which, int level, void** devPtr, int* count)
{
    if (!fp || !devPtr || !count)
        return set_error(X265HIP_EINVAL, "framepass_output: null argument");
    const bool cuLevel = which <= X265HIP_FP_SA8D;
    if (cuLevel && level >= 4 && level < 8 && (which == X265HIP_FP_MV || which == X265HIP_FP_MECOST))
    {
        // list 1 of the last B pass (x265hip_framepass_run_yuv_b): levels 4..7 = CU size 64, 32, 16, 8
        if (!fp->mv1[level - 4])
            return set_error(X265HIP_EINVAL, "framepass_output: no B pass has run on this handle");
        *devPtr = which == X265HIP_FP_MV ? (void*)fp->mv1[level - 4] : (void*)fp->cost1[level - 4];
        *count = fp->nLevel[level - 4];
        return X265HIP_OK;
    }
    if (level < 0 || level >= (cuLevel ? 4 : 6))
        return set_error(X265HIP_EINVAL, "framepass_output: level %d for output %d", level, which);
    if (!cuLevel && level >= 2)
    {
        // chroma: level 2/3 = Cb 16x16 / 4x4, 4/5 = Cr 16x16 / 4x4 (outputs of x265hip_framepass_run_yuv)
        const int pl = (level - 2) >> 1, t = (level - 2) & 1;
        *count = fp->nTu[t];
        switch (which)
        {
        case X265HIP_FP_LEVEL:  *devPtr = fp->clevel[pl][t]; break;
        case X265HIP_FP_NUMSIG: *devPtr = fp->cnumSig[pl][t]; break;
        case X265HIP_FP_DIST:   *devPtr = fp->cdist[pl][t]; break;
        default: return set_error(X265HIP_EINVAL, "framepass_output: output %d has no chroma form", which);
        }
        return X265HIP_OK;
    }
    switch (which)
    {
    case X265HIP_FP_PU_XY:  *devPtr = fp->puXY[level]; *count = fp->nLevel[level]; break;
    case X265HIP_FP_MV:     *devPtr = fp->mv[level]; *count = fp->nLevel[level]; break;
    case X265HIP_FP_MECOST: *devPtr = fp->cost[level]; *count = fp->nLevel[level]; break;
    case X265HIP_FP_SA8D:   *devPtr = fp->sa8d[level]; *count = fp->nLevel[level]; break;
    case X265HIP_FP_TU_OFF: *devPtr = fp->tuXY[level]; *count = fp->nTu[level]; break;
    case X265HIP_FP_LEVEL:  *devPtr = fp->level[level]; *count = fp->nTu[level]; break;
    case X265HIP_FP_NUMSIG: *devPtr = fp->numSig[level]; *count = fp->nTu[level]; break;
    case X265HIP_FP_DIST:   *devPtr = fp->dist[level]; *count = fp->nTu[level]; break;
    default: return set_error(X265HIP_EINVAL, "framepass_output: unknown output %d", which);
    }
    return X265HIP_OK;
}

} // extern "C"
