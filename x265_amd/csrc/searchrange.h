// searchrange.h — Search::setSearchRange (reference: source/encoder/search.cpp:2724-2770) with CUData::clipMv
// (cudata.cpp:1915-1928) as a device function, shared by search_range_kernel (motion.hip) and the fused form inside
// motion2_kernel.  Intra-refresh and multi-slice restrictions are at their x265 defaults (off).
#pragma once
#include "common.h"

namespace xh {

struct SearchRange { int minx, miny, maxx, maxy; };

__device__ __forceinline__ SearchRange search_range(int picW, int picH, int maxCUSize, int merange, int refLagPixels, int cx, int cy, int px, int py)
{
    const int dist = merange << 2;
    int minx = px - dist, miny = py - dist, maxx = px + dist, maxy = py + dist;
    const int offset = 8;
    const int xmax = (picW + offset - cx - 1) << 2, xmin = -((maxCUSize + offset + cx - 1) << 2);
    const int ymax = (picH + offset - cy - 1) << 2, ymin = -((maxCUSize + offset + cy - 1) << 2);
    minx = min(xmax, max(xmin, minx)); miny = min(ymax, max(ymin, miny));
    maxx = min(xmax, max(xmin, maxx)); maxy = min(ymax, max(ymin, maxy));
    const int maxMvLen = (1 << 15) - 1;
    minx = max(minx, -maxMvLen); miny = max(miny, -maxMvLen);
    maxx = min(maxx, maxMvLen); maxy = min(maxy, maxMvLen);
    minx >>= 2; miny >>= 2; maxx >>= 2; maxy >>= 2;
    miny = min(miny, refLagPixels);
    maxy = min(maxy, refLagPixels);
    maxy = max(maxy, miny);
    return SearchRange{ minx, miny, maxx, maxy };
}

// optional in-kernel derivation of (qmvp, mvmin, mvmax) for the fused frame-pass launch: qmvp = mvSrc[srcIdx[pu]] or (0,0)
struct DeriveRange
{
    int enable;
    const int32_t* mvSrc;
    const int32_t* srcIdx;
    int picW, picH, maxCUSize, refLagPixels;
    int32_t *qmvpO, *mvminO, *mvmaxO;       // also written, so the arrays hold what the separate entry point would produce
    // optional (row-team kernel, motion3.hip): write the PU's prediction — predInterLumaPixel at the winning vector, i.e. the block of the
    // winning vector's phase plane — to predOut at the PU position.  The search has just walked those cache lines; a separate copy kernel
    // over the whole frame re-fetches 13-27 MB for 2 MB of 8x8 blocks (r01 PMC).  *predDone (host) is set when the kernel honoured it.
    void* predOut;
    int64_t predStride;
    int* predDone;
};

// --me sea: the twelve window-sum planes of the reference picture (picture-origin pointers, the picture's stride), planeElems apart
struct SeaPlanes { const uint32_t* base; int64_t planeElems; int enable; };

// bChromaSATD inputs (4:2:0): source and reference chroma planes at the picture origin (motion.cpp:212, :1601-1660)
struct ChromaPlanes { const void* fencCb; const void* fencCr; int64_t strideFC; const void* refCb; const void* refCr; int64_t strideRC; int enable; };

// the same with the chroma SATD term on every sub-pel comparison: 8x8 / 16x16 / 32x32 PUs on the plane-based row-team kernel.
// Returns 1 when handled (rc = status), 0 when the caller must use the generic path.
int motion_estimate_fused_chroma(int depth, int size, const void* fencPlane, int64_t strideF, int64_t strideR, const void* planes, int64_t planeElems,
                                 const ChromaPlanes& cp, const int32_t* pu_xy, const DeriveRange& dr, int merange, int method, int subme,
                                 const uint16_t* mvcost, int n, int32_t* outMv, int32_t* outCost, hipStream_t st, int* rc);

// internal (framepass.hip): x265hip_set_search_range_batch + x265hip_motion_estimate_planes_batch in ONE launch, square PUs only
int motion_estimate_fused(int depth, int size, const void* fencPlane, int64_t strideF, const void* refPlane, int64_t strideR,
                          const void* planes, int64_t planeElems, const int32_t* pu_xy, const DeriveRange& dr, int merange, int method,
                          int subme, const uint16_t* mvcost, int n, int32_t* outMv, int32_t* outCost, hipStream_t st);

} // namespace xh
