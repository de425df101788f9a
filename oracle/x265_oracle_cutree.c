/* x265_oracle_cutree.c — TEST INFRASTRUCTURE ONLY (see x265_oracle.c).  CPU restatement of the lookahead's CU-tree propagation step:
 * Lookahead::estimateCUPropagate (reference source/encoder/slicetype.cpp:2641-2750) with its primitive estimateCUPropagateCost
 * (common/pixel.cpp:914-940), and Lookahead::cuTreeFinish (:2889-2937, no hevc-aq).  Pinned by tests/test_oracle_vs_ref.py + golden. */
#include <math.h>
#include <stdint.h>
#include <string.h>

static double clip_duration(double f) { return f < 0.01 ? 0.01 : (f > 1.00 ? 1.00 : f); }     /* CLIP_DURATION, ratecontrol.h:45-47 */

static void clip_add(uint16_t* s, int32_t x)
{
    const int32_t v = (int32_t)*s + x;
    *s = (uint16_t)(v < 65535 ? v : 65535);                                                   /* CLIP_ADD :2689 */
}

/* frames p0 < b <= p1 (isP: p1 == b, list 1 unused).  Frame b: propagateIn (its own propagateCost; a single zero row when !referenced),
 * intraCost, lowresCosts (cost | lists_used << 14), invQscale (invQscaleFactor, or invQscaleFactor8x8 for qgSize 8), the two lists' vectors.
 * refCosts0 / refCosts1: the reference frames' propagateCost, updated. */
void orc_cutree_propagate(int widthInCU, int heightInCU, int fpsNum, int fpsDenom, double averageDuration, int bMinusP0, int p1MinusP0, int referenced,
                          int weightedBiPred, const uint16_t* propagateIn, const int32_t* intraCost, const uint16_t* lowresCosts, const int32_t* invQscale,
                          const int32_t* mvs0, const int32_t* mvs1, uint16_t* refCosts0, uint16_t* refCosts1)
{
    uint16_t* refCosts[2] = { refCosts0, refCosts1 };
    const int32_t* mvl[2] = { mvs0, mvs1 };
    const int32_t distScaleFactor = ((bMinusP0 << 8) + (p1MinusP0 >> 1)) / p1MinusP0;
    const int32_t bipredWeight = weightedBiPred ? 64 - (distScaleFactor >> 2) : 32;
    const int32_t bipredWeights[2] = { bipredWeight, 64 - bipredWeight };
    const double fpsFactor = clip_duration((double)fpsDenom / fpsNum) / clip_duration(averageDuration);
    const double fps = fpsFactor / 256;
    const int W = widthInCU, H = heightInCU;
    for (int by = 0; by < H; by++)
        for (int bx = 0; bx < W; bx++)
        {
            const int cu = by * W + bx;
            /* estimateCUPropagateCost, pixel.cpp:914-940 */
            const int ic = intraCost[cu];
            const int interCost = ic < (lowresCosts[cu] & 0x3FFF) ? ic : (lowresCosts[cu] & 0x3FFF);
            const double propagateIntra = ic * invQscale[cu];
            const double propagateAmount = (double)(referenced ? propagateIn[cu] : 0) + propagateIntra * fps;   /* !referenced: one zeroed row is re-used (:2656-2658) */
            const double propagateNum = (double)(ic - interCost);
            const double propagateDenom = (double)ic;
            const int32_t amount = (int)(propagateAmount * propagateNum / propagateDenom + 0.5);
            if (amount <= 0)
                continue;
            const int lists = lowresCosts[cu] >> 14;
            for (int list = 0; list < 2; list++)
            {
                if (!((lists >> list) & 1))
                    continue;
                int32_t la = amount;
                if (lists == 3)
                    la = (la * bipredWeights[list] + 32) >> 6;
                int32_t x = mvl[list][2 * cu], y = mvl[list][2 * cu + 1];
                if (!(x | y))
                {
                    clip_add(&refCosts[list][cu], la);
                    continue;
                }
                const int32_t cux = (x >> 5) + bx, cuy = (y >> 5) + by;
                const int32_t i0 = cux + cuy * W;
                x &= 31;
                y &= 31;
                const int32_t w0 = (32 - y) * (32 - x), w1 = (32 - y) * x, w2 = y * (32 - x), w3 = y * x;
                if (cux >= 0 && cux < W && cuy >= 0 && cuy < H) clip_add(&refCosts[list][i0], (la * w0 + 512) >> 10);
                if (cux + 1 >= 0 && cux + 1 < W && cuy >= 0 && cuy < H) clip_add(&refCosts[list][i0 + 1], (la * w1 + 512) >> 10);
                if (cux >= 0 && cux < W && cuy + 1 >= 0 && cuy + 1 < H) clip_add(&refCosts[list][i0 + W], (la * w2 + 512) >> 10);
                if (cux + 1 >= 0 && cux + 1 < W && cuy + 1 >= 0 && cuy + 1 < H) clip_add(&refCosts[list][i0 + W + 1], (la * w3 + 512) >> 10);
            }
        }
}

/* cuTreeFinish :2889-2937: qpCuTreeOffset = qpAqOffset - strength * log2((intra + propagate) / intra), per 8x8 block (qgSize 16) or for the four
 * quantisation groups of a block (qgSize 8).  cuTreeStrength = 5.0 * (1 - qCompress) (:989). */
void orc_cutree_finish(int widthInCU, int heightInCU, int qgSize, int fpsNum, int fpsDenom, double averageDuration, double qCompress, double weightDelta,
                       const int32_t* intraCost, const int32_t* invQscale, const uint16_t* propagateCost, const double* qpAqOffset, double* qpCuTreeOffset)
{
    const int fpsFactor = (int)(clip_duration(averageDuration) / clip_duration((double)fpsDenom / fpsNum) * 256);
    const double strength = 5.0 * (1.0 - qCompress);
    const int W = widthInCU, H = heightInCU;
    for (int cy = 0; cy < H; cy++)
        for (int cx = 0; cx < W; cx++)
        {
            const int cu = cx + cy * W;
            if (qgSize == 8)
            {
                const int intracost = (intraCost[cu] / 4 * invQscale[cu] + 128) >> 8;
                if (intracost)
                {
                    const int prop = (propagateCost[cu] / 4 * fpsFactor + 128) >> 8;
                    const double r = log2((double)(intracost + prop)) - log2((double)intracost) + weightDelta;
                    const int base = cx * 2 + cy * W * 4, row = W * 2;
                    qpCuTreeOffset[base] = qpAqOffset[base] - strength * r;
                    qpCuTreeOffset[base + 1] = qpAqOffset[base + 1] - strength * r;
                    qpCuTreeOffset[base + row] = qpAqOffset[base + row] - strength * r;
                    qpCuTreeOffset[base + row + 1] = qpAqOffset[base + row + 1] - strength * r;
                }
            }
            else
            {
                const int intracost = (intraCost[cu] * invQscale[cu] + 128) >> 8;
                if (intracost)
                {
                    const int prop = (propagateCost[cu] * fpsFactor + 128) >> 8;
                    const double r = log2((double)(intracost + prop)) - log2((double)intracost) + weightDelta;
                    qpCuTreeOffset[cu] = qpAqOffset[cu] - strength * r;
                }
            }
        }
}
