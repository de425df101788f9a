/* x265_oracle_frame.c — TEST INFRASTRUCTURE ONLY (see x265_oracle.h): CPU restatement of the frame pass. */
#include "x265_oracle.h"
#include <stdlib.h>
#include <string.h>

static inline int fmin_i(int a, int b) { return a < b ? a : b; }
static inline int fmax_i(int a, int b) { return a > b ? a : b; }

/* encoder/search.cpp:2724-2770 Search::setSearchRange + common/cudata.cpp:1915-1928 CUData::clipMv
 * (bIntraRefresh off, maxSlices 1 — the x265 defaults). */
void orc_set_search_range(int picW, int picH, int maxCUSize, int merange, int refLagPixels, int cuX, int cuY,
                          const int32_t qmvp[2], int32_t mvmin[2], int32_t mvmax[2])
{
    int dist = merange << 2;
    int mn[2] = { qmvp[0] - dist, qmvp[1] - dist }, mx[2] = { qmvp[0] + dist, qmvp[1] + dist };
    const int offset = 8;
    int xmax = (picW + offset - cuX - 1) << 2, xmin = -((maxCUSize + offset + cuX - 1) << 2);
    int ymax = (picH + offset - cuY - 1) << 2, ymin = -((maxCUSize + offset + cuY - 1) << 2);
    mn[0] = fmin_i(xmax, fmax_i(xmin, mn[0])); mn[1] = fmin_i(ymax, fmax_i(ymin, mn[1]));
    mx[0] = fmin_i(xmax, fmax_i(xmin, mx[0])); mx[1] = fmin_i(ymax, fmax_i(ymin, mx[1]));
    const int maxMvLen = (1 << 15) - 1;
    mn[0] = fmax_i(mn[0], -maxMvLen); mn[1] = fmax_i(mn[1], -maxMvLen);
    mx[0] = fmin_i(mx[0], maxMvLen); mx[1] = fmin_i(mx[1], maxMvLen);
    mn[0] >>= 2; mn[1] >>= 2; mx[0] >>= 2; mx[1] >>= 2;
    mn[1] = fmin_i(mn[1], refLagPixels);
    mx[1] = fmin_i(mx[1], refLagPixels);
    mx[1] = fmax_i(mx[1], mn[1]);
    mvmin[0] = mn[0]; mvmin[1] = mn[1]; mvmax[0] = mx[0]; mvmax[1] = mx[1];
}

/* { bytes, primitive calls, PUs } of each motion-search level of the last orc_frame_pass_* call (see orc_me_stats) */
static uint64_t g_frame_me_stats[4][3];
void orc_frame_pass_me_stats(uint64_t out[12]) { memcpy(out, g_frame_me_stats, sizeof(g_frame_me_stats)); }

#define PIX uint8_t
#define FN(x) x##_8
#include "x265_oracle_frame.inc"
#undef PIX
#undef FN
#define PIX uint16_t
#define FN(x) x##_16
#include "x265_oracle_frame.inc"
#undef PIX
#undef FN
