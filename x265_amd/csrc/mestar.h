// mestar.h — X265_STAR_SEARCH (reference: source/encoder/motion.cpp:1132-1240, StarPatternSearch :362-604, COST_MV_PT_DIST
// :224-236, offsets[] :74-84), shared by the three motion kernels.  `C` provides
//     int fullpel_cost(int mx, int my, int shift)   =  sad(block at full-pel (mx, my)) + mvcost((mx, my) << shift)
// The reference evaluates the points of one distance either in sad_x4 groups (all in range) or one by one behind per-point
// range checks; the order is the same and a point's check is implied by the group check, so: for each point in reference
// order, if its own check holds, evaluate it and update (bcost, bmv, bPointNr, bDistance) on a strict improvement.
#pragma once
#include "common.h"

namespace xh {

struct StarState { int bx, by, bcost, bPointNr, bDistance; };

#define XH_STAR_PT(mx_, my_, cond_, point_, dist_) do { if (cond_) { const int c_ = c.fullpel_cost((mx_), (my_), 2); \
        if (c_ < st.bcost) { st.bcost = c_; st.bx = (mx_); st.by = (my_); st.bPointNr = (point_); st.bDistance = (dist_); } } } while (0)

template <class C>
__device__ __forceinline__ void star_pattern_search(C& c, int minx, int miny, int maxx, int maxy, StarState& st, int earlyExitIters, int merange)
{
    const int ox = st.bx, oy = st.by;
    int saved = st.bcost, rounds = 0;
    {
        const int dist = 1;
        const int top = oy - dist, bottom = oy + dist, left = ox - dist, right = ox + dist;
        XH_STAR_PT(ox, top, top >= miny, 2, dist);
        XH_STAR_PT(left, oy, left >= minx, 4, dist);
        XH_STAR_PT(right, oy, right <= maxx, 5, dist);
        XH_STAR_PT(ox, bottom, bottom <= maxy, 7, dist);
        if (st.bcost < saved) rounds = 0;
        else if (++rounds >= earlyExitIters) return;
    }
    for (int dist = 2; dist <= 8; dist <<= 1)
    {
        const int top = oy - dist, bottom = oy + dist, left = ox - dist, right = ox + dist;
        const int top2 = oy - (dist >> 1), bottom2 = oy + (dist >> 1), left2 = ox - (dist >> 1), right2 = ox + (dist >> 1);
        saved = st.bcost;
        XH_STAR_PT(ox, top, top >= miny, 2, dist);
        XH_STAR_PT(left2, top2, top2 >= miny && left2 >= minx, 1, dist >> 1);
        XH_STAR_PT(right2, top2, top2 >= miny && right2 <= maxx, 3, dist >> 1);
        XH_STAR_PT(left, oy, left >= minx, 4, dist);
        XH_STAR_PT(right, oy, right <= maxx, 5, dist);
        XH_STAR_PT(left2, bottom2, bottom2 <= maxy && left2 >= minx, 6, dist >> 1);
        XH_STAR_PT(right2, bottom2, bottom2 <= maxy && right2 <= maxx, 8, dist >> 1);
        XH_STAR_PT(ox, bottom, bottom <= maxy, 7, dist);
        if (st.bcost < saved) rounds = 0;
        else if (++rounds >= earlyExitIters) return;
    }
    for (int dist = 16; dist <= (int)(int16_t)merange; dist <<= 1)
    {
        const int top = oy - dist, bottom = oy + dist, left = ox - dist, right = ox + dist;
        saved = st.bcost;
        XH_STAR_PT(ox, top, top >= miny, 0, dist);
        XH_STAR_PT(left, oy, left >= minx, 0, dist);
        XH_STAR_PT(right, oy, right <= maxx, 0, dist);
        XH_STAR_PT(ox, bottom, bottom <= maxy, 0, dist);
        for (int index = 1; index < 4; index++)
        {
            const int posYT = top + ((dist >> 2) * index), posYB = bottom - ((dist >> 2) * index);
            const int posXL = ox - ((dist >> 2) * index), posXR = ox + ((dist >> 2) * index);
            XH_STAR_PT(posXL, posYT, posYT >= miny && posXL >= minx, 0, dist);
            XH_STAR_PT(posXR, posYT, posYT >= miny && posXR <= maxx, 0, dist);
            XH_STAR_PT(posXL, posYB, posYB <= maxy && posXL >= minx, 0, dist);
            XH_STAR_PT(posXR, posYB, posYB <= maxy && posXR <= maxx, 0, dist);
        }
        if (st.bcost < saved) rounds = 0;
        else if (++rounds >= earlyExitIters) return;
    }
}
#undef XH_STAR_PT

__device__ __constant__ const int8_t kStarOffsets[16][2] = { {-1,0}, {0,-1}, {-1,-1}, {1,-1}, {-1,0}, {1,0}, {-1,1}, {-1,-1},
                                                              {1,-1}, {1,1}, {-1,0}, {0,1}, {-1,1}, {1,1}, {1,0}, {0,1} };   // motion.cpp:74-84

// bmv (full-pel) / bcost in and out
template <class C>
__device__ __forceinline__ void star_search(C& c, int minx, int miny, int maxx, int maxy, int merange, int& bmvx, int& bmvy, int& bcost)
{
    StarState st = { bmvx, bmvy, bcost, 0, 0 };
#define XH_TRY(mx_, my_, shift_) do { const int c_ = c.fullpel_cost((mx_), (my_), (shift_)); \
        if (c_ < st.bcost) { st.bcost = c_; st.bx = (mx_); st.by = (my_); } } while (0)
#define XH_TWO_POINTS() do { \
        const int x1 = st.bx + kStarOffsets[(st.bPointNr - 1) * 2][0], y1 = st.by + kStarOffsets[(st.bPointNr - 1) * 2][1]; \
        const int x2 = st.bx + kStarOffsets[(st.bPointNr - 1) * 2 + 1][0], y2 = st.by + kStarOffsets[(st.bPointNr - 1) * 2 + 1][1]; \
        if (x1 >= minx && x1 <= maxx && y1 >= miny && y1 <= maxy) XH_TRY(x1, y1, 2); \
        if (x2 >= minx && x2 <= maxx && y2 >= miny && y2 <= maxy) XH_TRY(x2, y2, 2); } while (0)
    bool stop = false;
    star_pattern_search(c, minx, miny, maxx, maxy, st, 3, merange);            // EarlyExitIters = 3 (:1137)
    if (st.bDistance == 1)
    {
        if (st.bPointNr)
        {
            const int saved = st.bcost;
            XH_TWO_POINTS();
            if (st.bcost == saved)
                stop = true;
        }
        else
            stop = true;
    }
    if (!stop)
    {
        if (st.bDistance > 5)                                                  // RasterDistance = 5 (:1170)
        {
            for (int ty = miny; ty <= maxy; ty += 5)
                for (int tx = minx; tx <= maxx; tx += 5)
                {
                    if (tx + 15 <= maxx)
                    {
                        XH_TRY(tx, ty, 2);
                        XH_TRY(tx + 5, ty, 2);
                        XH_TRY(tx + 10, ty, 2);
                        XH_TRY(tx + 15, ty, 3);                                // the reference adds mvcost(tmv << 3) here (:1195)
                        tx += 15;
                    }
                    else
                        XH_TRY(tx, ty, 2);
                }
        }
        while (st.bDistance > 0)
        {
            st.bDistance = 0;
            st.bPointNr = 0;
            star_pattern_search(c, minx, miny, maxx, maxy, st, 32, merange);   // MaxIters = 32 (:1211)
            if (st.bDistance == 1)
            {
                if (!st.bPointNr)
                    break;
                XH_TWO_POINTS();
                break;
            }
        }
    }
#undef XH_TRY
#undef XH_TWO_POINTS
    bmvx = st.bx; bmvy = st.by; bcost = st.bcost;
}

} // namespace xh
