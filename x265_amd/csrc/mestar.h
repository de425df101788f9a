// mestar.h — X265_STAR_SEARCH (reference: source/encoder/motion.cpp:1132-1240, StarPatternSearch :362-604, COST_MV_PT_DIST
// :224-236, offsets[] :74-84), shared by the three motion kernels.  `C` provides
//     int fullpel_cost(int mx, int my, int shift)   =  sad(block at full-pel (mx, my)) + mvcost((mx, my) << shift)
//     void fullpel_costs<K>(mx[K], my[K], out[K])   =  the same for K points at once, shift 2
//     int mvcost(int qx, int qy)
// The reference evaluates the points of one distance either in sad_x4 groups (all in range) or one by one behind per-point
// range checks; the order is the same and a point's check is implied by the group check, so: for each point in reference
// order, if its own check holds, evaluate it and update (bcost, bmv, bPointNr, bDistance) on a strict improvement.
#pragma once
#include "common.h"

namespace xh {

struct StarState { int bx, by, bcost, bPointNr, bDistance; };

// K points of one ring measured together and replayed in reference order.  A point's cost does not depend on the running best, so
// batching them (all loads of the ring in flight at once) changes nothing but the latency.  Points that fail their range check are
// measured at the ring centre instead (always inside the range, so inside the padded plane) and ignored.
template <int K, class C>
__device__ __forceinline__ void star_points(C& c, StarState& st, int ox, int oy, const int (&px)[K], const int (&py)[K], const bool (&ok)[K],
                                            const int (&pnr)[K], const int (&dst)[K])
{
    int ex[K], ey[K], cost[K];
#pragma unroll
    for (int i = 0; i < K; i++)
    {
        ex[i] = ok[i] ? px[i] : ox;
        ey[i] = ok[i] ? py[i] : oy;
    }
    c.template fullpel_costs<K>(ex, ey, cost);
#pragma unroll
    for (int i = 0; i < K; i++)
        if (ok[i] && cost[i] < st.bcost) { st.bcost = cost[i]; st.bx = px[i]; st.by = py[i]; st.bPointNr = pnr[i]; st.bDistance = dst[i]; }
}

template <class C>
__device__ __forceinline__ void star_pattern_search(C& c, int minx, int miny, int maxx, int maxy, StarState& st, int earlyExitIters, int merange)
{
    const int ox = st.bx, oy = st.by;
    int saved = st.bcost, rounds = 0;
    {
        const int dist = 1;
        const int top = oy - dist, bottom = oy + dist, left = ox - dist, right = ox + dist;
        const int px[4] = { ox, left, right, ox }, py[4] = { top, oy, oy, bottom };
        const bool ok[4] = { top >= miny, left >= minx, right <= maxx, bottom <= maxy };
        const int pnr[4] = { 2, 4, 5, 7 }, dst[4] = { dist, dist, dist, dist };
        star_points<4>(c, st, ox, oy, px, py, ok, pnr, dst);
        if (st.bcost < saved) rounds = 0;
        else if (++rounds >= earlyExitIters) return;
    }
#pragma unroll 1
    for (int dist = 2; dist <= 8; dist <<= 1)
    {
        const int top = oy - dist, bottom = oy + dist, left = ox - dist, right = ox + dist;
        const int top2 = oy - (dist >> 1), bottom2 = oy + (dist >> 1), left2 = ox - (dist >> 1), right2 = ox + (dist >> 1);
        saved = st.bcost;
        const int px[8] = { ox, left2, right2, left, right, left2, right2, ox };
        const int py[8] = { top, top2, top2, oy, oy, bottom2, bottom2, bottom };
        const bool ok[8] = { top >= miny, top2 >= miny && left2 >= minx, top2 >= miny && right2 <= maxx, left >= minx, right <= maxx,
                             bottom2 <= maxy && left2 >= minx, bottom2 <= maxy && right2 <= maxx, bottom <= maxy };
        const int pnr[8] = { 2, 1, 3, 4, 5, 6, 8, 7 };
        const int dst[8] = { dist, dist >> 1, dist >> 1, dist, dist, dist >> 1, dist >> 1, dist };
        star_points<8>(c, st, ox, oy, px, py, ok, pnr, dst);
        if (st.bcost < saved) rounds = 0;
        else if (++rounds >= earlyExitIters) return;
    }
#pragma unroll 1
    for (int dist = 16; dist <= (int)(int16_t)merange; dist <<= 1)
    {
        const int top = oy - dist, bottom = oy + dist, left = ox - dist, right = ox + dist;
        saved = st.bcost;
        const int q = dist >> 2;
        // reference order: the four axis points, then for index 1..3 the four diagonal-side points (motion.cpp:560-600)
        {
            const int px[8] = { ox, left, right, ox, ox - q, ox + q, ox - q, ox + q };
            const int py[8] = { top, oy, oy, bottom, top + q, top + q, bottom - q, bottom - q };
            const bool ok[8] = { top >= miny, left >= minx, right <= maxx, bottom <= maxy,
                                 top + q >= miny && ox - q >= minx, top + q >= miny && ox + q <= maxx,
                                 bottom - q <= maxy && ox - q >= minx, bottom - q <= maxy && ox + q <= maxx };
            const int pnr[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, dst[8] = { dist, dist, dist, dist, dist, dist, dist, dist };
            star_points<8>(c, st, ox, oy, px, py, ok, pnr, dst);
        }
        {
            const int a = 2 * q, b = 3 * q;
            const int px[8] = { ox - a, ox + a, ox - a, ox + a, ox - b, ox + b, ox - b, ox + b };
            const int py[8] = { top + a, top + a, bottom - a, bottom - a, top + b, top + b, bottom - b, bottom - b };
            const bool ok[8] = { top + a >= miny && ox - a >= minx, top + a >= miny && ox + a <= maxx,
                                 bottom - a <= maxy && ox - a >= minx, bottom - a <= maxy && ox + a <= maxx,
                                 top + b >= miny && ox - b >= minx, top + b >= miny && ox + b <= maxx,
                                 bottom - b <= maxy && ox - b >= minx, bottom - b <= maxy && ox + b <= maxx };
            const int pnr[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, dst[8] = { dist, dist, dist, dist, dist, dist, dist, dist };
            star_points<8>(c, st, ox, oy, px, py, ok, pnr, dst);
        }
        if (st.bcost < saved) rounds = 0;
        else if (++rounds >= earlyExitIters) return;
    }
}

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
                    if (tx + 35 <= maxx)
                    {
                        // two consecutive sad_x4 groups of the reference measured together (eight loads in flight instead of four: the
                        // raster is a chain of dependent-latency batches, not of arithmetic), replayed in the same order
                        int px[8], py[8], cost[8];
#pragma unroll
                        for (int i = 0; i < 8; i++) { px[i] = tx + 5 * i; py[i] = ty; }
                        c.template fullpel_costs<8>(px, py, cost);
                        cost[3] += c.mvcost((tx + 15) << 3, ty << 3) - c.mvcost((tx + 15) << 2, ty << 2);      // the fourth of each group (:1195)
                        cost[7] += c.mvcost((tx + 35) << 3, ty << 3) - c.mvcost((tx + 35) << 2, ty << 2);
#pragma unroll
                        for (int i = 0; i < 8; i++)
                            if (cost[i] < st.bcost) { st.bcost = cost[i]; st.bx = px[i]; st.by = py[i]; }
                        tx += 35;
                    }
                    else if (tx + 15 <= maxx)
                    {
                        // one sad_x4 of the reference: four points measured together
                        const int px[4] = { tx, tx + 5, tx + 10, tx + 15 }, py[4] = { ty, ty, ty, ty };
                        int cost[4];
                        c.template fullpel_costs<4>(px, py, cost);
                        // the reference adds mvcost(tmv << 3) to the fourth (:1195)
                        cost[3] += c.mvcost((tx + 15) << 3, ty << 3) - c.mvcost((tx + 15) << 2, ty << 2);
#pragma unroll
                        for (int i = 0; i < 4; i++)
                            if (cost[i] < st.bcost) { st.bcost = cost[i]; st.bx = px[i]; st.by = py[i]; }
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
