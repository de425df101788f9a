// meumh.h — X265_UMH_SEARCH (reference: source/encoder/motion.cpp:946-1130; macros COST_MV :238, COST_MV_X4 :277-300, DIA1_ITER :330,
// CROSS :336-360, SAD_THRESH :61, sizeScale :126, hex4[] :66-72, predictorDifference :87-98), shared by the three motion kernels.
// `C` is the same contract as mestar.h:
//     int fullpel_cost(int mx, int my, int shift), void fullpel_costs<K>(mx[K], my[K], out[K]), int mvcost(int qx, int qy)
//
// Every step of the reference is "measure some points around omv, then compare them in a fixed order behind per-point range checks".
// A point's cost never depends on the running best, so each step becomes: four points measured together (all loads in flight), then
// replayed in reference order.  Where the reference uses sad_x4 without an x check (COST_MV_X4) the point is still measured — it lies
// at most two pixels outside the search window, inside the padded plane — and where it checks (COST_MV behind `if`, checkRange) a
// failing point is measured at omv instead and ignored.  The sad_x4 / single-sad split of CROSS and of the hexagon grid is only a
// matter of which checks are known to hold: the points and their order are the same in both forms.
#pragma once
#include "common.h"

namespace xh {

struct UmhState { int bx, by, bcost; };

template <class C>
__device__ __forceinline__ void umh_points4(C& c, UmhState& st, int ox, int oy, const int (&px)[4], const int (&py)[4], const bool (&ok)[4])
{
    int ex[4], ey[4], cost[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        ex[i] = ok[i] ? px[i] : ox;
        ey[i] = ok[i] ? py[i] : oy;
    }
    c.template fullpel_costs<4>(ex, ey, cost);
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (ok[i] && cost[i] < st.bcost) { st.bcost = cost[i]; st.bx = px[i]; st.by = py[i]; }
}

// COST_MV_X4(a, b, c, d) around (ox, oy): only the y range is tested (:293-300)
template <class C>
__device__ __forceinline__ void umh_x4(C& c, UmhState& st, int ox, int oy, int miny, int maxy, int ax, int ay, int bx, int by, int cx, int cy,
                                       int dx, int dy)
{
    const int px[4] = { ox + ax, ox + bx, ox + cx, ox + dx }, py[4] = { oy + ay, oy + by, oy + cy, oy + dy };
    const bool ok[4] = { py[0] >= miny && py[0] <= maxy, py[1] >= miny && py[1] <= maxy, py[2] >= miny && py[2] <= maxy,
                         py[3] >= miny && py[3] <= maxy };
    umh_points4(c, st, ox, oy, px, py, ok);
}

// CROSS(start, x_max, y_max) around (ox, oy): +i, -i for i = start, start + 2, … on the row, then on the column.  While the whole
// arm is known to fit (the guard) the reference takes two steps per sad_x4 and tests only the y range of each point (COST_MV_X4);
// otherwise, and for the tail, it tests the one bound the step moves towards and measures singly.  Same points, same order.
template <class C>
__device__ __forceinline__ void umh_cross(C& c, UmhState& st, int ox, int oy, int minx, int miny, int maxx, int maxy, int start, int x_max, int y_max)
{
    const bool fitsX = x_max <= min(maxx - ox, ox - minx), fitsY = y_max <= min(maxy - oy, oy - miny);
    const bool rowOk = oy >= miny && oy <= maxy;
#pragma unroll 1
    for (int i = start; i < x_max; i += 4)
    {
        const bool x4 = fitsX && i < x_max - 2, second = i + 2 < x_max;
        const int px[4] = { ox + i, ox - i, ox + i + 2, ox - i - 2 }, py[4] = { oy, oy, oy, oy };
        const bool ok[4] = { x4 ? rowOk : px[0] <= maxx, x4 ? rowOk : px[1] >= minx, x4 ? rowOk : second && px[2] <= maxx,
                             x4 ? rowOk : second && px[3] >= minx };
        umh_points4(c, st, ox, oy, px, py, ok);
    }
#pragma unroll 1
    for (int i = start; i < y_max; i += 4)
    {
        const bool x4 = fitsY && i < y_max - 2, second = i + 2 < y_max;
        const int px[4] = { ox, ox, ox, ox }, py[4] = { oy + i, oy - i, oy + i + 2, oy - i - 2 };
        const bool in0 = py[0] >= miny && py[0] <= maxy, in1 = py[1] >= miny && py[1] <= maxy;
        const bool in2 = py[2] >= miny && py[2] <= maxy, in3 = py[3] >= miny && py[3] <= maxy;
        const bool ok[4] = { x4 ? in0 : py[0] <= maxy, x4 ? in1 : py[1] >= miny, x4 ? in2 : second && py[2] <= maxy,
                             x4 ? in3 : second && py[3] >= miny };
        umh_points4(c, st, ox, oy, px, py, ok);
    }
}

__device__ __constant__ const int8_t kUmhHex4[16][2] = { {0,-4}, {0,4}, {-2,-3}, {2,-3}, {-4,-2}, {4,-2}, {-4,-1}, {4,-1},
                                                          {-4,0}, {4,0}, {-4,1}, {4,1}, {-4,2}, {4,2}, {-2,3}, {2,3} };       // motion.cpp:66-72

// bmv (full-pel) / bcost in and out; merange is scaled in place as the reference scales its argument (:1039), which the hexagon refine
// that follows then uses; (fpx, fpy) = pmv.roundToFPel() (:814); mvc = this PU's raw candidate list (numCand pairs).
// Returns true when the search continues into the hexagon refine of X265_HEX_SEARCH (goto me_hex2, :1126-1128).
template <class C>
__device__ __forceinline__ bool umh_search(C& c, int minx, int miny, int maxx, int maxy, int& merange, int& bmvx, int& bmvy, int& bcost, int fpx,
                                           int fpy, int numCand, const int32_t* __restrict__ mvc, int qmvpx, int qmvpy, int w, int h)
{
    UmhState st = { bmvx, bmvy, bcost };
    const int sizeScale = (h * h) >> 4;
#define XH_SAD_THRESH(v) (st.bcost < (((v) >> 4) * sizeScale))
    int cross_start = 1;
    // refine predictors, :951-963
    const int ucost1 = st.bcost;
    umh_x4(c, st, fpx, fpy, miny, maxy, 0, -1, 0, 1, -1, 0, 1, 0);
    if (fpx | fpy)
        umh_x4(c, st, 0, 0, miny, maxy, 0, -1, 0, 1, -1, 0, 1, 0);
    const int ucost2 = st.bcost;
    if ((st.bx | st.by) && !(st.bx == fpx && st.by == fpy))
        umh_x4(c, st, st.bx, st.by, miny, maxy, 0, -1, 0, 1, -1, 0, 1, 0);
    if (st.bcost == ucost2)
        cross_start = 3;
    // early termination, :965-983
    int ox = st.bx, oy = st.by;
    bool done = false;
    if (st.bcost == ucost2 && XH_SAD_THRESH(2000))
    {
        umh_x4(c, st, ox, oy, miny, maxy, 0, -2, -1, -1, 1, -1, -2, 0);
        umh_x4(c, st, ox, oy, miny, maxy, 2, 0, -1, 1, 1, 1, 0, 2);
        if (st.bcost == ucost1 && XH_SAD_THRESH(500))
            done = true;
        else if (st.bcost == ucost2)
        {
            const int range = (int)(int16_t)((int16_t)(merange >> 1) | 1);
            umh_cross(c, st, ox, oy, minx, miny, maxx, maxy, 3, range, range);
            umh_x4(c, st, ox, oy, miny, maxy, -1, -2, 1, -2, -2, -1, 2, -1);
            umh_x4(c, st, ox, oy, miny, maxy, -2, 1, 2, 1, -1, 2, 1, 2);
            if (st.bcost == ucost2)
                done = true;
            else
                cross_start = range + 2;
        }
    }
    if (done)
    {
        bmvx = st.bx; bmvy = st.by; bcost = st.bcost;
        return false;
    }
    // adaptive search range from the agreement of the predictors, :988-1040
    if (numCand)
    {
        const bool is64 = w == 64 && h == 64;
        int mvd, denom = 1;
        if (numCand == 1)
            mvd = is64 ? 25 : abs(qmvpx - mvc[0]) + abs(qmvpy - mvc[1]);
        else
        {
            denom = numCand - 1;
            mvd = 0;
            if (!is64)
            {
                mvd = abs(qmvpx - mvc[0]) + abs(qmvpy - mvc[1]);
                denom++;
            }
            for (int i = 0; i < numCand - 1; i++)
                mvd += abs(mvc[2 * i] - mvc[2 * i + 2]) + abs(mvc[2 * i + 1] - mvc[2 * i + 3]);
        }
        const int sad_ctx = XH_SAD_THRESH(1000) ? 0 : XH_SAD_THRESH(2000) ? 1 : XH_SAD_THRESH(4000) ? 2 : 3;
        const int mvd_ctx = mvd < 10 * denom ? 0 : mvd < 20 * denom ? 1 : mvd < 40 * denom ? 2 : 3;
        // range_mul[mvd_ctx][sad_ctx] (:994-1000) = { {3,3,4,4}, {3,4,4,4}, {4,4,4,5}, {4,4,5,6} }, one nibble each
        const uint32_t rows[4] = { 0x4433u, 0x4443u, 0x5444u, 0x6544u };
        merange = (merange * (int)((rows[mvd_ctx] >> (4 * sad_ctx)) & 15u)) >> 2;
    }
    // :1044-1045, still centred on the origin of the early-termination step
    umh_cross(c, st, ox, oy, minx, miny, maxx, maxy, cross_start, merange, merange >> 1);
    umh_x4(c, st, ox, oy, miny, maxy, -2, -2, -2, 2, 2, -2, 2, 2);
    // hexagon grid, :1047-1125
    ox = st.bx; oy = st.by;
    {
        const int room = min(min(maxx - ox, ox - minx), min(maxy - oy, oy - miny));
        int i = 1;
#pragma unroll 1
        do
        {
            const bool checked = 4 * i > room;          // single SADs behind checkRange; otherwise four sad_x4 with nothing to test
#pragma unroll 1
            for (int g = 0; g < 16; g += 4)
            {
                int px[4], py[4];
                bool ok[4];
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    px[k] = ox + kUmhHex4[g + k][0] * i;
                    py[k] = oy + kUmhHex4[g + k][1] * i;
                    ok[k] = !checked || (px[k] >= minx && px[k] <= maxx && py[k] >= miny && py[k] <= maxy);
                }
                umh_points4(c, st, ox, oy, px, py, ok);
            }
        }
        while (++i <= (merange >> 2));
    }
#undef XH_SAD_THRESH
    bmvx = st.bx; bmvy = st.by; bcost = st.bcost;
    return st.bx >= minx && st.bx <= maxx && st.by >= miny && st.by <= maxy;
}

} // namespace xh
