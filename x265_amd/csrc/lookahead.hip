// lookahead.hip — the lookahead's P-frame cost pass (CostEstimateGroup::estimateCUCost, reference
// source/encoder/slicetype.cpp:3218-3385, with the lowres flavour of MotionEstimate::motionEstimate, motion.cpp:775 and
// :1471-1501, and Lowres::lowresQPelCost, lowres.h:94-120) for batches of (frame, reference) pairs on gfx950.
//
// The reference walks the 8x8 blocks of the half-resolution frame bottom-up, right-to-left, and predicts each block's start
// vector from the blocks it has already done: right, below, below-left, below-right ("reverse-order MV prediction",
// slicetype.cpp:3271-3286).  A row is therefore a strictly serial chain, and row y trails row y+1 by two blocks.  Mapping:
//
//  * one wave = one block row of one pair, walking right to left; rows of a pair sit on ONE XCD in dispatch order
//    (blockIdx -> xcd = b % 8), bottom row first, so a row only ever waits for a workgroup dispatched before it;
//  * the row below publishes each finished vector as ONE 64-bit word {epoch, mv} with an agent-scope relaxed atomic store;
//    the row above spins on the word it needs last (the below-left neighbour) with agent-scope atomic loads — a single-word
//    handshake, no fences, no L2 invalidation;
//  * inside the wave the four DPP rows of 16 lanes are four CANDIDATE slots: the up-to-4 predicted vectors, the 3 start
//    points, the 6 hexagon points, the 8 square-refine points, the 4 half-pel and 4 quarter-pel points are each measured side
//    by side (one 8x8 block per slot: tile-major lanes, v_sad_u8 / quad-DPP Hadamard, row_ror all-reduce), gathered with
//    readlane and replayed in the reference's order on the scalar unit — every cost is independent of the running best, so
//    the result is bit-identical;
//  * quarter-pel blocks are the rounded byte average of two half-pel planes (pixelavg_pp); both loads are always issued
//    (B == A at half-pel positions, avg(a, a) == a), so the fetch is branch-free.
#include "common.h"
#include "internal.h"

namespace xh {

typedef x265hip_lookahead_pair LaPair;

template <typename P> struct LaPk;
template <> struct LaPk<uint8_t>
{
    typedef uint32_t T;
    static __device__ __forceinline__ T avg(T a, T b) { return (a | b) - (((a ^ b) >> 1) & 0x7f7f7f7fu); }
    static __device__ __forceinline__ unsigned sad(T a, T b) { return __builtin_amdgcn_sad_u8(a, b, 0u); }
    static __device__ __forceinline__ void unpack(T a, int v[4]) { v[0] = a & 255; v[1] = (a >> 8) & 255; v[2] = (a >> 16) & 255; v[3] = a >> 24; }
};
template <> struct LaPk<uint16_t>
{
    typedef uint2 T;
    static __device__ __forceinline__ uint32_t avg1(uint32_t a, uint32_t b) { return (a | b) - (((a ^ b) >> 1) & 0x7fff7fffu); }
    static __device__ __forceinline__ T avg(T a, T b) { return make_uint2(avg1(a.x, b.x), avg1(a.y, b.y)); }
    static __device__ __forceinline__ unsigned sad(T a, T b) { return __builtin_amdgcn_sad_u16(a.y, b.y, __builtin_amdgcn_sad_u16(a.x, b.x, 0u)); }
    static __device__ __forceinline__ void unpack(T a, int v[4]) { v[0] = a.x & 0xffff; v[1] = a.x >> 16; v[2] = a.y & 0xffff; v[3] = a.y >> 16; }
};

__device__ __forceinline__ int sfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <typename P>
struct LaCtx
{
    typedef typename LaPk<P>::T Q;
    const P* ref0;            // hpel plane 0 at this lane's quad of the current block
    int64_t planeElems;
    int stride;
    const uint16_t* cost;     // centred MVD cost row
    int px, py;               // mvp (scalar, quarter-pel)
    int slot;                 // candidate slot of this lane (0..3)
    int sel[4];               // sel[j] = slot == j ? ~0 : 0
    bool hi1, hi2;
    Q fq;
    int fu[4];

    // Lowres::lowresMC / lowresQPelCost block fetch (lowres.h:75-90, :102-118)
    __device__ __forceinline__ Q fetch(int qx, int qy) const
    {
        const int ia = (qy & 2) | ((qx & 2) >> 1);
        const int rx = qx + (qx & 1), ry = qy + (qy & 1);
        const int ib = (ry & 2) | ((rx & 2) >> 1);
        // 24-bit full-rate multiplies (lowres planes are far below 2^24 elements, strides below 2^23: checked at the entry point)
        const P* a = ref0 + (__umul24(ia, (int)planeElems) + __mul24(qy >> 2, stride) + (qx >> 2));
        const P* b = ref0 + (__umul24(ib, (int)planeElems) + __mul24(ry >> 2, stride) + (rx >> 2));
        return LaPk<P>::avg(ld_global_unaligned<Q>(a), ld_global_unaligned<Q>(b));
    }
    __device__ __forceinline__ int sad(int qx, int qy) const { return row_allsum((int)LaPk<P>::sad(fetch(qx, qy), fq)); }
    __device__ __forceinline__ int satd(int qx, int qy) const { return satd_of(fetch(qx, qy)); }
    // 8x8 SATD of the source block against a predicted block held one packed quad per lane
    __device__ __forceinline__ int satd_of(Q blk) const
    {
        int p[4];
        LaPk<P>::unpack(blk, p);
        const int d0 = fu[0] - p[0], d1 = fu[1] - p[1], d2 = fu[2] - p[2], d3 = fu[3] - p[3];
        const int s01 = d0 + d1, e01 = d0 - d1, s23 = d2 + d3, e23 = d2 - d3;
        int m[4] = { s01 + s23, s01 - s23, e01 + e23, e01 - e23 };
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            const int pr = __builtin_amdgcn_mov_dpp(m[i], 0xB1, 0xF, 0xF, true);
            m[i] = hi1 ? pr - m[i] : m[i] + pr;
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            const int pr = __builtin_amdgcn_mov_dpp(m[i], 0x4E, 0xF, 0xF, true);
            m[i] = hi2 ? pr - m[i] : m[i] + pr;
        }
        return row_allsum(iabs(m[0]) + iabs(m[1]) + iabs(m[2]) + iabs(m[3])) >> 1;
    }
    __device__ __forceinline__ int mvcost(int qx, int qy) const { return (int)(uint16_t)(cost[qx - px] + cost[qy - py]); }

    // measure 4*K candidates (quarter-pel vectors, scalar arrays) side by side: slot j of round k takes candidate 4k + j.
    // dist[] = SAD or SATD, mvc[] = lambda * bits of the vector.
    template <int K>
    __device__ __forceinline__ void eval(const int (&cx)[4 * K], const int (&cy)[4 * K], bool useSatd, int (&dist)[4 * K], int (&mvc)[4 * K]) const
    {
        int vd[K], vm[K];
#pragma unroll
        for (int k = 0; k < K; k++)
        {
            // this lane's candidate out of four wave-uniform values: mask-and-or (branch-free; the ?: chain compiled to exec-mask branches)
            const int qx = (cx[4 * k] & sel[0]) | (cx[4 * k + 1] & sel[1]) | (cx[4 * k + 2] & sel[2]) | (cx[4 * k + 3] & sel[3]);
            const int qy = (cy[4 * k] & sel[0]) | (cy[4 * k + 1] & sel[1]) | (cy[4 * k + 2] & sel[2]) | (cy[4 * k + 3] & sel[3]);
            vd[k] = useSatd ? satd(qx, qy) : sad(qx, qy);
            vm[k] = mvcost(qx, qy);
        }
#pragma unroll
        for (int k = 0; k < K; k++)
#pragma unroll
            for (int j = 0; j < 4; j++)
            {
                dist[4 * k + j] = __builtin_amdgcn_readlane(vd[k], 16 * j);
                mvc[4 * k + j] = __builtin_amdgcn_readlane(vm[k], 16 * j);
            }
    }
};

__device__ __forceinline__ int la_clip(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ uint64_t la_load(const uint64_t* p) { return __hip_atomic_load((const XH_GLOBAL_AS uint64_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void la_unpack(uint64_t w, int& x, int& y)
{
    x = sfl((int)(int16_t)(uint16_t)(w & 0xffff));
    y = sfl((int)(int16_t)(uint16_t)((w >> 16) & 0xffff));
}

__device__ __forceinline__ void la_add64(int64_t* p, int v) { atomicAdd((unsigned long long*)p, (unsigned long long)(long long)v); }

template <typename P>
__global__ __launch_bounds__(64) void lookahead_p_kernel(const LaPair* __restrict__ pairs, int nPairs, int64_t stride, int64_t planeElems, int W, int H,
                                                         int rowsPerSlice, int numSlices, const uint16_t* __restrict__ costTab, uint32_t epoch,
                                                         int64_t* __restrict__ est)
{
    typedef LaCtx<P> C;
    typedef typename C::Q Q;
    constexpr int N = 8;
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int pairIdx = (k / H) * 8 + xcd;
    if (pairIdx >= nPairs)
        return;
    const int cuY = H - 1 - (k % H);
    const LaPair pr = pairs[pairIdx];
    const bool bidirList = pr.bidirList != 0;
    if (pr.sliceGeom)
    {
        // this pair's own cooperative-slice geometry (x265hip_lookahead_pair::sliceGeom): batch-mode and single estimates share a launch
        rowsPerSlice = pr.sliceGeom & 0xffff;
        numSlices = pr.sliceGeom >> 16;
    }
    int slice = cuY / rowsPerSlice;
    if (slice > numSlices - 1) slice = numSlices - 1;
    const int lastY = slice == numSlices - 1 ? H - 1 : rowsPerSlice * (slice + 1) - 1;
    const bool lastRow = cuY == lastY;

    const int lane = threadIdx.x, s = lane & 15;
    const int t = s >> 2, r = s & 3;
    const int qrow = (t >> 1) * 4 + r, qcol = (t & 1) * 4;       // tile-major: the 4 rows of a 4x4 tile in one DPP quad
    C c;
    c.slot = lane >> 4;
#pragma unroll
    for (int j = 0; j < 4; j++)
        c.sel[j] = c.slot == j ? -1 : 0;
    c.hi1 = s & 1;
    c.hi2 = s & 2;
    c.stride = (int)stride;
    c.planeElems = planeElems;
    c.cost = costTab;
    const P* fencRow = (const P*)pr.fenc + (int64_t)(cuY * N + qrow) * stride + qcol;
    const P* refRow = (const P*)pr.ref + (int64_t)(cuY * N + qrow) * stride + qcol;
    const uint64_t* below = pr.sync + (int64_t)(cuY + 1) * W;
    uint64_t* mine = pr.sync + (int64_t)cuY * W;

    const int mvminY = -cuY * N - 8, mvmaxY = (H - cuY - 1) * N + 8;
    int rowSum = 0, scoreSum = 0, scoreAq = 0, intraCnt = 0;
    int prevX = 0, prevY = 0;
    bool stuck = false;

#pragma unroll 1
    for (int cuX = W - 1; cuX >= 0; cuX--)
    {
        const int cuXY = cuY * W + cuX;
        c.fq = ld_global_unaligned<Q>(fencRow + cuX * N);
        LaPk<P>::unpack(c.fq, c.fu);
        c.ref0 = refRow + cuX * N;
        const int mvminX = -cuX * N - 8, mvmaxX = (W - cuX - 1) * N + 8;
        const int qminX = mvminX * 4, qmaxX = mvmaxX * 4, qminY = mvminY * 4, qmaxY = mvmaxY * 4;

        // ---- reverse-order MV prediction (slicetype.cpp:3271-3307): SATD of each neighbour vector, cheapest wins ----
        // candidate order: right neighbour (cuX < W - 1), then below, below-left (cuX > 0), below-right (cuX < W - 1) when there is a row below.
        // Built with wave-uniform selects only: a dynamically indexed array would live in scratch memory, one more round trip per block.
        int mx[4] = { 0, 0, 0, 0 }, my[4] = { 0, 0, 0, 0 };
        int numc = 0;
        const bool hasRight = cuX < W - 1, hasLeft = cuX > 0;
        // the bookkeeping inputs of this block, requested now and used after the search (off the dependent chain)
        const int icLoad = bidirList ? 0 : ld_global_unaligned<int>(pr.intraCost + cuXY);
        const int iqLoad = (!bidirList && pr.invQscale) ? ld_global_unaligned<int>(pr.invQscale + cuXY) : 256;
        if (!lastRow)
        {
            // The row below moves right to left, so its below-left block (cuX - 1) is published last.  Every word carries the epoch next to
            // the vector and is written / read by ONE 64-bit agent-scope atomic, so each of the (up to) three words is validated on its own:
            // no ordering between different words is assumed.  Bounded: the row below is always dispatched earlier, so this normally takes
            // microseconds; if a word never arrives (a `sync` scratch that was not zeroed) give up after ~2 s instead of hanging the device
            // and report it through est[4 i + 3].
            const int lo = hasLeft ? cuX - 1 : cuX, hi = hasRight ? cuX + 1 : cuX;
            uint64_t w0 = 0, w1 = 0, w2 = 0;
            int spins = 0;
            for (;;)
            {
                w0 = la_load(below + cuX);
                w1 = la_load(below + lo);
                w2 = la_load(below + hi);
                const bool ok = (uint32_t)sfl((int)(w0 >> 32)) == epoch && (uint32_t)sfl((int)(w1 >> 32)) == epoch && (uint32_t)sfl((int)(w2 >> 32)) == epoch;
                if (ok || stuck)
                    break;
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1 << 22)) stuck = true;            // sticky: the rest of the row no longer waits
            }
            int bx, by, lx, ly, rx, ry;
            la_unpack(w0, bx, by);
            la_unpack(w1, lx, ly);
            la_unpack(w2, rx, ry);
            if (hasRight)
            {
                mx[0] = prevX; my[0] = prevY;
                mx[1] = bx; my[1] = by;
                mx[2] = hasLeft ? lx : rx; my[2] = hasLeft ? ly : ry;
                mx[3] = rx; my[3] = ry;                               // only counted when hasLeft
                numc = hasLeft ? 4 : 3;
            }
            else
            {
                mx[0] = bx; my[0] = by;
                mx[1] = lx; my[1] = ly;                               // only counted when hasLeft
                numc = hasLeft ? 2 : 1;
            }
        }
        else if (hasRight)
        {
            mx[0] = prevX; my[0] = prevY;
            numc = 1;
        }
        int mvpX = 0, mvpY = 0, skipCost = 0x7fffffff;
        c.px = 0; c.py = 0;
        if (numc)
        {
            int d[4], m[4];
            c.template eval<1>(mx, my, true, d, m);
            int mvpcost = 1 << 28;                                // MotionEstimate::COST_MAX (motion.h:65)
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (i < numc)
                {
                    if (d[i] < mvpcost) { mvpcost = d[i]; mvpX = mx[i]; mvpY = my[i]; }
                    if (bidirList && !(mvpX | mvpY))              // slicetype.cpp:3303-3305: this candidate's cost while the best so far is zero
                        skipCost = d[i];
                }
        }
        c.px = mvpX; c.py = mvpY;

        // ---- motionEstimate, lowres flavour (motion.cpp:761-795) ----
        const int pmvX = la_clip(mvpX, qminX, qmaxX), pmvY = la_clip(mvpY, qminY, qmaxY);
        int bmvX = (pmvX + 2) >> 2, bmvY = (pmvY + 2) >> 2;
        int bcost;
        int bprecost;
        {
            const int cx[4] = { pmvX, bmvX * 4, 0, 0 }, cy[4] = { pmvY, bmvY * 4, 0, 0 };
            int d[4], m[4];
            c.template eval<1>(cx, cy, false, d, m);
            bprecost = d[0];
            bcost = bprecost;
            if ((pmvX | pmvY) & 3)
                bcost = d[1] + m[1];
            if (pmvX | pmvY)
            {
                const int z = d[2] + m[2];
                if (z < bcost)
                {
                    bcost = z;
                    bmvX = 0;
                    bmvY = mvmaxY < 0 ? mvmaxY : 0;
                    if (bmvY < mvminY) bmvY = mvminY;
                }
            }
        }
#define YOK(yy) (((yy) >= mvminY) & ((yy) <= mvmaxY))
#define LT1(v) do { const int v_ = (v); if (v_ < bcost) bcost = v_; } while (0)
        // ---- hexagon search, merange 16 (motion.cpp:855-944) ----
        {
            const int cx[8] = { (bmvX - 2) * 4, (bmvX - 1) * 4, (bmvX + 1) * 4, (bmvX + 2) * 4, (bmvX + 1) * 4, (bmvX - 1) * 4, bmvX * 4, bmvX * 4 };
            const int cy[8] = { bmvY * 4, (bmvY + 2) * 4, (bmvY + 2) * 4, bmvY * 4, (bmvY - 2) * 4, (bmvY - 2) * 4, bmvY * 4, bmvY * 4 };
            int d[8], m[8];
            c.template eval<2>(cx, cy, false, d, m);
            bcost <<= 3;
            if (YOK(bmvY)) LT1(((d[0] + m[0]) << 3) + 2);
            if (YOK(bmvY + 2))
            {
                LT1(((d[1] + m[1]) << 3) + 3);
                LT1(((d[2] + m[2]) << 3) + 4);
            }
            if (YOK(bmvY)) LT1(((d[3] + m[3]) << 3) + 5);
            if (YOK(bmvY - 2))
            {
                LT1(((d[4] + m[4]) << 3) + 6);
                LT1(((d[5] + m[5]) << 3) + 7);
            }
        }
        if (bcost & 7)
        {
            // hex2[] = {-1,-2},{-2,0},{-1,2},{1,2},{2,0},{1,-2},{-1,-2},{-2,0}; mod6m1[] = 5,0,1,2,3,4,5,0 (motion.cpp:63-64)
            auto hx = [](int i) { return (int)((0x679A9767u >> (4 * i)) & 15) - 8; };
            auto hy = [](int i) { return (int)((0x8668AA86u >> (4 * i)) & 15) - 8; };
            auto m6 = [](int i) { return (int)((0x05432105u >> (4 * i)) & 15); };
            int dir = (bcost & 7) - 2;
            if (YOK(bmvY + hy(dir + 1)))
            {
                bmvX += hx(dir + 1);
                bmvY += hy(dir + 1);
#pragma unroll 1
                for (int i = (16 >> 1) - 1; i > 0 && bmvX >= mvminX && bmvX <= mvmaxX && bmvY >= mvminY && bmvY <= mvmaxY; i--)
                {
                    const int cx[4] = { (bmvX + hx(dir)) * 4, (bmvX + hx(dir + 1)) * 4, (bmvX + hx(dir + 2)) * 4, bmvX * 4 };
                    const int cy[4] = { (bmvY + hy(dir)) * 4, (bmvY + hy(dir + 1)) * 4, (bmvY + hy(dir + 2)) * 4, bmvY * 4 };
                    int d[4], m[4];
                    c.template eval<1>(cx, cy, false, d, m);
                    bcost &= ~7;
                    if (YOK(bmvY + hy(dir))) LT1(((d[0] + m[0]) << 3) + 1);
                    if (YOK(bmvY + hy(dir + 1))) LT1(((d[1] + m[1]) << 3) + 2);
                    if (YOK(bmvY + hy(dir + 2))) LT1(((d[2] + m[2]) << 3) + 3);
                    if (!(bcost & 7))
                        break;
                    dir += (bcost & 7) - 2;
                    dir = m6(dir + 1);
                    bmvX += hx(dir + 1);
                    bmvY += hy(dir + 1);
                }
            }
        }
        bcost >>= 3;
        {
            // square refine (motion.cpp:918-942): square1[1..8] = (0,-1),(0,1),(-1,0),(1,0),(-1,-1),(-1,1),(1,-1),(1,1)
            const int cx[8] = { bmvX * 4, bmvX * 4, (bmvX - 1) * 4, (bmvX + 1) * 4, (bmvX - 1) * 4, (bmvX - 1) * 4, (bmvX + 1) * 4, (bmvX + 1) * 4 };
            const int cy[8] = { (bmvY - 1) * 4, (bmvY + 1) * 4, bmvY * 4, bmvY * 4, (bmvY - 1) * 4, (bmvY + 1) * 4, (bmvY - 1) * 4, (bmvY + 1) * 4 };
            int d[8], m[8];
            c.template eval<2>(cx, cy, false, d, m);
            int dir = -1;
            const bool up = YOK(bmvY - 1), dn = YOK(bmvY + 1);
            const bool ok[8] = { up, dn, true, true, up, dn, up, dn };
#pragma unroll
            for (int i = 0; i < 8; i++)
            {
                const int v = d[i] + m[i];
                if (ok[i] && v < bcost) { bcost = v; dir = i; }
            }
            if (dir >= 0)
            {
                bmvX += (int)((0x99779788u >> (4 * dir)) & 15) - 8;   // x offsets {0,0,-1,1,-1,-1,1,1} + 8, one nibble each
                bmvY += (int)((0x97978897u >> (4 * dir)) & 15) - 8;   // y offsets {-1,1,0,0,-1,1,-1,1} + 8
            }
        }
        // ---- motion.cpp:1449-1455 ----
        if (bprecost < bcost)
        {
            bmvX = pmvX; bmvY = pmvY;
            bcost = bprecost;
        }
        else
        {
            bmvX *= 4; bmvY *= 4;
        }
        // ---- lowres sub-pel refine (motion.cpp:1466-1501, workload[1]: 4 half-pel and 4 quarter-pel directions) ----
        if (!bcost)
        {
            const int cx[4] = { bmvX, bmvX, bmvX, bmvX }, cy[4] = { bmvY, bmvY, bmvY, bmvY };
            int d[4], m[4];
            c.template eval<1>(cx, cy, false, d, m);
            bcost = m[0];
        }
        else
        {
            {
                const int cx[4] = { bmvX, bmvX, bmvX - 2, bmvX + 2 }, cy[4] = { bmvY - 2, bmvY + 2, bmvY, bmvY };
                int d[4], m[4];
                c.template eval<1>(cx, cy, false, d, m);
                int bdir = -1;
#pragma unroll
                for (int i = 0; i < 4; i++)
                {
                    const int v = d[i] + m[i];
                    if (cy[i] >= qminY && cy[i] <= qmaxY && v < bcost) { bcost = v; bdir = i; }
                }
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (bdir == i) { const int nx = cx[i], ny = cy[i]; bmvX = nx; bmvY = ny; }
            }
            {
                const int cx[8] = { bmvX, bmvX, bmvX, bmvX - 1, bmvX + 1, bmvX, bmvX, bmvX }, cy[8] = { bmvY, bmvY - 1, bmvY + 1, bmvY, bmvY, bmvY, bmvY, bmvY };
                int d[8], m[8];
                c.template eval<2>(cx, cy, true, d, m);
                bcost = d[0] + m[0];
                int bdir = -1;
#pragma unroll
                for (int i = 1; i < 5; i++)
                {
                    const int v = d[i] + m[i];
                    if (cy[i] >= qminY && cy[i] <= qmaxY && v < bcost) { bcost = v; bdir = i; }
                }
#pragma unroll
                for (int i = 1; i < 5; i++)
                    if (bdir == i) { const int nx = cx[i], ny = cy[i]; bmvX = nx; bmvY = ny; }
            }
        }
#undef YOK
#undef LT1
        if (bidirList && skipCost < 64 && skipCost < bcost)      // slicetype.cpp:3313-3317
        {
            bcost = skipCost;
            bmvX = 0; bmvY = 0;
        }
        // ---- publish, then the block's bookkeeping (slicetype.cpp:3318-3384, P frame: inter + 4 against intra) ----
        if (lane == 0)
            __hip_atomic_store(mine + cuX, ((uint64_t)epoch << 32) | (uint32_t)(uint16_t)bmvX | ((uint32_t)(uint16_t)bmvY << 16),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        prevX = bmvX; prevY = bmvY;
        const int fencCost = bcost;
        if (bidirList)
        {
            // one list of a B frame: only the vector and its cost; x265hip_lookahead_bidir_batch does the rest
            if (lane == 0)
            {
                pr.mvs[2 * cuXY] = bmvX;
                pr.mvs[2 * cuXY + 1] = bmvY;
                pr.mvCosts[cuXY] = fencCost;
            }
            continue;
        }
        int cuCost = fencCost + 4, listused = 1;
        const int ic = sfl(icLoad);
        if (ic < cuCost) { cuCost = ic; listused = 0; }
        const bool scored = (cuX > 0 && cuX < W - 1 && cuY > 0 && cuY < H - 1) || W <= 2 || H <= 2;
        int cuCostAq = cuCost;
        if (scored)
        {
            if (pr.invQscale)
                cuCostAq = (cuCost * sfl(iqLoad) + 128) >> 8;
            scoreSum += cuCost;
            scoreAq += cuCostAq;
            intraCnt += !listused;
        }
        rowSum += cuCostAq;
        if (lane == 0)
        {
            pr.mvs[2 * cuXY] = bmvX;
            pr.mvs[2 * cuXY + 1] = bmvY;
            pr.mvCosts[cuXY] = fencCost;
            pr.lowresCosts[cuXY] = (uint16_t)((cuCost < 16383 ? cuCost : 16383) | (listused << 14));
        }
    }
    if (lane == 0 && !bidirList)
    {
        pr.rowSatds[cuY] = rowSum;
        la_add64(est + 4 * pairIdx, scoreSum);
        la_add64(est + 4 * pairIdx + 1, scoreAq);
        la_add64(est + 4 * pairIdx + 2, intraCnt);
    }
    if (lane == 0 && stuck)
        atomicOr((unsigned long long*)(est + 4 * pairIdx + 3), 1ull);   // idempotent: any number of stuck rows leaves the same mark
}

// ---- a P estimate whose search was done before (bDoSearch[0] == false): only the bookkeeping of estimateCUCost ---------------
__global__ __launch_bounds__(64) void lookahead_pcost_kernel(const LaPair* __restrict__ pairs, int W, int H, int64_t* __restrict__ est)
{
    const int pairIdx = blockIdx.x / H, cuY = blockIdx.x % H;
    const LaPair pr = pairs[pairIdx];
    int rowSum = 0, scoreSum = 0, scoreAq = 0, intraCnt = 0;
    for (int cuX = threadIdx.x; cuX < W; cuX += 64)
    {
        const int cuXY = cuY * W + cuX;
        int cuCost = pr.mvCosts[cuXY] + 4, listused = 1;
        const int ic = pr.intraCost[cuXY];
        if (ic < cuCost) { cuCost = ic; listused = 0; }
        const bool scored = (cuX > 0 && cuX < W - 1 && cuY > 0 && cuY < H - 1) || W <= 2 || H <= 2;
        int cuCostAq = cuCost;
        if (scored)
        {
            if (pr.invQscale)
                cuCostAq = (cuCost * pr.invQscale[cuXY] + 128) >> 8;
            scoreSum += cuCost;
            scoreAq += cuCostAq;
            intraCnt += !listused;
        }
        rowSum += cuCostAq;
        pr.lowresCosts[cuXY] = (uint16_t)((cuCost < 16383 ? cuCost : 16383) | (listused << 14));
    }
    rowSum = wave_sum(rowSum);
    scoreSum = wave_sum(scoreSum);
    scoreAq = wave_sum(scoreAq);
    intraCnt = wave_sum(intraCnt);
    if (threadIdx.x == 0)
    {
        pr.rowSatds[cuY] = rowSum;
        la_add64(est + 4 * pairIdx, scoreSum);
        la_add64(est + 4 * pairIdx + 1, scoreAq);
        la_add64(est + 4 * pairIdx + 2, intraCnt);
    }
}

// ---- B frames: the part of estimateCUCost after the two list searches (slicetype.cpp:3320-3338, :3353-3384) ----------------
// One wave per block row; its four DPP rows take four neighbouring blocks at a time.  No dependencies between blocks.
template <typename P>
__global__ __launch_bounds__(64) void lookahead_bidir_kernel(const x265hip_lookahead_bframe* __restrict__ frames, int64_t stride, int64_t planeElems,
                                                             int W, int H, int64_t* __restrict__ est)
{
    typedef LaCtx<P> C;
    typedef typename C::Q Q;
    constexpr int N = 8;
    const int f = blockIdx.x / H, cuY = blockIdx.x % H;
    const x265hip_lookahead_bframe fr = frames[f];
    const int lane = threadIdx.x, s = lane & 15, slot = lane >> 4;
    const int t = s >> 2, r = s & 3;
    const int qrow = (t >> 1) * 4 + r, qcol = (t & 1) * 4;
    C c0, c1;
    c0.slot = c1.slot = slot;
    c0.hi1 = c1.hi1 = s & 1;
    c0.hi2 = c1.hi2 = s & 2;
    c0.stride = c1.stride = (int)stride;
    c0.planeElems = c1.planeElems = planeElems;
    const int64_t rowOff = (int64_t)(cuY * N + qrow) * stride + qcol;
    int rowSum = 0, scoreSum = 0, scoreAq = 0;
#pragma unroll 1
    for (int x0 = 0; x0 < W; x0 += 4)
    {
        const int cuX = x0 + slot;
        const bool live = cuX < W;
        const int cx = live ? cuX : W - 1;
        const int cuXY = cuY * W + cx;
        c0.fq = ld_unaligned<Q>((const P*)fr.fenc + rowOff + cx * N);
        LaPk<P>::unpack(c0.fq, c0.fu);
        c0.ref0 = (const P*)fr.ref0 + rowOff + cx * N;
        c1.ref0 = (const P*)fr.ref1 + rowOff + cx * N;
        int bcost = 1 << 28, listused = 0;
        const int m0 = fr.mvCosts0[cuXY], m1 = fr.mvCosts1[cuXY];
        if (m0 < bcost) { bcost = m0; listused = 1; }
        if (m1 < bcost) { bcost = m1; listused = 2; }
        // avg(l0-mv, l1-mv), then the co-located average (pixelavg_pp of the two motion-compensated blocks)
        const Q a = LaPk<P>::avg(c0.fetch(fr.mvs0[2 * cuXY], fr.mvs0[2 * cuXY + 1]), c1.fetch(fr.mvs1[2 * cuXY], fr.mvs1[2 * cuXY + 1]));
        const Q b = LaPk<P>::avg(c0.fetch(0, 0), c1.fetch(0, 0));
        const int ca = c0.satd_of(a), cb = c0.satd_of(b);
        if (ca < bcost) { bcost = ca; listused = 3; }
        if (cb < bcost) { bcost = cb; listused = 3; }
        bcost += 4;                                               // lowresPenalty
        if (live && s == 0)
        {
            fr.lowresCosts[cuXY] = (uint16_t)((bcost < 16383 ? bcost : 16383) | (listused << 14));
            int bcostAq = bcost;
            if ((cuX > 0 && cuX < W - 1 && cuY > 0 && cuY < H - 1) || W <= 2 || H <= 2)
            {
                if (fr.invQscale)
                    bcostAq = (bcost * fr.invQscale[cuXY] + 128) >> 8;
                scoreSum += bcost;
                scoreAq += bcostAq;
            }
            rowSum += bcostAq;
        }
    }
    // lanes 0, 16, 32, 48 hold the partial sums of their slots
    rowSum = __builtin_amdgcn_readlane(rowSum, 0) + __builtin_amdgcn_readlane(rowSum, 16) + __builtin_amdgcn_readlane(rowSum, 32) + __builtin_amdgcn_readlane(rowSum, 48);
    scoreSum = __builtin_amdgcn_readlane(scoreSum, 0) + __builtin_amdgcn_readlane(scoreSum, 16) + __builtin_amdgcn_readlane(scoreSum, 32) + __builtin_amdgcn_readlane(scoreSum, 48);
    scoreAq = __builtin_amdgcn_readlane(scoreAq, 0) + __builtin_amdgcn_readlane(scoreAq, 16) + __builtin_amdgcn_readlane(scoreAq, 32) + __builtin_amdgcn_readlane(scoreAq, 48);
    if (lane == 0)
    {
        fr.rowSatds[cuY] = rowSum;
        la_add64(est + 2 * f, scoreSum);
        la_add64(est + 2 * f + 1, scoreAq);
    }
}

} // namespace xh

using namespace xh;

extern "C" int x265hip_lookahead_cost_p_batch(int depth, const x265hip_lookahead_pair* pairs, int nPairs, int64_t stride, int64_t planeElems,
                                              int widthInCU, int heightInCU, int numRowsPerSlice, int numSlices,
                                              const uint16_t* mvcost, uint32_t epoch, int64_t* est, void* stream)
{
    XH_CHECK_DEV();
    if (!valid_depth(depth) || nPairs < 0 || widthInCU < 1 || heightInCU < 1 || numSlices < 1 || numRowsPerSlice < 1 || epoch == 0 ||
        stride >= (1 << 23) || planeElems >= (1 << 24) || planeElems < 0 || !est ||
        (long long)numRowsPerSlice * (numSlices - 1) >= heightInCU)
        return set_error(X265HIP_EINVAL, "lookahead_cost_p_batch: depth %d pairs %d grid %dx%d slices %d x %d rows epoch %u", depth, nPairs, widthInCU,
                         heightInCU, numSlices, numRowsPerSlice, epoch);
    if (nPairs == 0) return X265HIP_OK;
    int e = check_hip(hipMemsetAsync(est, 0, (size_t)nPairs * 4 * sizeof(int64_t), as_stream(stream)), "lookahead memset");
    if (e) return e;
    const int groups = (nPairs + 7) / 8;
    dim3 grid((unsigned)(groups * heightInCU * 8)), block(64);
    if (depth == 8)
        hipLaunchKernelGGL((lookahead_p_kernel<uint8_t>), grid, block, 0, as_stream(stream), pairs, nPairs, stride, planeElems, widthInCU, heightInCU,
                           numRowsPerSlice, numSlices, mvcost, epoch, est);
    else
        hipLaunchKernelGGL((lookahead_p_kernel<uint16_t>), grid, block, 0, as_stream(stream), pairs, nPairs, stride, planeElems, widthInCU, heightInCU,
                           numRowsPerSlice, numSlices, mvcost, epoch, est);
    XH_LAUNCH_CHECK("lookahead_p_kernel");
    return X265HIP_OK;
}

extern "C" int x265hip_lookahead_bidir_batch(int depth, const x265hip_lookahead_bframe* frames, int nFrames, int64_t stride, int64_t planeElems,
                                             int widthInCU, int heightInCU, int64_t* est, void* stream)
{
    XH_CHECK_DEV();
    if (!valid_depth(depth) || nFrames < 0 || widthInCU < 1 || heightInCU < 1 || stride >= (1 << 23) || planeElems >= (1 << 24) || planeElems < 0 || !est)
        return set_error(X265HIP_EINVAL, "lookahead_bidir_batch: depth %d frames %d grid %dx%d", depth, nFrames, widthInCU, heightInCU);
    if (nFrames == 0) return X265HIP_OK;
    int e = check_hip(hipMemsetAsync(est, 0, (size_t)nFrames * 2 * sizeof(int64_t), as_stream(stream)), "lookahead memset");
    if (e) return e;
    dim3 grid((unsigned)(nFrames * heightInCU)), block(64);
    if (depth == 8)
        hipLaunchKernelGGL((lookahead_bidir_kernel<uint8_t>), grid, block, 0, as_stream(stream), frames, stride, planeElems, widthInCU, heightInCU, est);
    else
        hipLaunchKernelGGL((lookahead_bidir_kernel<uint16_t>), grid, block, 0, as_stream(stream), frames, stride, planeElems, widthInCU, heightInCU, est);
    XH_LAUNCH_CHECK("lookahead_bidir_kernel");
    return X265HIP_OK;
}

extern "C" int x265hip_lookahead_pcost_batch(const x265hip_lookahead_pair* pairs, int nPairs, int widthInCU, int heightInCU, int64_t* est, void* stream)
{
    XH_CHECK_DEV();
    if (nPairs < 0 || widthInCU < 1 || heightInCU < 1 || !est)
        return set_error(X265HIP_EINVAL, "lookahead_pcost_batch: pairs %d grid %dx%d", nPairs, widthInCU, heightInCU);
    if (nPairs == 0) return X265HIP_OK;
    int e = check_hip(hipMemsetAsync(est, 0, (size_t)nPairs * 4 * sizeof(int64_t), as_stream(stream)), "lookahead memset");
    if (e) return e;
    hipLaunchKernelGGL(lookahead_pcost_kernel, dim3((unsigned)(nPairs * heightInCU)), dim3(64), 0, as_stream(stream), pairs, widthInCU, heightInCU, est);
    XH_LAUNCH_CHECK("lookahead_pcost_kernel");
    return X265HIP_OK;
}
