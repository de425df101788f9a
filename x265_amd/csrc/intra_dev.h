// intra_dev.h — the one sample function behind every HEVC intra prediction mode (shared by intra.hip's table primitives / mode scan and by the
// intra-scan jobs of cuserve.hip).  The L-shaped neighbour line is addressed by a signed coordinate j (0 = corner, +j = top / top-right,
// -j = left / bottom-left); reference: source/common/intrapred.cpp, constants.cpp:561.
#pragma once
#include "common.h"

namespace xh {

static __device__ __constant__ const int8_t kIntraAngle[17] = { -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32 };
static __device__ __constant__ const int16_t kIntraInvAngle[8] = { 4096, 1638, 910, 630, 482, 390, 315, 256 };   // 8192 / |angle|, |angle| = 2..32

// g_intraFilterFlags[mode] & n (constants.cpp:561) as the HEVC rule
__host__ __device__ __forceinline__ bool intra_uses_filtered(int n, int mode)
{
    if (mode == 1 || n < 8) return false;
    if (mode == 0) return true;
    const int dv = mode > 26 ? mode - 26 : 26 - mode, dh = mode > 10 ? mode - 10 : 10 - mode;
    const int d = dv < dh ? dv : dh;
    return d > (n == 8 ? 7 : (n == 16 ? 1 : 0));
}

// neighbour line in the reference's array layout: nb[0] corner, nb[1..2N] top, nb[2N+1..4N] left
template <typename P>
struct GlobalLine
{
    const P* nb;
    int n2;
    __device__ __forceinline__ int at(int j) const { return (int)nb[j >= 0 ? j : n2 - j]; }
};
// neighbour line centred in LDS: c[j], j in [-2N, 2N]
struct LdsLine
{
    const uint16_t* c;
    __device__ __forceinline__ int at(int j) const { return (int)c[j]; }
};

struct AngSetup { int sgn, angle, inv; };     // sgn = +1 vertical class, -1 horizontal class
__device__ __forceinline__ AngSetup ang_setup(int mode)
{
    const bool horiz = mode < 18;
    const int rel = horiz ? 10 - mode : mode - 26;
    AngSetup a;
    a.sgn = horiz ? -1 : 1;
    a.angle = kIntraAngle[8 + rel];
    a.inv = rel < 0 ? kIntraInvAngle[-rel - 1] : 0;
    return a;
}
// sample k of the main reference line of an angular mode (k = 0 is the corner)
template <typename L>
__device__ __forceinline__ int ang_main(const L& ln, const AngSetup& a, int k)
{
    const int j = k >= 0 ? a.sgn * k : -a.sgn * ((128 - k * a.inv) >> 8);
    return ln.at(j);
}
// angular sample at (x, y) before the mode-10 / mode-26 edge gradient (spec 8.4.4.2.6; intrapred.cpp:177-203)
template <typename L>
__device__ __forceinline__ int ang_sample(const L& ln, const AngSetup& a, int x, int y)
{
    const int u = a.sgn > 0 ? x : y, v = a.sgn > 0 ? y : x;
    const int t = (v + 1) * a.angle;
    const int k = (t >> 5) + u + 1, f = t & 31;
    const int s0 = ang_main(ln, a, k);
    if (!f)
        return s0;
    const int s1 = ang_main(ln, a, k + 1);
    return ((32 - f) * s0 + f * s1 + 16) >> 5;
}
// any mode, any position; dc = the block's DC value (only read when mode == 1)
template <typename L>
__device__ __forceinline__ int intra_sample(const L& ln, int n, int log2n, int mode, int bFilter, int x, int y, int maxv, int dc)
{
    if (mode == 0)    // planar, intrapred.cpp:88-104
        return ((n - 1 - x) * ln.at(-(y + 1)) + (x + 1) * ln.at(n + 1) + (n - 1 - y) * ln.at(x + 1) + (y + 1) * ln.at(-(n + 1)) + n) >> (log2n + 1);
    if (mode == 1)    // DC + edge smoothing, intrapred.cpp:57-86
    {
        if (!bFilter || (x && y))
            return dc;
        if (x == 0 && y == 0)
            return (ln.at(1) + ln.at(-1) + 2 * dc + 2) >> 2;
        return ((y == 0 ? ln.at(x + 1) : ln.at(-(y + 1))) + 3 * dc + 2) >> 2;
    }
    const AngSetup a = ang_setup(mode);
    int v = ang_sample(ln, a, x, y);
    if (bFilter && a.angle == 0)
    {
        // pure vertical / horizontal: first column / row follows the gradient of the other arm (intrapred.cpp:146-151)
        const int along = a.sgn > 0 ? x : y, across = a.sgn > 0 ? y : x;
        if (along == 0)
        {
            v = ln.at(a.sgn) + ((ln.at(-a.sgn * (across + 1)) - ln.at(0)) >> 1);
            v = v < 0 ? 0 : (v > maxv ? maxv : v);
        }
    }
    return v;
}

} // namespace xh
