// intra.hip — HEVC intra prediction (x265 cu[].intra_pred[35] / intra_filter / intra_pred_allangs, reference
// source/common/intrapred.cpp), the lookahead's half-resolution planes (frameInitLowres, pixel.cpp:604) and its per-8x8
// intra cost estimate (LookaheadTLD::lowresIntraEstimate, encoder/slicetype.cpp:696) for gfx950.
//
// One sample function serves every mode: the L-shaped neighbour line is addressed by a signed coordinate j
// (0 = corner, +j = top / top-right, -j = left / bottom-left).  A vertical-class angular mode walks along +j, a
// horizontal-class one along -j, and the "projected" samples of negative angles come from the other arm — so the
// reference's flip-copy + transpose of horizontal modes (intrapred.cpp:112-123, :207-221) is just a sign.
//
//  * intra_pred_kernel / intra_filter_kernel: the table primitives, one lane per 4 adjacent output samples, neighbour
//    line read from global memory (jobs = one block each, or 33 modes of a block for intra_pred_allangs).
//  * lowres_init_kernel: lane = 4 adjacent lowres samples of all four planes (2:1 pavg cascade), HBM-bound.
//  * lowres_intra_kernel: **row team** like motion3.hip — four 8x8 blocks per wave, one DPP row of 16 lanes each; the line
//    (raw + [1 2 1] filtered) lives in LDS, each lane predicts its 4 samples of a mode, the 8x8 SATD runs in-register
//    (4x4 Hadamards across DPP quads) and the 16-lane sum is a row_ror all-reduce: DC, planar, the 6 coarse angles and
//    the 4 refinements never leave the wave.
#include "common.h"
#include "internal.h"
#include "tiles.h"
#include "intra_dev.h"

namespace xh {

// ---- table primitives ------------------------------------------------------------------------------------------------------
struct IntraJobs
{
    const int32_t* nbOff;      // per block
    const int32_t* nbfOff;     // per block, allangs only
    const int32_t* modes;      // per job: mode | bFilter << 8 (ignored for allangs)
    const int32_t* dstOff;     // per job (ignored for allangs)
    int allangs, bLuma;
};

template <typename P>
__global__ __launch_bounds__(256) void intra_pred_kernel(const P* __restrict__ nbBase, P* __restrict__ dstBase, int64_t dstStride, IntraJobs jb,
                                                         int n, int log2n, int depth, long long totalQuads)
{
    const int qpb = (n * n) >> 2, qpr = n >> 2;
    const int maxv = (1 << depth) - 1;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < totalQuads; q += (long long)gridDim.x * blockDim.x)
    {
        const int job = (int)(q / qpb), qi = (int)(q % qpb);
        int y = qi / qpr, x0 = (qi % qpr) * 4;
        int mode, bFilter;
        const P* nb;
        P* dst;
        int64_t ds = dstStride;
        bool transpose = false;
        if (jb.allangs)
        {
            const int b = job / 33;
            mode = 2 + job % 33;
            bFilter = jb.bLuma;
            nb = nbBase + (intra_uses_filtered(n, mode) ? jb.nbfOff[b] : jb.nbOff[b]);
            dst = dstBase + (int64_t)job * n * n;
            ds = n;
            transpose = mode < 18;                           // "don't flip buffer" (intrapred.cpp:236-251)
        }
        else
        {
            mode = jb.modes[job] & 255;
            bFilter = (jb.modes[job] >> 8) & 1;
            nb = nbBase + jb.nbOff[job];
            dst = dstBase + jb.dstOff[job];
        }
        const GlobalLine<P> ln{ nb, 2 * n };
        int dc = 0;
        if (mode == 1)
        {
            dc = n;
            for (int i = 1; i <= n; i++)
                dc += ln.at(i) + ln.at(-i);
            dc >>= log2n + 1;
        }
        int v[4];
#pragma unroll
        for (int i = 0; i < 4; i++)
            v[i] = transpose ? intra_sample(ln, n, log2n, mode, bFilter, y, x0 + i, maxv, dc)
                             : intra_sample(ln, n, log2n, mode, bFilter, x0 + i, y, maxv, dc);
        store4(dst + (int64_t)y * ds + x0, v);
    }
}

template <typename P>
__global__ __launch_bounds__(256) void intra_filter_kernel(const P* __restrict__ inBase, const int32_t* __restrict__ inOff, P* __restrict__ outBase,
                                                           const int32_t* __restrict__ outOff, int n, long long total)
{
    const int len = 4 * n + 1, n2 = 2 * n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    {
        const int job = (int)(i / len), e = (int)(i % len);
        const GlobalLine<P> ln{ inBase + inOff[job], n2 };
        const int j = e <= n2 ? e : n2 - e;                   // array index -> line coordinate
        int v;
        if (j == n2 || j == -n2)
            v = ln.at(j);                                     // the two ends are kept (intrapred.cpp:44, :54)
        else
            v = (ln.at(j - 1) + 2 * ln.at(j) + ln.at(j + 1) + 2) >> 2;
        outBase[outOff[job] + e] = (P)v;
    }
}

// ---- lookahead: half-resolution planes ---------------------------------------------------------------------------------------
template <typename P>
__global__ __launch_bounds__(256) void lowres_init_kernel(const P* __restrict__ src, int64_t ss, P* __restrict__ d0, P* __restrict__ dh,
                                                          P* __restrict__ dv, P* __restrict__ dc, int64_t ds, int width, int height)
{
    const int qpr = (width + 3) >> 2;
    const long long total = (long long)qpr * height;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    {
        const int y = (int)(i / qpr), x0 = (int)(i % qpr) * 4;
        const P* r = src + (int64_t)(2 * y) * ss + 2 * x0;
        int a[3][12];
#pragma unroll
        for (int k = 0; k < 3; k++)
        {
            load4(r + k * ss, a[k]);
            load4(r + k * ss + 4, a[k] + 4);
            a[k][8] = r[k * ss + 8];
        }
        int c[9], e[9];                                        // rounded vertical pair averages: rows 0-1 and rows 1-2
#pragma unroll
        for (int j = 0; j < 9; j++)
        {
            c[j] = (a[0][j] + a[1][j] + 1) >> 1;
            e[j] = (a[1][j] + a[2][j] + 1) >> 1;
        }
        int o0[4], oh[4], ov[4], oc[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
            o0[j] = (c[2 * j] + c[2 * j + 1] + 1) >> 1;
            oh[j] = (c[2 * j + 1] + c[2 * j + 2] + 1) >> 1;
            ov[j] = (e[2 * j] + e[2 * j + 1] + 1) >> 1;
            oc[j] = (e[2 * j + 1] + e[2 * j + 2] + 1) >> 1;
        }
        const int64_t o = (int64_t)y * ds + x0;
        if (x0 + 4 <= width)
        {
            store4(d0 + o, o0); store4(dh + o, oh); store4(dv + o, ov); store4(dc + o, oc);
        }
        else
            for (int j = 0; x0 + j < width; j++)
            {
                d0[o + j] = (P)o0[j]; dh[o + j] = (P)oh[j]; dv[o + j] = (P)ov[j]; dc[o + j] = (P)oc[j];
            }
    }
}

// ---- lookahead: intra cost of every 8x8 block of a lowres plane ----------------------------------------------------------------
template <typename P>
__global__ __launch_bounds__(256) void lowres_intra_kernel(const P* __restrict__ plane, int64_t stride, int widthInCU, int heightInCU, int depth,
                                                           int32_t* __restrict__ intraCost, uint8_t* __restrict__ intraMode)
{
    constexpr int N = 8;
    __shared__ uint16_t lines[16][2][36];                      // [block in workgroup][raw / filtered][j + 2N], j in [-16, 16]
    const int s = threadIdx.x & 15, team = threadIdx.x >> 4;
    const int cu = blockIdx.x * 16 + team;
    const int ncu = widthInCU * heightInCU;
    const bool live = cu < ncu;
    const int cuc = live ? cu : ncu - 1;                       // dead rows shadow the last block (DPP needs all lanes in step)
    const int cx = cuc % widthInCU, cy = cuc / widthInCU;
    const P* cur = plane + (int64_t)cy * N * stride + cx * N;
    const int maxv = (1 << depth) - 1;
    uint16_t* raw = &lines[team][0][16];
    uint16_t* flt = &lines[team][1][16];

    // neighbour line (slicetype.cpp:728-733): lane s carries top[s] = line[s+1] and left[s] = line[-(s+1)]; lane 0 the corner too
    raw[s + 1] = (uint16_t)cur[-stride + s];
    raw[-(s + 1)] = (uint16_t)cur[(int64_t)s * stride - 1];
    if (s == 0)
        raw[0] = (uint16_t)cur[-stride - 1];
    __builtin_amdgcn_wave_barrier();
    {
        // intraFilter<8> (intrapred.cpp:32-55)
        const int jp = s + 1;
        const int tp = jp == 16 ? raw[16] : (raw[jp - 1] + 2 * raw[jp] + raw[jp + 1] + 2) >> 2;
        const int lf = jp == 16 ? raw[-16] : (raw[-jp + 1] + 2 * raw[-jp] + raw[-jp - 1] + 2) >> 2;
        flt[jp] = (uint16_t)tp;
        flt[-jp] = (uint16_t)lf;
        if (s == 0)
            flt[0] = (uint16_t)((raw[-1] + 2 * raw[0] + raw[1] + 2) >> 2);
    }
    __builtin_amdgcn_wave_barrier();

    // this lane's 4 source samples: tile-major so the 4 rows of a 4x4 tile sit in one DPP quad
    const int t = s >> 2, r = s & 3;
    const int py = (t >> 1) * 4 + r, px = (t & 1) * 4;
    int fe[4];
    load4(cur + (int64_t)py * stride + px, fe);
    const bool hi1 = s & 1, hi2 = s & 2;
    const int dc = (row_allsum(s < 8 ? (int)raw[s + 1] : (int)raw[-(s - 8 + 1)]) + N) >> 4;

    auto cost_of = [&](int mode) -> int {
        const LdsLine ln{ intra_uses_filtered(N, mode) ? flt : raw };
        int d[4];
#pragma unroll
        for (int i = 0; i < 4; i++)
            d[i] = fe[i] - intra_sample(ln, N, 3, mode, 1, px + i, py, maxv, dc);
        const int s01 = d[0] + d[1], e01 = d[0] - d[1], s23 = d[2] + d[3], e23 = d[2] - d[3];
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
        // satd 8x8 = four 4x4 tiles, each >> 1 (pixel.cpp:210-297); every tile sum is even, so one shift of the total
        return row_allsum(iabs(m[0]) + iabs(m[1]) + iabs(m[2]) + iabs(m[3])) >> 1;
    };

    int best = cost_of(1), bestMode = 1;                       // DC first, planar only if strictly cheaper (slicetype.cpp:741-747)
    {
        const int c = cost_of(0);
        if (c < best) { best = c; bestMode = 0; }
    }
    int abest = 0x7fffffff, amode = 4;
    for (int m = 5; m < 35; m += 5)
    {
        const int c = cost_of(m);
        if (c < abest) { abest = c; amode = m; }
    }
#pragma unroll 1
    for (int dist = 2; dist >= 1; dist--)
    {
        const int lo = amode - dist, hi = amode + dist;
        const int c0 = cost_of(lo);
        if (c0 < abest) { abest = c0; amode = lo; }
        const int c1 = cost_of(hi);
        if (c1 < abest) { abest = c1; amode = hi; }
    }
    if (abest < best) { best = abest; bestMode = amode; }
    best += 5 * (1 << (2 * (depth - 8))) + 4;                  // intraPenalty = 5 * (int)x265_lambda_tab[X265_LOOKAHEAD_QP] + lowresPenalty
    if (live && s == 0)
    {
        intraCost[cu] = best;
        intraMode[cu] = (uint8_t)bestMode;
    }
}

// rowSatds[0][0][cy] = sum of the row's costs; costEst = sum over the non-edge blocks (slicetype.cpp:777-800, AQ off).  One wave per row.
__global__ __launch_bounds__(64) void lowres_intra_sums_kernel(const int32_t* __restrict__ intraCost, int widthInCU, int heightInCU,
                                                               int32_t* __restrict__ rowSatd, int32_t* __restrict__ costEst)
{
    const bool all = widthInCU <= 2 || heightInCU <= 2;
    const int cy = blockIdx.x;
    int row = 0, in = 0;
    for (int cx = threadIdx.x; cx < widthInCU; cx += 64)
    {
        const int c = intraCost[cy * widthInCU + cx];
        row += c;
        if (all || (cx > 0 && cx < widthInCU - 1 && cy > 0 && cy < heightInCU - 1))
            in += c;
    }
    for (int o = 32; o; o >>= 1)
    {
        row += __shfl_xor(row, o);
        in += __shfl_xor(in, o);
    }
    if (threadIdx.x == 0)
    {
        rowSatd[cy] = row;
        if (in) atomicAdd(costEst, in);
    }
}

// ---- intra mode scan: sa8d(source block, prediction of mode m) for all 35 modes of a block -----------------------------------------
// The distortion half of Search::checkIntraInInter / estIntraPredQT (search.cpp:1344-1420, :1568-1640): DC (edge-smoothed when N <= 16),
// planar (filtered line when N >= 8), the 33 angles each from the raw or filtered line per g_intraFilterFlags; cost = cu[].sa8d (satd_4x4 for
// 4x4, sa8d_8x8, one rounding per 16x16 above).  The reference compares horizontal modes transposed (allangs buffer layout, :1388-1391);
// a 2-D Hadamard magnitude sum is transpose-invariant, so every mode is measured upright here.  Lane = one 4x4 tile whose 16 predicted
// samples are generated in registers — no prediction buffer exists; an 8x8 Hadamard spans a DPP quad as in pixel.hip.
template <typename P>
__global__ __launch_bounds__(256) void intra_scan_kernel(const P* __restrict__ lines, const int32_t* __restrict__ lineOff, const int32_t* __restrict__ filtOff,
                                                         const P* __restrict__ fenc, int64_t fs, const int32_t* __restrict__ fencOff,
                                                         int n, int log2n, int depth, long long jobs, int32_t* __restrict__ costs)
{
    const int lane = threadIdx.x & 63;
    const int tiles = (n >> 2) * (n >> 2);
    const int T = tiles >= 64 ? 64 : tiles;
    const int jpw = 64 / T, sub = lane & (T - 1);
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long job = wave * jpw + lane / T;
    const bool ok = job < jobs;
    const long long jc = ok ? job : jobs - 1;
    const int b = (int)(jc / 35), mode = (int)(jc % 35);
    const int maxv = (1 << depth) - 1;
    const int bFilter = n <= 16;
    const GlobalLine<P> ln{ lines + (intra_uses_filtered(n, mode) ? filtOff[b] : lineOff[b]), 2 * n };
    int dc = 0;
    if (mode == 1)
    {
        dc = n;
        for (int i = 1; i <= n; i++)
            dc += ln.at(i) + ln.at(-i);
        dc >>= log2n + 1;
    }
    // tile order as in pixel.hip: 16x16 blocks in raster order, inside them 8x8 blocks, inside them the four 4x4 quadrants (one DPP quad)
    const int n16x = n >= 16 ? (n >> 4) : 1;
    const int b16 = sub >> 4, b8 = (sub >> 2) & 3, q = sub & 3;
    int x, y;
    if (n >= 8)
    {
        x = (b16 % n16x) * 16 + (b8 & 1) * 8 + (q & 1) * 4;
        y = (b16 / n16x) * 16 + (b8 >> 1) * 8 + (q >> 1) * 4;
        if (n == 8) { x = (q & 1) * 4; y = (q >> 1) * 4; }
    }
    else
        x = y = 0;
    const P* f = fenc + fencOff[b] + (int64_t)y * fs + x;
    int m[16];
#pragma unroll
    for (int yy = 0; yy < 4; yy++)
    {
        int v[4];
        load4(f + yy * fs, v);
#pragma unroll
        for (int xx = 0; xx < 4; xx++)
            m[4 * yy + xx] = v[xx] - intra_sample(ln, n, log2n, mode, bFilter, x + xx, y + yy, maxv, dc);
    }
    hadamard4x4(m);
    int acc;
    if (n == 4)
        acc = abs_sum16(m) >> 1;                              // satd_4x4 (pixel.cpp:210-237), cu[BLOCK_4x4].sa8d alias
    else
    {
        const int raw8 = quad_sa8d_raw(m, lane);
        if (n >= 16)
        {
            const int s16 = group_sum((lane & 3) == 0 ? raw8 : 0, 16);
            acc = (lane & 15) == 0 ? ((s16 + 2) >> 2) : 0;     // sa8d_16x16: four raw 8x8, one rounding (pixel.cpp:341-350)
        }
        else
            acc = (lane & 3) == 0 ? ((raw8 + 2) >> 2) : 0;     // sa8d_8x8 (pixel.cpp:336)
        acc = group_sum(acc, T);
    }
    if (ok && sub == 0)
        costs[job] = acc;
}

static bool valid_intra_size(int n) { return n == 4 || n == 8 || n == 16 || n == 32; }
static int log2_of(int n) { return n == 4 ? 2 : n == 8 ? 3 : n == 16 ? 4 : 5; }

} // namespace xh

using namespace xh;

extern "C" {

int x265hip_intra_pred_batch(int depth, int n, const void* nb, const int32_t* nbOff, const int32_t* modes, void* dst, const int32_t* dstOff,
                             int64_t dstStride, int count, void* stream)
{
    XH_CHECK_DEV();
    if (!valid_depth(depth) || !valid_intra_size(n) || count < 0)
        return set_error(X265HIP_EINVAL, "intra_pred_batch: depth %d size %d count %d", depth, n, count);
    if (count == 0) return X265HIP_OK;
    const long long quads = (long long)count * n * n / 4;
    IntraJobs jb{ nbOff, nullptr, modes, dstOff, 0, 0 };
    dim3 grid(grid_for((quads + 255) / 256)), block(256);
    if (depth == 8)
        hipLaunchKernelGGL((intra_pred_kernel<uint8_t>), grid, block, 0, as_stream(stream), (const uint8_t*)nb, (uint8_t*)dst, dstStride, jb, n, log2_of(n), depth, quads);
    else
        hipLaunchKernelGGL((intra_pred_kernel<uint16_t>), grid, block, 0, as_stream(stream), (const uint16_t*)nb, (uint16_t*)dst, dstStride, jb, n, log2_of(n), depth, quads);
    XH_LAUNCH_CHECK("intra_pred_kernel");
    return X265HIP_OK;
}

int x265hip_intra_allangs_batch(int depth, int n, const void* nb, const int32_t* nbOff, const int32_t* nbfOff, int bLuma, void* dest, int count, void* stream)
{
    XH_CHECK_DEV();
    if (!valid_depth(depth) || !valid_intra_size(n) || count < 0)
        return set_error(X265HIP_EINVAL, "intra_allangs_batch: depth %d size %d count %d", depth, n, count);
    if (count == 0) return X265HIP_OK;
    const long long quads = (long long)count * 33 * n * n / 4;
    IntraJobs jb{ nbOff, nbfOff, nullptr, nullptr, 1, bLuma ? 1 : 0 };
    dim3 grid(grid_for((quads + 255) / 256)), block(256);
    if (depth == 8)
        hipLaunchKernelGGL((intra_pred_kernel<uint8_t>), grid, block, 0, as_stream(stream), (const uint8_t*)nb, (uint8_t*)dest, (int64_t)n, jb, n, log2_of(n), depth, quads);
    else
        hipLaunchKernelGGL((intra_pred_kernel<uint16_t>), grid, block, 0, as_stream(stream), (const uint16_t*)nb, (uint16_t*)dest, (int64_t)n, jb, n, log2_of(n), depth, quads);
    XH_LAUNCH_CHECK("intra_pred_kernel(allangs)");
    return X265HIP_OK;
}

int x265hip_intra_filter_batch(int depth, int n, const void* in, const int32_t* inOff, void* out, const int32_t* outOff, int count, void* stream)
{
    XH_CHECK_DEV();
    if (!valid_depth(depth) || !valid_intra_size(n) || count < 0)
        return set_error(X265HIP_EINVAL, "intra_filter_batch: depth %d size %d count %d", depth, n, count);
    if (count == 0) return X265HIP_OK;
    const long long total = (long long)count * (4 * n + 1);
    dim3 grid(grid_for((total + 255) / 256)), block(256);
    if (depth == 8)
        hipLaunchKernelGGL((intra_filter_kernel<uint8_t>), grid, block, 0, as_stream(stream), (const uint8_t*)in, inOff, (uint8_t*)out, outOff, n, total);
    else
        hipLaunchKernelGGL((intra_filter_kernel<uint16_t>), grid, block, 0, as_stream(stream), (const uint16_t*)in, inOff, (uint16_t*)out, outOff, n, total);
    XH_LAUNCH_CHECK("intra_filter_kernel");
    return X265HIP_OK;
}

int x265hip_intra_scan_batch(int depth, int n, const void* lines, const int32_t* lineOff, const int32_t* filteredOff, const void* fenc,
                             int64_t fencStride, const int32_t* fencOff, int count, int32_t* costs, void* stream)
{
    XH_CHECK_DEV();
    if (!valid_depth(depth) || !valid_intra_size(n) || count < 0)
        return set_error(X265HIP_EINVAL, "intra_scan_batch: depth %d size %d count %d", depth, n, count);
    if (count == 0) return X265HIP_OK;
    const long long jobs = (long long)count * 35;
    const int tiles = (n / 4) * (n / 4), T = tiles >= 64 ? 64 : tiles;
    const long long waves = (jobs + 64 / T - 1) / (64 / T);
    dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    if (depth == 8)
        hipLaunchKernelGGL((intra_scan_kernel<uint8_t>), grid, block, 0, as_stream(stream), (const uint8_t*)lines, lineOff, filteredOff, (const uint8_t*)fenc,
                           fencStride, fencOff, n, log2_of(n), depth, jobs, costs);
    else
        hipLaunchKernelGGL((intra_scan_kernel<uint16_t>), grid, block, 0, as_stream(stream), (const uint16_t*)lines, lineOff, filteredOff, (const uint16_t*)fenc,
                           fencStride, fencOff, n, log2_of(n), depth, jobs, costs);
    XH_LAUNCH_CHECK("intra_scan_kernel");
    return X265HIP_OK;
}

int x265hip_frame_init_lowres(int depth, const void* src, int64_t srcStride, void* dst0, void* dstH, void* dstV, void* dstC, int64_t dstStride,
                              int width, int height, void* stream)
{
    XH_CHECK_DEV();
    if (!valid_depth(depth) || width < 1 || height < 1)
        return set_error(X265HIP_EINVAL, "frame_init_lowres: depth %d size %dx%d", depth, width, height);
    const long long total = (long long)((width + 3) / 4) * height;
    dim3 grid(grid_for((total + 255) / 256)), block(256);
    if (depth == 8)
        hipLaunchKernelGGL((lowres_init_kernel<uint8_t>), grid, block, 0, as_stream(stream), (const uint8_t*)src, srcStride, (uint8_t*)dst0, (uint8_t*)dstH,
                           (uint8_t*)dstV, (uint8_t*)dstC, dstStride, width, height);
    else
        hipLaunchKernelGGL((lowres_init_kernel<uint16_t>), grid, block, 0, as_stream(stream), (const uint16_t*)src, srcStride, (uint16_t*)dst0, (uint16_t*)dstH,
                           (uint16_t*)dstV, (uint16_t*)dstC, dstStride, width, height);
    XH_LAUNCH_CHECK("lowres_init_kernel");
    return X265HIP_OK;
}

int x265hip_lowres_init(int depth, const void* src, int64_t srcStride, void* const planes[4], int64_t dstStride, int width, int height,
                        int marginX, int marginY, void* stream)
{
    if (!planes || marginX < 0 || marginY < 0)
        return set_error(X265HIP_EINVAL, "lowres_init: planes %p margins %d,%d", (const void*)planes, marginX, marginY);
    int e = x265hip_frame_init_lowres(depth, src, srcStride, planes[0], planes[1], planes[2], planes[3], dstStride, width, height, stream);
    if (e) return e;
    const int64_t st[4] = { dstStride, dstStride, dstStride, dstStride };
    const int w[4] = { width, width, width, width }, h[4] = { height, height, height, height };
    const int mx[4] = { marginX, marginX, marginX, marginX }, my[4] = { marginY, marginY, marginY, marginY };
    return extend_border_planes(depth, 4, planes, st, w, h, mx, my, as_stream(stream));
}

int x265hip_lowres_intra_estimate(int depth, const void* plane, int64_t stride, int widthInCU, int heightInCU, int32_t* intraCost, uint8_t* intraMode,
                                  int32_t* rowSatd, int32_t* costEst, void* stream)
{
    XH_CHECK_DEV();
    if (!valid_depth(depth) || widthInCU < 1 || heightInCU < 1 || !intraCost || !intraMode)
        return set_error(X265HIP_EINVAL, "lowres_intra_estimate: depth %d grid %dx%d", depth, widthInCU, heightInCU);
    const int ncu = widthInCU * heightInCU;
    dim3 grid((ncu + 15) / 16), block(256);
    if (depth == 8)
        hipLaunchKernelGGL((lowres_intra_kernel<uint8_t>), grid, block, 0, as_stream(stream), (const uint8_t*)plane, stride, widthInCU, heightInCU, depth, intraCost, intraMode);
    else
        hipLaunchKernelGGL((lowres_intra_kernel<uint16_t>), grid, block, 0, as_stream(stream), (const uint16_t*)plane, stride, widthInCU, heightInCU, depth, intraCost, intraMode);
    XH_LAUNCH_CHECK("lowres_intra_kernel");
    if (rowSatd && costEst)
    {
        int e = check_hip(hipMemsetAsync(costEst, 0, sizeof(int32_t), as_stream(stream)), "lowres_intra memset");
        if (e) return e;
        hipLaunchKernelGGL(lowres_intra_sums_kernel, dim3(heightInCU), dim3(64), 0, as_stream(stream), intraCost, widthInCU, heightInCU, rowSatd, costEst);
        XH_LAUNCH_CHECK("lowres_intra_sums_kernel");
    }
    return X265HIP_OK;
}

} // extern "C"
