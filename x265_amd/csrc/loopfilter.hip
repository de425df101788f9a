// loopfilter.hip — the in-loop filter primitives: deblocking edge filters pelFilterLumaStrong / pelFilterChroma
// (reference source/common/loopfilter.cpp:139-185), SAO offset application saoCuOrgE0 / E1 / E1_2Rows / E2 / E3 / B0 and calSign
// (loopfilter.cpp:38-137), and the SAO statistics saoCuStatsBO / E0 / E1 / E2 / E3 (source/encoder/sao.cpp:1762-1925); callers
// Deblock::edgeFilterLuma / Chroma (deblock.cpp), SAO::applyPixelOffsets and SAO::calcSaoStatsCTU (sao.cpp).  One job = one call of the
// reference's primitive, many jobs per launch.
//
// The reference threads a sign buffer from pixel to pixel and row to row (signLeft = -signRight, upBuff1[x] = -signDown); every entry of
// it is the sign of a difference of two UNMODIFIED reconstruction samples, so on the device each sample's class comes straight from the
// picture: only the first row (or column) of a call takes its signs from the caller's buffer, and the buffers the call leaves behind are
// the signs of its last row(s).  Offset application is in place with neighbours read across the samples being written, so a job is one
// workgroup: load, barrier, write.  Statistics are per-class sums over up to 64x64 samples: wave-level partial sums, then LDS atomics.
#include "common.h"

namespace xh {

__device__ __forceinline__ int sgn(int v) { return (v > 0) - (v < 0); }
__device__ __forceinline__ int lf_clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
template <typename P> __device__ __forceinline__ P clip_pix(int v, int maxv) { return (P)(v < 0 ? 0 : (v > maxv ? maxv : v)); }

// ---- deblocking: one thread per line, four lines per job
template <typename P>
__global__ __launch_bounds__(256) void pel_filter_luma_strong_kernel(P* __restrict__ plane, const int64_t* __restrict__ off, int64_t srcStep, int64_t offset,
                                                                     const int32_t* __restrict__ tcPA, const int32_t* __restrict__ tcQA, int n)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x, j = t >> 2;
    if (j >= n)
        return;
    P* src = plane + off[j] + (t & 3) * srcStep;
    const int tcP = tcPA[j], tcQ = tcQA[j];
    const int m4 = src[0], m3 = src[-offset], m5 = src[offset], m2 = src[-offset * 2], m6 = src[offset * 2], m1 = src[-offset * 3], m7 = src[offset * 3],
              m0 = src[-offset * 4];
    src[-offset * 3] = (P)(lf_clip3(-tcP, tcP, ((2 * m0 + 3 * m1 + m2 + m3 + m4 + 4) >> 3) - m1) + m1);
    src[-offset * 2] = (P)(lf_clip3(-tcP, tcP, ((m1 + m2 + m3 + m4 + 2) >> 2) - m2) + m2);
    src[-offset] = (P)(lf_clip3(-tcP, tcP, ((m1 + 2 * m2 + 2 * m3 + 2 * m4 + m5 + 4) >> 3) - m3) + m3);
    src[0] = (P)(lf_clip3(-tcQ, tcQ, ((m2 + 2 * m3 + 2 * m4 + 2 * m5 + m6 + 4) >> 3) - m4) + m4);
    src[offset] = (P)(lf_clip3(-tcQ, tcQ, ((m3 + m4 + m5 + m6 + 2) >> 2) - m5) + m5);
    src[offset * 2] = (P)(lf_clip3(-tcQ, tcQ, ((m3 + m4 + m5 + 3 * m6 + 2 * m7 + 4) >> 3) - m6) + m6);
}

template <typename P>
__global__ __launch_bounds__(256) void pel_filter_chroma_kernel(P* __restrict__ plane, const int64_t* __restrict__ off, int64_t srcStep, int64_t offset,
                                                                const int32_t* __restrict__ tcA, const int32_t* __restrict__ maskPA,
                                                                const int32_t* __restrict__ maskQA, int n, int maxv)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x, j = t >> 2;
    if (j >= n)
        return;
    P* src = plane + off[j] + (t & 3) * srcStep;
    const int tc = tcA[j];
    const int m4 = src[0], m3 = src[-offset], m5 = src[offset], m2 = src[-offset * 2];
    const int delta = lf_clip3(-tc, tc, (((m4 - m3) * 4) + m2 - m5 + 4) >> 3);
    src[-offset] = clip_pix<P>(m3 + (delta & maskPA[j]), maxv);
    src[0] = clip_pix<P>(m4 - (delta & maskQA[j]), maxv);
}

template <typename P>
__global__ __launch_bounds__(256) void sao_sign_kernel(int8_t* __restrict__ dst, const P* __restrict__ a, const P* __restrict__ b, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        dst[i] = (int8_t)sgn((int)a[i] - (int)b[i]);
}

// ---- Deblock::edgeFilterLuma / edgeFilterChroma (deblock.cpp:317-513): the whole per-unit decision (boundary strength and QPs in, beta / tc
// from Table 8-12, the d / strong / side decisions from lines 0 and 3, then the strong or the normal filter) — one thread per line, four
// threads per 4-line unit; every thread reads lines 0 and 3 of its unit for the decisions, so no exchange is needed.  All vertical edges
// of a picture are independent of each other (8-sample grid, at most 3 samples modified either side), and so are all horizontal ones:
// a frame is two launches.
__device__ __forceinline__ int db_beta(int q) { return q < 16 ? 0 : (q < 29 ? q - 10 : 2 * q - 38); }
__device__ __forceinline__ int db_tc(int q)
{
    // tc' of Table 8-12 for Q = 18 .. 53 (zero below)
    const uint8_t tail[36] = { 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24 };
    return q < 18 ? 0 : tail[q - 18];
}
__device__ __forceinline__ int db_chroma_qp(int qp)
{
    const uint8_t map[14] = { 29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37 };           // Table 8-10, qPi 30..43
    return qp < 30 ? qp : (qp > 43 ? qp - 6 : map[qp - 30]);
}

template <typename P>
__global__ __launch_bounds__(256) void deblock_luma_kernel(P* __restrict__ plane, int64_t stride, int dir, const int32_t* __restrict__ xy,
                                                           const uint8_t* __restrict__ bsA, const int8_t* __restrict__ qpPA, const int8_t* __restrict__ qpQA,
                                                           const uint8_t* __restrict__ bypass, int betaOffsetDiv2, int tcOffsetDiv2, int n, int depth)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x, j = t >> 2, line = t & 3;
    if (j >= n)
        return;
    const int bs = bsA[j];
    if (!bs)
        return;
    int maskP = -1, maskQ = -1;
    if (bypass)
    {
        maskP = (int)bypass[2 * j] - 1;
        maskQ = (int)bypass[2 * j + 1] - 1;
        if (!(maskP | maskQ))
            return;
    }
    const int64_t srcStep = dir == 0 ? stride : 1, offset = dir == 0 ? 1 : stride;
    P* u = plane + (int64_t)xy[2 * j + 1] * stride + xy[2 * j];
    const int qp = ((int)qpPA[j] + (int)qpQA[j] + 1) >> 1, shift = depth - 8, maxv = (1 << depth) - 1;
    const int beta = db_beta(lf_clip3(0, 51, qp + 2 * betaOffsetDiv2)) << shift;
    auto dP = [&](const P* s) { return abs((int)s[-offset * 3] - 2 * (int)s[-offset * 2] + (int)s[-offset]); };
    auto dQ = [&](const P* s) { return abs((int)s[0] - 2 * (int)s[offset] + (int)s[offset * 2]); };
    const P* l0 = u;
    const P* l3 = u + srcStep * 3;
    const int dp0 = dP(l0), dq0 = dQ(l0), dp3 = dP(l3), dq3 = dQ(l3);
    const int d0 = dp0 + dq0, d3 = dp3 + dq3;
    if (d0 + d3 >= beta)
        return;
    const int tc = db_tc(lf_clip3(0, 53, qp + 2 * (bs - 1) + 2 * tcOffsetDiv2)) << shift;
    auto strong = [&](const P* s) {
        return (abs((int)s[-offset * 4] - (int)s[-offset]) + abs((int)s[offset * 3] - (int)s[0]) < (beta >> 3)) &&
               (abs((int)s[-offset] - (int)s[0]) < ((tc * 5 + 1) >> 1));
    };
    P* src = u + srcStep * line;
    const int m4 = src[0], m3 = src[-offset], m5 = src[offset], m2 = src[-offset * 2], m6 = src[offset * 2], m1 = src[-offset * 3];
    if (2 * d0 < (beta >> 2) && 2 * d3 < (beta >> 2) && strong(l0) && strong(l3))
    {
        const int m7 = src[offset * 3], m0 = src[-offset * 4], tcP = (2 * tc) & maskP, tcQ = (2 * tc) & maskQ;
        // the decisions above read lines 0 and 3, which their owner threads now modify: the four threads of a unit sit in one wave and
        // took the same branches, so every such load was issued before any of these stores (one instruction stream, in-order memory issue)
        src[-offset * 3] = (P)(lf_clip3(-tcP, tcP, ((2 * m0 + 3 * m1 + m2 + m3 + m4 + 4) >> 3) - m1) + m1);
        src[-offset * 2] = (P)(lf_clip3(-tcP, tcP, ((m1 + m2 + m3 + m4 + 2) >> 2) - m2) + m2);
        src[-offset] = (P)(lf_clip3(-tcP, tcP, ((m1 + 2 * m2 + 2 * m3 + 2 * m4 + m5 + 4) >> 3) - m3) + m3);
        src[0] = (P)(lf_clip3(-tcQ, tcQ, ((m2 + 2 * m3 + 2 * m4 + 2 * m5 + m6 + 4) >> 3) - m4) + m4);
        src[offset] = (P)(lf_clip3(-tcQ, tcQ, ((m3 + m4 + m5 + m6 + 2) >> 2) - m5) + m5);
        src[offset * 2] = (P)(lf_clip3(-tcQ, tcQ, ((m3 + m4 + m5 + 3 * m6 + 2 * m7 + 4) >> 3) - m6) + m6);
        return;
    }
    // pelFilterLuma (deblock.cpp:276-314)
    const int side = (beta + (beta >> 1)) >> 3;
    const int maskP1 = (dp0 + dp3 < side ? -1 : 0) & maskP, maskQ1 = (dq0 + dq3 < side ? -1 : 0) & maskQ;
    int delta = (9 * (m4 - m3) - 3 * (m5 - m2) + 8) >> 4;
    if (abs(delta) < tc * 10)
    {
        const int tc2 = tc >> 1;
        delta = lf_clip3(-tc, tc, delta);
        src[-offset] = clip_pix<P>(m3 + (delta & maskP), maxv);
        src[0] = clip_pix<P>(m4 - (delta & maskQ), maxv);
        if (maskP1)
            src[-offset * 2] = clip_pix<P>(m2 + lf_clip3(-tc2, tc2, ((((m1 + m3 + 1) >> 1) - m2 + delta) >> 1)), maxv);
        if (maskQ1)
            src[offset] = clip_pix<P>(m5 + lf_clip3(-tc2, tc2, ((((m6 + m4 + 1) >> 1) - m5 - delta) >> 1)), maxv);
    }
}

template <typename P>
__global__ __launch_bounds__(256) void deblock_chroma_kernel(P* __restrict__ cb, P* __restrict__ cr, int64_t stride, int dir, const int32_t* __restrict__ xy,
                                                             const uint8_t* __restrict__ bsA, const int8_t* __restrict__ qpPA,
                                                             const int8_t* __restrict__ qpQA, const uint8_t* __restrict__ bypass, int tcOffsetDiv2,
                                                             int cbQpOffset, int crQpOffset, int n, int depth)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x, j = t >> 3, line = t & 3, c = (t >> 2) & 1;
    if (j >= n || bsA[j] <= 1)
        return;
    int maskP = -1, maskQ = -1;
    if (bypass)
    {
        maskP = bypass[2 * j] ? 0 : -1;
        maskQ = bypass[2 * j + 1] ? 0 : -1;
        if (!(maskP | maskQ))
            return;
    }
    const int64_t srcStep = dir == 0 ? stride : 1, offset = dir == 0 ? 1 : stride;
    P* src = (c ? cr : cb) + (int64_t)xy[2 * j + 1] * stride + xy[2 * j] + srcStep * line;
    const int qpA = ((int)qpPA[j] + (int)qpQA[j] + 1) >> 1;
    const int qp = db_chroma_qp(qpA + (c ? crQpOffset : cbQpOffset));
    const int tc = db_tc(lf_clip3(0, 53, qp + 2 + 2 * tcOffsetDiv2)) << (depth - 8), maxv = (1 << depth) - 1;
    const int m4 = src[0], m3 = src[-offset], m5 = src[offset], m2 = src[-offset * 2];
    const int delta = lf_clip3(-tc, tc, (((m4 - m3) * 4) + m2 - m5 + 4) >> 3);
    src[-offset] = clip_pix<P>(m3 + (delta & maskP), maxv);
    src[0] = clip_pix<P>(m4 - (delta & maskQ), maxv);
}

// ---- SAO offset application: one workgroup per job
enum { SAO_E0 = 0, SAO_E1 = 1, SAO_E1_2ROWS = 2, SAO_E2 = 3, SAO_E3 = 4, SAO_B0 = 5 };

template <typename P>
__global__ __launch_bounds__(256) void sao_apply_kernel(int kind, P* __restrict__ plane, int64_t stride, int8_t* __restrict__ aux,
                                                        const x265hip_sao_job* __restrict__ jobs, int maxv, int boShift)
{
    const x265hip_sao_job jb = jobs[blockIdx.x];
    P* rec = plane + jb.recOff;
    const int t = threadIdx.x;
    if (kind == SAO_B0)
    {
        // band offset: every sample on its own (offsets = the 32-entry band table)
        for (int i = t; i < jb.width * jb.height; i += blockDim.x)
        {
            const int y = i / jb.width, x = i - y * jb.width;
            const int v = rec[(int64_t)y * stride + x];
            rec[(int64_t)y * stride + x] = clip_pix<P>(v + jb.offsets[v >> boShift], maxv);
        }
        return;
    }
    const int rows = (kind == SAO_E0 || kind == SAO_E1_2ROWS) ? 2 : 1;
    const int x0 = kind == SAO_E3 ? jb.startX + 1 : 0, x1 = jb.width;                                   // E3: width holds endX
    int outv[2] = { 0, 0 }, auxv[2] = { 0, 0 };
    bool act[2] = { false, false };
    // a thread owns column x0 + t of each row.  The reference calls these per CTU (width <= 64); one pass of the 256 threads covers that,
    // and the loop beyond is only correct for the kinds that do not read a left neighbour (everything but E0)
    for (int xb = x0; xb < x1; xb += blockDim.x)
    {
        const int x = xb + t;
#pragma unroll
        for (int y = 0; y < 2; y++)
        {
            act[y] = y < rows && x < x1;
            if (!act[y])
                continue;
            const P* r = rec + (int64_t)y * stride;
            const int c = r[x];
            int s0, s1;                                     // sign towards the "next" neighbour, sign from the "previous" one
            if (kind == SAO_E0)
            {
                s0 = sgn(c - (int)r[x + 1]);
                s1 = x == 0 ? jb.signLeft[y] : sgn(c - (int)r[x - 1]);
            }
            else if (kind == SAO_E1 || kind == SAO_E1_2ROWS)
            {
                s0 = sgn(c - (int)r[x + stride]);
                s1 = y == 0 ? aux[jb.aux0 + x] : sgn(c - (int)r[x - stride]);
            }
            else if (kind == SAO_E2)
            {
                s0 = sgn(c - (int)r[x + stride + 1]);
                s1 = aux[jb.aux1 + x];
            }
            else
            {
                s0 = sgn(c - (int)r[x + stride]);
                s1 = aux[jb.aux0 + x];
            }
            outv[y] = (int)clip_pix<P>(c + jb.offsets[s0 + s1 + 2], maxv);
            auxv[y] = -s0;
        }
        __syncthreads();                                    // every neighbour and every sign buffer entry has been read
#pragma unroll
        for (int y = 0; y < 2; y++)
        {
            if (!act[y])
                continue;
            rec[(int64_t)y * stride + x] = (P)outv[y];
            if ((kind == SAO_E1 || kind == SAO_E1_2ROWS) && y == rows - 1) aux[jb.aux0 + x] = (int8_t)auxv[y];
            if (kind == SAO_E2) aux[jb.aux0 + x + 1] = (int8_t)auxv[y];
            if (kind == SAO_E3) aux[jb.aux0 + x - 1] = (int8_t)auxv[y];
        }
        __syncthreads();
    }
}

// ---- SAO statistics: one workgroup per job (one CTU call of the reference)
enum { SAO_ST_BO = 0, SAO_ST_E0 = 1, SAO_ST_E1 = 2, SAO_ST_E2 = 3, SAO_ST_E3 = 4 };

template <typename P>
__global__ __launch_bounds__(256) void sao_stats_kernel(int kind, const int16_t* __restrict__ diffBase, const P* __restrict__ plane, int64_t stride,
                                                        int8_t* __restrict__ aux, const x265hip_sao_stats_job* __restrict__ jobs,
                                                        int32_t* __restrict__ stats, int32_t* __restrict__ count, int boShift)
{
    __shared__ int32_t sSum[32], sCnt[32];
    const x265hip_sao_stats_job jb = jobs[blockIdx.x];
    const P* rec = plane + jb.recOff;
    const int16_t* diff = diffBase + jb.diffOff;
    const int t = threadIdx.x, endX = jb.endX, endY = jb.endY;
    if (t < 32) { sSum[t] = 0; sCnt[t] = 0; }
    __syncthreads();
    const int eo[5] = { 1, 2, 0, 3, 4 };                   // SAO::s_eoTable (sao.cpp:65)
    for (int i = t; i < endX * endY; i += blockDim.x)
    {
        const int y = i / endX, x = i - y * endX;
        const P* r = rec + (int64_t)y * stride;
        const int c = r[x], d = diff[y * 64 + x];
        int cls;
        if (kind == SAO_ST_BO)
            cls = c >> boShift;
        else
        {
            int s0, s1;
            if (kind == SAO_ST_E0)
            {
                s0 = sgn(c - (int)r[x + 1]);
                s1 = sgn(c - (int)r[x - 1]);
            }
            else if (kind == SAO_ST_E1)
            {
                s0 = sgn(c - (int)r[x + stride]);
                s1 = y == 0 ? aux[jb.aux0 + x] : sgn(c - (int)r[x - stride]);
            }
            else if (kind == SAO_ST_E2)
            {
                s0 = sgn(c - (int)r[x + stride + 1]);
                s1 = y == 0 ? aux[jb.aux0 + x] : sgn(c - (int)r[x - stride - 1]);
            }
            else
            {
                s0 = sgn(c - (int)r[x + stride - 1]);
                s1 = y == 0 ? aux[jb.aux0 + x] : sgn(c - (int)r[x - stride + 1]);
            }
            cls = eo[s0 + s1 + 2];
        }
        atomicAdd(&sSum[cls], d);
        atomicAdd(&sCnt[cls], 1);
    }
    __syncthreads();                                        // all reads of the caller's first-row signs are done
    // the sign buffers the reference leaves behind
    if (kind == SAO_ST_E1)
    {
        for (int x = t; x < endX; x += blockDim.x)
        {
            const P* r = rec + (int64_t)(endY - 1) * stride;
            aux[jb.aux0 + x] = (int8_t)(-sgn((int)r[x] - (int)r[x + stride]));
        }
    }
    else if (kind == SAO_ST_E2)
    {
        // row y writes entries [0 .. endX] of the buffer that is "upBufft" at that row; the two buffers swap every row (sao.cpp:1865-1876)
        for (int k = 0; k < 2; k++)
        {
            const int y = endY - 1 - k;
            if (y < 0)
                break;
            const int64_t dst = (y & 1) ? jb.aux0 : jb.aux1;           // row 0 writes upBufft (aux1), row 1 upBuff1 (aux0), …
            const P* r = rec + (int64_t)y * stride;
            for (int x = t; x <= endX; x += blockDim.x)
                aux[dst + x] = (int8_t)(x == 0 ? sgn((int)r[stride] - (int)r[-1]) : -sgn((int)r[x - 1] - (int)r[x - 1 + stride + 1]));
        }
    }
    else if (kind == SAO_ST_E3)
    {
        const P* r = rec + (int64_t)(endY - 1) * stride;
        for (int x = t; x <= endX; x += blockDim.x)
        {
            // upBuff1[x - 1] = -signDown(x) for x in [0, endX), then upBuff1[endX - 1] = sign(rec[endX - 1 + stride] - rec[endX]) (:1909-1914)
            if (x < endX)
                aux[jb.aux0 + x - 1] = (int8_t)(-sgn((int)r[x] - (int)r[x + stride - 1]));
            if (x == endX)
                aux[jb.aux0 + endX - 1] = (int8_t)sgn((int)r[endX - 1 + stride] - (int)r[endX]);
        }
    }
    const int ncls = kind == SAO_ST_BO ? 32 : 5;
    if (t < ncls)
    {
        stats[(int64_t)blockIdx.x * 32 + t] += sSum[t];
        count[(int64_t)blockIdx.x * 32 + t] += sCnt[t];
    }
}

} // namespace xh

using namespace xh;

extern "C" int x265hip_pel_filter_luma_strong_batch(int depth, void* plane, const int64_t* off, int64_t srcStep, int64_t offset, const int32_t* tcP,
                                                    const int32_t* tcQ, int n, void* stream)
{
    XH_CHECK_DEV();
    if (!valid_depth(depth) || n < 0)
        return set_error(X265HIP_EINVAL, "pel_filter_luma_strong: depth %d n %d", depth, n);
    if (!n) return X265HIP_OK;
    dim3 grid((4 * n + 255) / 256), block(256);
    if (depth == 8)
        hipLaunchKernelGGL((pel_filter_luma_strong_kernel<uint8_t>), grid, block, 0, as_stream(stream), (uint8_t*)plane, off, srcStep, offset, tcP, tcQ, n);
    else
        hipLaunchKernelGGL((pel_filter_luma_strong_kernel<uint16_t>), grid, block, 0, as_stream(stream), (uint16_t*)plane, off, srcStep, offset, tcP, tcQ, n);
    XH_LAUNCH_CHECK("pel_filter_luma_strong_kernel");
    return X265HIP_OK;
}

extern "C" int x265hip_pel_filter_chroma_batch(int depth, void* plane, const int64_t* off, int64_t srcStep, int64_t offset, const int32_t* tc,
                                               const int32_t* maskP, const int32_t* maskQ, int n, void* stream)
{
    XH_CHECK_DEV();
    if (!valid_depth(depth) || n < 0)
        return set_error(X265HIP_EINVAL, "pel_filter_chroma: depth %d n %d", depth, n);
    if (!n) return X265HIP_OK;
    dim3 grid((4 * n + 255) / 256), block(256);
    const int maxv = (1 << depth) - 1;
    if (depth == 8)
        hipLaunchKernelGGL((pel_filter_chroma_kernel<uint8_t>), grid, block, 0, as_stream(stream), (uint8_t*)plane, off, srcStep, offset, tc, maskP, maskQ, n, maxv);
    else
        hipLaunchKernelGGL((pel_filter_chroma_kernel<uint16_t>), grid, block, 0, as_stream(stream), (uint16_t*)plane, off, srcStep, offset, tc, maskP, maskQ, n, maxv);
    XH_LAUNCH_CHECK("pel_filter_chroma_kernel");
    return X265HIP_OK;
}

extern "C" int x265hip_sao_sign(int depth, int8_t* dst, const void* src1, const void* src2, int n, void* stream)
{
    XH_CHECK_DEV();
    if (!valid_depth(depth) || n < 0)
        return set_error(X265HIP_EINVAL, "sao_sign: depth %d n %d", depth, n);
    if (!n) return X265HIP_OK;
    dim3 grid((n + 255) / 256), block(256);
    if (depth == 8)
        hipLaunchKernelGGL((sao_sign_kernel<uint8_t>), grid, block, 0, as_stream(stream), dst, (const uint8_t*)src1, (const uint8_t*)src2, n);
    else
        hipLaunchKernelGGL((sao_sign_kernel<uint16_t>), grid, block, 0, as_stream(stream), dst, (const uint16_t*)src1, (const uint16_t*)src2, n);
    XH_LAUNCH_CHECK("sao_sign_kernel");
    return X265HIP_OK;
}

extern "C" int x265hip_sao_apply_batch(int depth, int kind, void* plane, int64_t stride, int8_t* aux, const x265hip_sao_job* jobs, int n, void* stream)
{
    XH_CHECK_DEV();
    if (!valid_depth(depth) || kind < 0 || kind > 5 || n < 0)
        return set_error(X265HIP_EINVAL, "sao_apply: depth %d kind %d n %d", depth, kind, n);
    if (!n) return X265HIP_OK;
    const int maxv = (1 << depth) - 1, boShift = depth - 5;            // SAO_BO_BITS = 5
    dim3 grid(n), block(256);
    if (depth == 8)
        hipLaunchKernelGGL((sao_apply_kernel<uint8_t>), grid, block, 0, as_stream(stream), kind, (uint8_t*)plane, stride, aux, jobs, maxv, boShift);
    else
        hipLaunchKernelGGL((sao_apply_kernel<uint16_t>), grid, block, 0, as_stream(stream), kind, (uint16_t*)plane, stride, aux, jobs, maxv, boShift);
    XH_LAUNCH_CHECK("sao_apply_kernel");
    return X265HIP_OK;
}

extern "C" int x265hip_sao_stats_batch(int depth, int kind, const int16_t* diff, const void* plane, int64_t stride, int8_t* aux,
                                       const x265hip_sao_stats_job* jobs, int n, int32_t* stats, int32_t* count, void* stream)
{
    XH_CHECK_DEV();
    if (!valid_depth(depth) || kind < 0 || kind > 4 || n < 0)
        return set_error(X265HIP_EINVAL, "sao_stats: depth %d kind %d n %d", depth, kind, n);
    if (!n) return X265HIP_OK;
    dim3 grid(n), block(256);
    if (depth == 8)
        hipLaunchKernelGGL((sao_stats_kernel<uint8_t>), grid, block, 0, as_stream(stream), kind, diff, (const uint8_t*)plane, stride, aux, jobs, stats, count,
                           depth - 5);
    else
        hipLaunchKernelGGL((sao_stats_kernel<uint16_t>), grid, block, 0, as_stream(stream), kind, diff, (const uint16_t*)plane, stride, aux, jobs, stats, count,
                           depth - 5);
    XH_LAUNCH_CHECK("sao_stats_kernel");
    return X265HIP_OK;
}

extern "C" int x265hip_deblock_luma_batch(int depth, void* plane, int64_t stride, int dir, const int32_t* xy, const uint8_t* bs, const int8_t* qpP,
                                          const int8_t* qpQ, const uint8_t* bypass, int betaOffsetDiv2, int tcOffsetDiv2, int n, void* stream)
{
    XH_CHECK_DEV();
    if (!valid_depth(depth) || (dir != 0 && dir != 1) || n < 0 || betaOffsetDiv2 < -6 || betaOffsetDiv2 > 6 || tcOffsetDiv2 < -6 || tcOffsetDiv2 > 6)
        return set_error(X265HIP_EINVAL, "deblock_luma: depth %d dir %d n %d offsets %d %d", depth, dir, n, betaOffsetDiv2, tcOffsetDiv2);
    if (!n) return X265HIP_OK;
    dim3 grid((4 * n + 255) / 256), block(256);
    if (depth == 8)
        hipLaunchKernelGGL((deblock_luma_kernel<uint8_t>), grid, block, 0, as_stream(stream), (uint8_t*)plane, stride, dir, xy, bs, qpP, qpQ, bypass,
                           betaOffsetDiv2, tcOffsetDiv2, n, depth);
    else
        hipLaunchKernelGGL((deblock_luma_kernel<uint16_t>), grid, block, 0, as_stream(stream), (uint16_t*)plane, stride, dir, xy, bs, qpP, qpQ, bypass,
                           betaOffsetDiv2, tcOffsetDiv2, n, depth);
    XH_LAUNCH_CHECK("deblock_luma_kernel");
    return X265HIP_OK;
}

extern "C" int x265hip_deblock_chroma_batch(int depth, void* cb, void* cr, int64_t strideC, int dir, const int32_t* xy, const uint8_t* bs,
                                            const int8_t* qpP, const int8_t* qpQ, const uint8_t* bypass, int tcOffsetDiv2, int cbQpOffset, int crQpOffset,
                                            int n, void* stream)
{
    XH_CHECK_DEV();
    if (!valid_depth(depth) || (dir != 0 && dir != 1) || n < 0 || tcOffsetDiv2 < -6 || tcOffsetDiv2 > 6 || cbQpOffset < -12 || cbQpOffset > 12 ||
        crQpOffset < -12 || crQpOffset > 12)
        return set_error(X265HIP_EINVAL, "deblock_chroma: depth %d dir %d n %d offsets %d %d %d", depth, dir, n, tcOffsetDiv2, cbQpOffset, crQpOffset);
    if (!n) return X265HIP_OK;
    dim3 grid((8 * n + 255) / 256), block(256);
    if (depth == 8)
        hipLaunchKernelGGL((deblock_chroma_kernel<uint8_t>), grid, block, 0, as_stream(stream), (uint8_t*)cb, (uint8_t*)cr, strideC, dir, xy, bs, qpP, qpQ,
                           bypass, tcOffsetDiv2, cbQpOffset, crQpOffset, n, depth);
    else
        hipLaunchKernelGGL((deblock_chroma_kernel<uint16_t>), grid, block, 0, as_stream(stream), (uint16_t*)cb, (uint16_t*)cr, strideC, dir, xy, bs, qpP, qpQ,
                           bypass, tcOffsetDiv2, cbQpOffset, crQpOffset, n, depth);
    XH_LAUNCH_CHECK("deblock_chroma_kernel");
    return X265HIP_OK;
}
