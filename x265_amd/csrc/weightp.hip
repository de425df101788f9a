// weightp.hip — the lookahead's weighted-prediction analysis (the part left open): LookaheadTLD::weightCostLuma
// (reference source/encoder/slicetype.cpp:807-840) for any number of candidate weights in one launch, and LookaheadTLD::weightsAnalyse
// (:860-960) on top of it.
//
// weightCostLuma weights the whole padded lowres reference into a scratch plane (weight_pp) and then sums, over every 8x8 block, the
// smaller of the block's SATD against the frame and its intra cost.  A weighted sample depends on its own reference sample only, so the
// kernel weights the 64 samples of a block on the fly and never writes the scratch plane: one lane per block, the 8x8 SATD as the
// reference's two 8x4 halves (four 4x4 Hadamards, (left + right) >> 1 per half, pixel.cpp:236-282), a wave-level sum, one atomic per wave.
// weightsAnalyse itself is a few float decisions around two cost evaluations; it runs on the host side of the library (the same IEEE
// single-precision expressions in the same order), launches the two evaluations, and — when it decides to weight — the weighting of the
// four lowres planes (x265hip_weight_pp over the padded buffers).
#include "common.h"
#include <cmath>
#include <algorithm>
#include <vector>

namespace xh {

struct WeightCand { int present, w0, round, shift, offset; };
constexpr int kCandsPerLaunch = 16;
struct WeightCands { WeightCand c[kCandsPerLaunch]; };

__device__ __forceinline__ int had4_abs_sum(const int (&d)[4][4])
{
    int t[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        const int s01 = d[i][0] + d[i][1], d01 = d[i][0] - d[i][1], s23 = d[i][2] + d[i][3], d23 = d[i][2] - d[i][3];
        t[i][0] = s01 + s23; t[i][1] = d01 + d23; t[i][2] = s01 - s23; t[i][3] = d01 - d23;
    }
    int sum = 0;
#pragma unroll
    for (int j = 0; j < 4; j++)
    {
        const int s01 = t[0][j] + t[1][j], d01 = t[0][j] - t[1][j], s23 = t[2][j] + t[3][j], d23 = t[2][j] - t[3][j];
        sum += abs(s01 + s23) + abs(d01 + d23) + abs(s01 - s23) + abs(d01 - d23);
    }
    return sum;
}

template <typename P>
__global__ __launch_bounds__(256) void weight_cost_kernel(const P* __restrict__ fenc, const P* __restrict__ ref, int64_t stride, int widthInCU, int ncu,
                                                          const int32_t* __restrict__ intraCost, WeightCands cands,
                                                          uint32_t* __restrict__ costs, int maxv, int correction)
{
    const int c = blockIdx.y;
    const WeightCand k = cands.c[c];
    const int mb = blockIdx.x * blockDim.x + threadIdx.x;
    int cost = 0;
    if (mb < ncu)
    {
        const int by = mb / widthInCU, bx = mb - by * widthInCU;
        const P* r = ref + (int64_t)by * 8 * stride + bx * 8;
        const P* f = fenc + (int64_t)by * 8 * stride + bx * 8;
        int satd = 0;
#pragma unroll
        for (int half = 0; half < 2; half++)
        {
            int sums = 0;
#pragma unroll
            for (int side = 0; side < 2; side++)
            {
                int d[4][4];
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int i = 0; i < 4; i++)
                    {
                        const int64_t o = (int64_t)(half * 4 + j) * stride + side * 4 + i;
                        int v = r[o];
                        if (k.present)
                        {
                            v = ((k.w0 * (int)(int16_t)(v << correction) + k.round) >> k.shift) + k.offset;      // weight_pp_c, pixel.cpp:518-543
                            v = v < 0 ? 0 : (v > maxv ? maxv : v);
                        }
                        d[j][i] = v - (int)f[o];
                    }
                sums += had4_abs_sum(d);
            }
            satd += sums >> 1;
        }
        cost = min(satd, intraCost[mb]);
    }
    cost = wave_sum(cost);
    if ((threadIdx.x & 63) == 0 && cost)
        atomicAdd(&costs[c], (uint32_t)cost);
}

static int launch_weight_costs(int depth, const void* fenc, const void* ref, int64_t stride, int width, int lines, const int32_t* intraCost,
                               const x265hip_weight_param* wp, int n, uint32_t* costs, hipStream_t st)
{
    // the candidates travel as kernel arguments, sixteen per launch: no scratch buffer to own, so any thread, any device, any stream
    const int correction = 14 - depth;
    if (hipMemsetAsync(costs, 0, sizeof(uint32_t) * n, st) != hipSuccess)
        return set_error(X265HIP_EHIP, "weight_cost: memset");
    const int wcu = (width + 7) >> 3, hcu = (lines + 7) >> 3, ncu = wcu * hcu;
    for (int i0 = 0; i0 < n; i0 += kCandsPerLaunch)
    {
        const int m = n - i0 < kCandsPerLaunch ? n - i0 : kCandsPerLaunch;
        WeightCands h{};
        for (int i = 0; i < m; i++)
        {
            const int denom = wp[i0 + i].log2WeightDenom;
            h.c[i].present = wp[i0 + i].wtPresent;
            h.c[i].w0 = wp[i0 + i].inputWeight;
            h.c[i].round = (denom ? 1 << (denom - 1) : 0) << correction;
            h.c[i].shift = denom + correction;
            h.c[i].offset = wp[i0 + i].inputOffset << (depth - 8);
        }
        dim3 grid((ncu + 255) / 256, m), block(256);
        if (depth == 8)
            hipLaunchKernelGGL((weight_cost_kernel<uint8_t>), grid, block, 0, st, (const uint8_t*)fenc, (const uint8_t*)ref, stride, wcu, ncu, intraCost, h, costs + i0,
                               (1 << depth) - 1, correction);
        else
            hipLaunchKernelGGL((weight_cost_kernel<uint16_t>), grid, block, 0, st, (const uint16_t*)fenc, (const uint16_t*)ref, stride, wcu, ncu, intraCost, h,
                               costs + i0, (1 << depth) - 1, correction);
        XH_LAUNCH_CHECK("weight_cost_kernel");
    }
    return X265HIP_OK;
}

} // namespace xh

using namespace xh;

extern "C" int x265hip_lookahead_weight_cost_batch(int depth, const void* fencPlane, const void* refPlane, int64_t stride, int width, int lines,
                                                   const int32_t* intraCost, const x265hip_weight_param* wp, int n, uint32_t* costs, void* stream)
{
    XH_CHECK_DEV();
    if (!valid_depth(depth) || width < 1 || lines < 1 || n < 0 || n > 4096)
        return set_error(X265HIP_EINVAL, "lookahead_weight_cost: depth %d %dx%d n %d", depth, width, lines, n);
    for (int i = 0; i < n; i++)
        if (wp[i].log2WeightDenom < 0 || wp[i].log2WeightDenom > 7 || wp[i].inputWeight < 0 || wp[i].inputWeight > 255)
            return set_error(X265HIP_EINVAL, "lookahead_weight_cost: weight %d denominator %d", wp[i].inputWeight, wp[i].log2WeightDenom);
    if (!n) return X265HIP_OK;
    return launch_weight_costs(depth, fencPlane, refPlane, stride, width, lines, intraCost, wp, n, costs, as_stream(stream));
}

extern "C" int x265hip_lookahead_weights_analyse(int depth, const void* fencPlane, const void* refBuffers, int64_t planeElems, int64_t stride,
                                                 int64_t padOffset, int paddedLines, int width, int lines, const int32_t* intraCost,
                                                 uint64_t fencSsd, uint64_t fencSum, uint64_t refSsd, uint64_t refSum, void* weightedBuffers,
                                                 x265hip_weight_param* chosen, int* isWeighted, void* stream)
{
    XH_CHECK_DEV();
    if (!valid_depth(depth) || width < 1 || lines < 1 || !chosen || !isWeighted || paddedLines < lines || planeElems < stride * paddedLines)
        return set_error(X265HIP_EINVAL, "lookahead_weights_analyse: depth %d %dx%d", depth, width, lines);
    hipStream_t st = as_stream(stream);
    const int B = depth == 8 ? 1 : 2;
    const char* ref0 = (const char*)refBuffers + padOffset * B;
    *isWeighted = 0;
    chosen->inputWeight = chosen->inputOffset = chosen->log2WeightDenom = chosen->wtPresent = 0;
    // four words of device scratch per (thread, device): the function blocks, so a thread never has two calls in flight
    static thread_local uint32_t* t_cost[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
        return set_error(X265HIP_EHIP, "lookahead_weights_analyse: no current device");
    if (!t_cost[dev] && hipMalloc((void**)&t_cost[dev], sizeof(uint32_t) * 4) != hipSuccess)
        return set_error(X265HIP_ENOMEM, "lookahead_weights_analyse: scratch");
    uint32_t* dCost = t_cost[dev];
    auto cost_of = [&](const x265hip_weight_param& wp, unsigned int* out) -> int {
        int e = launch_weight_costs(depth, fencPlane, ref0, stride, width, lines, intraCost, &wp, 1, dCost, st);
        if (e) return e;
        uint32_t v = 0;
        if (hipMemcpyAsync(&v, dCost, sizeof(v), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            return set_error(X265HIP_EHIP, "lookahead_weights_analyse: cost readback");
        *out = v;
        return X265HIP_OK;
    };
    // slicetype.cpp:885-897
    static const float epsilon = 1.f / 128.f;
    float guessScale, fencMean, refMean;
    if (fencSsd && refSsd)
        guessScale = sqrtf((float)fencSsd / refSsd);
    else
        guessScale = 1.0f;
    fencMean = (float)fencSum / (lines * width) / (1 << (depth - 8));
    refMean = (float)refSum / (lines * width) / (1 << (depth - 8));
    if (fabsf(refMean - fencMean) < 0.5f && fabsf(1.f - guessScale) < epsilon)
        return X265HIP_OK;
    // :903-910, WeightParam::setFromWeightAndOffset (slice.h:304-316) with bNormalize; wtPresent is still 0 for this first evaluation
    int w = (int)(guessScale * 128 + 0.5f), mindenom = 7;
    while (mindenom > 0 && w > 127) { mindenom--; w >>= 1; }
    int minscale = w < 127 ? w : 127, minoff = 0, found = 0;
    if (minscale < 0) minscale = 0;
    unsigned int minscore = 0, origscore = 1, s = 0;
    x265hip_weight_param wp = { minscale, 0, mindenom, 0 };
    int e = cost_of(wp, &minscore);
    if (e) return e;
    origscore = minscore;
    if (!minscore)
        return X265HIP_OK;
    // :912-926
    int curScale = minscale;
    int curOffset = (int)(fencMean - refMean * curScale / (1 << mindenom) + 0.5f);
    if (curOffset < -128 || curOffset > 127)
    {
        curOffset = curOffset < -128 ? -128 : 127;
        curScale = (int)((1 << mindenom) * (fencMean - curOffset) / refMean + 0.5f);
        curScale = curScale < 0 ? 0 : (curScale > 127 ? 127 : curScale);
    }
    wp.inputWeight = curScale; wp.inputOffset = curOffset; wp.log2WeightDenom = mindenom; wp.wtPresent = 1;
    e = cost_of(wp, &s);
    if (e) return e;
    if (s < minscore) { minscore = s; minscale = curScale; minoff = curOffset; found = 1; }
    // :928-936
    if (mindenom > 0 && !(minscale & 1))
    {
        int idx = 0;
        while (!((minscale >> idx) & 1) && idx < 31) idx++;
        const int sh = idx < mindenom ? idx : mindenom;
        mindenom -= sh;
        minscale >>= sh;
    }
    if (!found || (minscale == 1 << mindenom && minoff == 0) || (float)minscore / origscore > 0.998f)
        return X265HIP_OK;
    // :942-958: weight the four lowres planes, whole padded buffers
    chosen->inputWeight = minscale; chosen->inputOffset = minoff; chosen->log2WeightDenom = mindenom; chosen->wtPresent = 1;
    const int correction = 14 - depth, round = (mindenom ? 1 << (mindenom - 1) : 0) << correction;
    for (int i = 0; i < 4; i++)
    {
        e = x265hip_weight_pp(depth, (const char*)refBuffers + (int64_t)i * planeElems * B, (char*)weightedBuffers + (int64_t)i * planeElems * B, stride,
                              (int)stride, paddedLines, minscale, round, mindenom + correction, minoff << (depth - 8), stream);
        if (e) return e;
    }
    *isWeighted = 1;
    return X265HIP_OK;
}

// ---- adaptive quantisation of the lookahead: LookaheadTLD::calcAdaptiveQuantFrame (slicetype.cpp:444-700), 4:2:0, aqMode 0..3.
// The device computes what is data-parallel — the AC energy of every qgSize block (acEnergyCu :256: luma block + the two half-size chroma
// blocks, var = (sum, ssd), energy = ssd - (sum^2 >> shift) per plane, 32-bit wrap-around like the reference) and the frame sums the
// weighted-prediction analysis reads; one wave per block.  The per-block offsets are a few double-precision expressions with a running
// average between two passes: they run on the host side of the library in the reference's order (same libm), from the downloaded energies.
namespace xh {

template <typename P>
__global__ __launch_bounds__(256) void aq_energy_kernel(const P* __restrict__ y, const P* __restrict__ cb, const P* __restrict__ cr, int64_t stride,
                                                        int64_t strideC, int blocksX, int nblocks, int qg, uint32_t* __restrict__ energy,
                                                        unsigned long long* __restrict__ sums)
{
    // a wave walks over blocks (grid-stride) and keeps the frame sums in registers; the workgroup's four waves meet in LDS and issue one
    // atomic per plane and quantity — thousands of atomics on six addresses would serialise the whole launch
    __shared__ unsigned long long sAcc[4][6];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wavesTotal = gridDim.x * 4;
    unsigned long long acc[6] = { 0, 0, 0, 0, 0, 0 };
    for (int blk = blockIdx.x * 4 + wv; blk < nblocks; blk += wavesTotal)
    {
        const int by = (blk / blocksX) * qg, bx = (blk % blocksX) * qg;
        uint32_t var = 0;
#pragma unroll
        for (int p = 0; p < 3; p++)
        {
            const int n = p ? qg >> 1 : qg;                              // block edge in this plane
            const int shift = p ? (qg == 8 ? 4 : 6) : (qg == 8 ? 6 : 8);
            const P* src = p == 0 ? y + (int64_t)by * stride + bx : (p == 1 ? cb : cr) + (int64_t)(by >> 1) * strideC + (bx >> 1);
            const int64_t ss = p ? strideC : stride;
            uint32_t sum = 0, ssd = 0;
            for (int i = lane; i < n * n; i += 64)
            {
                const uint32_t v = src[(int64_t)(i / n) * ss + (i % n)];
                sum += v;
                ssd += v * v;
            }
            sum = (uint32_t)wave_sum((int)sum);
            ssd = (uint32_t)wave_sum((int)ssd);
            acc[p] += sum;
            acc[3 + p] += ssd;
            var += ssd - (uint32_t)(((uint64_t)sum * sum) >> shift);
        }
        if (!lane)
            energy[blk] = var;
    }
    if (!lane)
    {
#pragma unroll
        for (int k = 0; k < 6; k++) sAcc[wv][k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < 6)
        atomicAdd(&sums[threadIdx.x], sAcc[0][threadIdx.x] + sAcc[1][threadIdx.x] + sAcc[2][threadIdx.x] + sAcc[3][threadIdx.x]);
}

static int aq_exp2fix8(double x)
{
    const int i = (int)(x * (-64.f / 6.f) + 512.5f);
    if (i < 0) return 0;
    if (i > 1023) return 0xffff;
    const int lut = (int)(256.0 * (pow(2.0, (i & 63) / 64.0) - 1.0) + 0.5);       // x265_exp2_lut[i & 63] (constants.cpp)
    return (lut + 256) << (i >> 6) >> 8;
}

} // namespace xh

extern "C" int x265hip_aq_block_energy(int depth, const x265hip_yuv* pic, int width, int height, int qgSize, uint32_t* energy, uint64_t* sums, void* stream)
{
    XH_CHECK_DEV();
    if (!valid_depth(depth) || !pic || width < 1 || height < 1 || (qgSize != 8 && qgSize != 16))
        return set_error(X265HIP_EINVAL, "aq_block_energy: depth %d %dx%d qgSize %d", depth, width, height, qgSize);
    const int bxs = (width + qgSize - 1) / qgSize, bys = (height + qgSize - 1) / qgSize, n = bxs * bys;
    dim3 grid(std::min((n + 3) / 4, 512)), block(256);
    if (depth == 8)
        hipLaunchKernelGGL((aq_energy_kernel<uint8_t>), grid, block, 0, as_stream(stream), (const uint8_t*)pic->y, (const uint8_t*)pic->cb, (const uint8_t*)pic->cr,
                           pic->strideY, pic->strideC, bxs, n, qgSize, energy, (unsigned long long*)sums);
    else
        hipLaunchKernelGGL((aq_energy_kernel<uint16_t>), grid, block, 0, as_stream(stream), (const uint16_t*)pic->y, (const uint16_t*)pic->cb,
                           (const uint16_t*)pic->cr, pic->strideY, pic->strideC, bxs, n, qgSize, energy, (unsigned long long*)sums);
    XH_LAUNCH_CHECK("aq_energy_kernel");
    return X265HIP_OK;
}

extern "C" int x265hip_lookahead_aq_frame(int depth, const x265hip_yuv* pic, int width, int height, int qgSize, int aqMode, double aqStrength, int weightp,
                                          double* qpAqOffset, int32_t* invQscaleFactor, int32_t* invQscaleFactor8x8, uint64_t* wpStats, int* blockCountOut,
                                          void* stream)
{
    XH_CHECK_DEV();
    if (!valid_depth(depth) || !pic || width < 16 || height < 16 || (qgSize != 8 && qgSize != 16) || aqMode < 0 || aqMode > 3 || !wpStats || !blockCountOut ||
        (aqMode && (!qpAqOffset || !invQscaleFactor || (qgSize == 8 && !invQscaleFactor8x8))))
        return set_error(X265HIP_EINVAL, "lookahead_aq_frame: depth %d %dx%d qgSize %d aqMode %d", depth, width, height, qgSize, aqMode);
    hipStream_t st = as_stream(stream);
    const int lw = ((width / 2 + 7) >> 3) << 3, lh = ((height / 2 + 7) >> 3) << 3, wcu = lw >> 3, hcu = lh >> 3;
    const int incr = qgSize == 8 ? 8 : 16;
    const int blockCount = qgSize == 8 ? (wcu * 2) * (hcu * 2) : wcu * hcu;
    const int bxs = (width + incr - 1) / incr, bys = (height + incr - 1) / incr, n = bxs * bys;       // the blocks the reference's loops visit
    *blockCountOut = blockCount;
    if (n > blockCount)
        return set_error(X265HIP_EINVAL, "lookahead_aq_frame: %d blocks visited, %d allocated by the reference", n, blockCount);
    uint64_t sums[6] = { 0, 0, 0, 0, 0, 0 };
    std::vector<uint32_t> energy((size_t)n);
    const bool needEnergy = (aqMode != 0 && aqStrength != 0) || weightp;
    if (needEnergy)
    {
        uint32_t* dE = nullptr;
        uint64_t* dS = nullptr;
        if (hipMalloc((void**)&dE, sizeof(uint32_t) * n + 64) != hipSuccess)
            return set_error(X265HIP_ENOMEM, "lookahead_aq_frame: scratch");
        dS = (uint64_t*)((char*)dE + (((size_t)sizeof(uint32_t) * n + 15) & ~(size_t)15));
        int e = hipMemsetAsync(dS, 0, 48, st) == hipSuccess ? X265HIP_OK : X265HIP_EHIP;
        if (!e) e = x265hip_aq_block_energy(depth, pic, width, height, qgSize, dE, dS, stream);
        if (!e && (hipMemcpyAsync(energy.data(), dE, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, st) != hipSuccess ||
                   hipMemcpyAsync(sums, dS, 48, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess))
            e = set_error(X265HIP_EHIP, "lookahead_aq_frame: readback");
        (void)device_free(dE);
        if (e) return e;
    }
    const float modeOneConst = qgSize == 8 ? 11.427f : 14.427f, modeTwoConst = qgSize == 8 ? 8.f : 11.f;
    if (aqMode == 0 || aqStrength == 0)
    {
        if (aqMode && aqStrength == 0)
            for (int i = 0; i < blockCount; i++) { qpAqOffset[i] = 0; invQscaleFactor[i] = 256; }
        // with AQ off the energies are only computed for the statistics (:497-503); in mode VARIANCE each block is measured once (second
        // loop), in the AUTO modes once (first loop): the sums above count every block exactly once in all three cases
    }
    else
    {
        // slicetype.cpp:514-632, double arithmetic in the reference's order
        double avg_adj_pow2 = 0, avg_adj = 0, qp_adj = 0, bias_strength = 0.f, strength = 0.f;
        if (aqMode == 2 || aqMode == 3)
        {
            const double bit_depth_correction = 1.f / (1 << (2 * (depth - 8)));
            for (int i = 0; i < n; i++)
            {
                qp_adj = pow(energy[i] * bit_depth_correction + 1, 0.1);
                qpAqOffset[i] = qp_adj;
                avg_adj += qp_adj;
                avg_adj_pow2 += qp_adj * qp_adj;
            }
            avg_adj /= blockCount;
            avg_adj_pow2 /= blockCount;
            strength = aqStrength * avg_adj;
            avg_adj = avg_adj - 0.5f * (avg_adj_pow2 - modeTwoConst) / avg_adj;
            bias_strength = aqStrength;
        }
        else
            strength = aqStrength * 1.0397f;
        for (int i = 0; i < n; i++)
        {
            if (aqMode == 3)
            {
                qp_adj = qpAqOffset[i];
                qp_adj = strength * (qp_adj - avg_adj) + bias_strength * (1.f - modeTwoConst / (qp_adj * qp_adj));
            }
            else if (aqMode == 2)
            {
                qp_adj = qpAqOffset[i];
                qp_adj = strength * (qp_adj - avg_adj);
            }
            else
                qp_adj = strength * (log2((double)(energy[i] > 1 ? energy[i] : 1)) - (modeOneConst + 2 * (depth - 8)));
            qpAqOffset[i] = qp_adj;
            invQscaleFactor[i] = aq_exp2fix8(qp_adj);
        }
    }
    if (qgSize == 8 && aqMode)
        for (int cuY = 0; cuY < hcu; cuY++)
            for (int cuX = 0; cuX < wcu; cuX++)
                invQscaleFactor8x8[cuX + cuY * wcu] = (invQscaleFactor[cuX * 2 + cuY * wcu * 4] + invQscaleFactor[cuX * 2 + cuY * wcu * 4 + 1] +
                                                       invQscaleFactor[cuX * 2 + cuY * wcu * 4 + wcu * 2] + invQscaleFactor[cuX * 2 + cuY * wcu * 4 + wcu * 2 + 1]) / 4;
    if (weightp)
    {
        // :662-676
        const int maxCol = ((width + 8) >> 4) << 4, maxRow = ((height + 8) >> 4) << 4;
        const int w3[3] = { maxCol, maxCol >> 1, maxCol >> 1 }, h3[3] = { maxRow, maxRow >> 1, maxRow >> 1 };
        for (int i = 0; i < 3; i++)
            sums[3 + i] = sums[3 + i] - (sums[i] * sums[i] + (uint64_t)((w3[i] * h3[i]) / 2)) / (uint64_t)(w3[i] * h3[i]);
    }
    for (int i = 0; i < 6; i++) wpStats[i] = sums[i];
    return X265HIP_OK;
}

// ---- CU-tree: Lookahead::estimateCUPropagate (slicetype.cpp:2641-2750) with its primitive estimateCUPropagateCost (pixel.cpp:914-940).
// One lane per 8x8 block of frame b: the amount it passes on (double precision, the reference's operations one by one — explicit _rn
// intrinsics so that nothing is contracted into an FMA), split over the lists it used and over the four blocks its vector points at in
// each reference.  The reference adds into the references' uint16 propagateCost with saturation, block after block; every addend is
// non-negative, so the saturated result is min(initial + sum of addends, 65535) in any order: addends go to 64-bit accumulators with
// atomics, a second kernel folds them in.
namespace xh {

__global__ __launch_bounds__(256) void cutree_propagate_kernel(int W, int H, double fps, int referenced, int bw0, int bw1,
                                                               const uint16_t* __restrict__ propagateIn, const int32_t* __restrict__ intraCost,
                                                               const uint16_t* __restrict__ lowresCosts, const int32_t* __restrict__ invQscale,
                                                               const int32_t* __restrict__ mvs0, const int32_t* __restrict__ mvs1,
                                                               unsigned long long* __restrict__ acc)
{
    const int cu = blockIdx.x * blockDim.x + threadIdx.x, ncu = W * H;
    if (cu >= ncu)
        return;
    const int by = cu / W, bx = cu - by * W;
    const int ic = intraCost[cu], lc = lowresCosts[cu];
    const int interCost = min(ic, lc & 0x3FFF);
    const double propagateIntra = __dmul_rn((double)ic, (double)invQscale[cu]);           // int * int in the reference, converted: same value (< 2^53)
    const double propagateAmount = __dadd_rn((double)(referenced ? propagateIn[cu] : 0), __dmul_rn(propagateIntra, fps));
    const double num = (double)(ic - interCost);
    const int amount = (int)__dadd_rn(__ddiv_rn(__dmul_rn(propagateAmount, num), (double)ic), 0.5);
    if (amount <= 0)
        return;
    const int lists = lc >> 14;
#pragma unroll
    for (int list = 0; list < 2; list++)
    {
        if (!((lists >> list) & 1))
            continue;
        int la = amount;
        if (lists == 3)
            la = (la * (list ? bw1 : bw0) + 32) >> 6;
        const int32_t* mv = list ? mvs1 : mvs0;
        unsigned long long* a = acc + (size_t)list * ncu;
        int x = mv[2 * cu], y = mv[2 * cu + 1];
        if (!(x | y))
        {
            atomicAdd(&a[cu], (unsigned long long)la);
            continue;
        }
        const int cux = (x >> 5) + bx, cuy = (y >> 5) + by, i0 = cux + cuy * W;
        x &= 31;
        y &= 31;
        const int w0 = (32 - y) * (32 - x), w1 = (32 - y) * x, w2 = y * (32 - x), w3 = y * x;
        const bool xin = cux >= 0 && cux < W, x1in = cux + 1 >= 0 && cux + 1 < W, yin = cuy >= 0 && cuy < H, y1in = cuy + 1 >= 0 && cuy + 1 < H;
        if (xin && yin) atomicAdd(&a[i0], (unsigned long long)((la * w0 + 512) >> 10));
        if (x1in && yin) atomicAdd(&a[i0 + 1], (unsigned long long)((la * w1 + 512) >> 10));
        if (xin && y1in) atomicAdd(&a[i0 + W], (unsigned long long)((la * w2 + 512) >> 10));
        if (x1in && y1in) atomicAdd(&a[i0 + W + 1], (unsigned long long)((la * w3 + 512) >> 10));
    }
}

__global__ __launch_bounds__(256) void cutree_fold_kernel(int ncu, const unsigned long long* __restrict__ acc, uint16_t* __restrict__ ref0, uint16_t* __restrict__ ref1)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncu)
        return;
    const unsigned long long a0 = acc[i] + ref0[i];
    ref0[i] = (uint16_t)(a0 < 65535ull ? a0 : 65535ull);
    if (ref1)
    {
        const unsigned long long a1 = acc[ncu + i] + ref1[i];
        ref1[i] = (uint16_t)(a1 < 65535ull ? a1 : 65535ull);
    }
}

} // namespace xh

extern "C" int x265hip_cutree_propagate(int widthInCU, int heightInCU, int fpsNum, int fpsDenom, double averageDuration, int bMinusP0, int p1MinusP0,
                                        int referenced, int weightedBiPred, const uint16_t* propagateIn, const int32_t* intraCost, const uint16_t* lowresCosts,
                                        const int32_t* invQscale, const int32_t* mvs0, const int32_t* mvs1, uint16_t* refCosts0, uint16_t* refCosts1,
                                        uint64_t* scratch, void* stream)
{
    XH_CHECK_DEV();
    if (widthInCU < 1 || heightInCU < 1 || fpsNum < 1 || fpsDenom < 1 || bMinusP0 < 1 || p1MinusP0 < bMinusP0 || !scratch || !refCosts0 ||
        (p1MinusP0 > bMinusP0 && (!refCosts1 || !mvs1)))
        return set_error(X265HIP_EINVAL, "cutree_propagate: %dx%d fps %d/%d distances %d %d", widthInCU, heightInCU, fpsNum, fpsDenom, bMinusP0, p1MinusP0);
    hipStream_t st = as_stream(stream);
    const int ncu = widthInCU * heightInCU;
    auto clipd = [](double f) { return f < 0.01 ? 0.01 : (f > 1.00 ? 1.00 : f); };                 // CLIP_DURATION (ratecontrol.h:42-47)
    const double fpsFactor = clipd((double)fpsDenom / fpsNum) / clipd(averageDuration);
    const double fps = fpsFactor / 256;
    const int dsf = ((bMinusP0 << 8) + (p1MinusP0 >> 1)) / p1MinusP0;
    const int bw = weightedBiPred ? 64 - (dsf >> 2) : 32;
    if (hipMemsetAsync(scratch, 0, sizeof(uint64_t) * 2 * (size_t)ncu, st) != hipSuccess)
        return set_error(X265HIP_EHIP, "cutree_propagate: memset");
    dim3 grid((ncu + 255) / 256), block(256);
    hipLaunchKernelGGL(cutree_propagate_kernel, grid, block, 0, st, widthInCU, heightInCU, fps, referenced, bw, 64 - bw, propagateIn, intraCost, lowresCosts, invQscale,
                       mvs0, mvs1 ? mvs1 : mvs0, (unsigned long long*)scratch);
    XH_LAUNCH_CHECK("cutree_propagate_kernel");
    hipLaunchKernelGGL(cutree_fold_kernel, grid, block, 0, st, ncu, (const unsigned long long*)scratch, refCosts0, p1MinusP0 > bMinusP0 ? refCosts1 : (uint16_t*)nullptr);
    XH_LAUNCH_CHECK("cutree_fold_kernel");
    return X265HIP_OK;
}
