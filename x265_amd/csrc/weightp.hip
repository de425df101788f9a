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

namespace xh {

struct WeightCand { int present, w0, round, shift, offset; };

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
                                                          const int32_t* __restrict__ intraCost, const WeightCand* __restrict__ cands,
                                                          uint32_t* __restrict__ costs, int maxv, int correction)
{
    const int c = blockIdx.y;
    const WeightCand k = cands[c];
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
    static WeightCand* dCand = nullptr;
    static int cap = 0;
    if (n > cap)
    {
        if (dCand) (void)hipFree(dCand);
        cap = n < 16 ? 16 : n;
        if (hipMalloc((void**)&dCand, sizeof(WeightCand) * cap) != hipSuccess) { cap = 0; dCand = nullptr; return set_error(X265HIP_ENOMEM, "weight_cost: candidates"); }
    }
    const int correction = 14 - depth;
    WeightCand* h = (WeightCand*)alloca(sizeof(WeightCand) * n);
    for (int i = 0; i < n; i++)
    {
        const int denom = wp[i].log2WeightDenom;
        h[i].present = wp[i].wtPresent;
        h[i].w0 = wp[i].inputWeight;
        h[i].round = (denom ? 1 << (denom - 1) : 0) << correction;
        h[i].shift = denom + correction;
        h[i].offset = wp[i].inputOffset << (depth - 8);
    }
    if (hipMemcpyAsync(dCand, h, sizeof(WeightCand) * n, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)   // h is on this stack
        return set_error(X265HIP_EHIP, "weight_cost: candidate upload");
    if (hipMemsetAsync(costs, 0, sizeof(uint32_t) * n, st) != hipSuccess)
        return set_error(X265HIP_EHIP, "weight_cost: memset");
    const int wcu = (width + 7) >> 3, hcu = (lines + 7) >> 3, ncu = wcu * hcu;
    dim3 grid((ncu + 255) / 256, n), block(256);
    if (depth == 8)
        hipLaunchKernelGGL((weight_cost_kernel<uint8_t>), grid, block, 0, st, (const uint8_t*)fenc, (const uint8_t*)ref, stride, wcu, ncu, intraCost, dCand, costs,
                           (1 << depth) - 1, correction);
    else
        hipLaunchKernelGGL((weight_cost_kernel<uint16_t>), grid, block, 0, st, (const uint16_t*)fenc, (const uint16_t*)ref, stride, wcu, ncu, intraCost, dCand, costs,
                           (1 << depth) - 1, correction);
    XH_LAUNCH_CHECK("weight_cost_kernel");
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
    static uint32_t* dCost = nullptr;
    if (!dCost && hipMalloc((void**)&dCost, sizeof(uint32_t) * 4) != hipSuccess)
        return set_error(X265HIP_ENOMEM, "lookahead_weights_analyse: scratch");
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
