// bipred.hip — motion compensation through the 14-bit domain.  First the unweighted bi-predictive branch of Predict::motionCompensation (reference
// source/common/predict.cpp:131-199): predInterLumaShort / predInterChromaShort (:268-306, :364-420) of both references into the 14-bit
// domain, then Yuv::addAvg (yuv.cpp:189-211 -> pu[].addAvg, pixel.cpp:842-862).  One launch per PU shape covers luma, Cb and Cr.
//
// A lane produces a few adjacent output samples: it forms the two 14-bit predictions in registers and combines them, so the short
// intermediates (2 x 64 x 64 x int16 per PU in the reference) never exist in memory.  The 14-bit prediction has four cases in the
// reference (convert_p2s / hps / vps / hps(rows extended) + vss, chosen by the fractions); they are one formula: with a zero fraction the
// tap set is {.., 64, ..}, hps gives p << (14 - depth) - 8192 without loss, and vss of a single 64-tap is the identity — so the kernel
// evaluates "horizontal stage (or the shift), then vertical stage (or nothing)" and is bit-identical in all four.
#include "common.h"
#include "filters.h"

namespace xh {

struct BiPlanes
{
    const void* ref[2][3];        // [list][Y, Cb, Cr] picture origins
    void* dst[3];
    int64_t strideRY, strideRC, strideDY, strideDC;
};

// V adjacent 14-bit prediction samples at (qx, qy) fractional vector; NT taps, `fracBits` 2 (luma) or 3 (chroma)
template <typename P, int NT, int V>
__device__ __forceinline__ void short_pred(const P* r, int64_t rs, int mvx, int mvy, int fracBits, int depth, int out[V])
{
    constexpr int HALF = NT / 2 - 1;
    const int mask = (1 << fracBits) - 1, xF = mvx & mask, yF = mvy & mask;
    const P* base = r + (int64_t)(mvy >> fracBits) * rs + (mvx >> fracBits);
    const Stage s1 = stage_for(IF_HPS, depth);
    int c1[NT], c2[NT];
#pragma unroll
    for (int i = 0; i < NT; i++) { c1[i] = filter_tap<NT>(xF, i); c2[i] = filter_tap<NT>(yF, i); }
    auto hrow = [&](const P* row, int h[V]) {
        if (!xF)
        {
#pragma unroll
            for (int o = 0; o < V; o++) h[o] = ((int)row[o] << (14 - depth)) - 8192;          // convert_p2s (ipfilter.cpp:40-57) == hps with taps {0,..,64,..}
        }
        else
        {
            int v[V + NT - 1];
#pragma unroll
            for (int i = 0; i < V + NT - 1; i++) v[i] = (int)row[i - HALF];
#pragma unroll
            for (int o = 0; o < V; o++)
            {
                int sum = 0;
#pragma unroll
                for (int i = 0; i < NT; i++) sum += v[o + i] * c1[i];
                h[o] = finish(sum, s1);
            }
        }
    };
    if (!yF)
    {
        hrow(base, out);
        return;
    }
    int sum[V];
#pragma unroll
    for (int o = 0; o < V; o++) sum[o] = 0;
#pragma unroll
    for (int k = 0; k < NT; k++)
    {
        int h[V];
        hrow(base + (int64_t)(k - HALF) * rs, h);
#pragma unroll
        for (int o = 0; o < V; o++) sum[o] += h[o] * c2[k];
    }
    // vss (ipfilter.cpp:284-317): shift 6, no offset; for xF == 0 this equals vps ((sum - 8192 << s) >> s), see the header comment
#pragma unroll
    for (int o = 0; o < V; o++) out[o] = (int)(int16_t)(sum[o] >> 6);
}

// How the 14-bit prediction(s) become pixels (predict.cpp): MC_AVG Yuv::addAvg; MC_WBI addWeightBi (:411-522, weightBidir :52-55);
// MC_UNI one prediction through weight_sp's formula (pixel.cpp weight_sp_c) — addWeightUni (:525-576) with the slice's weights, and with
// w = 1, shift = 14 - depth, no offset it is exactly predInterLumaPixel / predInterChromaPixel (hpp / vpp / hvpp round the same sum the
// same way: floor((floor(s / 64) + c) / 2^k) = floor((s + 64 c) / 2^(k + 6))), so the unweighted uni branch needs no kernel of its own.
enum { MC_AVG = 0, MC_WBI = 1, MC_UNI = 2 };
struct McWeights { int w0[3], w1[3], round[3], shift[3], offset[3]; };

template <int MODE>
__device__ __forceinline__ int mc_combine(int a, int b, const McWeights& k, int pl, int avgOffset, int avgShift, int maxv)
{
    int v;
    if (MODE == MC_AVG)
        v = (a + b + avgOffset) >> avgShift;
    else if (MODE == MC_WBI)
        v = (k.w0[pl] * (a + 8192) + k.w1[pl] * (b + 8192) + k.round[pl] + (k.offset[pl] * (1 << (k.shift[pl] - 1)))) >> k.shift[pl];
    else
        v = ((k.w0[pl] * (a + 8192) + k.round[pl]) >> k.shift[pl]) + k.offset[pl];
    return v < 0 ? 0 : (v > maxv ? maxv : v);
}

template <typename P, int MODE>
__global__ __launch_bounds__(256) void pred_bi_kernel(BiPlanes bp, const int32_t* __restrict__ pu_xy, const int32_t* __restrict__ mv0, const int32_t* __restrict__ mv1,
                                                      int w, int h, int n, int depth, McWeights wk)
{
    // work items per PU: luma quads, then Cb pairs, then Cr pairs
    const int lq = (w >> 2) * h, cp = (w >> 2) * (h >> 1);                 // chroma row = w/2 samples = w/4 pairs
    const int per = lq + 2 * cp;
    const long long total = (long long)n * per;
    const int shift = 15 - depth, offset = (1 << (shift - 1)) + 2 * 8192, maxv = (1 << depth) - 1;      // addAvg (pixel.cpp:845-847)
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x)
    {
        const int pu = (int)(idx / per), it = (int)(idx - (long long)pu * per);
        const int bx = pu_xy[2 * pu], by = pu_xy[2 * pu + 1];
        const int ax = mv0[2 * pu], ay = mv0[2 * pu + 1];
        const int cx = MODE == MC_UNI ? 0 : mv1[2 * pu], cy = MODE == MC_UNI ? 0 : mv1[2 * pu + 1];
        if (it < lq)
        {
            const int y = it / (w >> 2), x = (it % (w >> 2)) * 4;
            int a[4], b[4], o[4];
            short_pred<P, 8, 4>((const P*)bp.ref[0][0] + (int64_t)(by + y) * bp.strideRY + bx + x, bp.strideRY, ax, ay, 2, depth, a);
            if (MODE != MC_UNI)
                short_pred<P, 8, 4>((const P*)bp.ref[1][0] + (int64_t)(by + y) * bp.strideRY + bx + x, bp.strideRY, cx, cy, 2, depth, b);
#pragma unroll
            for (int i = 0; i < 4; i++)
                o[i] = mc_combine<MODE>(a[i], MODE == MC_UNI ? 0 : b[i], wk, 0, offset, shift, maxv);
            store4((P*)bp.dst[0] + (int64_t)(by + y) * bp.strideDY + bx + x, o);
        }
        else
        {
            const int k = it - lq, pl = 1 + k / cp, kk = k % cp;
            const int y = kk / (w >> 2), x = (kk % (w >> 2)) * 2;
            const int64_t ro = (int64_t)((by >> 1) + y) * bp.strideRC + (bx >> 1) + x;
            int a[2], b[2];
            short_pred<P, 4, 2>((const P*)bp.ref[0][pl] + ro, bp.strideRC, ax, ay, 3, depth, a);
            if (MODE != MC_UNI)
                short_pred<P, 4, 2>((const P*)bp.ref[1][pl] + ro, bp.strideRC, cx, cy, 3, depth, b);
            P* d = (P*)bp.dst[pl] + (int64_t)((by >> 1) + y) * bp.strideDC + (bx >> 1) + x;
#pragma unroll
            for (int i = 0; i < 2; i++)
                d[i] = (P)mc_combine<MODE>(a[i], MODE == MC_UNI ? 0 : b[i], wk, pl, offset, shift, maxv);
        }
    }
}

template <int MODE>
static int launch_mc(int depth, int w, int h, const BiPlanes& bp, const int32_t* pu_xy, const int32_t* mv0, const int32_t* mv1, int n,
                     const McWeights& k, void* stream)
{
    const long long total = (long long)n * ((w / 4) * h + 2 * (w / 4) * (h / 2));
    dim3 grid(grid_for((total + 255) / 256)), block(256);
    if (depth == 8)
        hipLaunchKernelGGL((pred_bi_kernel<uint8_t, MODE>), grid, block, 0, as_stream(stream), bp, pu_xy, mv0, mv1, w, h, n, depth, k);
    else
        hipLaunchKernelGGL((pred_bi_kernel<uint16_t, MODE>), grid, block, 0, as_stream(stream), bp, pu_xy, mv0, mv1, w, h, n, depth, k);
    XH_LAUNCH_CHECK("pred_bi_kernel");
    return X265HIP_OK;
}

} // namespace xh

using namespace xh;

extern "C" int x265hip_pred_inter_bi_batch(int depth, int w, int h, const x265hip_yuv* ref0, const x265hip_yuv* ref1, const x265hip_yuv* dst,
                                           const int32_t* pu_xy, const int32_t* mv0, const int32_t* mv1, int n, void* stream)
{
    XH_CHECK_DEV();
    if (!valid_depth(depth) || !valid_block(w, h) || (w & 3) || (h & 1) || n < 0 || !ref0 || !ref1 || !dst)
        return set_error(X265HIP_EINVAL, "pred_inter_bi: depth %d PU %dx%d n %d", depth, w, h, n);
    if (ref0->strideY != ref1->strideY || ref0->strideC != ref1->strideC)
        return set_error(X265HIP_EINVAL, "pred_inter_bi: the two references must share their strides");
    if (!n) return X265HIP_OK;
    BiPlanes bp;
    bp.ref[0][0] = ref0->y; bp.ref[0][1] = ref0->cb; bp.ref[0][2] = ref0->cr;
    bp.ref[1][0] = ref1->y; bp.ref[1][1] = ref1->cb; bp.ref[1][2] = ref1->cr;
    bp.dst[0] = dst->y; bp.dst[1] = dst->cb; bp.dst[2] = dst->cr;
    bp.strideRY = ref0->strideY; bp.strideRC = ref0->strideC; bp.strideDY = dst->strideY; bp.strideDC = dst->strideC;
    return launch_mc<MC_AVG>(depth, w, h, bp, pu_xy, mv0, mv1, n, McWeights{}, stream);
}

extern "C" int x265hip_motion_compensation_batch(int depth, int w, int h, const x265hip_yuv* ref0, const x265hip_yuv* ref1, const x265hip_yuv* dst,
                                                 const int32_t* pu_xy, const int32_t* mv0, const int32_t* mv1, int n,
                                                 const x265hip_weight_param* wp0, const x265hip_weight_param* wp1, void* stream)
{
    XH_CHECK_DEV();
    if (!valid_depth(depth) || !valid_block(w, h) || (w & 3) || (h & 1) || n < 0 || !ref0 || !dst || (ref1 && !mv1))
        return set_error(X265HIP_EINVAL, "motion_compensation: depth %d PU %dx%d n %d", depth, w, h, n);
    if (ref1 && (ref0->strideY != ref1->strideY || ref0->strideC != ref1->strideC))
        return set_error(X265HIP_EINVAL, "motion_compensation: the two references must share their strides");
    for (int l = 0; l < 2; l++)
        for (int pl = 0; pl < 3; pl++)
        {
            const x265hip_weight_param* p = l ? wp1 : wp0;
            if (p && (p[pl].log2WeightDenom < 0 || p[pl].log2WeightDenom > 7 || p[pl].inputWeight < -128 || p[pl].inputWeight > 255))
                return set_error(X265HIP_EINVAL, "motion_compensation: weight %d denominator %d out of range", p[pl].inputWeight, p[pl].log2WeightDenom);
        }
    if (!n) return X265HIP_OK;
    BiPlanes bp;
    const x265hip_yuv* r1 = ref1 ? ref1 : ref0;
    bp.ref[0][0] = ref0->y; bp.ref[0][1] = ref0->cb; bp.ref[0][2] = ref0->cr;
    bp.ref[1][0] = r1->y; bp.ref[1][1] = r1->cb; bp.ref[1][2] = r1->cr;
    bp.dst[0] = dst->y; bp.dst[1] = dst->cb; bp.dst[2] = dst->cr;
    bp.strideRY = ref0->strideY; bp.strideRC = ref0->strideC; bp.strideDY = dst->strideY; bp.strideDC = dst->strideC;
    const int shiftNum = 14 - depth;                                   // IF_INTERNAL_PREC - X265_DEPTH
    McWeights k{};
    if (!ref1)
    {
        // uni-prediction: predict.cpp:84-119 (P slice) and :201-265 (one list of a B slice)
        const bool weighted = wp0 && wp0[0].wtPresent;
        for (int pl = 0; pl < 3; pl++)
        {
            k.w0[pl] = weighted ? wp0[pl].inputWeight : 1;
            k.offset[pl] = weighted ? wp0[pl].inputOffset * (1 << (depth - 8)) : 0;
            k.shift[pl] = (weighted ? wp0[pl].log2WeightDenom : 0) + shiftNum;
            k.round[pl] = k.shift[pl] ? 1 << (k.shift[pl] - 1) : 0;    // addWeightUni :541 (wv.round is not what it uses)
        }
        return launch_mc<MC_UNI>(depth, w, h, bp, pu_xy, mv0, mv0, n, k, stream);
    }
    if (!(wp0 && wp1 && (wp0[0].wtPresent || wp1[0].wtPresent)))
        return launch_mc<MC_AVG>(depth, w, h, bp, pu_xy, mv0, mv1, n, k, stream);
    for (int pl = 0; pl < 3; pl++)
    {
        // predict.cpp:143-156 (wv1.shift = wv0.shift) and addWeightBi :425-430
        k.w0[pl] = wp0[pl].inputWeight;
        k.w1[pl] = wp1[pl].inputWeight;
        k.offset[pl] = wp0[pl].inputOffset * (1 << (depth - 8)) + wp1[pl].inputOffset * (1 << (depth - 8));
        k.shift[pl] = wp0[pl].log2WeightDenom + shiftNum + 1;
        k.round[pl] = 1 << (k.shift[pl] - 1);
    }
    return launch_mc<MC_WBI>(depth, w, h, bp, pu_xy, mv0, mv1, n, k, stream);
}
