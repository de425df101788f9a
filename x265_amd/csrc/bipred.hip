// bipred.hip — bi-predictive motion compensation: the unweighted B-slice branch of Predict::motionCompensation (reference
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

template <typename P>
__global__ __launch_bounds__(256) void pred_bi_kernel(BiPlanes bp, const int32_t* __restrict__ pu_xy, const int32_t* __restrict__ mv0, const int32_t* __restrict__ mv1,
                                                      int w, int h, int n, int depth)
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
        const int ax = mv0[2 * pu], ay = mv0[2 * pu + 1], cx = mv1[2 * pu], cy = mv1[2 * pu + 1];
        if (it < lq)
        {
            const int y = it / (w >> 2), x = (it % (w >> 2)) * 4;
            int a[4], b[4], o[4];
            short_pred<P, 8, 4>((const P*)bp.ref[0][0] + (int64_t)(by + y) * bp.strideRY + bx + x, bp.strideRY, ax, ay, 2, depth, a);
            short_pred<P, 8, 4>((const P*)bp.ref[1][0] + (int64_t)(by + y) * bp.strideRY + bx + x, bp.strideRY, cx, cy, 2, depth, b);
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                const int v = (a[i] + b[i] + offset) >> shift;
                o[i] = v < 0 ? 0 : (v > maxv ? maxv : v);
            }
            store4((P*)bp.dst[0] + (int64_t)(by + y) * bp.strideDY + bx + x, o);
        }
        else
        {
            const int k = it - lq, pl = 1 + k / cp, kk = k % cp;
            const int y = kk / (w >> 2), x = (kk % (w >> 2)) * 2;
            const int64_t ro = (int64_t)((by >> 1) + y) * bp.strideRC + (bx >> 1) + x;
            int a[2], b[2];
            short_pred<P, 4, 2>((const P*)bp.ref[0][pl] + ro, bp.strideRC, ax, ay, 3, depth, a);
            short_pred<P, 4, 2>((const P*)bp.ref[1][pl] + ro, bp.strideRC, cx, cy, 3, depth, b);
            P* d = (P*)bp.dst[pl] + (int64_t)((by >> 1) + y) * bp.strideDC + (bx >> 1) + x;
#pragma unroll
            for (int i = 0; i < 2; i++)
            {
                const int v = (a[i] + b[i] + offset) >> shift;
                d[i] = (P)(v < 0 ? 0 : (v > maxv ? maxv : v));
            }
        }
    }
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
    const long long total = (long long)n * ((w / 4) * h + 2 * (w / 4) * (h / 2));
    dim3 grid(grid_for((total + 255) / 256)), block(256);
    if (depth == 8)
        hipLaunchKernelGGL((pred_bi_kernel<uint8_t>), grid, block, 0, as_stream(stream), bp, pu_xy, mv0, mv1, w, h, n, depth);
    else
        hipLaunchKernelGGL((pred_bi_kernel<uint16_t>), grid, block, 0, as_stream(stream), bp, pu_xy, mv0, mv1, w, h, n, depth);
    XH_LAUNCH_CHECK("pred_bi_kernel");
    return X265HIP_OK;
}
