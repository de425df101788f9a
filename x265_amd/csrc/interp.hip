// interp.hip — batched sub-pel interpolation (8-tap luma / 4-tap chroma DCT-IF) for gfx950.
//
// Reference semantics (bit-exact): source/common/ipfilter.cpp — interp_horiz_pp_c :79, interp_horiz_ps_c :120,
// interp_vert_pp_c :164, interp_vert_ps_c :205, interp_vert_sp_c :241, interp_vert_ss_c :284, interp_hv_pp_c :362;
// taps from common/constants.cpp:250-268; IF_INTERNAL_PREC 14, IF_FILTER_PREC 6, IF_INTERNAL_OFFS 8192
// (common/constants.h:68-70).
//
// One lane produces V (4, or 2 for the 2/6-wide chroma shapes) horizontally adjacent outputs of one row.  The
// separable hv filter keeps its 14-bit intermediate in LDS (one wave per block), never in HBM.
#include "common.h"
#include "filters.h"

namespace xh {


// ---- single-stage filters -----------------------------------------------------------------------------------------
template <typename PS, typename PD, int NT, bool HORIZ, int V>
__global__ __launch_bounds__(256) void interp_kernel(const PS* __restrict__ src, int64_t sS, PD* __restrict__ dst, int64_t sD,
                                                     const int32_t* __restrict__ offS, const int32_t* __restrict__ offD,
                                                     const int32_t* __restrict__ coeff, int n, int w, int h, int rowExt, Stage st)
{
    const int vecX = w / V;
    const int rows = rowExt ? h + NT - 1 : h;
    const int per = vecX * rows;
    const long long total = (long long)n * per;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x)
    {
        const int job = (int)(idx / per);
        const int p = (int)(idx - (long long)job * per);
        const int y = p / vecX, x = (p - y * vecX) * V;
        const int ci = coeff[job] & 15;
        int c[NT];
#pragma unroll
        for (int i = 0; i < NT; i++)
            c[i] = filter_tap<NT>(ci, i);
        const PS* s = src + offS[job] + (int64_t)(y - (rowExt ? NT / 2 - 1 : 0)) * sS + x;
        int out[V];
        if (HORIZ)
        {
            int v[V + NT - 1];
            load_span<V + NT - 1>(s - (NT / 2 - 1), v);
#pragma unroll
            for (int o = 0; o < V; o++)
            {
                int sum = 0;
#pragma unroll
                for (int i = 0; i < NT; i++)
                    sum += v[o + i] * c[i];
                out[o] = finish(sum, st);
            }
        }
        else
        {
            int sum[V];
#pragma unroll
            for (int o = 0; o < V; o++) sum[o] = 0;
#pragma unroll
            for (int i = 0; i < NT; i++)
            {
                int v[V];
                load_span<V>(s + (int64_t)(i - (NT / 2 - 1)) * sS, v);
#pragma unroll
                for (int o = 0; o < V; o++)
                    sum[o] += v[o] * c[i];
            }
#pragma unroll
            for (int o = 0; o < V; o++)
                out[o] = finish(sum[o], st);
        }
        store_span<V>(dst + offD[job] + (int64_t)y * sD + x, out);
    }
}

// ---- hv: hps (with row extension) into LDS, then vsp out of LDS — one wave per block ---------------------------------
template <typename P, int V>
__global__ __launch_bounds__(256) void interp_hv_kernel(const P* __restrict__ src, int64_t sS, P* __restrict__ dst, int64_t sD,
                                                        const int32_t* __restrict__ offS, const int32_t* __restrict__ offD,
                                                        const int32_t* __restrict__ coeff, int n, int w, int h, int depth)
{
    constexpr int NT = 8;
    __shared__ int16_t lds[4][(64 + NT - 1) * 64];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int16_t* im = lds[wv];
    const Stage s1 = stage_for(IF_HPS, depth), s2 = stage_for(IF_VSP, depth);
    const int vecX = w / V;
    const int wavesTotal = gridDim.x * 4;
    for (int job = blockIdx.x * 4 + wv; job < n; job += wavesTotal)
    {
        const int cx = coeff[job] & 15, cy = (coeff[job] >> 4) & 15;
        int c1[NT], c2[NT];
#pragma unroll
        for (int i = 0; i < NT; i++) { c1[i] = kLumaFilter[cx][i]; c2[i] = kLumaFilter[cy][i]; }
        const P* s = src + offS[job] - (NT / 2 - 1) * sS - (NT / 2 - 1);
        const int rows1 = h + NT - 1;
        for (int p = lane; p < rows1 * vecX; p += 64)
        {
            const int y = p / vecX, x = (p - y * vecX) * V;
            int v[V + NT - 1], out[V];
            load_span<V + NT - 1>(s + (int64_t)y * sS + x, v);
#pragma unroll
            for (int o = 0; o < V; o++)
            {
                int sum = 0;
#pragma unroll
                for (int i = 0; i < NT; i++) sum += v[o + i] * c1[i];
                out[o] = finish(sum, s1);
            }
#pragma unroll
            for (int o = 0; o < V; o++) im[y * w + x + o] = (int16_t)out[o];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);          // wave-private LDS tile: wave-local ordering is enough
        __builtin_amdgcn_wave_barrier();
        for (int p = lane; p < h * vecX; p += 64)
        {
            const int y = p / vecX, x = (p - y * vecX) * V;
            int out[V];
#pragma unroll
            for (int o = 0; o < V; o++)
            {
                int sum = 0;
#pragma unroll
                for (int i = 0; i < NT; i++) sum += (int)im[(y + i) * w + x + o] * c2[i];
                out[o] = finish(sum, s2);
            }
            store_span<V>(dst + offD[job] + (int64_t)y * sD + x, out);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
}

template <typename PS, typename PD, int NT, bool HORIZ>
static int launch_interp(const void* src, int64_t sS, void* dst, int64_t sD, const int32_t* offS, const int32_t* offD,
                         const int32_t* coeff, int n, int w, int h, int rowExt, Stage stg, hipStream_t st)
{
    const int V = (w & 3) ? 2 : 4;
    const long long total = (long long)n * (w / V) * (rowExt ? h + NT - 1 : h);
    dim3 grid(grid_for((total + 255) / 256)), block(256);
    if (V == 4)
        hipLaunchKernelGGL((interp_kernel<PS, PD, NT, HORIZ, 4>), grid, block, 0, st, (const PS*)src, sS, (PD*)dst, sD, offS, offD, coeff, n, w, h, rowExt, stg);
    else
        hipLaunchKernelGGL((interp_kernel<PS, PD, NT, HORIZ, 2>), grid, block, 0, st, (const PS*)src, sS, (PD*)dst, sD, offS, offD, coeff, n, w, h, rowExt, stg);
    XH_LAUNCH_CHECK("interp_kernel");
    return X265HIP_OK;
}

template <typename P, int NT>
static int dispatch_interp(int kind, int depth, int w, int h, const void* src, int64_t sS, void* dst, int64_t sD,
                           const int32_t* offS, const int32_t* offD, const int32_t* coeff, int flags, int n, hipStream_t st)
{
    const Stage stg = stage_for(kind, depth);
    switch (kind)
    {
    case IF_HPP: return launch_interp<P, P, NT, true>(src, sS, dst, sD, offS, offD, coeff, n, w, h, 0, stg, st);
    case IF_HPS: return launch_interp<P, int16_t, NT, true>(src, sS, dst, sD, offS, offD, coeff, n, w, h, flags & 1, stg, st);
    case IF_VPP: return launch_interp<P, P, NT, false>(src, sS, dst, sD, offS, offD, coeff, n, w, h, 0, stg, st);
    case IF_VPS: return launch_interp<P, int16_t, NT, false>(src, sS, dst, sD, offS, offD, coeff, n, w, h, 0, stg, st);
    case IF_VSP: return launch_interp<int16_t, P, NT, false>(src, sS, dst, sD, offS, offD, coeff, n, w, h, 0, stg, st);
    case IF_VSS: return launch_interp<int16_t, int16_t, NT, false>(src, sS, dst, sD, offS, offD, coeff, n, w, h, 0, stg, st);
    default: return set_error(X265HIP_EINVAL, "interp: kind %d", kind);
    }
}

} // namespace xh

using namespace xh;

extern "C" int x265hip_interp_batch(int kind, int taps, int depth, int w, int h, const void* src, int64_t strideS,
                                    void* dst, int64_t strideD, const int32_t* offS, const int32_t* offD,
                                    const int32_t* coeff, int flags, int n, void* stream)
{
    XH_CHECK_DEV();
    if (!valid_depth(depth) || !valid_block(w, h) || n < 0 || (taps != 4 && taps != 8) || kind < 0 || kind > IF_HVPP)
        return set_error(X265HIP_EINVAL, "interp: kind %d taps %d depth %d %dx%d n %d", kind, taps, depth, w, h, n);
    if (!n) return X265HIP_OK;
    hipStream_t st = as_stream(stream);
    if (kind == IF_HVPP)
    {
        if (taps != 8)
            return set_error(X265HIP_EINVAL, "interp: hv_pp is a luma (8-tap) slot (primitives.h:181)");
        dim3 grid(grid_for((n + 3) / 4)), block(256);
        const bool v4 = !(w & 3);
        if (depth == 8)
        {
            if (v4) hipLaunchKernelGGL((interp_hv_kernel<uint8_t, 4>), grid, block, 0, st, (const uint8_t*)src, strideS, (uint8_t*)dst, strideD, offS, offD, coeff, n, w, h, depth);
            else hipLaunchKernelGGL((interp_hv_kernel<uint8_t, 2>), grid, block, 0, st, (const uint8_t*)src, strideS, (uint8_t*)dst, strideD, offS, offD, coeff, n, w, h, depth);
        }
        else
        {
            if (v4) hipLaunchKernelGGL((interp_hv_kernel<uint16_t, 4>), grid, block, 0, st, (const uint16_t*)src, strideS, (uint16_t*)dst, strideD, offS, offD, coeff, n, w, h, depth);
            else hipLaunchKernelGGL((interp_hv_kernel<uint16_t, 2>), grid, block, 0, st, (const uint16_t*)src, strideS, (uint16_t*)dst, strideD, offS, offD, coeff, n, w, h, depth);
        }
        XH_LAUNCH_CHECK("interp_hv_kernel");
        return X265HIP_OK;
    }
    if (depth == 8)
        return taps == 8 ? dispatch_interp<uint8_t, 8>(kind, depth, w, h, src, strideS, dst, strideD, offS, offD, coeff, flags, n, st)
                         : dispatch_interp<uint8_t, 4>(kind, depth, w, h, src, strideS, dst, strideD, offS, offD, coeff, flags, n, st);
    return taps == 8 ? dispatch_interp<uint16_t, 8>(kind, depth, w, h, src, strideS, dst, strideD, offS, offD, coeff, flags, n, st)
                     : dispatch_interp<uint16_t, 4>(kind, depth, w, h, src, strideS, dst, strideD, offS, offD, coeff, flags, n, st);
}
