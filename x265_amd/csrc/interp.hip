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
#include <cstdlib>
#include "filters.h"
#include "internal.h"

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

// ======================================================================================================================
// Sub-pel planes: all 16 quarter-pel phases of a whole reference picture, computed once per reference
// ======================================================================================================================
// MotionEstimate::subpelCompare (motion.cpp:1571-1600) filters a W x H block with luma_hpp / luma_vpp / luma_hvpp for every
// sub-pel candidate.  Those filters are position-independent per-pixel functions of the reference picture, so a device with
// 288 GB of HBM computes them ONCE per reference picture: plane[yFrac*4 + xFrac](x, y) = the value the reference's filter
// produces for that pixel (plane 0 = the picture itself).  16 planes of a padded 1080p picture are 43 MB.  The motion search
// then reads sub-pel candidates as plain blocks: no filtering in its serial chain at all.
namespace xh {

template <typename P>
__global__ __launch_bounds__(256) void subpel_planes_kernel(const P* __restrict__ ref, int64_t stride, P* __restrict__ planes, int64_t planeElems,
                                                            int x0, int y0, int x1, int y1, int depth)
{
    // buffer bounds: the computed region is inset by 4 from the padded buffer on every side (see the launcher)
    const int xlo = x0 - 4, xhi = x1 + 4, ylo = y0 - 4, yhi = y1 + 4;
    constexpr int TW = 64, TH = 16, IW = TW + 8, IH = TH + 7;          // IW padded to 72 (71 needed)
    __shared__ __attribute__((aligned(16))) P in[IH][IW];
    __shared__ __attribute__((aligned(16))) int16_t im[3][IH][TW];
    const int tilesX = (x1 - x0 + TW - 1) / TW;
    const int bx = x0 + (blockIdx.x % tilesX) * TW, by = y0 + (blockIdx.x / tilesX) * TH;
    const int t = threadIdx.x;
    const Stage sHpp = stage_for(IF_HPP, depth), sHps = stage_for(IF_HPS, depth), sVpp = stage_for(IF_VPP, depth), sVsp = stage_for(IF_VSP, depth);
    // ---- stage the input tile: rows by-3 .. by+TH+3, cols bx-3 .. bx+TW+4 (quads of 4; 18 quads per row)
    for (int i = t; i < IH * (IW / 4); i += 256)
    {
        const int r = i / (IW / 4), q = i % (IW / 4);
        int v[4];
        // tiles overhang the computed region at the right / bottom: clamp the READ position into the buffer (the clamped
        // values only feed outputs that are masked out below)
        const int yy = min(max(by - 3 + r, ylo), yhi - 1), xs = bx - 3 + 4 * q;
        const P* row = ref + (int64_t)yy * stride;
        if (xs >= xlo && xs + 3 < xhi)
            load4(row + xs, v);
        else
        {
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = (int)row[min(max(xs + e, xlo), xhi - 1)];
        }
        store4(&in[r][4 * q], v);
    }
    __syncthreads();
    // ---- phase A: horizontal sums for xFrac = 1..3 on every staged row; hps intermediates to LDS, hpp planes out
    for (int i = t; i < 3 * IH * (TW / 4); i += 256)
    {
        const int xf = 1 + i / (IH * (TW / 4)), rem = i % (IH * (TW / 4)), r = rem / (TW / 4), q = rem % (TW / 4);
        int v[11], hps[4], hpp[4];
        load_span<11>(&in[r][4 * q], v);                               // in[][c] holds column bx-3+c
#pragma unroll
        for (int o = 0; o < 4; o++)
        {
            int sum = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) sum += v[o + k] * kLumaFilter[xf][k];
            hps[o] = finish(sum, sHps);
            hpp[o] = finish(sum, sHpp);
        }
        store4(&im[xf - 1][r][4 * q], hps);
        const int y = by + r - 3, x = bx + 4 * q;
        if (r >= 3 && r < 3 + TH && y < y1 && x < x1)
            store4(planes + (int64_t)xf * planeElems + (int64_t)y * stride + x, hpp);
    }
    __syncthreads();
    // ---- phase B: plane 0, vertical-only planes and the nine hv planes for one output quad per thread
    {
        const int r = t / (TW / 4), q = t % (TW / 4);
        const int y = by + r, x = bx + 4 * q;
        if (y < y1 && x < x1)
        {
            int px[8][4];
#pragma unroll
            for (int k = 0; k < 8; k++) load4(&in[r + k][4 * q + 3], px[k]);      // rows y-3..y+4 at columns x..x+3
            store4(planes + (int64_t)y * stride + x, px[3]);
#pragma unroll
            for (int yf = 1; yf < 4; yf++)
            {
                int out[4];
#pragma unroll
                for (int o = 0; o < 4; o++)
                {
                    int sum = 0;
#pragma unroll
                    for (int k = 0; k < 8; k++) sum += px[k][o] * kLumaFilter[yf][k];
                    out[o] = finish(sum, sVpp);
                }
                store4(planes + (int64_t)(yf * 4) * planeElems + (int64_t)y * stride + x, out);
            }
#pragma unroll
            for (int xf = 1; xf < 4; xf++)
            {
                int h[8][4];
#pragma unroll
                for (int k = 0; k < 8; k++) load4(&im[xf - 1][r + k][4 * q], h[k]);
#pragma unroll
                for (int yf = 1; yf < 4; yf++)
                {
                    int out[4];
#pragma unroll
                    for (int o = 0; o < 4; o++)
                    {
                        int sum = 0;
#pragma unroll
                        for (int k = 0; k < 8; k++) sum += h[k][o] * kLumaFilter[yf][k];
                        out[o] = finish(sum, sVsp);
                    }
                    store4(planes + (int64_t)(yf * 4 + xf) * planeElems + (int64_t)y * stride + x, out);
                }
            }
        }
    }
}


// ---- 8-bit specialisation: the same planes with packed dot products ---------------------------------------------------------
// Pixels are staged biased by -128 (xor 0x80) so that v_dot4_i32_i8 applies: sum(c * p) = dot4(c, p - 128) + 128 * sum(c) and
// sum(c) = 64 for every phase, i.e. "+ 8192" — which is exactly the -8192 offset of the 14-bit intermediate (ipfilter.cpp:124-126),
// so at 8 bit the hps value IS the biased dot product.  Horizontal windows come from v_alignbyte_b32, the vertical pass
// transposes 4x4 bytes with v_perm_b32, and the second (vertical, 14-bit) stage runs on v_dot2_i32_i16 over row pairs.
constexpr int pack4c(int a, int b, int c, int d) { return (a & 255) | ((b & 255) << 8) | ((c & 255) << 16) | (int)((unsigned)(d & 255) << 24); }
constexpr int pack2c(int a, int b) { return (a & 0xffff) | (int)((unsigned)(b & 0xffff) << 16); }
struct LumaPacked
{
    int lo4[4], hi4[4];          // taps 0-3 / 4-7 as signed bytes, per phase
    int pr[4][4];                // tap pairs (0,1) (2,3) (4,5) (6,7) as signed shorts, per phase
};
constexpr LumaPacked make_luma_packed()
{
    const int f[4][8] = { { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 }, { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
    LumaPacked t{};
    for (int p = 0; p < 4; p++)
    {
        t.lo4[p] = pack4c(f[p][0], f[p][1], f[p][2], f[p][3]);
        t.hi4[p] = pack4c(f[p][4], f[p][5], f[p][6], f[p][7]);
        for (int k = 0; k < 4; k++)
            t.pr[p][k] = pack2c(f[p][2 * k], f[p][2 * k + 1]);
    }
    return t;
}
constexpr LumaPacked kLumaPk = make_luma_packed();

typedef short short2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int dot4(int a, int b, int acc) { return __builtin_amdgcn_sdot4(a, b, acc, false); }
__device__ __forceinline__ int dot2(int a, int b, int acc)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b), acc, false);
}
// four values >> SH, saturated to u8 and packed: gfx950's v_ashr_pk_u8_i32 does two at a time; v_perm_b32 joins the halves.
// (Written with the builtin on purpose: left to pattern matching, this compiler emits the same instruction WITHOUT masking the
// undefined upper half and corrupts bytes 2-3 whenever the first value is negative.)
template <int SH>
__device__ __forceinline__ uint32_t sat_pack4(int a, int b, int c, int d)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_ashr_pk_u8_i32(a, b, SH), hi = (uint32_t)__builtin_amdgcn_ashr_pk_u8_i32(c, d, SH);
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
}

__global__ __launch_bounds__(256) void subpel_planes8_kernel(const uint8_t* __restrict__ ref, int64_t stride, uint8_t* __restrict__ planes, int64_t planeElems,
                                                             int x0, int y0, int x1, int y1)
{
    const int xlo = x0 - 4, xhi = x1 + 4, ylo = y0 - 4, yhi = y1 + 4;
    constexpr int TW = 64, TH = 16, IWD = 18, IH = TH + 7;              // staged row: 18 dwords = columns bx-4 .. bx+67
    __shared__ uint32_t in[IH][IWD];                                     // biased pixels (p ^ 0x80)
    __shared__ __attribute__((aligned(8))) int16_t im[3][IH][TW];
    const int tilesX = (x1 - x0 + TW - 1) / TW;
    const int bx = x0 + (blockIdx.x % tilesX) * TW, by = y0 + (blockIdx.x / tilesX) * TH;
    const int t = threadIdx.x;
    for (int i = t; i < IH * IWD; i += 256)
    {
        const int r = i / IWD, q = i % IWD;
        const int yy = min(max(by - 3 + r, ylo), yhi - 1), xs = bx - 4 + 4 * q;
        const uint8_t* row = ref + (int64_t)yy * stride;
        uint32_t v;
        if (xs >= xlo && xs + 3 < xhi)
            v = ld_unaligned<uint32_t>(row + xs);
        else
        {
            v = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) v |= (uint32_t)row[min(max(xs + e, xlo), xhi - 1)] << (8 * e);
        }
        in[r][q] = v ^ 0x80808080u;
    }
    __syncthreads();
    // ---- phase A: the three horizontal phases of one staged row quad per item
    for (int i = t; i < IH * (TW / 4); i += 256)
    {
        const int r = i / (TW / 4), q = i % (TW / 4);
        const uint32_t d0 = in[r][q], d1 = in[r][q + 1], d2 = in[r][q + 2];
        // output o (column bx + 4q + o) needs bytes 4q + 1 + o .. + 8 of the staged row
        uint32_t wl[4], wh[4];
        wl[0] = __builtin_amdgcn_alignbyte(d1, d0, 1); wh[0] = __builtin_amdgcn_alignbyte(d2, d1, 1);
        wl[1] = __builtin_amdgcn_alignbyte(d1, d0, 2); wh[1] = __builtin_amdgcn_alignbyte(d2, d1, 2);
        wl[2] = __builtin_amdgcn_alignbyte(d1, d0, 3); wh[2] = __builtin_amdgcn_alignbyte(d2, d1, 3);
        wl[3] = d1; wh[3] = d2;
        const int y = by + r - 3, x = bx + 4 * q;
        const bool out = r >= 3 && r < 3 + TH && y < y1 && x < x1;
#pragma unroll
        for (int xf = 1; xf < 4; xf++)
        {
            int h[4];
#pragma unroll
            for (int o = 0; o < 4; o++)
                h[o] = dot4((int)wl[o], kLumaPk.lo4[xf], dot4((int)wh[o], kLumaPk.hi4[xf], 0));     // = true sum - 8192 = the hps value
            uint2 hv;
            hv.x = (uint32_t)(h[0] & 0xffff) | ((uint32_t)h[1] << 16);
            hv.y = (uint32_t)(h[2] & 0xffff) | ((uint32_t)h[3] << 16);
            *reinterpret_cast<uint2*>(&im[xf - 1][r][4 * q]) = hv;
            if (out)
                st_unaligned<uint32_t>(planes + (int64_t)xf * planeElems + (int64_t)y * stride + x,
                                       sat_pack4<6>(h[0] + 8224, h[1] + 8224, h[2] + 8224, h[3] + 8224));           // (sum + 32) >> 6, clipped
        }
    }
    __syncthreads();
    // ---- phase B: plane 0, the vertical planes and the nine hv planes of one output quad per thread
    {
        const int r = t / (TW / 4), q = t % (TW / 4);
        const int y = by + r, x = bx + 4 * q;
        if (y < y1 && x < x1)
        {
            uint32_t R[8];
#pragma unroll
            for (int k = 0; k < 8; k++) R[k] = in[r + k][q + 1];                // rows y-3 .. y+4 at columns x .. x+3 (biased)
            st_unaligned<uint32_t>(planes + (int64_t)y * stride + x, R[3] ^ 0x80808080u);
            uint32_t cl[4], ch[4];                                               // column o: rows 0-3 / rows 4-7 as bytes
            {
                const uint32_t t0 = __builtin_amdgcn_perm(R[1], R[0], 0x05010400u), t1 = __builtin_amdgcn_perm(R[1], R[0], 0x07030602u);
                const uint32_t u0 = __builtin_amdgcn_perm(R[3], R[2], 0x05010400u), u1 = __builtin_amdgcn_perm(R[3], R[2], 0x07030602u);
                cl[0] = __builtin_amdgcn_perm(u0, t0, 0x05040100u); cl[1] = __builtin_amdgcn_perm(u0, t0, 0x07060302u);
                cl[2] = __builtin_amdgcn_perm(u1, t1, 0x05040100u); cl[3] = __builtin_amdgcn_perm(u1, t1, 0x07060302u);
                const uint32_t t2 = __builtin_amdgcn_perm(R[5], R[4], 0x05010400u), t3 = __builtin_amdgcn_perm(R[5], R[4], 0x07030602u);
                const uint32_t u2 = __builtin_amdgcn_perm(R[7], R[6], 0x05010400u), u3 = __builtin_amdgcn_perm(R[7], R[6], 0x07030602u);
                ch[0] = __builtin_amdgcn_perm(u2, t2, 0x05040100u); ch[1] = __builtin_amdgcn_perm(u2, t2, 0x07060302u);
                ch[2] = __builtin_amdgcn_perm(u3, t3, 0x05040100u); ch[3] = __builtin_amdgcn_perm(u3, t3, 0x07060302u);
            }
#pragma unroll
            for (int yf = 1; yf < 4; yf++)
            {
                int o4[4];
#pragma unroll
                for (int o = 0; o < 4; o++)
                    o4[o] = dot4((int)cl[o], kLumaPk.lo4[yf], dot4((int)ch[o], kLumaPk.hi4[yf], 8192 + 32));
                st_unaligned<uint32_t>(planes + (int64_t)(yf * 4) * planeElems + (int64_t)y * stride + x, sat_pack4<6>(o4[0], o4[1], o4[2], o4[3]));
            }
#pragma unroll
            for (int xf = 1; xf < 4; xf++)
            {
                uint32_t pr[4][4];                                               // [row pair][column]: (h[2j][o], h[2j+1][o])
#pragma unroll
                for (int j = 0; j < 4; j++)
                {
                    const uint2 a = *reinterpret_cast<const uint2*>(&im[xf - 1][r + 2 * j][4 * q]);
                    const uint2 b = *reinterpret_cast<const uint2*>(&im[xf - 1][r + 2 * j + 1][4 * q]);
                    pr[j][0] = __builtin_amdgcn_perm(b.x, a.x, 0x05040100u); pr[j][1] = __builtin_amdgcn_perm(b.x, a.x, 0x07060302u);
                    pr[j][2] = __builtin_amdgcn_perm(b.y, a.y, 0x05040100u); pr[j][3] = __builtin_amdgcn_perm(b.y, a.y, 0x07060302u);
                }
#pragma unroll
                for (int yf = 1; yf < 4; yf++)
                {
                    int o4[4];
#pragma unroll
                    for (int o = 0; o < 4; o++)
                    {
                        int sum = (1 << 11) + (8192 << 6);                               // interp_vert_sp at depth 8: offset, shift 12 (ipfilter.cpp:244-246)
#pragma unroll
                        for (int j = 0; j < 4; j++) sum = dot2((int)pr[j][o], kLumaPk.pr[yf][j], sum);
                        o4[o] = sum;
                    }
                    st_unaligned<uint32_t>(planes + (int64_t)(yf * 4 + xf) * planeElems + (int64_t)y * stride + x, sat_pack4<12>(o4[0], o4[1], o4[2], o4[3]));
                }
            }
        }
    }
}

} // namespace xh

namespace xh {
// the region [x0, x1) x [y0, y1) (picture coordinates) of all 15 fractional planes + plane 0; reads rows y0 - 3 .. y1 + 3 and columns
// x0 - 4 .. x1 + 3 of the reference.  The whole picture is one call; a band of rows is the same launch with a smaller y range (refpic.hip).
int build_subpel_rows(int depth, const void* refOrigin, int64_t stride, int x0, int x1, int y0, int y1, void* planesOrigin, int64_t planeElems, hipStream_t st)
{
    if (x1 <= x0 || y1 <= y0) return X265HIP_OK;
    const int tilesX = (x1 - x0 + 63) / 64, tilesY = (y1 - y0 + 15) / 16;
    dim3 grid(tilesX * tilesY), block(256);
    static const bool generic8 = getenv("X265HIP_PLANES_GENERIC") != nullptr;
    if (depth == 8 && !generic8)
        hipLaunchKernelGGL(subpel_planes8_kernel, grid, block, 0, st, (const uint8_t*)refOrigin, stride, (uint8_t*)planesOrigin, planeElems, x0, y0, x1, y1);
    else if (depth == 8)
        hipLaunchKernelGGL((subpel_planes_kernel<uint8_t>), grid, block, 0, st, (const uint8_t*)refOrigin, stride, (uint8_t*)planesOrigin, planeElems, x0, y0, x1, y1,
                           depth);
    else
        hipLaunchKernelGGL((subpel_planes_kernel<uint16_t>), grid, block, 0, st, (const uint16_t*)refOrigin, stride, (uint16_t*)planesOrigin, planeElems, x0, y0, x1,
                           y1, depth);
    XH_LAUNCH_CHECK("subpel_planes_kernel");
    return X265HIP_OK;
}
} // namespace xh

extern "C" int x265hip_build_subpel_planes(int depth, const void* refOrigin, int64_t stride, int picW, int picH, int marginX, int marginY,
                                           void* planesOrigin, int64_t planeElems, void* stream)
{
    XH_CHECK_DEV();
    if (!valid_depth(depth) || picW < 8 || picH < 8 || marginX < 8 || marginY < 8 || (picW & 3) || (marginX & 3))
        return set_error(X265HIP_EINVAL, "build_subpel_planes: depth %d pic %dx%d margins %d,%d", depth, picW, picH, marginX, marginY);
    // the 8-tap support needs 3 / 4 pixels around every output: compute everything except the outermost 4 columns / rows
    return xh::build_subpel_rows(depth, refOrigin, stride, -marginX + 4, picW + marginX - 4, -marginY + 4, picH + marginY - 4, planesOrigin, planeElems,
                                 as_stream(stream));
}

// ---- prediction out of the quarter-pel planes: predInterLumaPixel (predict.cpp:245-266) becomes a phase-selected block copy ----
namespace xh {
template <typename P>
__global__ __launch_bounds__(256) void pred_from_planes_kernel(const P* __restrict__ planes, int64_t planeElems, int64_t sR, P* __restrict__ dst, int64_t sD,
                                                               const int32_t* __restrict__ pu_xy, const int32_t* __restrict__ qmv, int size, int n)
{
    const int qx = size >> 2, per = qx * size;
    const long long total = (long long)n * per;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x)
    {
        const int pu = (int)(idx / per), p = (int)(idx - (long long)pu * per), y = p / qx, x = (p - y * qx) * 4;
        const int bx = pu_xy[2 * pu], by = pu_xy[2 * pu + 1], mx = qmv[2 * pu], my = qmv[2 * pu + 1];
        const P* s = planes + (int64_t)((my & 3) * 4 + (mx & 3)) * planeElems + (int64_t)(by + (my >> 2) + y) * sR + bx + (mx >> 2) + x;
        int v[4];
        load4(s, v);
        store4(dst + (int64_t)(by + y) * sD + bx + x, v);
    }
}

int pred_from_planes(int depth, int size, const void* planes, int64_t planeElems, int64_t strideR, void* dst, int64_t strideD,
                     const int32_t* pu_xy, const int32_t* qmv, int n, hipStream_t st)
{
    if (!n) return X265HIP_OK;
    const long long total = (long long)n * (size / 4) * size;
    dim3 grid(grid_for((total + 255) / 256)), block(256);
    if (depth == 8)
        hipLaunchKernelGGL((pred_from_planes_kernel<uint8_t>), grid, block, 0, st, (const uint8_t*)planes, planeElems, strideR, (uint8_t*)dst, strideD, pu_xy, qmv, size, n);
    else
        hipLaunchKernelGGL((pred_from_planes_kernel<uint16_t>), grid, block, 0, st, (const uint16_t*)planes, planeElems, strideR, (uint16_t*)dst, strideD, pu_xy, qmv, size, n);
    XH_LAUNCH_CHECK("pred_from_planes_kernel");
    return X265HIP_OK;
}
} // namespace xh

// ======================================================================================================================
// Predict::predInterChromaPixel (reference: source/common/predict.cpp:306-352), 4:2:0
// ======================================================================================================================
// One lane = one row quad of a chroma block; the four cases (copy / filter_hpp / filter_vpp / filter_hps(rowExt)+filter_vsp) are
// selected per PU by the eighth-pel fraction of the chroma vector exactly as the reference does.  Cb and Cr in one launch.
namespace xh {

template <typename P>
__global__ __launch_bounds__(256) void pred_chroma_kernel(const P* __restrict__ refCb, const P* __restrict__ refCr, int64_t sR,
                                                          P* __restrict__ dstCb, P* __restrict__ dstCr, int64_t sD,
                                                          const int32_t* __restrict__ pu_xy, const int32_t* __restrict__ qmv,
                                                          int cw, int ch, int n, int depth)
{
    const int qx = cw >> 2, per = qx * ch;
    const long long total = 2LL * n * per;
    const Stage sHpp = stage_for(IF_HPP, depth), sHps = stage_for(IF_HPS, depth), sVpp = stage_for(IF_VPP, depth), sVsp = stage_for(IF_VSP, depth);
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x)
    {
        const int plane = (int)(idx / ((long long)n * per));
        const long long rem = idx - (long long)plane * n * per;
        const int pu = (int)(rem / per), p = (int)(rem - (long long)pu * per), y = p / qx, x = (p - y * qx) * 4;
        // chroma block origin = luma PU origin / 2; chroma vector (1/8 pel) = luma quarter-pel vector for 4:2:0 (predict.cpp:311-312)
        const int bx = pu_xy[2 * pu] >> 1, by = pu_xy[2 * pu + 1] >> 1, mvx = qmv[2 * pu], mvy = qmv[2 * pu + 1];
        const int xFrac = mvx & 7, yFrac = mvy & 7;
        const P* s = (plane ? refCr : refCb) + (int64_t)(by + (mvy >> 3) + y) * sR + bx + (mvx >> 3) + x;
        P* d = (plane ? dstCr : dstCb) + (int64_t)(by + y) * sD + bx + x;
        int out[4];
        if (!(xFrac | yFrac))
            load4(s, out);
        else if (!yFrac)
        {
            int v[7];
            load_span<7>(s - 1, v);
#pragma unroll
            for (int o = 0; o < 4; o++)
            {
                int sum = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) sum += v[o + k] * kChromaFilter[xFrac][k];
                out[o] = finish(sum, sHpp);
            }
        }
        else if (!xFrac)
        {
            int sum[4] = { 0, 0, 0, 0 };
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                int v[4];
                load4(s + (int64_t)(k - 1) * sR, v);
#pragma unroll
                for (int o = 0; o < 4; o++) sum[o] += v[o] * kChromaFilter[yFrac][k];
            }
#pragma unroll
            for (int o = 0; o < 4; o++) out[o] = finish(sum[o], sVpp);
        }
        else
        {
            int sum[4] = { 0, 0, 0, 0 };
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                int v[7];
                load_span<7>(s + (int64_t)(k - 1) * sR - 1, v);
#pragma unroll
                for (int o = 0; o < 4; o++)
                {
                    int hs = 0;
#pragma unroll
                    for (int t = 0; t < 4; t++) hs += v[o + t] * kChromaFilter[xFrac][t];
                    sum[o] += finish(hs, sHps) * kChromaFilter[yFrac][k];
                }
            }
#pragma unroll
            for (int o = 0; o < 4; o++) out[o] = finish(sum[o], sVsp);
        }
        store4(d, out);
    }
}

} // namespace xh

extern "C" int x265hip_pred_inter_chroma_batch(int depth, int lumaW, int lumaH, const void* refCb, const void* refCr, int64_t strideR,
                                               void* dstCb, void* dstCr, int64_t strideD, const int32_t* pu_xy, const int32_t* qmv, int n,
                                               void* stream)
{
    XH_CHECK_DEV();
    if (!valid_depth(depth) || !valid_block(lumaW, lumaH) || (lumaW & 7) || (lumaH & 1) || n < 0)
        return set_error(X265HIP_EINVAL, "pred_inter_chroma: depth %d luma PU %dx%d n %d (4:2:0 chroma blocks must be a multiple of 4 wide)", depth, lumaW, lumaH, n);
    if (!n) return X265HIP_OK;
    const int cw = lumaW >> 1, ch = lumaH >> 1;
    const long long total = 2LL * n * (cw / 4) * ch;
    dim3 grid(grid_for((total + 255) / 256)), block(256);
    if (depth == 8)
        hipLaunchKernelGGL((pred_chroma_kernel<uint8_t>), grid, block, 0, as_stream(stream), (const uint8_t*)refCb, (const uint8_t*)refCr, strideR,
                           (uint8_t*)dstCb, (uint8_t*)dstCr, strideD, pu_xy, qmv, cw, ch, n, depth);
    else
        hipLaunchKernelGGL((pred_chroma_kernel<uint16_t>), grid, block, 0, as_stream(stream), (const uint16_t*)refCb, (const uint16_t*)refCr, strideR,
                           (uint16_t*)dstCb, (uint16_t*)dstCr, strideD, pu_xy, qmv, cw, ch, n, depth);
    XH_LAUNCH_CHECK("pred_chroma_kernel");
    return X265HIP_OK;
}
