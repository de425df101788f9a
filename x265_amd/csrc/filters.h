// filters.h — DCT-IF building blocks shared by interp.hip and the fused motion search (motion.hip).
// Semantics follow source/common/ipfilter.cpp:79-369 of the reference (cited per kernel).
#pragma once
#include "common.h"

namespace xh {

enum { IF_HPP = X265HIP_IF_HPP, IF_HPS = X265HIP_IF_HPS, IF_VPP = X265HIP_IF_VPP, IF_VPS = X265HIP_IF_VPS,
       IF_VSP = X265HIP_IF_VSP, IF_VSS = X265HIP_IF_VSS, IF_HVPP = X265HIP_IF_HVPP };

// exact-length row segment load (no read past the last element the reference itself would touch)
template <int L, typename P>
__device__ __forceinline__ void load_span(const P* p, int* v)
{
    constexpr int Q = L / 4;
#pragma unroll
    for (int q = 0; q < Q; q++)
        load4(p + 4 * q, v + 4 * q);
#pragma unroll
    for (int i = 4 * Q; i < L; i++)
        v[i] = (int)p[i];
}
template <int V, typename P>
__device__ __forceinline__ void store_span(P* p, const int* v)
{
    if (V == 4)
        store4(p, v);
    else
    {
#pragma unroll
        for (int i = 0; i < V; i++)
            p[i] = (P)v[i];
    }
}

// rounding / range stage of each filter flavour.  `isPixelOut`: clip to [0, maxVal]; else plain int16 store.
struct Stage
{
    int offset, shift, maxVal;
    bool clip;
};
__device__ __forceinline__ int finish(int sum, const Stage& s)
{
    int v = (int)(int16_t)((sum + s.offset) >> s.shift);
    if (s.clip)
        v = v < 0 ? 0 : (v > s.maxVal ? s.maxVal : v);
    return v;
}
__host__ __device__ inline Stage stage_for(int kind, int depth)
{
    const int headRoom = 14 - depth;
    Stage s;
    s.maxVal = (1 << depth) - 1;
    switch (kind)
    {
    case IF_HPP: case IF_VPP: s.offset = 32; s.shift = 6; s.clip = true; break;                           // ipfilter.cpp:83-84, :168-169
    case IF_HPS: case IF_VPS: s.shift = 6 - headRoom; s.offset = (int)((unsigned)-8192 << s.shift); s.clip = false; break;   // :124-126, :208-210
    case IF_VSP: s.shift = 6 + headRoom; s.offset = (1 << (s.shift - 1)) + (8192 << 6); s.clip = true; break;            // :244-246
    default: s.shift = 6; s.offset = 0; s.clip = false; break;                                             // vss :288
    }
    return s;
}

// One quad (4 horizontally adjacent samples) of a 4:2:0 chroma prediction at eighth-pel vector (mvx, mvy); `r` = the quad at vector (0,0).
// 4-tap H to the 14-bit intermediate, 4-tap V back to pixels.  This single hv form reproduces the reference's four cases exactly (copy /
// filter_hpp / filter_vpp / filter_hps + filter_vsp, motion.cpp:1618-1658, predict.cpp:328-362): with a zero fraction the filter is
// {0, 64, 0, 0}, the intermediate is 64 p - offset without loss, and (64 (S >> s1) + round) >> s2 == (S + 32) >> 6 because s1 + s2 = 12 and
// the inner floor nests inside the outer one.
template <typename P>
__device__ __forceinline__ void chroma_quad_hv(const P* r, int strideC, int mvx, int mvy, int depth, int out[4])
{
    const int xF = mvx & 7, yF = mvy & 7;
    const P* base = r + (mvy >> 3) * strideC + (mvx >> 3);
    const Stage s1 = stage_for(IF_HPS, depth), s2 = stage_for(IF_VSP, depth);
    int c1[4], c2[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { c1[i] = kChromaFilter[xF][i]; c2[i] = kChromaFilter[yF][i]; }
    int sum[4] = { 0, 0, 0, 0 };
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
        int v[8];
        const P* row = base + (k - 1) * strideC - 1;
        load4(row, v);
        load4(row + 4, v + 4);
#pragma unroll
        for (int o = 0; o < 4; o++)
        {
            const int h = finish(v[o] * c1[0] + v[o + 1] * c1[1] + v[o + 2] * c1[2] + v[o + 3] * c1[3], s1);
            sum[o] += h * c2[k];
        }
    }
#pragma unroll
    for (int o = 0; o < 4; o++) out[o] = finish(sum[o], s2);
}

// |H4 D H4^T| contribution of one lane holding one row (4 samples) of a 4x4 tile whose four rows sit in one DPP quad
__device__ __forceinline__ int quad_row_hadamard_abs(const int d[4], bool hi1, bool hi2)
{
    const int s01 = d[0] + d[1], e01 = d[0] - d[1], s23 = d[2] + d[3], e23 = d[2] - d[3];
    int m[4] = { s01 + s23, s01 - s23, e01 + e23, e01 - e23 };
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        const int pr = __builtin_amdgcn_mov_dpp(m[i], 0xB1, 0xF, 0xF, true);       // quad_perm [1,0,3,2]
        m[i] = hi1 ? pr - m[i] : m[i] + pr;
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        const int pr = __builtin_amdgcn_mov_dpp(m[i], 0x4E, 0xF, 0xF, true);       // quad_perm [2,3,0,1]
        m[i] = hi2 ? pr - m[i] : m[i] + pr;
    }
    return iabs(m[0]) + iabs(m[1]) + iabs(m[2]) + iabs(m[3]);
}

} // namespace xh
