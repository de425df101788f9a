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


} // namespace xh
