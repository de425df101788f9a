// tiles.h — 4x4-tile building blocks shared by the comparison kernels (pixel.hip), the fused motion search
// (motion.hip) and the residual chain (frame.hip): packed SAD, tile differences, 4x4 Hadamard and the quad-level
// 8x8 Hadamard.  Semantics follow source/common/pixel.cpp:40-376 of the reference (cited per kernel).
#pragma once
#include "common.h"

namespace xh {

// ---- per-tile primitives ------------------------------------------------------------------------------------
__device__ __forceinline__ int tile_sad(const uint8_t* a, int64_t sa, const uint8_t* b, int64_t sb)
{
    unsigned s = 0;
#pragma unroll
    for (int y = 0; y < 4; y++)
        s = __builtin_amdgcn_sad_u8(ld_unaligned<uint32_t>(a + y * sa), ld_unaligned<uint32_t>(b + y * sb), s);
    return (int)s;
}
__device__ __forceinline__ int tile_sad(const uint16_t* a, int64_t sa, const uint16_t* b, int64_t sb)
{
    unsigned s = 0;
#pragma unroll
    for (int y = 0; y < 4; y++)
    {
        uint2 x = ld_unaligned<uint2>(a + y * sa), z = ld_unaligned<uint2>(b + y * sb);
        s = __builtin_amdgcn_sad_u16(x.x, z.x, s);
        s = __builtin_amdgcn_sad_u16(x.y, z.y, s);
    }
    return (int)s;
}

template <typename PA, typename PB>
__device__ __forceinline__ void tile_diff(const PA* a, int64_t sa, const PB* b, int64_t sb, int d[16])
{
#pragma unroll
    for (int y = 0; y < 4; y++)
    {
        int va[4], vb[4];
        load4(a + y * sa, va);
        load4(b + y * sb, vb);
#pragma unroll
        for (int x = 0; x < 4; x++)
            d[4 * y + x] = va[x] - vb[x];
    }
}
template <typename PA>
__device__ __forceinline__ void tile_load(const PA* a, int64_t sa, int d[16])
{
#pragma unroll
    for (int y = 0; y < 4; y++)
        load4(a + y * sa, &d[4 * y]);
}

// in-place 4x4 Hadamard (rows then columns), no normalisation: m <- H4 * m * H4^T
__device__ __forceinline__ void hadamard4x4(int m[16])
{
#pragma unroll
    for (int y = 0; y < 4; y++)
    {
        int s01 = m[4 * y] + m[4 * y + 1], d01 = m[4 * y] - m[4 * y + 1];
        int s23 = m[4 * y + 2] + m[4 * y + 3], d23 = m[4 * y + 2] - m[4 * y + 3];
        m[4 * y] = s01 + s23; m[4 * y + 1] = s01 - s23; m[4 * y + 2] = d01 + d23; m[4 * y + 3] = d01 - d23;
    }
#pragma unroll
    for (int x = 0; x < 4; x++)
    {
        int s01 = m[x] + m[4 + x], d01 = m[x] - m[4 + x];
        int s23 = m[8 + x] + m[12 + x], d23 = m[8 + x] - m[12 + x];
        m[x] = s01 + s23; m[4 + x] = s01 - s23; m[8 + x] = d01 + d23; m[12 + x] = d01 - d23;
    }
}
__device__ __forceinline__ int abs_sum16(const int m[16])
{
    int s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++)
        s += iabs(m[i]);
    return s;
}
// 2x2 butterfly across the DPP quad that holds the four quadrants of one 8x8, then |.| sum over the whole 8x8.
// Every lane of the quad returns the un-rounded 8x8 Hadamard magnitude sum (pixel.cpp:299-334 `_sa8d_8x8`).
__device__ __forceinline__ int quad_sa8d_raw(int m[16], int lane)
{
    const bool hi1 = lane & 1, hi2 = lane & 2;
#pragma unroll
    for (int i = 0; i < 16; i++)
    {
        int p = quad_xor1(m[i]);
        m[i] = hi1 ? p - m[i] : m[i] + p;
    }
#pragma unroll
    for (int i = 0; i < 16; i++)
    {
        int p = quad_xor2(m[i]);
        m[i] = hi2 ? p - m[i] : m[i] + p;
    }
    return quad_sum(abs_sum16(m));
}


} // namespace xh
