// dctcore.h — the MFMA transform pass shared by transform.hip (batched dct/idct) and frame.hip (fused residual chain).
// See the header comment of transform.hip for the int16 = 256*hi + lo split and the block-diagonal operand.
#pragma once
#include "common.h"

namespace xh {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// ---- HEVC core transform matrices, regenerated from the 32 distinct cosine magnitudes (normative constants) ----
struct DctTables
{
    int8_t t32[32][32];
};
constexpr DctTables make_dct_tables()
{
    // c[m] = HEVC integer approximation of 64*sqrt(2)*cos(m*pi/64), m = 0..32 (c[0] is the DC value 64)
    const int c[33] = { 64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                        61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0 };
    DctTables t{};
    for (int k = 0; k < 32; k++)
        for (int n = 0; n < 32; n++)
        {
            int m = (k * (2 * n + 1)) & 127;   // angle index modulo the cosine period, folded by symmetry below
            int v = m <= 32 ? c[m] : m <= 64 ? -c[64 - m] : m <= 96 ? -c[m - 64] : c[128 - m];
            t.t32[k][n] = (int8_t)v;
        }
    return t;
}
__device__ __constant__ const DctTables kDct = make_dct_tables();
// T_N[k][n] = T_32[k * 32 / N][n]
template <int N>
__device__ __forceinline__ int dct_coef(int k, int n) { return kDct.t32[k * (32 / N)][n]; }

__device__ __constant__ const int8_t kDst4[4][4] = { { 29, 55, 74, 84 }, { 74, 74, 0, -74 }, { 84, -29, -74, 55 }, { 55, -84, 74, -29 } };
__device__ __constant__ const int8_t kDct4[4][4] = { { 64, 64, 64, 64 }, { 83, 36, -36, -83 }, { 64, -64, -64, 64 }, { 36, -83, 83, -36 } };

// ---- one MFMA pass ------------------------------------------------------------------------------------------------
// bsel: this lane's 16 bytes of the block-diagonal coefficient operand; corr: 128 * column sum for this lane's column
template <int N, bool INV>
__device__ __forceinline__ void make_b_operand(int lane, v4i& bsel, int& corr)
{
    const int n = lane & 31, g2 = lane >> 5;
    int sum = 0;
    uint32_t w[4] = { 0, 0, 0, 0 };
#pragma unroll
    for (int j = 0; j < 16; j++)
    {
        const int k = 16 * g2 + j;
        int v = 0;
        if (k / N == n / N)
            v = INV ? dct_coef<N>(k % N, n % N) : dct_coef<N>(n % N, k % N);
        sum += v;
        w[j >> 2] |= (uint32_t)(v & 255) << (8 * (j & 3));
    }
    sum += __shfl_xor(sum, 32, kWave);
    corr = 128 * sum;
    bsel = v4i{ (int)w[0], (int)w[1], (int)w[2], (int)w[3] };
}

// in/out: per-wave LDS tiles of 1024 int16 holding (32/N)^2 TUs as [tu][row][col]
template <int N, bool INV>
__device__ __forceinline__ void mfma_pass(const int16_t* in, int16_t* out, int lane,
                                          const v4i& bsel, int corr, int shift)
{
    constexpr int TPS = 32 / N;                 // TUs per 32-row stack
    const int i = lane & 31, g2 = lane >> 5;
    const int tuRow = i / N, r = i % N;
    uint32_t hi[4], lo[4];
#pragma unroll
    for (int half = 0; half < 2; half++)
    {
        const int k0 = 16 * g2 + 8 * half;      // 8 consecutive k never straddle a TU (N >= 8)
        const int s = k0 / N, c0 = k0 % N;
        const int tu = s * TPS + tuRow;
        uint32_t w[4];
        if (!INV)
        {
            // contraction runs along the row: 8 contiguous int16, 16-byte aligned inside the tile
            const uint4 v = *reinterpret_cast<const uint4*>(in + tu * N * N + r * N + c0);
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        }
        else
        {
            // contraction runs down the column r of the TU: stride N
            const int16_t* p = in + tu * N * N + c0 * N + r;
#pragma unroll
            for (int m = 0; m < 4; m++)
                w[m] = (uint32_t)(uint16_t)p[(2 * m) * N] | ((uint32_t)(uint16_t)p[(2 * m + 1) * N] << 16);
        }
        // bytes of x0..x3 from (w0, w1), x4..x7 from (w2, w3): v_perm_b32 picks the low / high byte of each int16
        lo[2 * half]     = __builtin_amdgcn_perm(w[1], w[0], 0x06040200u) ^ 0x80808080u;
        lo[2 * half + 1] = __builtin_amdgcn_perm(w[3], w[2], 0x06040200u) ^ 0x80808080u;
        hi[2 * half]     = __builtin_amdgcn_perm(w[1], w[0], 0x07050301u);
        hi[2 * half + 1] = __builtin_amdgcn_perm(w[3], w[2], 0x07050301u);
    }
    const v4i ahi = { (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3] };
    const v4i alo = { (int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3] };
    v16i acc = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(ahi, bsel, acc, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 16; q++)
        acc[q] <<= 8;
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(alo, bsel, acc, 0, 0, 0);

    // D[row][col]: col = lane & 31, row = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5)
    const int n = lane & 31;
    const int so = n / N, co = n % N;           // output stack / coefficient index
    const int add = (1 << (shift - 1)) + corr;
#pragma unroll
    for (int qb = 0; qb < 4; qb++)
    {
        const int row0 = 8 * qb + 4 * (lane >> 5);          // 4 consecutive rows row0..row0+3, same TU
        const int tu = so * TPS + row0 / N, r0 = row0 % N;
        int v[4];
#pragma unroll
        for (int e = 0; e < 4; e++)
        {
            int t = (acc[4 * qb + e] + add) >> shift;
            v[e] = INV ? clip3i(-32768, 32767, t) : t;       // forward truncates to int16 like the reference's cast
        }
        if (!INV)
        {
            // dst[k*line + j]: coefficient-major, rows contiguous -> one 8-byte LDS store
            store4(out + tu * N * N + co * N + r0, v);
        }
        else
        {
            // dst[j*N + k]: row r0+e, column co
#pragma unroll
            for (int e = 0; e < 4; e++)
                out[tu * N * N + (r0 + e) * N + co] = (int16_t)v[e];
        }
    }
}


} // namespace xh
