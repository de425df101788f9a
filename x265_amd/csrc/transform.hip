// transform.hip — batched HEVC integer transforms and quantisation for gfx950.
//
// Reference semantics (bit-exact): source/common/dct.cpp — dct4_c..dct32_c :459-525 (partialButterfly* :83-240,
// :418), idct4_c..idct32_c :544-610 (partialButterflyInverse* :242-416), dst4_c :442 / idst4_c :527, quant_c :664,
// nquant_c :688, dequant_normal_c :612, dequant_scaling_c :636, count_nonzero_c :714.
//
// 8/16/32-point transforms run on the matrix cores.  One 1-D pass of the reference is, exactly,
//     dst[k][j] = (sum_n T[k][n] * src[j][n] + add) >> shift              (forward; output transposed)
//     dst[j][k] = clip16((sum_n T[n][k] * src[n][j] + add) >> shift)      (inverse)
// i.e. a (rows x N) * (N x N) integer GEMM with an int16 data operand and an int8-range coefficient operand
// (|T| <= 90).  CDNA4 has no int16 MFMA, so the data operand is split x = 256*hi + lo_u, hi = x >> 8 (signed byte),
// lo_u = x & 255 = lo_s + 128 with lo_s = lo_u ^ 0x80 read as a signed byte:
//     sum T*x = 256 * (T . hi) + (T . lo_s) + 128 * colsum(T)
// — two v_mfma_i32_32x32x32_i8 with int32 accumulation (exact; |sum| < 2^31 for every int16 input) and a per-column
// constant.  One MFMA covers a 32-row x 32-k slab: a 32x32 TU natively, FOUR 16x16 TUs or SIXTEEN 8x8 TUs through a
// block-diagonal coefficient operand (B[k][n] = T[..] when k/N == n/N, else 0).  The K index permutation inside the
// MFMA operands is irrelevant as long as A and B use the same one, which they do by construction below.
//
// Data flow per wave (4 waves per workgroup, each independent): global -> LDS (coalesced 16-B loads, arbitrary source
// stride) -> pass 1 (LDS -> MFMA -> LDS, transposed as the reference stores it) -> pass 2 -> coalesced global store.
// Algorithmic HBM traffic: 2 N^2 bytes in + 2 N^2 bytes out per TU; everything else stays in LDS/registers.
#include "common.h"
#include "dctcore.h"

namespace xh {

// ---- 8/16/32-point kernels ---------------------------------------------------------------------------------------------
// forward: src + offS[tu] (row stride strideS) -> dst + tu*N*N ; inverse: src + tu*N*N -> dst + offD[tu] (row stride)
template <int N, bool INV>
__global__ __launch_bounds__(256) void dct_mfma_kernel(const int16_t* __restrict__ src, int64_t stride, const int32_t* __restrict__ offs,
                                                       int16_t* __restrict__ dst, int n, int shift1, int shift2)
{
    constexpr int G = (32 / N) * (32 / N);      // TUs per wave-group; G * N * N == 1024
    __shared__ __attribute__((aligned(16))) int16_t lds[4][2][1024];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int16_t* buf0 = lds[wv][0];
    int16_t* buf1 = lds[wv][1];
    v4i bsel;
    int corr;
    make_b_operand<N, INV>(lane, bsel, corr);

    const int groups = (n + G - 1) / G;
    const int wavesTotal = gridDim.x * 4;
    for (int grp = blockIdx.x * 4 + wv; grp < groups; grp += wavesTotal)
    {
        const int tu0 = grp * G;
        // ---- load: lane covers 16 elements of the 1024-element tile, as two runs of 8 (one row segment each)
#pragma unroll
        for (int half = 0; half < 2; half++)
        {
            const int e = lane * 16 + half * 8;
            const int g = e / (N * N), rr = (e % (N * N)) / N, cc = e % N;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (tu0 + g < n)
            {
                const int16_t* p = INV ? src + (int64_t)(tu0 + g) * N * N + rr * N + cc
                                       : src + offs[tu0 + g] + rr * stride + cc;
                v = ld_unaligned<uint4>(p);
            }
            *reinterpret_cast<uint4*>(buf0 + e) = v;
        }
        // each wave owns its LDS tiles: only wave-local ordering is needed (LDS ops of one wave complete in order)
        __builtin_amdgcn_s_waitcnt(0xc07f);     // lgkmcnt(0)
        mfma_pass<N, INV>(buf0, buf1, lane, bsel, corr, shift1);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        mfma_pass<N, INV>(buf1, buf0, lane, bsel, corr, shift2);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        // ---- store
#pragma unroll
        for (int half = 0; half < 2; half++)
        {
            const int e = lane * 16 + half * 8;
            const int g = e / (N * N), rr = (e % (N * N)) / N, cc = e % N;
            if (tu0 + g < n)
            {
                const uint4 v = *reinterpret_cast<const uint4*>(buf0 + e);
                int16_t* p = INV ? dst + offs[tu0 + g] + rr * stride + cc
                                 : dst + (int64_t)(tu0 + g) * N * N + rr * N + cc;
                st_unaligned<uint4>(p, v);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
}

// ---- 4x4 DCT / DST: one lane per TU, both passes in registers ------------------------------------------------------
template <bool INV>
__global__ __launch_bounds__(256) void dct4_kernel(const int16_t* __restrict__ src, int64_t stride, const int32_t* __restrict__ offs,
                                                   int16_t* __restrict__ dst, int n, int useDst, int shift1, int shift2)
{
    for (int tu = blockIdx.x * blockDim.x + threadIdx.x; tu < n; tu += gridDim.x * blockDim.x)
    {
        int x[16], t[16];
        int mtx[16];
#pragma unroll
        for (int i = 0; i < 16; i++)
            mtx[i] = useDst ? kDst4[i >> 2][i & 3] : kDct4[i >> 2][i & 3];
#pragma unroll
        for (int r = 0; r < 4; r++)
            load4(INV ? src + (int64_t)tu * 16 + r * 4 : src + offs[tu] + r * stride, &x[4 * r]);
        const int add1 = 1 << (shift1 - 1), add2 = 1 << (shift2 - 1);
        if (!INV)
        {
            // dst[k*4 + j] = (sum_n M[k][n] * src[j*4 + n] + add) >> shift, truncated to int16
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    int s = 0;
#pragma unroll
                    for (int m = 0; m < 4; m++) s += mtx[4 * k + m] * x[4 * j + m];
                    t[4 * k + j] = (int)(int16_t)((s + add1) >> shift1);
                }
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    int s = 0;
#pragma unroll
                    for (int m = 0; m < 4; m++) s += mtx[4 * k + m] * t[4 * j + m];
                    x[4 * k + j] = (s + add2) >> shift2;
                }
#pragma unroll
            for (int r = 0; r < 4; r++)
                store4(dst + (int64_t)tu * 16 + r * 4, &x[4 * r]);
        }
        else
        {
            // dst[j*4 + k] = clip16((sum_n M[n][k] * src[n*4 + j] + add) >> shift)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    int s = 0;
#pragma unroll
                    for (int m = 0; m < 4; m++) s += mtx[4 * m + k] * x[4 * m + j];
                    t[4 * j + k] = clip3i(-32768, 32767, (s + add1) >> shift1);
                }
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    int s = 0;
#pragma unroll
                    for (int m = 0; m < 4; m++) s += mtx[4 * m + k] * t[4 * m + j];
                    x[4 * j + k] = clip3i(-32768, 32767, (s + add2) >> shift2);
                }
#pragma unroll
            for (int r = 0; r < 4; r++)
                store4(dst + offs[tu] + r * stride, &x[4 * r]);
        }
    }
}

// ---- quantisation -----------------------------------------------------------------------------------------------------
// T lanes per TU, 4 coefficients per lane per pass; numSig by DPP/shuffle sum over the T lanes
template <bool NQUANT>
__global__ __launch_bounds__(256) void quant_kernel(const int16_t* __restrict__ coef, const int32_t* __restrict__ quantCoeff,
                                                    int32_t* __restrict__ deltaU, int16_t* __restrict__ qCoef,
                                                    int qBits, int add, int numCoeff, int n, uint32_t* __restrict__ numSig)
{
    const int lane = threadIdx.x & 63;
    const int quads = numCoeff >> 2;
    const int T = quads >= 64 ? 64 : quads;          // 4, 16, 64 (numCoeff 16, 64, >= 256)
    const int tpw = 64 / T, iters = quads / T;
    const int sub = lane & (T - 1);
    const int wavesTotal = gridDim.x * (blockDim.x >> 6);
    const int gwave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int qBits8 = qBits - 8;
    for (long long tu0 = (long long)gwave * tpw; tu0 < n; tu0 += (long long)wavesTotal * tpw)
    {
        const long long tu = tu0 + lane / T;
        const bool ok = tu < n;
        int cnt = 0;
        if (ok)
        {
            for (int it = 0; it < iters; it++)
            {
                const int i0 = (sub + it * T) * 4;
                const int64_t g = tu * numCoeff + i0;
                int c[4], q[4], du[4];
                load4(coef + g, c);
                const int4 qc = ld_unaligned<int4>(quantCoeff + i0);
                const int qcv[4] = { qc.x, qc.y, qc.z, qc.w };
#pragma unroll
                for (int e = 0; e < 4; e++)
                {
                    const int sign = c[e] < 0 ? -1 : 1;
                    const int tmplevel = iabs(c[e]) * qcv[e];
                    int level = (tmplevel + add) >> qBits;
                    du[e] = (tmplevel - (level << qBits)) >> qBits8;
                    cnt += level != 0;
                    level *= sign;
                    level = clip3i(-32768, 32767, level);
                    q[e] = NQUANT ? iabs(level) : level;
                }
                store4(qCoef + g, q);
                if (!NQUANT && deltaU)
                    st_unaligned<int4>(deltaU + g, make_int4(du[0], du[1], du[2], du[3]));
            }
        }
        cnt = group_sum(cnt, T);
        if (ok && sub == 0)
            numSig[tu] = (uint32_t)cnt;
    }
}

__global__ __launch_bounds__(256) void dequant_normal_kernel(const int16_t* __restrict__ q, int16_t* __restrict__ coef,
                                                             int64_t num, int scale, int shift)
{
    const int add = 1 << (shift - 1);
    const int64_t quads = num >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += (int64_t)gridDim.x * blockDim.x)
    {
        int v[4];
        load4(q + 4 * i, v);
#pragma unroll
        for (int e = 0; e < 4; e++)
            v[e] = clip3i(-32768, 32767, (v[e] * scale + add) >> shift);
        store4(coef + 4 * i, v);
    }
}

__global__ __launch_bounds__(256) void dequant_scaling_kernel(const int16_t* __restrict__ q, const int32_t* __restrict__ dqc,
                                                              int16_t* __restrict__ coef, int numCoeff, int64_t total, int per, int shift)
{
    shift += 4;
    const int64_t quads = total >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += (int64_t)gridDim.x * blockDim.x)
    {
        int v[4];
        load4(q + 4 * i, v);
        const int4 d4 = ld_unaligned<int4>(dqc + (int)((4 * i) % numCoeff));
        const int d[4] = { d4.x, d4.y, d4.z, d4.w };
#pragma unroll
        for (int e = 0; e < 4; e++)
        {
            if (shift > per)
            {
                const int add = 1 << (shift - per - 1);
                v[e] = clip3i(-32768, 32767, (v[e] * d[e] + add) >> (shift - per));
            }
            else
            {
                const int c = clip3i(-32768, 32767, v[e] * d[e]);
                v[e] = clip3i(-32768, 32767, (int)((unsigned)c << (per - shift)));
            }
        }
        store4(coef + 4 * i, v);
    }
}

__global__ __launch_bounds__(256) void count_nonzero_kernel(const int16_t* __restrict__ q, int numCoeff, int n, uint32_t* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int quads = numCoeff >> 2;
    const int T = quads >= 64 ? 64 : quads;
    const int tpw = 64 / T, iters = quads / T;
    const int sub = lane & (T - 1);
    const int wavesTotal = gridDim.x * (blockDim.x >> 6);
    const int gwave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    for (long long tu0 = (long long)gwave * tpw; tu0 < n; tu0 += (long long)wavesTotal * tpw)
    {
        const long long tu = tu0 + lane / T;
        const bool ok = tu < n;
        int cnt = 0;
        if (ok)
            for (int it = 0; it < iters; it++)
            {
                int v[4];
                load4(q + tu * numCoeff + (sub + it * T) * 4, v);
                cnt += (v[0] != 0) + (v[1] != 0) + (v[2] != 0) + (v[3] != 0);
            }
        cnt = group_sum(cnt, T);
        if (ok && sub == 0)
            out[tu] = (uint32_t)cnt;
    }
}

static bool tu_count_ok(int numCoeff) { return numCoeff == 16 || numCoeff == 64 || numCoeff == 256 || numCoeff == 1024; }

} // namespace xh

using namespace xh;

#define XH_ARGS_CHECK(cond, ...) do { if (!(cond)) return set_error(X265HIP_EINVAL, __VA_ARGS__); } while (0)

extern "C" {

int x265hip_dct_batch(int size, int dst4, int depth, const int16_t* src, int64_t strideS, const int32_t* offS,
                      int16_t* dst, int n, void* stream)
{
    XH_CHECK_DEV();
    XH_ARGS_CHECK(valid_depth(depth) && n >= 0, "dct: depth %d n %d", depth, n);
    XH_ARGS_CHECK(size == 4 || size == 8 || size == 16 || size == 32, "dct: size %d", size);
    XH_ARGS_CHECK(!dst4 || size == 4, "dct: DST is 4x4 only");
    if (!n) return X265HIP_OK;
    hipStream_t st = as_stream(stream);
    const int log2n = size == 4 ? 2 : size == 8 ? 3 : size == 16 ? 4 : 5;
    const int s1 = log2n - 1 + depth - 8, s2 = log2n + 6;          // dct.cpp:444-445, :461-462, :477-478, :493-494, :511-512
    if (size == 4)
        hipLaunchKernelGGL((dct4_kernel<false>), dim3(grid_for((n + 255) / 256)), dim3(256), 0, st, src, strideS, offS, dst, n, dst4, s1, s2);
    else
    {
        const int G = (32 / size) * (32 / size);
        dim3 grid(grid_for(((n + G - 1) / G + 3) / 4)), block(256);
        if (size == 8) hipLaunchKernelGGL((dct_mfma_kernel<8, false>), grid, block, 0, st, src, strideS, offS, dst, n, s1, s2);
        else if (size == 16) hipLaunchKernelGGL((dct_mfma_kernel<16, false>), grid, block, 0, st, src, strideS, offS, dst, n, s1, s2);
        else hipLaunchKernelGGL((dct_mfma_kernel<32, false>), grid, block, 0, st, src, strideS, offS, dst, n, s1, s2);
    }
    XH_LAUNCH_CHECK("dct kernel");
    return X265HIP_OK;
}

int x265hip_idct_batch(int size, int dst4, int depth, const int16_t* src, int16_t* dst, int64_t strideD,
                       const int32_t* offD, int n, void* stream)
{
    XH_CHECK_DEV();
    XH_ARGS_CHECK(valid_depth(depth) && n >= 0, "idct: depth %d n %d", depth, n);
    XH_ARGS_CHECK(size == 4 || size == 8 || size == 16 || size == 32, "idct: size %d", size);
    XH_ARGS_CHECK(!dst4 || size == 4, "idct: DST is 4x4 only");
    if (!n) return X265HIP_OK;
    hipStream_t st = as_stream(stream);
    const int s1 = 7, s2 = 12 - (depth - 8);                          // dct.cpp:529-530, :546-547
    if (size == 4)
        hipLaunchKernelGGL((dct4_kernel<true>), dim3(grid_for((n + 255) / 256)), dim3(256), 0, st, src, strideD, offD, dst, n, dst4, s1, s2);
    else
    {
        const int G = (32 / size) * (32 / size);
        dim3 grid(grid_for(((n + G - 1) / G + 3) / 4)), block(256);
        if (size == 8) hipLaunchKernelGGL((dct_mfma_kernel<8, true>), grid, block, 0, st, src, strideD, offD, dst, n, s1, s2);
        else if (size == 16) hipLaunchKernelGGL((dct_mfma_kernel<16, true>), grid, block, 0, st, src, strideD, offD, dst, n, s1, s2);
        else hipLaunchKernelGGL((dct_mfma_kernel<32, true>), grid, block, 0, st, src, strideD, offD, dst, n, s1, s2);
    }
    XH_LAUNCH_CHECK("idct kernel");
    return X265HIP_OK;
}

int x265hip_quant_batch(const int16_t* coef, const int32_t* quantCoeff, int32_t* deltaU, int16_t* qCoef,
                        int qBits, int add, int numCoeff, int n, uint32_t* numSig, void* stream)
{
    XH_CHECK_DEV();
    XH_ARGS_CHECK(tu_count_ok(numCoeff) && n >= 0 && qBits >= 8, "quant: numCoeff %d n %d qBits %d", numCoeff, n, qBits);
    if (!n) return X265HIP_OK;
    const int T = numCoeff / 4 >= 64 ? 64 : numCoeff / 4;
    const long long waves = ((long long)n + 64 / T - 1) / (64 / T);
    hipLaunchKernelGGL((quant_kernel<false>), dim3(grid_for((waves + 3) / 4)), dim3(256), 0, as_stream(stream),
                       coef, quantCoeff, deltaU, qCoef, qBits, add, numCoeff, n, numSig);
    XH_LAUNCH_CHECK("quant_kernel");
    return X265HIP_OK;
}

int x265hip_nquant_batch(const int16_t* coef, const int32_t* quantCoeff, int16_t* qCoef,
                         int qBits, int add, int numCoeff, int n, uint32_t* numSig, void* stream)
{
    XH_CHECK_DEV();
    XH_ARGS_CHECK(tu_count_ok(numCoeff) && n >= 0, "nquant: numCoeff %d n %d", numCoeff, n);
    if (!n) return X265HIP_OK;
    const int T = numCoeff / 4 >= 64 ? 64 : numCoeff / 4;
    const long long waves = ((long long)n + 64 / T - 1) / (64 / T);
    hipLaunchKernelGGL((quant_kernel<true>), dim3(grid_for((waves + 3) / 4)), dim3(256), 0, as_stream(stream),
                       coef, quantCoeff, (int32_t*)nullptr, qCoef, qBits, add, numCoeff, n, numSig);
    XH_LAUNCH_CHECK("nquant_kernel");
    return X265HIP_OK;
}

int x265hip_dequant_normal(const int16_t* quantCoef, int16_t* coef, int64_t num, int scale, int shift, void* stream)
{
    XH_CHECK_DEV();
    XH_ARGS_CHECK(num >= 0 && !(num & 3) && shift >= 1, "dequant_normal: num %lld shift %d", (long long)num, shift);
    if (!num) return X265HIP_OK;
    hipLaunchKernelGGL(dequant_normal_kernel, dim3(grid_for((num / 4 + 255) / 256)), dim3(256), 0, as_stream(stream),
                       quantCoef, coef, num, scale, shift);
    XH_LAUNCH_CHECK("dequant_normal_kernel");
    return X265HIP_OK;
}

int x265hip_dequant_scaling_batch(const int16_t* quantCoef, const int32_t* deQuantCoef, int16_t* coef,
                                  int numCoeff, int n, int per, int shift, void* stream)
{
    XH_CHECK_DEV();
    XH_ARGS_CHECK(tu_count_ok(numCoeff) && n >= 0, "dequant_scaling: numCoeff %d n %d", numCoeff, n);
    if (!n) return X265HIP_OK;
    const int64_t total = (int64_t)numCoeff * n;
    hipLaunchKernelGGL(dequant_scaling_kernel, dim3(grid_for((total / 4 + 255) / 256)), dim3(256), 0, as_stream(stream),
                       quantCoef, deQuantCoef, coef, numCoeff, total, per, shift);
    XH_LAUNCH_CHECK("dequant_scaling_kernel");
    return X265HIP_OK;
}

int x265hip_count_nonzero_batch(const int16_t* qCoef, int numCoeff, int n, uint32_t* out, void* stream)
{
    XH_CHECK_DEV();
    XH_ARGS_CHECK(tu_count_ok(numCoeff) && n >= 0, "count_nonzero: numCoeff %d n %d", numCoeff, n);
    if (!n) return X265HIP_OK;
    const int T = numCoeff / 4 >= 64 ? 64 : numCoeff / 4;
    const long long waves = ((long long)n + 64 / T - 1) / (64 / T);
    hipLaunchKernelGGL(count_nonzero_kernel, dim3(grid_for((waves + 3) / 4)), dim3(256), 0, as_stream(stream), qCoef, numCoeff, n, out);
    XH_LAUNCH_CHECK("count_nonzero_kernel");
    return X265HIP_OK;
}

} // extern "C"
