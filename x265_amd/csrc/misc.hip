// misc.hip — the remaining small slots of SURVEY.md §8a rows a11 / a16: layout shuffles around the transform
// (cpy2Dto1D_shl/shr, cpy1Dto2D_shl/shr, copy_cnt, blockfill_s), denoiseDct and the RDOQ per-coefficient-group cost
// helpers (nonPsyRdoQuant, psyRdoQuant, psyRdoQuant_1p/_2p).
//
// Reference semantics (bit-exact): source/common/pixel.cpp:393-467 (blockfill_s_c, cpy2Dto1D_shl/shr, cpy1Dto2D_shl/shr),
// source/common/dct.cpp:728 (copy_count), :743 (denoiseDct_c), :985-1069 (nonPsyRdoQuant_c, psyRdoQuant_c, psyRdoQuant_c_1,
// psyRdoQuant_c_2).  All of it is HBM-bound elementwise work on int16 blocks.
#include "common.h"

namespace xh {

enum { SH_2D1D_SHL, SH_2D1D_SHR, SH_1D2D_SHL, SH_1D2D_SHR };

// job i: 2-D block at plane + off[i] (row stride `stride`)  <->  dense [i][size*size]
template <int KIND>
__global__ __launch_bounds__(256) void shuffle_kernel(int16_t* __restrict__ dst, const int16_t* __restrict__ src, int64_t stride,
                                                      const int32_t* __restrict__ off, int shift, int size, int n)
{
    const int qx = size >> 2, per = qx * size;
    const long long total = (long long)n * per;
    const int round = shift > 0 ? (int)(int16_t)(1 << (shift - 1)) : 0;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x)
    {
        const int job = (int)(idx / per), p = (int)(idx - (long long)job * per), y = p / qx, x = (p - y * qx) * 4;
        const bool from2d = KIND == SH_2D1D_SHL || KIND == SH_2D1D_SHR;
        const int16_t* s = from2d ? src + off[job] + y * stride + x : src + (int64_t)job * size * size + y * size + x;
        int16_t* d = from2d ? dst + (int64_t)job * size * size + y * size + x : dst + off[job] + y * stride + x;
        int v[4];
        load4(s, v);
#pragma unroll
        for (int e = 0; e < 4; e++)
            v[e] = (KIND == SH_2D1D_SHL || KIND == SH_1D2D_SHL) ? (int)(int16_t)(v[e] << shift) : ((v[e] + round) >> shift);
        store4(d, v);
    }
}

// copy_count (dct.cpp:728): coeff[i][k*size+j] = resi[k*stride+j], numSig per block
__global__ __launch_bounds__(256) void copy_cnt_kernel(int16_t* __restrict__ coeff, const int16_t* __restrict__ resi, int64_t stride,
                                                       const int32_t* __restrict__ off, int size, int n, uint32_t* __restrict__ numSig)
{
    const int lane = threadIdx.x & 63;
    const int qx = size >> 2, quads = qx * size;
    const int T = quads >= 64 ? 64 : quads, bpw = 64 / T, iters = quads / T, sub = lane & (T - 1);
    const int wavesTotal = gridDim.x * (blockDim.x >> 6), gwave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    for (long long j0 = (long long)gwave * bpw; j0 < n; j0 += (long long)wavesTotal * bpw)
    {
        const long long job = j0 + lane / T;
        const bool ok = job < n;
        int cnt = 0;
        if (ok)
            for (int it = 0; it < iters; it++)
            {
                const int q = sub + it * T, y = q / qx, x = (q - y * qx) * 4;
                int v[4];
                load4(resi + off[job] + y * stride + x, v);
                store4(coeff + job * size * size + y * size + x, v);
                cnt += (v[0] != 0) + (v[1] != 0) + (v[2] != 0) + (v[3] != 0);
            }
        cnt = group_sum(cnt, T);
        if (ok && sub == 0)
            numSig[job] = (uint32_t)cnt;
    }
}

__global__ __launch_bounds__(256) void blockfill_kernel(int16_t* __restrict__ dst, int64_t stride, const int32_t* __restrict__ off,
                                                        const int16_t* __restrict__ val, int size, int n)
{
    const int qx = size >> 2, per = qx * size;
    const long long total = (long long)n * per;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x)
    {
        const int job = (int)(idx / per), p = (int)(idx - (long long)job * per), y = p / qx, x = (p - y * qx) * 4;
        const int v = val[job];
        const int q[4] = { v, v, v, v };
        store4(dst + off[job] + y * stride + x, q);
    }
}

// denoiseDct_c (dct.cpp:743): in place on dctCoef [n][numCoeff]; resSum / offset are [numCoeff] shared by the n TUs, so resSum
// accumulates with atomics when n > 1
__global__ __launch_bounds__(256) void denoise_kernel(int16_t* __restrict__ dctCoef, uint32_t* __restrict__ resSum, const uint16_t* __restrict__ offset,
                                                      int numCoeff, int n)
{
    const long long total = (long long)n * numCoeff;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x)
    {
        const int i = (int)(idx % numCoeff);
        int level = dctCoef[idx];
        const int sign = level >> 31;
        level = (level + sign) ^ sign;
        if (n == 1) resSum[i] += (uint32_t)level; else atomicAdd(&resSum[i], (uint32_t)level);
        level -= offset[i];
        dctCoef[idx] = (int16_t)(level < 0 ? 0 : (level ^ sign) - sign);
    }
}

// RDOQ cost helpers for coefficient group jobs: job i = 4x4 group at blkPos[i] of TU tu[i] (row pitch = trSize)
// kind 0 nonPsy (dct.cpp:985), 1 psy (:1005), 2 psy_1p (:1030 == nonPsy), 3 psy_2p (:1049: only the psy term, applied to the
// costUncoded already stored).  cgUncoded / cgRd receive the group's contribution to *totalUncodedCost / *totalRdCost.
__global__ __launch_bounds__(256) void rdoq_cost_kernel(int kind, int log2n, int depth, const int16_t* __restrict__ resiDct,
                                                        const int16_t* __restrict__ fencDct, const int64_t* __restrict__ psyScale,
                                                        const int32_t* __restrict__ tu, const int32_t* __restrict__ blkPos, int n,
                                                        int64_t* __restrict__ costUncoded, int64_t* __restrict__ cgUncoded, int64_t* __restrict__ cgRd)
{
    const int transformShift = 15 - depth - log2n;                  // MAX_TR_DYNAMIC_RANGE - X265_DEPTH - log2TrSize
    const int scaleBits = 15 - 2 * transformShift;                  // SCALE_BITS - 2 * transformShift
    const int trSize = 1 << log2n, nc = trSize * trSize;
    const int maxs = max(0, 2 * transformShift + 1);
    for (int job = blockIdx.x * blockDim.x + threadIdx.x; job < n; job += gridDim.x * blockDim.x)
    {
        const int64_t base = (int64_t)tu[job] * nc;
        int64_t sum = 0;
        const int64_t ps = (kind == 1 || kind == 3) ? psyScale[0] : 0;
        for (int y = 0; y < 4; y++)
            for (int x = 0; x < 4; x++)
            {
                const int64_t pos = base + blkPos[job] + y * trSize + x;
                const int64_t s = resiDct[pos];
                int64_t c = (kind == 3) ? costUncoded[pos] : (int64_t)((uint64_t)(s * s) << scaleBits);
                if (kind == 1 || kind == 3)
                {
                    const int64_t pred = (int64_t)fencDct[pos] - s;
                    c -= (ps * pred) >> maxs;
                }
                costUncoded[pos] = c;
                sum += c;
            }
        cgUncoded[job] = sum;
        cgRd[job] = sum;
    }
}

static bool tu_size_ok(int s) { return s == 4 || s == 8 || s == 16 || s == 32; }

} // namespace xh

using namespace xh;

extern "C" {

int x265hip_cpy_shift_batch(int kind, int size, int16_t* dst, const int16_t* src, int64_t stride, const int32_t* off, int shift, int n, void* stream)
{
    XH_CHECK_DEV();
    if (kind < 0 || kind > 3 || !tu_size_ok(size) || n < 0 || shift < 0 || shift > 15 || ((kind & 1) && shift < 1))
        return set_error(X265HIP_EINVAL, "cpy_shift: kind %d size %d shift %d n %d", kind, size, shift, n);
    if (!n) return X265HIP_OK;
    const long long total = (long long)n * (size / 4) * size;
    dim3 grid(grid_for((total + 255) / 256)), block(256);
    hipStream_t st = as_stream(stream);
    switch (kind)
    {
    case 0: hipLaunchKernelGGL((shuffle_kernel<SH_2D1D_SHL>), grid, block, 0, st, dst, src, stride, off, shift, size, n); break;
    case 1: hipLaunchKernelGGL((shuffle_kernel<SH_2D1D_SHR>), grid, block, 0, st, dst, src, stride, off, shift, size, n); break;
    case 2: hipLaunchKernelGGL((shuffle_kernel<SH_1D2D_SHL>), grid, block, 0, st, dst, src, stride, off, shift, size, n); break;
    default: hipLaunchKernelGGL((shuffle_kernel<SH_1D2D_SHR>), grid, block, 0, st, dst, src, stride, off, shift, size, n); break;
    }
    XH_LAUNCH_CHECK("shuffle_kernel");
    return X265HIP_OK;
}

int x265hip_copy_cnt_batch(int size, int16_t* coeff, const int16_t* resi, int64_t stride, const int32_t* off, int n, uint32_t* numSig, void* stream)
{
    XH_CHECK_DEV();
    if (!tu_size_ok(size) || n < 0) return set_error(X265HIP_EINVAL, "copy_cnt: size %d n %d", size, n);
    if (!n) return X265HIP_OK;
    const int quads = (size / 4) * size, T = quads >= 64 ? 64 : quads;
    const long long waves = ((long long)n + 64 / T - 1) / (64 / T);
    hipLaunchKernelGGL(copy_cnt_kernel, dim3(grid_for((waves + 3) / 4)), dim3(256), 0, as_stream(stream), coeff, resi, stride, off, size, n, numSig);
    XH_LAUNCH_CHECK("copy_cnt_kernel");
    return X265HIP_OK;
}

int x265hip_blockfill_s_batch(int size, int16_t* dst, int64_t stride, const int32_t* off, const int16_t* val, int n, void* stream)
{
    XH_CHECK_DEV();
    if (!(tu_size_ok(size) || size == 64) || n < 0) return set_error(X265HIP_EINVAL, "blockfill_s: size %d n %d", size, n);
    if (!n) return X265HIP_OK;
    const long long total = (long long)n * (size / 4) * size;
    hipLaunchKernelGGL(blockfill_kernel, dim3(grid_for((total + 255) / 256)), dim3(256), 0, as_stream(stream), dst, stride, off, val, size, n);
    XH_LAUNCH_CHECK("blockfill_kernel");
    return X265HIP_OK;
}

int x265hip_denoise_dct_batch(int16_t* dctCoef, uint32_t* resSum, const uint16_t* offset, int numCoeff, int n, void* stream)
{
    XH_CHECK_DEV();
    if (numCoeff < 1 || n < 0) return set_error(X265HIP_EINVAL, "denoise_dct: numCoeff %d n %d", numCoeff, n);
    if (!n) return X265HIP_OK;
    const long long total = (long long)n * numCoeff;
    hipLaunchKernelGGL(denoise_kernel, dim3(grid_for((total + 255) / 256)), dim3(256), 0, as_stream(stream), dctCoef, resSum, offset, numCoeff, n);
    XH_LAUNCH_CHECK("denoise_kernel");
    return X265HIP_OK;
}

int x265hip_rdoq_cost_batch(int kind, int size, int depth, const int16_t* resiDct, const int16_t* fencDct, const int64_t* psyScale,
                            const int32_t* tu, const int32_t* blkPos, int n, int64_t* costUncoded, int64_t* cgUncoded, int64_t* cgRd, void* stream)
{
    XH_CHECK_DEV();
    if (kind < 0 || kind > 3 || !tu_size_ok(size) || !valid_depth(depth) || n < 0)
        return set_error(X265HIP_EINVAL, "rdoq_cost: kind %d size %d depth %d n %d", kind, size, depth, n);
    if (!n) return X265HIP_OK;
    const int log2n = size == 4 ? 2 : size == 8 ? 3 : size == 16 ? 4 : 5;
    hipLaunchKernelGGL(rdoq_cost_kernel, dim3(grid_for((n + 255) / 256)), dim3(256), 0, as_stream(stream), kind, log2n, depth, resiDct, fencDct,
                       psyScale, tu, blkPos, n, costUncoded, cgUncoded, cgRd);
    XH_LAUNCH_CHECK("rdoq_cost_kernel");
    return X265HIP_OK;
}

} // extern "C"
