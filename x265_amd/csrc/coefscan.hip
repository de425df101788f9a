// coefscan.hip — the coefficient-scan cost primitives of the RDOQ / bit-estimation loop (reference source/common/dct.cpp:757-1006:
// scanPosLast_c, findPosFirstLast_c, costCoeffNxN_c, costCoeffRemain_c, costC1C2Flag_c; callers Quant::rdoQuant quant.cpp:610-1420 and
// Entropy::codeCoeffNxN entropy.cpp), one job per transform unit or coefficient group, many jobs per launch.
//
// The scan orders are generated on the host by the rule of the standard (6.5.3-6.5.5: up-right diagonal / horizontal / vertical over 4x4
// groups; the group grid of 16x16 and 32x32 units always diagonal) and live in device memory.  The CABAC cost table (x265_entropyStateBits,
// constants.cpp: (next state << 24) | bits per state^bin) is the encoder's data: x265hip_set_entropy_state_bits uploads it once, and the cost
// kernels refuse to run without it.
//
// scanPosLast is the one with parallelism inside a job: a lane takes one 4x4 group of the scan (a 32x32 unit has 64), the wave agrees on
// the last significant position with one max-reduction, and each lane then builds its group's count / significance / sign words — the
// reference's serial walk stops at that same position (numSig reaching zero), so the words are identical.  The other four are short serial
// walks through a CABAC context array, one lane per job.
#include "common.h"
#include <cstring>
#include <mutex>

namespace xh {

__device__ uint32_t g_stateBits[128];
__device__ uint16_t g_scanTab[3][16 + 64 + 256 + 1024];          // [type][offset(log2)..]
// __device__ symbols exist once PER DEVICE: which devices hold the tables is tracked per device id, under a mutex (slot calls arrive from
// every pool worker of the encoder, and a process may drive several GPUs)
static constexpr int kMaxDevices = 64;
static std::mutex s_tabLock;
static bool s_scanSet[kMaxDevices] = {}, s_stateBitsOn[kMaxDevices] = {};
static bool s_haveStateBits = false;
static uint32_t s_stateBitsHost[128];

static int current_device()
{
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices)
        return -1;
    return d;
}

__host__ __device__ constexpr int scan_off(int log2) { return log2 == 2 ? 0 : log2 == 3 ? 16 : log2 == 4 ? 80 : 336; }

static void scan_positions(int type, int n, int* xs, int* ys)
{
    int i = 0;
    if (type == 0)
    {
        int x = 0, y = 0;
        while (i < n * n)
        {
            while (y >= 0)
            {
                if (x < n && y < n) { xs[i] = x; ys[i] = y; i++; }
                y--; x++;
            }
            y = x; x = 0;
        }
    }
    else if (type == 1)
    {
        for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) { xs[i] = x; ys[i] = y; i++; }
    }
    else
    {
        for (int x = 0; x < n; x++) for (int y = 0; y < n; y++) { xs[i] = x; ys[i] = y; i++; }
    }
}

static int ensure_scan_tables()
{
    const int dev = current_device();
    if (dev < 0)
        return set_error(X265HIP_EHIP, "coefscan: no current device");
    std::lock_guard<std::mutex> g(s_tabLock);
    if (s_scanSet[dev])
        return X265HIP_OK;
    static uint16_t tab[3][16 + 64 + 256 + 1024];
    for (int type = 0; type < 3; type++)
        for (int log2 = 2; log2 <= 5; log2++)
        {
            const int size = 1 << log2, cgs = size >> 2, t = log2 > 3 ? 0 : type;       // MDCS_LOG2_MAX_SIZE = 3 (common.h:316)
            int cx[64], cy[64], px[16], py[16];
            scan_positions(t, 4, px, py);
            cx[0] = cy[0] = 0;
            if (cgs > 1) scan_positions(t, cgs, cx, cy);
            for (int g = 0; g < cgs * cgs; g++)
                for (int k = 0; k < 16; k++)
                    tab[type][scan_off(log2) + g * 16 + k] = (uint16_t)((cy[g] * 4 + py[k]) * size + cx[g] * 4 + px[k]);
        }
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_scanTab), tab, sizeof(tab)) != hipSuccess)
        return set_error(X265HIP_EHIP, "coefscan: scan table upload failed");
    s_scanSet[dev] = true;
    return X265HIP_OK;
}

// one CABAC bin against *ctx (dct.cpp:884-890 == sbacNext / sbacGetEntropyBits of contexts.h)
__device__ __forceinline__ uint32_t bin_cost(uint8_t* ctx, uint32_t bin)
{
    const uint32_t mstate = *ctx, mps = mstate & 1;
    const uint32_t sb = g_stateBits[mstate ^ bin];
    uint32_t next = (sb >> 24) + mps;
    if ((mstate ^ bin) == 1)
        next = bin;
    *ctx = (uint8_t)next;
    return sb;
}

// scanPosLast_c, dct.cpp:757-792.  One wave per transform unit, lane = coefficient group in scan order.
__global__ __launch_bounds__(256) void scan_pos_last_kernel(const int16_t* __restrict__ coeff, int log2, int type, int n, uint16_t* __restrict__ coeffSign,
                                                            uint16_t* __restrict__ coeffFlag, uint8_t* __restrict__ coeffNum, int32_t* __restrict__ lastPos)
{
    const int lane = threadIdx.x & 63, tu = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (tu >= n)
        return;
    const int ncg = 1 << (2 * log2 - 4);
    const int16_t* c = coeff + ((int64_t)tu << (2 * log2));
    const uint16_t* scan = g_scanTab[type] + scan_off(log2) + lane * 16;
    int16_t v[16];
    int last = -1;
    if (lane < ncg)
    {
#pragma unroll
        for (int k = 0; k < 16; k++)
        {
            v[k] = c[scan[k]];
            if (v[k]) last = lane * 16 + k;
        }
    }
    int wl = last;
#pragma unroll
    for (int off = 32; off; off >>= 1)
        wl = max(wl, __shfl_xor(wl, off));
    if (wl < 0)
        wl = 0;                                        // an all-zero unit: the reference's do-while still takes position 0
    const int limit = lane < (wl >> 4) ? 16 : (lane == (wl >> 4) ? (wl & 15) + 1 : 0);
    uint32_t sign = 0, flag = 0, num = 0;
    if (lane < ncg)
    {
#pragma unroll
        for (int k = 0; k < 16; k++)
            if (k < limit)
            {
                const uint32_t nz = v[k] != 0;
                sign += (uint32_t)(v[k] < 0) << num;
                flag = (flag << 1) + nz;
                num += nz;
            }
    }
    coeffSign[(int64_t)tu * 64 + lane] = (uint16_t)sign;
    coeffFlag[(int64_t)tu * 64 + lane] = (uint16_t)flag;
    coeffNum[(int64_t)tu * 64 + lane] = (uint8_t)num;
    if (!lane)
        lastPos[tu] = wl;
}

// findPosFirstLast_c, dct.cpp:795-838
__global__ __launch_bounds__(256) void find_pos_first_last_kernel(const int16_t* __restrict__ coeff, const int64_t* __restrict__ cgOffsets, int64_t trSize,
                                                                  int type, int n, uint32_t* __restrict__ out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n)
        return;
    const int16_t* c = coeff + cgOffsets[j];
    const uint16_t* scan = g_scanTab[type];            // the 4x4 table
    int16_t v[16];
#pragma unroll
    for (int k = 0; k < 16; k++)
        v[k] = c[(scan[k] >> 2) * trSize + (scan[k] & 3)];
    int last = -1, first = 16;
#pragma unroll
    for (int k = 0; k < 16; k++)
        if (v[k]) last = k;
#pragma unroll
    for (int k = 15; k >= 0; k--)
        if (v[k]) first = k;
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < 16; k++)
        if (k >= first && k <= last) sum += (uint32_t)(int32_t)v[k];
    out[j] = (sum << 31) | ((uint32_t)last << 8) | (uint32_t)first;
}

// costCoeffNxN_c, dct.cpp:841-899
__global__ __launch_bounds__(256) void cost_coeff_nxn_kernel(const int16_t* __restrict__ coeff, const x265hip_coeff_group_job* __restrict__ jobs, int n,
                                                             uint8_t* __restrict__ baseCtx, int ctxStride, uint16_t* __restrict__ absCoeff,
                                                             uint32_t* __restrict__ bits)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n)
        return;
    const x265hip_coeff_group_job jb = jobs[j];
    const int16_t* c = coeff + jb.coeffOffset;
    const uint16_t* scan = g_scanTab[jb.scanType];
    uint8_t* ctx = baseCtx + (int64_t)j * ctxStride;
    uint16_t* out = absCoeff + (int64_t)j * 16;
    uint16_t tmp[16];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            const int t = c[(int64_t)i * jb.trSize + k];
            tmp[i * 4 + k] = (uint16_t)(t < 0 ? -t : t);
        }
    int pos = jb.scanPosSigOff;
    uint32_t first = pos < 15 ? 1 : 0, numNonZero = first, sum = 0, mask = jb.scanFlagMask;
    do
    {
        const uint32_t blkPos = scan[pos];
        const uint32_t posZeroMask = (jb.subPosBase + pos) ? ~0u : 0u;
        const uint32_t sig = mask & 1;
        mask >>= 1;
        if (pos != 0 || jb.subPosBase == 0 || numNonZero)
        {
            const uint32_t ctxSig = (uint32_t)(jb.tabSigCtx[blkPos] + jb.offset) & posZeroMask;
            sum += bin_cost(&ctx[ctxSig], sig);
        }
        out[numNonZero - first] = tmp[blkPos];
        numNonZero += sig;
        pos--;
    }
    while (pos >= 0);
    bits[j] = sum & 0xFFFFFF;
}

// costCoeffRemain_c, dct.cpp:901-946
__global__ __launch_bounds__(256) void cost_coeff_remain_kernel(const uint16_t* __restrict__ absCoeff, const int32_t* __restrict__ numNonZero,
                                                                const int32_t* __restrict__ firstIdx, int n, uint32_t* __restrict__ bits)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n)
        return;
    const uint16_t* a = absCoeff + (int64_t)j * 16;
    const int nnz = numNonZero[j];
    int idx = firstIdx[j], baseLevel = 3;
    uint32_t rice = 0, sum = 0;
    do
    {
        if (idx >= 8)                                  // C1FLAG_NUMBER
            baseLevel = 1;
        const uint32_t lvl = a[idx];
        int code = (int)lvl - baseLevel;
        if (code >= 0)
        {
            code = (int)((uint32_t)code >> rice) - 3;  // COEF_REMAIN_BIN_REDUCTION
            if (code >= 0)
                code = 2 * (31 - __clz(code + 1));
            sum += 3 + 1 + rice + (uint32_t)code;          // a negative code (short prefix) is added as is, dct.cpp:933
            if (lvl > (3u << rice))
                rice = (rice + 1) - (rice >> 2);
        }
        baseLevel = 2;
        idx++;
    }
    while (idx < nnz);
    bits[j] = sum;
}

// costC1C2Flag_c, dct.cpp:949-1006
__global__ __launch_bounds__(256) void cost_c1c2_kernel(const uint16_t* __restrict__ absCoeff, const int32_t* __restrict__ numC1Flag,
                                                        uint8_t* __restrict__ baseCtxMod, int ctxStride, int ctxOffset, int n, uint32_t* __restrict__ out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n)
        return;
    const uint16_t* a = absCoeff + (int64_t)j * 16;
    uint8_t* ctx = baseCtxMod + (int64_t)j * ctxStride;
    const int cnt = numC1Flag[j];
    uint32_t sum = 0, c1 = 1, firstC2Idx = 8, firstC2Flag = 2, c1Next = 0xFFFFFFFEu;
    int idx = 0;
    do
    {
        const uint32_t s1 = a[idx] > 1, s2 = a[idx] > 2;
        sum += bin_cost(&ctx[c1], s1) & 0xFFFFFF;
        if (s1)
            c1Next = 0;
        if (s1 + firstC2Flag == 3)
            firstC2Flag = s2;
        if (s1 + firstC2Idx == 9)
            firstC2Idx = (uint32_t)idx;
        c1 = c1Next & 3;
        c1Next >>= 2;
        idx++;
    }
    while (idx < cnt);
    if (!c1)
        sum += bin_cost(&ctx[ctxOffset], firstC2Flag) & 0xFFFFFF;
    out[j] = (sum & 0x00FFFFFF) + (c1 << 26) + (firstC2Idx << 28);
}

// the CABAC table reaches each device the first time a cost kernel is about to run there
static int need_state_bits(const char* who)
{
    const int dev = current_device();
    if (dev < 0)
        return set_error(X265HIP_EHIP, "%s: no current device", who);
    std::lock_guard<std::mutex> g(s_tabLock);
    if (!s_haveStateBits)
        return set_error(X265HIP_EINVAL, "%s: x265hip_set_entropy_state_bits has not been called", who);
    if (!s_stateBitsOn[dev])
    {
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_stateBits), s_stateBitsHost, sizeof(s_stateBitsHost)) != hipSuccess)
            return set_error(X265HIP_EHIP, "%s: CABAC table upload failed", who);
        s_stateBitsOn[dev] = true;
    }
    return X265HIP_OK;
}

} // namespace xh

using namespace xh;

extern "C" int x265hip_set_entropy_state_bits(const uint32_t* bits)
{
    XH_CHECK_DEV();
    if (!bits)
        return set_error(X265HIP_EINVAL, "set_entropy_state_bits: null table");
    std::lock_guard<std::mutex> g(s_tabLock);
    memcpy(s_stateBitsHost, bits, sizeof(s_stateBitsHost));
    s_haveStateBits = true;
    for (int d = 0; d < kMaxDevices; d++)
        s_stateBitsOn[d] = false;                  // every device re-uploads before its next cost kernel (need_state_bits)
    return X265HIP_OK;
}

extern "C" int x265hip_scan_pos_last_batch(int log2TrSize, int scanType, const int16_t* coeff, int n, uint16_t* coeffSign, uint16_t* coeffFlag,
                                           uint8_t* coeffNum, int32_t* lastPos, void* stream)
{
    XH_CHECK_DEV();
    if (log2TrSize < 2 || log2TrSize > 5 || scanType < 0 || scanType > 2 || n < 0)
        return set_error(X265HIP_EINVAL, "scan_pos_last: log2TrSize %d scanType %d n %d", log2TrSize, scanType, n);
    int e = ensure_scan_tables();
    if (e) return e;
    if (!n) return X265HIP_OK;
    hipLaunchKernelGGL(scan_pos_last_kernel, dim3((n + 3) / 4), dim3(256), 0, as_stream(stream), coeff, log2TrSize, scanType, n, coeffSign, coeffFlag,
                       coeffNum, lastPos);
    XH_LAUNCH_CHECK("scan_pos_last_kernel");
    return X265HIP_OK;
}

extern "C" int x265hip_find_pos_first_last_batch(const int16_t* coeff, const int64_t* cgOffsets, int64_t trSize, int scanType, int n, uint32_t* out,
                                                 void* stream)
{
    XH_CHECK_DEV();
    if (scanType < 0 || scanType > 2 || n < 0 || trSize < 4)
        return set_error(X265HIP_EINVAL, "find_pos_first_last: scanType %d n %d trSize %lld", scanType, n, (long long)trSize);
    int e = ensure_scan_tables();
    if (e) return e;
    if (!n) return X265HIP_OK;
    hipLaunchKernelGGL(find_pos_first_last_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), coeff, cgOffsets, trSize, scanType, n, out);
    XH_LAUNCH_CHECK("find_pos_first_last_kernel");
    return X265HIP_OK;
}

extern "C" int x265hip_cost_coeff_nxn_batch(const int16_t* coeff, const x265hip_coeff_group_job* jobs, int n, uint8_t* baseCtx, int ctxStride,
                                            uint16_t* absCoeff, uint32_t* bits, void* stream)
{
    XH_CHECK_DEV();
    if (n < 0 || ctxStride < 16)
        return set_error(X265HIP_EINVAL, "cost_coeff_nxn: n %d ctxStride %d", n, ctxStride);
    int e = need_state_bits("cost_coeff_nxn");
    if (e) return e;
    e = ensure_scan_tables();
    if (e) return e;
    if (!n) return X265HIP_OK;
    hipLaunchKernelGGL(cost_coeff_nxn_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), coeff, jobs, n, baseCtx, ctxStride, absCoeff, bits);
    XH_LAUNCH_CHECK("cost_coeff_nxn_kernel");
    return X265HIP_OK;
}

extern "C" int x265hip_cost_coeff_remain_batch(const uint16_t* absCoeff, const int32_t* numNonZero, const int32_t* firstIdx, int n, uint32_t* bits,
                                               void* stream)
{
    XH_CHECK_DEV();
    if (n < 0)
        return set_error(X265HIP_EINVAL, "cost_coeff_remain: n %d", n);
    if (!n) return X265HIP_OK;
    hipLaunchKernelGGL(cost_coeff_remain_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), absCoeff, numNonZero, firstIdx, n, bits);
    XH_LAUNCH_CHECK("cost_coeff_remain_kernel");
    return X265HIP_OK;
}

extern "C" int x265hip_cost_c1c2_flag_batch(const uint16_t* absCoeff, const int32_t* numC1Flag, uint8_t* baseCtxMod, int ctxStride, int ctxOffset, int n,
                                            uint32_t* out, void* stream)
{
    XH_CHECK_DEV();
    if (n < 0 || ctxStride < 4 || ctxOffset < 0 || ctxOffset >= ctxStride)
        return set_error(X265HIP_EINVAL, "cost_c1c2_flag: n %d ctxStride %d ctxOffset %d", n, ctxStride, ctxOffset);
    int e = need_state_bits("cost_c1c2_flag");
    if (e) return e;
    if (!n) return X265HIP_OK;
    hipLaunchKernelGGL(cost_c1c2_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), absCoeff, numC1Flag, baseCtxMod, ctxStride, ctxOffset, n, out);
    XH_LAUNCH_CHECK("cost_c1c2_kernel");
    return X265HIP_OK;
}
