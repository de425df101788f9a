// frame.hip — the fused residual chain of Search::estimateResidualQT for a batch of TUs (gfx950).
//
// Reference call sequence (search.cpp:3178-3330 -> quant.cpp:397 transformNxN, :543 invtransformNxN):
//   sub_ps (pixel.cpp:815) -> cu[].dct (dct.cpp:459-525) -> quant (dct.cpp:664; numSig) -> dequant_normal (dct.cpp:612)
//   -> cu[].idct (dct.cpp:544-610) -> add_ps (pixel.cpp:829) -> sse_pp (pixel.cpp:167)
// Each stage reproduces the corresponding primitive bit for bit; the int16 residual, the coefficients and the
// reconstructed residual never leave LDS/registers, which is the whole point: per TU the HBM traffic is
// N^2 * (fenc B + pred B + recon B + level 2) bytes (+ the shared quantCoeff row from L2) instead of the ~10 round trips
// of the per-primitive path (SURVEY.md §8d "fused dct->quant->dequant->idct->recon->sse").
// x265 short-cuts (numSig == 0 -> no inverse; DC-only -> blockfill) are arithmetic identities of the full inverse
// transform (HEVC conformance), so running the full path gives the same recon.
#include "common.h"
#include "dctcore.h"
#include "internal.h"

namespace xh {

template <typename P> __device__ __forceinline__ void load8(const P* p, int v[8]) { load4(p, v); load4(p + 4, v + 4); }
template <typename P> __device__ __forceinline__ void store8(P* p, const int v[8]) { store4(p, v); store4(p + 4, v + 4); }

struct QParams { int qBits, add, dqScale, dqShift, maxVal; };

typedef int16_t ChainLds[4][2][1024];

template <typename P, int N>
__device__ __forceinline__ void residual_chain_body(ChainLds& lds, int block, int blocks,
                                                    const P* __restrict__ fenc, int64_t sF, const P* __restrict__ pred, int64_t sP,
                                                    P* __restrict__ recon, int64_t sR,
                                                    const int32_t* __restrict__ offF, const int32_t* __restrict__ offP,
                                                    const int32_t* __restrict__ offR, const int32_t* __restrict__ quantCoeff,
                                                    QParams qp, int s1f, int s2f, int s1i, int s2i,
                                                    int16_t* __restrict__ level, uint32_t* __restrict__ numSig,
                                                    uint64_t* __restrict__ dist, int n)
{
    constexpr int G = (32 / N) * (32 / N);
    constexpr int LPT = N * N / 16;             // lanes per TU (each lane owns 16 elements)
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int16_t* buf0 = lds[wv][0];
    int16_t* buf1 = lds[wv][1];
    v4i bF, bI;
    int corrF, corrI;
    make_b_operand<N, false>(lane, bF, corrF);
    make_b_operand<N, true>(lane, bI, corrI);

    const int groups = (n + G - 1) / G;
    const int wavesTotal = blocks * 4;
    for (int grp = block * 4 + wv; grp < groups; grp += wavesTotal)
    {
        const int tu0 = grp * G;
        int fv[16], pv[16];
        // ---- residual = fenc - pred  (two runs of 8 pixels per lane)
#pragma unroll
        for (int half = 0; half < 2; half++)
        {
            const int e = lane * 16 + half * 8;
            const int g = e / (N * N), rr = (e % (N * N)) / N, cc = e % N;
            int r[8];
            if (tu0 + g < n)
            {
                load8(fenc + offF[tu0 + g] + rr * sF + cc, &fv[8 * half]);
                load8(pred + offP[tu0 + g] + rr * sP + cc, &pv[8 * half]);
            }
            else
            {
#pragma unroll
                for (int i = 0; i < 8; i++) { fv[8 * half + i] = 0; pv[8 * half + i] = 0; }
            }
#pragma unroll
            for (int i = 0; i < 8; i++) r[i] = fv[8 * half + i] - pv[8 * half + i];
            store8(buf0 + e, r);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        // ---- forward transform
        mfma_pass<N, false>(buf0, buf1, lane, bF, corrF, s1f);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        mfma_pass<N, false>(buf1, buf0, lane, bF, corrF, s2f);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        // ---- quant + dequant on the lane's 16 coefficients (contiguous inside one TU)
        int cnt = 0;
        const int tuL = tu0 + (lane * 16) / (N * N);
        const bool okL = tuL < n;
        const int dqAdd = 1 << (qp.dqShift - 1);
#pragma unroll
        for (int half = 0; half < 2; half++)
        {
            const int e = lane * 16 + half * 8;
            const int idx = e % (N * N);
            int cf[8], lv[8], dq[8], qc[8];
            load8(buf0 + e, cf);
            const int4 q0 = ld_unaligned<int4>(quantCoeff + idx), q1 = ld_unaligned<int4>(quantCoeff + idx + 4);
            qc[0] = q0.x; qc[1] = q0.y; qc[2] = q0.z; qc[3] = q0.w; qc[4] = q1.x; qc[5] = q1.y; qc[6] = q1.z; qc[7] = q1.w;
#pragma unroll
            for (int i = 0; i < 8; i++)
            {
                const int sign = cf[i] < 0 ? -1 : 1;
                int l = (iabs(cf[i]) * qc[i] + qp.add) >> qp.qBits;
                cnt += l != 0;
                l = clip3i(-32768, 32767, l * sign);
                lv[i] = l;
                dq[i] = clip3i(-32768, 32767, (l * qp.dqScale + dqAdd) >> qp.dqShift);
            }
            if (okL)
                st_unaligned<uint4>(level + (int64_t)tuL * N * N + idx,
                                    make_uint4(((uint32_t)lv[0] & 0xffff) | ((uint32_t)lv[1] << 16), ((uint32_t)lv[2] & 0xffff) | ((uint32_t)lv[3] << 16),
                                               ((uint32_t)lv[4] & 0xffff) | ((uint32_t)lv[5] << 16), ((uint32_t)lv[6] & 0xffff) | ((uint32_t)lv[7] << 16)));
            store8(buf0 + e, dq);
        }
        cnt = group_sum(cnt, LPT);
        if (okL && (lane & (LPT - 1)) == 0)
            numSig[tuL] = (uint32_t)cnt;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        // ---- inverse transform
        mfma_pass<N, true>(buf0, buf1, lane, bI, corrI, s1i);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        mfma_pass<N, true>(buf1, buf0, lane, bI, corrI, s2i);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        // ---- recon = clip(pred + resi'), distortion = sum (fenc - recon)^2
        unsigned long long sse = 0;
#pragma unroll
        for (int half = 0; half < 2; half++)
        {
            const int e = lane * 16 + half * 8;
            const int g = e / (N * N), rr = (e % (N * N)) / N, cc = e % N;
            int r[8], rec[8];
            load8(buf0 + e, r);
#pragma unroll
            for (int i = 0; i < 8; i++)
            {
                rec[i] = clip3i(0, qp.maxVal, pv[8 * half + i] + r[i]);
                const unsigned d = (unsigned)(fv[8 * half + i] - rec[i]);
                sse += d * d;
            }
            if (tu0 + g < n)
                store8(recon + offR[tu0 + g] + rr * sR + cc, rec);
        }
        sse = group_sum64(sse, LPT);
        if (okL && (lane & (LPT - 1)) == 0)
            dist[tuL] = sse;
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
}

template <typename P, int N>
__global__ __launch_bounds__(256) void residual_chain_kernel(const P* __restrict__ fenc, int64_t sF, const P* __restrict__ pred, int64_t sP,
                                                             P* __restrict__ recon, int64_t sR,
                                                             const int32_t* __restrict__ offF, const int32_t* __restrict__ offP,
                                                             const int32_t* __restrict__ offR, const int32_t* __restrict__ quantCoeff,
                                                             QParams qp, int s1f, int s2f, int s1i, int s2i,
                                                             int16_t* __restrict__ level, uint32_t* __restrict__ numSig,
                                                             uint64_t* __restrict__ dist, int n)
{
    __shared__ __attribute__((aligned(16))) ChainLds lds;
    residual_chain_body<P, N>(lds, blockIdx.x, gridDim.x, fenc, sF, pred, sP, recon, sR, offF, offP, offR, quantCoeff, qp, s1f, s2f, s1i, s2i, level, numSig, dist, n);
}

// 4x4 TUs: one lane per TU, the whole chain in registers (DCT4; the DST variant is intra-luma only and not part of
// the inter residual chain this entry point serves)
template <typename P>
__device__ __forceinline__ void residual_chain4_body(int block, int blocks,
                                                     const P* __restrict__ fenc, int64_t sF, const P* __restrict__ pred, int64_t sP,
                                                     P* __restrict__ recon, int64_t sR,
                                                     const int32_t* __restrict__ offF, const int32_t* __restrict__ offP,
                                                     const int32_t* __restrict__ offR, const int32_t* __restrict__ quantCoeff,
                                                     QParams qp, int s1f, int s2f, int s1i, int s2i,
                                                     int16_t* __restrict__ level, uint32_t* __restrict__ numSig,
                                                     uint64_t* __restrict__ dist, int n)
{
    for (int tu = block * blockDim.x + threadIdx.x; tu < n; tu += blocks * blockDim.x)
    {
        int f[16], p[16], x[16], t[16];
#pragma unroll
        for (int r = 0; r < 4; r++)
        {
            load4(fenc + offF[tu] + r * sF, &f[4 * r]);
            load4(pred + offP[tu] + r * sP, &p[4 * r]);
        }
#pragma unroll
        for (int i = 0; i < 16; i++) x[i] = f[i] - p[i];
        const int a1 = 1 << (s1f - 1), a2 = 1 << (s2f - 1);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                int s = 0;
#pragma unroll
                for (int m = 0; m < 4; m++) s += kDct4[k][m] * x[4 * j + m];
                t[4 * k + j] = (int)(int16_t)((s + a1) >> s1f);
            }
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                int s = 0;
#pragma unroll
                for (int m = 0; m < 4; m++) s += kDct4[k][m] * t[4 * j + m];
                x[4 * k + j] = (int)(int16_t)((s + a2) >> s2f);
            }
        int cnt = 0;
        const int dqAdd = 1 << (qp.dqShift - 1);
        int lv[16];
#pragma unroll
        for (int i = 0; i < 16; i++)
        {
            const int sign = x[i] < 0 ? -1 : 1;
            int l = (iabs(x[i]) * quantCoeff[i] + qp.add) >> qp.qBits;
            cnt += l != 0;
            l = clip3i(-32768, 32767, l * sign);
            lv[i] = l;
            x[i] = clip3i(-32768, 32767, (l * qp.dqScale + dqAdd) >> qp.dqShift);
        }
#pragma unroll
        for (int r = 0; r < 4; r++)
            store4(level + (int64_t)tu * 16 + 4 * r, &lv[4 * r]);
        numSig[tu] = (uint32_t)cnt;
        const int b1 = 1 << (s1i - 1), b2 = 1 << (s2i - 1);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                int s = 0;
#pragma unroll
                for (int m = 0; m < 4; m++) s += kDct4[m][k] * x[4 * m + j];
                t[4 * j + k] = clip3i(-32768, 32767, (s + b1) >> s1i);
            }
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                int s = 0;
#pragma unroll
                for (int m = 0; m < 4; m++) s += kDct4[m][k] * t[4 * m + j];
                x[4 * j + k] = clip3i(-32768, 32767, (s + b2) >> s2i);
            }
        unsigned long long sse = 0;
        int rec[16];
#pragma unroll
        for (int i = 0; i < 16; i++)
        {
            rec[i] = clip3i(0, qp.maxVal, p[i] + x[i]);
            const unsigned d = (unsigned)(f[i] - rec[i]);
            sse += d * d;
        }
#pragma unroll
        for (int r = 0; r < 4; r++)
            store4(recon + offR[tu] + r * sR, &rec[4 * r]);
        dist[tu] = sse;
    }
}

template <typename P>
__global__ __launch_bounds__(256) void residual_chain4_kernel(const P* __restrict__ fenc, int64_t sF, const P* __restrict__ pred, int64_t sP,
                                                              P* __restrict__ recon, int64_t sR,
                                                              const int32_t* __restrict__ offF, const int32_t* __restrict__ offP,
                                                              const int32_t* __restrict__ offR, const int32_t* __restrict__ quantCoeff,
                                                              QParams qp, int s1f, int s2f, int s1i, int s2i,
                                                              int16_t* __restrict__ level, uint32_t* __restrict__ numSig,
                                                              uint64_t* __restrict__ dist, int n)
{
    residual_chain4_body<P>(blockIdx.x, gridDim.x, fenc, sF, pred, sP, recon, sR, offF, offP, offR, quantCoeff, qp, s1f, s2f, s1i, s2i, level, numSig, dist, n);
}

// ---- several chains in ONE launch: the frame pass runs luma 32x32, luma 8x8, chroma 16x16 and chroma 4x4 TUs back to back; as separate
// launches the three small ones are ~10 us of pure latency each.  A workgroup finds its segment from its block index; segments may differ in
// planes, strides, TU size and quantiser.
struct ChainSeg
{
    const void* fenc; const void* pred; void* recon;
    int64_t sF, sP, sR;
    const int32_t* offF; const int32_t* offP; const int32_t* offR; const int32_t* quantCoeff;
    QParams qp;
    int s1f, s2f, s1i, s2i;
    int16_t* level; uint32_t* numSig; uint64_t* dist;
    int n, size, firstBlock, blocks;
};
struct ChainSegs { ChainSeg s[4]; int count; };

template <typename P>
__global__ __launch_bounds__(256) void residual_chain_multi_kernel(ChainSegs segs)
{
    __shared__ __attribute__((aligned(16))) ChainLds lds;
    int k = 0;
#pragma unroll
    for (int i = 1; i < 4; i++)
        if (i < segs.count && (int)blockIdx.x >= segs.s[i].firstBlock) k = i;
    const ChainSeg& g = segs.s[k];
    const int block = blockIdx.x - g.firstBlock;
    const P* f = (const P*)g.fenc;
    const P* p = (const P*)g.pred;
    P* r = (P*)g.recon;
#define CH_ARGS f, g.sF, p, g.sP, r, g.sR, g.offF, g.offP, g.offR, g.quantCoeff, g.qp, g.s1f, g.s2f, g.s1i, g.s2i, g.level, g.numSig, g.dist, g.n
    if (g.size == 32) residual_chain_body<P, 32>(lds, block, g.blocks, CH_ARGS);
    else if (g.size == 16) residual_chain_body<P, 16>(lds, block, g.blocks, CH_ARGS);
    else if (g.size == 8) residual_chain_body<P, 8>(lds, block, g.blocks, CH_ARGS);
    else residual_chain4_body<P>(block, g.blocks, CH_ARGS);
#undef CH_ARGS
}

template <typename P>
static int launch_chain(int size, int depth, const void* fenc, int64_t sF, const void* pred, int64_t sP, void* recon, int64_t sR,
                        const int32_t* offF, const int32_t* offP, const int32_t* offR, const int32_t* quantCoeff,
                        QParams qp, int16_t* level, uint32_t* numSig, uint64_t* dist, int n, hipStream_t st)
{
    const int log2n = size == 4 ? 2 : size == 8 ? 3 : size == 16 ? 4 : 5;
    const int s1f = log2n - 1 + depth - 8, s2f = log2n + 6, s1i = 7, s2i = 12 - (depth - 8);
    const P* f = (const P*)fenc;
    const P* p = (const P*)pred;
    P* r = (P*)recon;
    if (size == 4)
    {
        hipLaunchKernelGGL((residual_chain4_kernel<P>), dim3(grid_for((n + 255) / 256)), dim3(256), 0, st, f, sF, p, sP, r, sR,
                           offF, offP, offR, quantCoeff, qp, s1f, s2f, s1i, s2i, level, numSig, dist, n);
    }
    else
    {
        const int G = (32 / size) * (32 / size);
        dim3 grid(grid_for(((n + G - 1) / G + 3) / 4)), block(256);
        if (size == 8)
            hipLaunchKernelGGL((residual_chain_kernel<P, 8>), grid, block, 0, st, f, sF, p, sP, r, sR, offF, offP, offR, quantCoeff, qp, s1f, s2f, s1i, s2i, level, numSig, dist, n);
        else if (size == 16)
            hipLaunchKernelGGL((residual_chain_kernel<P, 16>), grid, block, 0, st, f, sF, p, sP, r, sR, offF, offP, offR, quantCoeff, qp, s1f, s2f, s1i, s2i, level, numSig, dist, n);
        else
            hipLaunchKernelGGL((residual_chain_kernel<P, 32>), grid, block, 0, st, f, sF, p, sP, r, sR, offF, offP, offR, quantCoeff, qp, s1f, s2f, s1i, s2i, level, numSig, dist, n);
    }
    XH_LAUNCH_CHECK("residual_chain_kernel");
    return X265HIP_OK;
}

// internal (framepass.hip): up to four chains of different TU sizes / planes / quantisers in one launch
int residual_chain_multi(int depth, const ChainJob* jobs, int count, hipStream_t st)
{
    if (count < 1 || count > 4)
        return set_error(X265HIP_EINVAL, "residual_chain_multi: %d segments", count);
    ChainSegs segs{};
    int blocks = 0, used = 0;
    for (int i = 0; i < count; i++)
    {
        const ChainJob& j = jobs[i];
        if (!j.n) continue;
        if (!(j.size == 4 || j.size == 8 || j.size == 16 || j.size == 32) || j.qBits < 1 || j.dqShift < 1)
            return set_error(X265HIP_EINVAL, "residual_chain_multi: segment %d size %d", i, j.size);
        ChainSeg& g = segs.s[used++];
        const int log2n = j.size == 4 ? 2 : j.size == 8 ? 3 : j.size == 16 ? 4 : 5;
        g.fenc = j.fenc; g.pred = j.pred; g.recon = j.recon; g.sF = j.sF; g.sP = j.sP; g.sR = j.sR;
        g.offF = j.offF; g.offP = j.offP; g.offR = j.offR; g.quantCoeff = j.quantCoeff;
        g.qp = QParams{ j.qBits, j.add, j.dqScale, j.dqShift, (1 << depth) - 1 };
        g.s1f = log2n - 1 + depth - 8; g.s2f = log2n + 6; g.s1i = 7; g.s2i = 12 - (depth - 8);
        g.level = j.level; g.numSig = j.numSig; g.dist = j.dist; g.n = j.n; g.size = j.size;
        const int G = j.size == 4 ? 1 : (32 / j.size) * (32 / j.size);
        g.blocks = j.size == 4 ? grid_for((j.n + 255) / 256) : grid_for(((j.n + G - 1) / G + 3) / 4);
        g.firstBlock = blocks;
        blocks += g.blocks;
    }
    segs.count = used;
    if (!used) return X265HIP_OK;
    if (depth == 8)
        hipLaunchKernelGGL((residual_chain_multi_kernel<uint8_t>), dim3(blocks), dim3(256), 0, st, segs);
    else
        hipLaunchKernelGGL((residual_chain_multi_kernel<uint16_t>), dim3(blocks), dim3(256), 0, st, segs);
    XH_LAUNCH_CHECK("residual_chain_multi_kernel");
    return X265HIP_OK;
}

} // namespace xh

using namespace xh;

extern "C" int x265hip_residual_chain_batch(int size, int depth, const void* fenc, int64_t strideF, const void* pred,
                                            int64_t strideP, void* recon, int64_t strideR, const int32_t* offF,
                                            const int32_t* offP, const int32_t* offR, const int32_t* quantCoeff,
                                            int qBits, int add, int dqScale, int dqShift, int16_t* level,
                                            uint32_t* numSig, uint64_t* dist, int n, void* stream)
{
    XH_CHECK_DEV();
    if (!valid_depth(depth) || n < 0 || !(size == 4 || size == 8 || size == 16 || size == 32) || qBits < 1 || dqShift < 1)
        return set_error(X265HIP_EINVAL, "residual_chain: size %d depth %d n %d qBits %d dqShift %d", size, depth, n, qBits, dqShift);
    if (!n) return X265HIP_OK;
    QParams qp = { qBits, add, dqScale, dqShift, (1 << depth) - 1 };
    return depth == 8 ? launch_chain<uint8_t>(size, depth, fenc, strideF, pred, strideP, recon, strideR, offF, offP, offR, quantCoeff, qp, level, numSig, dist, n, as_stream(stream))
                      : launch_chain<uint16_t>(size, depth, fenc, strideF, pred, strideP, recon, strideR, offF, offP, offR, quantCoeff, qp, level, numSig, dist, n, as_stream(stream));
}
