// cuserve.hip — CU residual quad-tree jobs (include/x265hip.h, x265hip_cuserve_*): the transform arithmetic of one inter CU handed over by a
// host thread that WAITS for it, so the figure of merit is the round trip, not throughput.
//
// Reference arithmetic (per transform unit; search.cpp:3178-3560 is the caller on the host):
//   Quant::transformNxN    quant.cpp:397-470   cu[].dct (dct.cpp:459-525) -> quant_c (dct.cpp:664-686) -> signBitHidingHDQ (quant.cpp:246-395,
//                                              with scanPosLast_c dct.cpp:757-788 on the up-right diagonal scan: inter units always scan diagonally,
//                                              cudata.cpp:2088)
//   Quant::invtransformNxN quant.cpp:543-603   dequant_normal (dct.cpp:612-630) -> cu[].idct (dct.cpp:544-610); the DC-only shortcut (:586-596) is
//                                              an arithmetic identity of the full inverse transform
//   sse_pp(source, prediction), sse_pp(source, clip(prediction + residual'))   pixel.cpp:167-186, add_ps :829
//
// Hand-off: a SLOT is a block of page-locked, host-coherent memory (job header + pixel block in, unit headers + levels + reconstructed residual
// out).  The device reads and writes it directly over PCIe; there is no staging copy, no stream synchronisation and no event on the path:
//   mode 0 (resident)   one workgroup per slot, started on the server stream, polls the slot's doorbell word (system-scope atomic load,
//                       s_sleep between polls); a host thread submits by storing the next sequence number.  A server that has seen no work for
//                       `idleUs` ends by itself (nothing in the process may wait on it for long: hipFree synchronises every stream) and is
//                       started again by the next submitter that finds it gone.
//   mode 1 (launch)     one 1-workgroup launch per job on the slot's own stream.
// Completion: every unit's `ready` word takes the job's sequence number when its results are in host memory (released with a system-scope
// fence); luma units of the largest size first, so the host's entropy coder can start while chroma is still on the device.
//
// One workgroup = 4 waves; a wave owns a 32x32 MFMA tile = one 32x32 unit, four 16x16 or sixteen 8x8 units of ONE plane (dctcore.h); the CU's
// source and prediction are staged in LDS once (every level re-reads them).  Sign-bit hiding: one lane per 4x4 coefficient group (64 per tile).
#include "common.h"
#include "dctcore.h"
#include "tiles.h"
#include "intra_dev.h"
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>

namespace xh {

void clock_add(int clk, uint64_t spans, uint64_t ns, uint64_t bytes);     // runtime.hip
int place_device(int place);

namespace {

constexpr int kInts = X265HIP_CUJOB_MAX_ELEMS;

// A slot = a mailbox of two halves, each living where its READER is:
//   SlotIn   host -> device.  DEVICE memory (uncached for the GPU's L2: hipDeviceMallocUncached) that the host writes through the large BAR — posted
//            writes, no round trip; the server polls the doorbell and fetches header + pixels from its own HBM.  Against a mailbox in host memory the
//            hand-off loses two PCIe read round trips (tools/micro/bar_mailbox: doorbell -> echo 10.6 us -> 6.6 us on the MI355X box).  The host never
//            READS this half (a load through the BAR is a microsecond): x265hip_cuserve_slot hands out a shadow header in ordinary memory.
//            Without a large BAR (or X265HIP_CUSERVE_MAILBOX=host): host-coherent page-locked memory, as in the first version.
//   SlotOut  device -> host.  Host-coherent page-locked memory: the device's stores are posted writes as well, the host polls its own memory.
struct SlotIn
{
    uint32_t doorbell;                           // host -> device: sequence number of the job below (mode 0)
    uint32_t pad0[31];
    alignas(128) x265hip_cujob job;              // 80 bytes; the pixel block follows at +128 so that header and pixels are ONE run of 16-byte chunks
    alignas(128) unsigned char pixels[X265HIP_CUJOB_PIXEL_BYTES];
};
struct SlotOut
{
    uint32_t failed;                             // device -> host: a job the device could not do (never expected)
    uint32_t pad1[15];
    alignas(64) x265hip_cujob_unit units[X265HIP_CUJOB_MAX_UNITS];
    alignas(64) int16_t levels[kInts];
    alignas(64) int16_t resi[kInts];
};

// ---- up-right diagonal scans (6.5.3): position k of the scan of an n x n grid -> (x, y) ---------------------------------------------------------
struct DiagScans { uint8_t s2[4], s4[16], s8[64]; };      // value = y * n + x
constexpr DiagScans make_diag()
{
    DiagScans d{};
    for (int which = 0; which < 3; which++)
    {
        const int n = which == 0 ? 2 : which == 1 ? 4 : 8;
        int i = 0, x = 0, y = 0;
        while (i < n * n)
        {
            while (y >= 0)
            {
                if (x < n && y < n)
                {
                    const uint8_t v = (uint8_t)(y * n + x);
                    if (which == 0) d.s2[i] = v; else if (which == 1) d.s4[i] = v; else d.s8[i] = v;
                    i++;
                }
                y--; x++;
            }
            y = x; x = 0;
        }
    }
    return d;
}
__device__ __constant__ const DiagScans kDiag = make_diag();

struct HostCtl                                    // host-coherent page-locked memory, one per x265hip_cuserve
{
    uint32_t serverState;                        // 0: no server; 0x80000000 | g: generation g is being started; g: generation g is polling
    uint32_t pad[15];
    uint32_t leave;                              // host -> device: the server is asked to leave (somebody has to synchronise the device: servers_pause)
};
struct DevCtl                                     // device memory
{
    uint32_t quit;                               // set by the workgroup that finds the server idle (or told to leave): every workgroup leaves at its next poll
    uint32_t left;                               // workgroups that have left
    uint64_t lastWork;                           // wall_clock64() of the last job any workgroup has taken
    uint64_t busyTicks[512];                     // per workgroup (two per slot): 100 MHz ticks spent on its jobs
};

struct TileLds { int16_t a[1024], b[1024], c[1024]; };            // per wave: transform ping-pong + deltaU
struct BOperand { int b[4]; int corr; int pad[3]; };             // make_b_operand's result for one lane
// the team form of a 32x32 unit (team_chain32 below: the four waves of a workgroup on ONE unit): a wave's 16x16x32 coefficient operand, and the words the
// waves exchange at the barriers
struct TeamOperand { long b; int corr; int pad; };
struct TeamLds
{
    TeamOperand op[2][256];                      // [forward, inverse][thread]: built once per kernel
    TeamOperand op16[2][64];                     // the same for a 16x16 unit on one wave (solo_chain16)
    uint32_t part[16];                           // per 16-lane row of the team: partial sums (significant levels, squared differences)
    uint32_t part2[16];
    int energy[2][16];                           // psy energies of the unit's sixteen 8x8 blocks: [source, reconstruction]
    int numSig;
};
struct JobLds
{
    alignas(16) x265hip_cujob job;               // + padding up to 128 bytes, then the pixels: the same run of chunks as in the slot
    uint32_t pad[(128 - sizeof(x265hip_cujob)) / 4];
    alignas(16) unsigned char pix[X265HIP_CUJOB_PIXEL_BYTES];
    alignas(16) TileLds tile[4];
    int saoExtra[3520];                          // directly behind tile[]: an SAO statistics job lays its histograms over both (run_sao: 9 664 ints)
    alignas(16) BOperand bop[3][2][64];          // [log2n - 3][forward, inverse][lane]: built once per kernel
    alignas(16) TeamLds team;
    uint32_t seq;
};
static_assert(sizeof(x265hip_cujob) <= 128, "job header");

// ticket = what the host rings and the units' ready words take: bits 31..8 a running number (never 0, never 0xffffff), bits 7..0 what the device
// needs to know before it has read anything: log2CUSize - 4 (bits 1..0), chroma (bit 2), 16-bit samples (bit 3), an inverse job's levels block (bit 4)
// bits 1..0 == 3: an SAO statistics job (x265hip_saojob): bits 7..2 = the job's size, header included, in 512-byte steps
__host__ __device__ inline uint32_t ticket_bytes(uint32_t t)
{
    if ((t & 3) == 3) return ((t >> 2) & 63u) * 512u - 128u;
    const uint32_t n2 = 1u << (2 * ((t & 3) + 4)), elems = (t & 4) ? n2 + n2 / 2 : n2;
    return 2 * elems * ((t & 8) ? 2 : 1);
}

struct PlaneParams { int qBits, add, quantScale, dqScale, dqShift, s1f, s2f, s1i, s2i, maxVal; };

__device__ __forceinline__ PlaneParams plane_params(const x265hip_cujob& j, int plane, int log2n)
{
    // quant.cpp:408 transformShift = MAX_TR_DYNAMIC_RANGE(15) - depth - log2TrSize; :461 qbits = QUANT_SHIFT(14) + per + transformShift;
    // :466 add = offset << (qbits - 9); :556 shift = QUANT_IQUANT_SHIFT(20) - QUANT_SHIFT - transformShift; :567 scale = invQuantScales[rem] << per
    // dct.cpp:459-525: forward shifts log2n - 1 + (depth - 8), log2n + 6; :544-610 inverse shifts 7, 12 - (depth - 8)
    PlaneParams p;
    const int depth = (int)j.bitDepth, transformShift = 15 - depth - log2n;
    p.qBits = 14 + j.qpPer[plane] + transformShift;
    p.add = (int)j.quantOffset << (p.qBits - 9);
    p.quantScale = j.quantScale[plane];
    p.dqScale = j.dequantScale[plane] << j.qpPer[plane];
    p.dqShift = 20 - 14 - transformShift;
    p.s1f = log2n - 1 + (depth - 8); p.s2f = log2n + 6;
    p.s1i = 7; p.s2i = 12 - (depth - 8);
    p.maxVal = (1 << depth) - 1;
    return p;
}

// One tile: units u0 .. u0 + G - 1 (G = (32 / N)^2, raster order, `count` of them exist) of size N x N of one plane.
//   src / prd: the plane's source and prediction in LDS, `pw` elements per row; the plane has (pw / N)^2 units
template <typename P, int N>
__device__ __forceinline__ void tile_chain(TileLds& t, const BOperand (*bop)[64], const P* src, const P* prd, int pw, int u0, int count, const PlaneParams qp,
                                           bool signHide, x265hip_cujob_unit* units, int unitBase, int16_t* levels, int16_t* resi, int elemBase, uint32_t seq, uint64_t t0, bool stamps,
                                           int coef)
{
    // job.reserved != 0 (tools/micro/cuserve_rt): 100 MHz ticks since the doorbell was seen at six points of the chain, two per reserved word of the unit
    uint32_t stamp[6] = { 0, 0, 0, 0, 0, 0 };
#define XH_STAMP(i) do { if (stamps) stamp[i] = (uint32_t)(wall_clock64() - t0) & 0xffffu; } while (0)
    XH_STAMP(0);
    constexpr int G = (32 / N) * (32 / N);
    constexpr int LPT = N * N / 16;              // lanes per unit: 16 coefficients each in the quantiser, one 4x4 group each in the sign hiding
    constexpr int CGW = N / 4;                   // coefficient groups per row of a unit
    const int lane = threadIdx.x & 63;
    const int perRow = pw / N;
    const BOperand& oF = bop[0][lane];
    const BOperand& oI = bop[1][lane];
    const v4i bF = { oF.b[0], oF.b[1], oF.b[2], oF.b[3] }, bI = { oI.b[0], oI.b[1], oI.b[2], oI.b[3] };
    const int corrF = oF.corr, corrI = oI.corr;

    // coef (x265hip_cujob::coefMode): bit 0 the residual's transform coefficients go out, bit 1 the source block's; bit 2: the two parts are separate work
    // items on different waves (the residual part releases `ready`, the source part `readyInv`)
    const bool srcOnly = (coef & 3) == 2;
    // ---- residual = source - prediction (two runs of 8 per lane; kept in registers for the distortions)
    int fv[16], pv[16];
#pragma unroll
    for (int half = 0; half < 2; half++)
    {
        const int e = lane * 16 + half * 8;
        const int g = e / (N * N), rr = (e % (N * N)) / N, cc = e % N;
        int r[8];
        if (g < count)
        {
            const int u = u0 + g, ux = u % perRow, uy = u / perRow;
            const int off = (uy * N + rr) * pw + ux * N + cc;
            load4(src + off, &fv[8 * half]); load4(src + off + 4, &fv[8 * half + 4]);
            load4(prd + off, &pv[8 * half]); load4(prd + off + 4, &pv[8 * half + 4]);
        }
        else
        {
#pragma unroll
            for (int i = 0; i < 8; i++) { fv[8 * half + i] = 0; pv[8 * half + i] = 0; }
        }
        // (coefficient mode, source part: the SOURCE samples are transformed — m_fencDctCoeff, quant.cpp:436-442 — by the same two passes)
#pragma unroll
        for (int i = 0; i < 8; i++) r[i] = srcOnly ? fv[8 * half + i] : fv[8 * half + i] - pv[8 * half + i];
        store4(t.a + e, r); store4(t.a + e + 4, r + 4);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    // ---- forward transform: a -> b -> a
    mfma_pass<N, false>(t.a, t.b, lane, bF, corrF, qp.s1f);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    mfma_pass<N, false>(t.b, t.a, lane, bF, corrF, qp.s2f);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    XH_STAMP(1);
    if (coef)
    {
        // ---- coefficient mode (x265hip_cujob::coefMode): the host quantises (Quant::rdoQuant).  The unit's transform coefficients go out where the levels would
        // (residual part) or where the reconstructed residual would (source part: m_fencDctCoeff for psy-rdoq)
        const int gC = (lane * 16) / (N * N);
        const bool okC = gC < count;
        int16_t* dstC = srcOnly ? resi : levels;
#pragma unroll
        for (int half = 0; half < 2; half++)
        {
            const int e = lane * 16 + half * 8;
            if (okC)
                *reinterpret_cast<uint4*>(dstC + elemBase + (u0 + gC) * N * N + (e % (N * N))) = *reinterpret_cast<const uint4*>(t.a + e);
        }
        x265hip_cujob_unit* unC = units + unitBase + u0 + gC;
        const bool writerC = okC && (lane & (LPT - 1)) == 0;
        if (srcOnly)
        {
            // (the release store waits for every store this wave has issued: the block above is in host memory when the word is seen)
            if (writerC)
                __hip_atomic_store(&unC->readyInv, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            return;
        }
        unsigned long long zeroC = 0;
#pragma unroll
        for (int i = 0; i < 16; i++)
        {
            const int d0 = fv[i] - pv[i];
            zeroC += (unsigned)(d0 * d0);
        }
        zeroC = group_sum64(zeroC, LPT);
        if (writerC)
        {
            unC->numSig = 0;
            unC->zeroDist = zeroC;
            unC->fwdTicks = (uint32_t)(wall_clock64() - t0);
            XH_STAMP(5);
            if (stamps) { unC->reserved[0] = stamp[0] | (stamp[1] << 16); unC->reserved[1] = 0; unC->reserved[2] = stamp[5] << 16; }
            if (!(coef & 4))
                __hip_atomic_store(&unC->readyInv, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&unC->ready, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        return;
    }
    // ---- quant (quant_c): levels -> b, deltaU -> c; the transform coefficients stay in a (sign hiding reads their signs)
    int cnt = 0;
    const int qBits8 = qp.qBits - 8;
#pragma unroll
    for (int half = 0; half < 2; half++)
    {
        const int e = lane * 16 + half * 8;
        int cf[8], lv[8], du[8];
        load4(t.a + e, cf); load4(t.a + e + 4, cf + 4);
#pragma unroll
        for (int i = 0; i < 8; i++)
        {
            const int tmp = iabs(cf[i]) * qp.quantScale;
            const int l = (tmp + qp.add) >> qp.qBits;
            du[i] = (tmp - (l << qp.qBits)) >> qBits8;
            cnt += l != 0;
            lv[i] = clip3i(-32768, 32767, cf[i] < 0 ? -l : l);
        }
        store4(t.b + e, lv); store4(t.b + e + 4, lv + 4);
        store4(t.c + e, du); store4(t.c + e + 4, du + 4);
    }
    int numSig = group_sum(cnt, LPT);                       // of the unit this lane belongs to (lanes g * LPT .. g * LPT + LPT - 1)
    __builtin_amdgcn_s_waitcnt(0xc07f);
    XH_STAMP(2);
    // ---- sign-bit hiding (signBitHidingHDQ): lane = coefficient group `cg` (scan order) of unit lane / LPT
    {
        const int g = lane / LPT, cg = lane % LPT;
        const int cgPos = CGW == 2 ? kDiag.s2[cg] : CGW == 4 ? kDiag.s4[cg] : kDiag.s8[cg];
        const int cgx = cgPos % CGW, cgy = cgPos / CGW;
        const int base = g * N * N + (cgy * 4) * N + cgx * 4;
        int lv[16];
        uint32_t flags = 0;                                  // bit (15 - n) = level at scan position n of the group is non-zero
#pragma unroll
        for (int n = 0; n < 16; n++)
        {
            const int p4 = kDiag.s4[n];
            lv[n] = t.b[base + (p4 >> 2) * N + (p4 & 3)];
            flags |= (uint32_t)(lv[n] != 0) << (15 - n);
        }
        // the unit's last non-zero group in scan order (scanPosLast_c walks the scan until numSig non-zero levels were seen)
        const unsigned long long nz = __ballot(flags != 0);
        unsigned long long mine = nz;
        if constexpr (LPT < 64) mine = (nz >> (g * LPT)) & ((1ull << LPT) - 1);
        const int cgLast = mine ? 63 - __builtin_clzll(mine) : -1;
        int delta = 0;
        if (signHide && numSig >= 2 && flags && cg <= cgLast)
        {
            const int firstNZ = 15 ^ (31 - __builtin_clz(flags));            // quant.cpp:303-307
            const int lastNZ = 15 ^ __builtin_ctz(flags);
            if (lastNZ - firstNZ >= 4)                                       // SBH_THRESHOLD
            {
                const uint32_t signbit = lv[firstNZ] > 0 ? 0 : 1;
                int absSum = 0;
#pragma unroll
                for (int n = 0; n < 16; n++)
                    if (n >= firstNZ && n <= lastNZ) absSum += lv[n];
                if (signbit != ((uint32_t)absSum & 1))
                {
                    int minCostInc = 0x7fffffff, minN = -1, finalChange = 0, curChange = 0;
                    const int start = cg == cgLast ? lastNZ : 15;
                    uint32_t cgFlags = flags >> (15 - start);               // bit 0 = position `start`
#pragma unroll
                    for (int n = 15; n >= 0; n--)
                    {
                        if (n > start) continue;
                        const int p4 = kDiag.s4[n];
                        const int at = base + (p4 >> 2) * N + (p4 & 3);
                        const int dU = t.c[at];
                        int curCost;
                        if (cgFlags & 1)
                        {
                            if (dU > 0) { curCost = -dU; curChange = 1; }
                            else if (cgFlags == 1 && iabs(lv[n]) == 1) curCost = 0x7fffffff;
                            else { curCost = dU; curChange = -1; }
                        }
                        else if (cgFlags == 0)
                        {
                            const uint32_t thisSignBit = t.a[at] >= 0 ? 0 : 1;
                            if (thisSignBit != signbit) curCost = 0x7fffffff;
                            else { curCost = -dU; curChange = 1; }
                        }
                        else { curCost = -dU; curChange = 1; }
                        if (curCost < minCostInc) { minCostInc = curCost; finalChange = curChange; minN = n; }
                        cgFlags >>= 1;
                    }
                    if (minN >= 0)
                    {
                        const int p4 = kDiag.s4[minN];
                        const int at = base + (p4 >> 2) * N + (p4 & 3);
                        int v = t.b[at];
                        if (v == 32767 || v == -32768) finalChange = -1;
                        if (!v) delta = 1;
                        else if (finalChange == -1 && iabs(v) == 1) delta = -1;
                        const int sigMask = t.a[at] < 0 ? -1 : 0;
                        v += (finalChange ^ sigMask) - sigMask;
                        t.b[at] = (int16_t)v;
                    }
                }
            }
        }
        numSig += group_sum(delta, LPT);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    XH_STAMP(3);
    // ---- levels out (16 contiguous per lane), dequant_normal -> a
    const int gL = (lane * 16) / (N * N);
    const bool okL = gL < count;
    {
        const int dqAdd = 1 << (qp.dqShift - 1);
#pragma unroll
        for (int half = 0; half < 2; half++)
        {
            const int e = lane * 16 + half * 8;
            int lv[8], dq[8];
            load4(t.b + e, lv); load4(t.b + e + 4, lv + 4);
#pragma unroll
            for (int i = 0; i < 8; i++) dq[i] = clip3i(-32768, 32767, (lv[i] * qp.dqScale + dqAdd) >> qp.dqShift);
            if (okL)
                *reinterpret_cast<uint4*>(levels + elemBase + (u0 + gL) * N * N + (e % (N * N))) = *reinterpret_cast<const uint4*>(t.b + e);
            store4(t.a + e, dq); store4(t.a + e + 4, dq + 4);
        }
    }
    // ---- the forward half is complete: numSig, zeroDist and the levels go out first (the host's entropy coder starts on them)
    unsigned long long zero = 0;
#pragma unroll
    for (int i = 0; i < 16; i++)
    {
        const int d0 = fv[i] - pv[i];
        zero += (unsigned)(d0 * d0);
    }
    zero = group_sum64(zero, LPT);
    x265hip_cujob_unit* un = units + unitBase + u0 + gL;
    const bool writer = okL && (lane & (LPT - 1)) == 0;
    if (writer)
    {
        un->numSig = (uint32_t)numSig;
        un->zeroDist = zero;
    }
    // the unit's data before its ready word: the release store waits for every store this wave has issued (s_waitcnt vmcnt(0) is per wave) — no
    // separate __threadfence_system(), whose L2 invalidate is of no use here (the slot is uncached host memory) and costs the other kernels their lines
    if (writer)
    {
        un->fwdTicks = (uint32_t)(wall_clock64() - t0);                      // 100 MHz ticks from the job's start to this unit's forward half
        __hip_atomic_store(&un->ready, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // no level left in any unit of the tile (the common case of chroma at everyday QPs): nobody asks for these units' inverse half — Quant::invtransformNxN
    // is only called on a unit with a coded block flag, and codedDist / codedEnergy / the reconstructed residual are defined for numSig != 0 only — so
    // the half is declared done at once (the host's end-of-scope wait and the next tile of this wave start ~3 us earlier)
    if (__ballot(okL && numSig != 0) == 0)
    {
        if (writer)
            __hip_atomic_store(&un->readyInv, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        return;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    // ---- inverse transform: a -> b -> a
    mfma_pass<N, true>(t.a, t.b, lane, bI, corrI, qp.s1i);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    mfma_pass<N, true>(t.b, t.a, lane, bI, corrI, qp.s2i);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    XH_STAMP(4);
    // ---- reconstructed residual out; distortion of the coded alternative
    unsigned long long coded = 0;
#pragma unroll
    for (int half = 0; half < 2; half++)
    {
        const int e = lane * 16 + half * 8;
        int r[8];
        load4(t.a + e, r); load4(t.a + e + 4, r + 4);
#pragma unroll
        for (int i = 0; i < 8; i++)
        {
            const int d1 = fv[8 * half + i] - clip3i(0, qp.maxVal, pv[8 * half + i] + r[i]);
            coded += (unsigned)(d1 * d1);
        }
        if (okL)
            *reinterpret_cast<uint4*>(resi + elemBase + (u0 + gL) * N * N + (e % (N * N))) = *reinterpret_cast<const uint4*>(t.a + e);
    }
    coded = group_sum64(coded, LPT);
    // ---- psy_cost_pp(source, reconstruction) (pixel.cpp:726-748): per 8x8 block |E(source) - E(reconstruction)|, E = sa8d_8x8 against zero - (sum >> 2).
    // The reconstruction goes to LDS as int16; the lanes then take 4x4 tiles, four consecutive lanes = the quadrants of one 8x8 block
    // (quad_sa8d_raw: the 8x8 Hadamard across the DPP quad), LPT tiles = one unit, exactly one pass
#pragma unroll
    for (int half = 0; half < 2; half++)
    {
        const int e = lane * 16 + half * 8;
        int r[8], rec[8];
        load4(t.a + e, r); load4(t.a + e + 4, r + 4);
#pragma unroll
        for (int i = 0; i < 8; i++) rec[i] = clip3i(0, qp.maxVal, pv[8 * half + i] + r[i]);
        store4(t.b + e, rec); store4(t.b + e + 4, rec + 4);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    int energy;
    {
        constexpr int B8 = N / 8;                                            // 8x8 blocks per row of a unit
        const int g2 = lane / LPT, tt = lane % LPT, b8 = tt >> 2, q = tt & 3;
        const int tx = (b8 % B8) * 8 + (q & 1) * 4, ty = (b8 / B8) * 8 + (q >> 1) * 4;
        const int u = u0 + (g2 < count ? g2 : 0), ux = u % perRow, uy = u / perRow;
        int ms[16], mr[16];
        tile_load(src + (uy * N + ty) * pw + ux * N + tx, (int64_t)pw, ms);
        tile_load(t.b + g2 * N * N + ty * N + tx, (int64_t)N, mr);
        int sumS = 0, sumR = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) { sumS += ms[i]; sumR += mr[i]; }
        hadamard4x4(ms);
        hadamard4x4(mr);
        const int rawS = quad_sa8d_raw(ms, lane), rawR = quad_sa8d_raw(mr, lane);
        const int es = ((rawS + 2) >> 2) - (quad_sum(sumS) >> 2), er = ((rawR + 2) >> 2) - (quad_sum(sumR) >> 2);
        energy = group_sum((lane & 3) == 0 ? iabs(es - er) : 0, LPT);
    }
    if (writer)
    {
        un->codedDist = coded;
        un->codedEnergy = (uint32_t)energy;
        XH_STAMP(5);
        if (stamps) { un->reserved[0] = stamp[0] | (stamp[1] << 16); un->reserved[1] = stamp[2] | (stamp[3] << 16); un->reserved[2] = stamp[4] | (stamp[5] << 16); }
        __hip_atomic_store(&un->readyInv, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
}

// ---- the team form: the four waves of a workgroup on ONE 32x32 unit ---------------------------------------------------------------------------------
// A 32x32 luma unit is what the submitting thread waits for first, and on one wave every stage of its chain is ~150 dependent VALU instructions at 16
// coefficients per lane (tile_chain above: 2.8 us from the pixels in LDS to the ready word).  Here the unit is spread over the workgroup: every
// elementwise stage works on FOUR coefficients per thread, a transform pass is one 16x16 output quadrant per wave (two v_mfma_i32_16x16x32_i8 with the same
// exact high-byte / low-byte split as dctcore.h), the stages meet at workgroup barriers that wait for LDS only.  Sign-bit hiding stays one lane per 4x4
// coefficient group (64 groups: one wave).  Same arithmetic, same LDS layouts, same results as tile_chain<P, 32>.
__device__ __forceinline__ void team_barrier()
{
    // s_barrier alone orders nothing for the compiler (the intrinsic is IntrNoMem: loads and stores may move across it): LDS-only fences either side — this
    // wave's LDS traffic is complete before the barrier and none of the next stage's is issued before it; stores to host memory are NOT waited for here
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// Stores to host memory are NEVER ordered by these barriers: a unit's blocks leave through wave 0 alone, whose release store of the ready word orders them.  (The
// first team version let every wave store its quarter and wait for it — s_waitcnt vmcnt(0) — in front of a barrier and wave 0's ready word: X265HIP_VERIFY at
// BASELINE configs[2] caught coefficients read by the host before another wave's quarter had landed, 5 runs of 7; a system-scope release fence per wave closes
// it and costs 1.5 us of the chain: profiles/r06_v1_team_fence.txt)

// thread tid's operand of the 16x16x32 form: wave w makes output rows 16 * (w >> 1) .., columns 16 * (w & 1) ..; lane l holds column C0 + (l & 15) of the
// coefficient matrix for the contraction indices 8 * (l >> 4) .. + 7
template <bool INV>
__device__ __forceinline__ void make_team_operand(int tid, TeamOperand& o)
{
    const int w = tid >> 6, lane = tid & 63;
    const int n = 16 * (w & 1) + (lane & 15), k0 = 8 * (lane >> 4);
    int sum = 0;
    unsigned long long bits = 0;
#pragma unroll
    for (int j = 0; j < 8; j++)
    {
        const int k = k0 + j;
        const int v = INV ? dct_coef<32>(k, n) : dct_coef<32>(n, k);
        sum += v;
        bits |= (unsigned long long)(v & 255) << (8 * j);
    }
    sum += __shfl_xor(sum, 16, kWave);
    sum += __shfl_xor(sum, 32, kWave);
    o.b = (long)bits;
    o.corr = 128 * sum;
    o.pad = 0;
}

// one pass of the 32x32 transform by the team: mfma_pass<32, INV> with the output quadrants dealt to the waves; in / out as there
template <bool INV>
__device__ __forceinline__ void team_pass(const int16_t* in, int16_t* out, int tid, const TeamOperand& op, int shift)
{
    const int w = tid >> 6, lane = tid & 63;
    const int i = 16 * (w >> 1) + (lane & 15), k0 = 8 * (lane >> 4);
    uint32_t x[4];
    if (!INV)
    {
        const uint4 v = *reinterpret_cast<const uint4*>(in + i * 32 + k0);
        x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
    }
    else
    {
        const int16_t* p = in + k0 * 32 + i;
#pragma unroll
        for (int m = 0; m < 4; m++)
            x[m] = (uint32_t)(uint16_t)p[(2 * m) * 32] | ((uint32_t)(uint16_t)p[(2 * m + 1) * 32] << 16);
    }
    const uint32_t lo0 = __builtin_amdgcn_perm(x[1], x[0], 0x06040200u) ^ 0x80808080u, lo1 = __builtin_amdgcn_perm(x[3], x[2], 0x06040200u) ^ 0x80808080u;
    const uint32_t hi0 = __builtin_amdgcn_perm(x[1], x[0], 0x07050301u), hi1 = __builtin_amdgcn_perm(x[3], x[2], 0x07050301u);
    const long ahi = (long)((unsigned long long)hi0 | ((unsigned long long)hi1 << 32)), alo = (long)((unsigned long long)lo0 | ((unsigned long long)lo1 << 32));
    v4i acc = { 0, 0, 0, 0 };
    acc = __builtin_amdgcn_mfma_i32_16x16x32_i8(ahi, op.b, acc, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; q++)
        acc[q] <<= 8;
    acc = __builtin_amdgcn_mfma_i32_16x16x32_i8(alo, op.b, acc, 0, 0, 0);
    // D[row][col]: col = lane & 15, row = 4 * (lane >> 4) + q
    const int n = 16 * (w & 1) + (lane & 15), row0 = 16 * (w >> 1) + 4 * (lane >> 4);
    const int add = (1 << (shift - 1)) + op.corr;
    int v[4];
#pragma unroll
    for (int e = 0; e < 4; e++)
    {
        const int t = (acc[e] + add) >> shift;
        v[e] = INV ? clip3i(-32768, 32767, t) : t;
    }
    if (!INV)
        store4(out + n * 32 + row0, v);
    else
    {
#pragma unroll
        for (int e = 0; e < 4; e++)
            out[(row0 + e) * 32 + n] = (int16_t)v[e];
    }
}

// the sum of v over the team, for every thread: 16-lane rows by DPP, the sixteen row totals through LDS (the caller's barrier follows the store)
__device__ __forceinline__ void team_part(uint32_t* part, int tid, uint32_t v)
{
    const uint32_t r = (uint32_t)row_allsum((int)v);
    if ((tid & 15) == 0) part[tid >> 4] = r;
}
__device__ __forceinline__ unsigned long long team_total(const uint32_t* part)
{
    unsigned long long t = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) t += part[i];
    return t;
}

// the inverse half of a team unit: the dequantised coefficients are in t.a, the unit's source and prediction samples of this thread in fv / pv
template <typename P>
__device__ __forceinline__ void team_inverse32(TileLds& t, TeamLds& tm, const P* src, int pw, int ux, int uy, const PlaneParams qp, const int fv[4], const int pv[4],
                                               x265hip_cujob_unit* un, int16_t* resi, uint32_t seq, uint64_t t0, bool stamps, uint32_t stamp[6])
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, e = tid * 4;
    const TeamOperand& oI = tm.op[1][tid];
    // ---- inverse transform: a -> b -> a
    team_pass<true>(t.a, t.b, tid, oI, qp.s1i);
    team_barrier();
    team_pass<true>(t.b, t.a, tid, oI, qp.s2i);
    team_barrier();
    XH_STAMP(4);
    // ---- distortion of the coded alternative; the reconstruction to b as int16 for the energies (the reconstructed residual stays in a)
    {
        int r[4], rec[4];
        uint32_t codedP = 0;
        load4(t.a + e, r);
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            rec[i] = clip3i(0, qp.maxVal, pv[i] + r[i]);
            const int d1 = fv[i] - rec[i];
            codedP += (uint32_t)(d1 * d1);
        }
        store4(t.b + e, rec);
        team_part(tm.part, tid, codedP);
    }
    team_barrier();
    // ---- wave 0 carries the reconstructed residual out; psy_cost_pp(source, reconstruction) meanwhile: per 8x8 block |E(source) - E(reconstruction)| (tile_chain) —
    // wave 2 measures the reconstruction, wave 3 the source: a lane = one 4x4 tile, four consecutive lanes the quadrants of one 8x8 block
    if (wv == 0)
    {
        const uint4 r0 = *reinterpret_cast<const uint4*>(t.a + lane * 16), r1 = *reinterpret_cast<const uint4*>(t.a + lane * 16 + 8);
        *reinterpret_cast<uint4*>(resi + lane * 16) = r0;
        *reinterpret_cast<uint4*>(resi + lane * 16 + 8) = r1;
    }
    else if (wv >= 2)
    {
        const int b8 = lane >> 2, q = lane & 3;
        const int tx = (b8 & 3) * 8 + (q & 1) * 4, ty = (b8 >> 2) * 8 + (q >> 1) * 4;
        int m[16];
        if (wv == 2) tile_load(t.b + ty * 32 + tx, (int64_t)32, m);
        else tile_load(src + (uy * 32 + ty) * pw + ux * 32 + tx, (int64_t)pw, m);
        int sum = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) sum += m[i];
        hadamard4x4(m);
        const int raw = quad_sa8d_raw(m, lane);
        const int en = ((raw + 2) >> 2) - (quad_sum(sum) >> 2);
        if (q == 0) tm.energy[wv == 2 ? 1 : 0][b8] = en;
    }
    team_barrier();
    // (wave 0's release store orders its own stores of the residual in front of the word)
    if (tid == 0)
    {
        int energy = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) energy += iabs(tm.energy[0][i] - tm.energy[1][i]);
        un->codedDist = team_total(tm.part);
        un->codedEnergy = (uint32_t)energy;
        XH_STAMP(5);
        if (stamps) { un->reserved[0] = stamp[0] | (stamp[1] << 16); un->reserved[1] = stamp[2] | (stamp[3] << 16); un->reserved[2] = stamp[4] | (stamp[5] << 16); }
        __hip_atomic_store(&un->readyInv, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    team_barrier();                              // (the tile and the team's words are free for the next unit)
}

// unit `u` (raster order; `pw` elements per row of the plane in LDS) of a plane whose transform size is 32: tile_chain<P, 32>'s work for one unit
template <typename P>
__device__ __forceinline__ void team_chain32(TileLds& t, TeamLds& tm, const P* src, const P* prd, int pw, int u, const PlaneParams qp, bool signHide,
                                             x265hip_cujob_unit* un, int16_t* levels, int16_t* resi, uint32_t seq, uint64_t t0, bool stamps, int coef)
{
    uint32_t stamp[6] = { 0, 0, 0, 0, 0, 0 };
    XH_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int perRow = pw / 32, ux = u % perRow, uy = u / perRow;
    const TeamOperand& oF = tm.op[0][tid];
    const TeamOperand& oI = tm.op[1][tid];
    const bool srcOnly = (coef & 3) == 2;
    // ---- residual: four consecutive samples of a row per thread, kept in registers for the distortions
    const int e = tid * 4;
    int fv[4], pv[4];
    {
        const int off = (uy * 32 + (e >> 5)) * pw + ux * 32 + (e & 31);
        load4(src + off, fv);
        load4(prd + off, pv);
        int r[4];
#pragma unroll
        for (int i = 0; i < 4; i++) r[i] = srcOnly ? fv[i] : fv[i] - pv[i];
        store4(t.a + e, r);
    }
    uint32_t zeroP = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { const int d0 = fv[i] - pv[i]; zeroP += (uint32_t)(d0 * d0); }
    team_part(tm.part2, tid, zeroP);             // (read after the barriers below; nothing else writes part2 before the forward half is published)
    team_barrier();
    // ---- forward transform: a -> b -> a
    team_pass<false>(t.a, t.b, tid, oF, qp.s1f);
    team_barrier();
    team_pass<false>(t.b, t.a, tid, oF, qp.s2f);
    team_barrier();
    XH_STAMP(1);
    if (coef)
    {
        // ---- coefficient mode (tile_chain): the transform coefficients go out where the levels would (residual part) or where the reconstructed residual would
        // (source part)
        int16_t* dstC = srcOnly ? resi : levels;
        // (one wave carries the block out and publishes behind it, see the forward half below)
        if (wv == 0)
        {
            const uint4 c0 = *reinterpret_cast<const uint4*>(t.a + lane * 16), c1 = *reinterpret_cast<const uint4*>(t.a + lane * 16 + 8);
            *reinterpret_cast<uint4*>(dstC + lane * 16) = c0;
            *reinterpret_cast<uint4*>(dstC + lane * 16 + 8) = c1;
            if (lane == 0)
            {
                if (srcOnly)
                    __hip_atomic_store(&un->readyInv, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                else
                {
                    un->numSig = 0;
                    un->zeroDist = team_total(tm.part2);
                    un->fwdTicks = (uint32_t)(wall_clock64() - t0);
                    XH_STAMP(5);
                    if (stamps) { un->reserved[0] = stamp[0] | (stamp[1] << 16); un->reserved[1] = 0; un->reserved[2] = stamp[5] << 16; }
                    if (!(coef & 4))
                        __hip_atomic_store(&un->readyInv, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(&un->ready, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
        team_barrier();                          // (part2 and the tile are free again)
        return;
    }
    // ---- quant (quant_c): levels -> b, deltaU -> c; the coefficients stay in a
    {
        int cf[4], lv[4], du[4];
        uint32_t cnt = 0;
        const int qBits8 = qp.qBits - 8;
        load4(t.a + e, cf);
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            const int tmp = iabs(cf[i]) * qp.quantScale;
            const int l = (tmp + qp.add) >> qp.qBits;
            du[i] = (tmp - (l << qp.qBits)) >> qBits8;
            cnt += l != 0;
            lv[i] = clip3i(-32768, 32767, cf[i] < 0 ? -l : l);
        }
        store4(t.b + e, lv);
        store4(t.c + e, du);
        team_part(tm.part, tid, cnt);
    }
    team_barrier();
    XH_STAMP(2);
    int numSig = (int)team_total(tm.part);
    // ---- sign-bit hiding (signBitHidingHDQ): wave 0, lane = coefficient group (scan order); tile_chain's code for N = 32
    if (signHide && numSig >= 2)
    {
        if (wv == 0)
        {
            constexpr int N = 32, CGW = 8;
            const int cg = lane;
            const int cgPos = kDiag.s8[cg];
            const int cgx = cgPos % CGW, cgy = cgPos / CGW;
            const int base = (cgy * 4) * N + cgx * 4;
            int lv[16];
            uint32_t flags = 0;
#pragma unroll
            for (int n = 0; n < 16; n++)
            {
                const int p4 = kDiag.s4[n];
                lv[n] = t.b[base + (p4 >> 2) * N + (p4 & 3)];
                flags |= (uint32_t)(lv[n] != 0) << (15 - n);
            }
            const unsigned long long nz = __ballot(flags != 0);
            const int cgLast = nz ? 63 - __builtin_clzll(nz) : -1;
            int delta = 0;
            if (flags && cg <= cgLast)
            {
                const int firstNZ = 15 ^ (31 - __builtin_clz(flags));
                const int lastNZ = 15 ^ __builtin_ctz(flags);
                if (lastNZ - firstNZ >= 4)
                {
                    const uint32_t signbit = lv[firstNZ] > 0 ? 0 : 1;
                    int absSum = 0;
#pragma unroll
                    for (int n = 0; n < 16; n++)
                        if (n >= firstNZ && n <= lastNZ) absSum += lv[n];
                    if (signbit != ((uint32_t)absSum & 1))
                    {
                        int minCostInc = 0x7fffffff, minN = -1, finalChange = 0, curChange = 0;
                        const int start = cg == cgLast ? lastNZ : 15;
                        uint32_t cgFlags = flags >> (15 - start);
#pragma unroll
                        for (int n = 15; n >= 0; n--)
                        {
                            if (n > start) continue;
                            const int p4 = kDiag.s4[n];
                            const int at = base + (p4 >> 2) * N + (p4 & 3);
                            const int dU = t.c[at];
                            int curCost;
                            if (cgFlags & 1)
                            {
                                if (dU > 0) { curCost = -dU; curChange = 1; }
                                else if (cgFlags == 1 && iabs(lv[n]) == 1) curCost = 0x7fffffff;
                                else { curCost = dU; curChange = -1; }
                            }
                            else if (cgFlags == 0)
                            {
                                const uint32_t thisSignBit = t.a[at] >= 0 ? 0 : 1;
                                if (thisSignBit != signbit) curCost = 0x7fffffff;
                                else { curCost = -dU; curChange = 1; }
                            }
                            else { curCost = -dU; curChange = 1; }
                            if (curCost < minCostInc) { minCostInc = curCost; finalChange = curChange; minN = n; }
                            cgFlags >>= 1;
                        }
                        if (minN >= 0)
                        {
                            const int p4 = kDiag.s4[minN];
                            const int at = base + (p4 >> 2) * N + (p4 & 3);
                            int v = t.b[at];
                            if (v == 32767 || v == -32768) finalChange = -1;
                            if (!v) delta = 1;
                            else if (finalChange == -1 && iabs(v) == 1) delta = -1;
                            const int sigMask = t.a[at] < 0 ? -1 : 0;
                            v += (finalChange ^ sigMask) - sigMask;
                            t.b[at] = (int16_t)v;
                        }
                    }
                }
            }
            const int d = wave_sum(delta);
            if (lane == 0) tm.numSig = numSig + d;
        }
        team_barrier();
        numSig = tm.numSig;
    }
    XH_STAMP(3);
    // ---- dequant_normal -> a (every thread its four); wave 0 takes the unit's levels into registers: they leave through ONE wave, whose release store of the ready
    // word then orders them (stores of another wave are not ordered against it by anything cheaper than a system-scope release fence per wave: profiles/r06_v1_team_fence.txt)
    uint4 keep0 = { 0, 0, 0, 0 }, keep1 = { 0, 0, 0, 0 };
    {
        int lv[4], dq[4];
        const int dqAdd = 1 << (qp.dqShift - 1);
        load4(t.b + e, lv);
#pragma unroll
        for (int i = 0; i < 4; i++) dq[i] = clip3i(-32768, 32767, (lv[i] * qp.dqScale + dqAdd) >> qp.dqShift);
        store4(t.a + e, dq);
        if (wv == 0)
        {
            keep0 = *reinterpret_cast<const uint4*>(t.b + lane * 16);
            keep1 = *reinterpret_cast<const uint4*>(t.b + lane * 16 + 8);
        }
    }
    team_barrier();
    // ---- the forward half is complete: levels, numSig, zeroDist, then the ready word (the other waves are already in the inverse passes)
    if (wv == 0)
    {
        *reinterpret_cast<uint4*>(levels + lane * 16) = keep0;
        *reinterpret_cast<uint4*>(levels + lane * 16 + 8) = keep1;
        if (lane == 0)
        {
            un->numSig = (uint32_t)numSig;
            un->zeroDist = team_total(tm.part2);
            un->fwdTicks = (uint32_t)(wall_clock64() - t0);
            __hip_atomic_store(&un->ready, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            // nobody asks for the inverse half of a unit without a level (tile_chain)
            if (numSig == 0)
                __hip_atomic_store(&un->readyInv, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (numSig == 0)
    {
        team_barrier();
        return;
    }
    team_inverse32<P>(t, tm, src, pw, ux, uy, qp, fv, pv, un, resi, seq, t0, stamps, stamp);
}

// ---- a 16x16 unit on ONE wave, four coefficients per lane (the chroma units of a 32x32 CU: one unit per plane) ----------------------------------------------
// tile_chain<P, 16> gives a wave a 32x32 MFMA tile = four 16x16 units; a 32x32 CU has ONE per chroma plane, and the wave then runs the whole tile's instruction
// stream for a quarter of its data.  Here the unit is the wave's only work: a pass is two v_mfma_i32_16x16x32_i8 (the contraction index padded from 16 to 32
// with zeros), every elementwise stage has four coefficients per lane, sign-bit hiding one lane per 4x4 group (16 lanes).  Same arithmetic and results.
template <bool INV>
__device__ __forceinline__ void make_solo16_operand(int lane, TeamOperand& o)
{
    const int n = lane & 15, k0 = 8 * (lane >> 4);
    int sum = 0;
    unsigned long long bits = 0;
#pragma unroll
    for (int j = 0; j < 8; j++)
    {
        const int k = k0 + j;
        const int v = k < 16 ? (INV ? dct_coef<16>(k, n) : dct_coef<16>(n, k)) : 0;
        sum += v;
        bits |= (unsigned long long)(v & 255) << (8 * j);
    }
    sum += __shfl_xor(sum, 16, kWave);
    sum += __shfl_xor(sum, 32, kWave);
    o.b = (long)bits;
    o.corr = 128 * sum;
    o.pad = 0;
}
template <bool INV>
__device__ __forceinline__ void solo_pass16(const int16_t* in, int16_t* out, int lane, const TeamOperand& op, int shift)
{
    const int i = lane & 15, k0 = 8 * (lane >> 4);
    uint32_t x[4] = { 0, 0, 0, 0 };
    if (k0 < 16)
    {
        if (!INV)
        {
            const uint4 v = *reinterpret_cast<const uint4*>(in + i * 16 + k0);
            x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
        }
        else
        {
            const int16_t* p = in + k0 * 16 + i;
#pragma unroll
            for (int m = 0; m < 4; m++)
                x[m] = (uint32_t)(uint16_t)p[(2 * m) * 16] | ((uint32_t)(uint16_t)p[(2 * m + 1) * 16] << 16);
        }
    }
    const uint32_t lo0 = __builtin_amdgcn_perm(x[1], x[0], 0x06040200u) ^ 0x80808080u, lo1 = __builtin_amdgcn_perm(x[3], x[2], 0x06040200u) ^ 0x80808080u;
    const uint32_t hi0 = __builtin_amdgcn_perm(x[1], x[0], 0x07050301u), hi1 = __builtin_amdgcn_perm(x[3], x[2], 0x07050301u);
    const long ahi = (long)((unsigned long long)hi0 | ((unsigned long long)hi1 << 32)), alo = (long)((unsigned long long)lo0 | ((unsigned long long)lo1 << 32));
    v4i acc = { 0, 0, 0, 0 };
    acc = __builtin_amdgcn_mfma_i32_16x16x32_i8(ahi, op.b, acc, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; q++)
        acc[q] <<= 8;
    acc = __builtin_amdgcn_mfma_i32_16x16x32_i8(alo, op.b, acc, 0, 0, 0);
    const int n = lane & 15, row0 = 4 * (lane >> 4);
    const int add = (1 << (shift - 1)) + op.corr;
    int v[4];
#pragma unroll
    for (int e = 0; e < 4; e++)
    {
        const int t = (acc[e] + add) >> shift;
        v[e] = INV ? clip3i(-32768, 32767, t) : t;
    }
    if (!INV)
        store4(out + n * 16 + row0, v);
    else
    {
#pragma unroll
        for (int e = 0; e < 4; e++)
            out[(row0 + e) * 16 + n] = (int16_t)v[e];
    }
}
// the sum of v over the wave (v <= 2^26 per lane): rows of 16 lanes by DPP, the four row totals through the scalar unit
__device__ __forceinline__ unsigned long long solo_total(uint32_t v)
{
    const uint32_t r = (uint32_t)row_allsum((int)v);
    return (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)r, 0) + (uint32_t)__builtin_amdgcn_readlane((int)r, 16) +
           (uint32_t)__builtin_amdgcn_readlane((int)r, 32) + (uint32_t)__builtin_amdgcn_readlane((int)r, 48);
}

template <typename P>
__device__ __forceinline__ void solo_chain16(TileLds& t, const TeamOperand (*op)[64], const P* src, const P* prd, const PlaneParams qp, bool signHide,
                                             x265hip_cujob_unit* un, int16_t* levels, int16_t* resi, uint32_t seq, uint64_t t0, bool stamps, int coef)
{
    uint32_t stamp[6] = { 0, 0, 0, 0, 0, 0 };
    XH_STAMP(0);
    const int lane = threadIdx.x & 63, e = lane * 4;
    const TeamOperand& oF = op[0][lane];
    const TeamOperand& oI = op[1][lane];
    int fv[4], pv[4];
    {
        load4(src + e, fv);
        load4(prd + e, pv);
        int r[4];
#pragma unroll
        for (int i = 0; i < 4; i++) r[i] = fv[i] - pv[i];
        store4(t.a + e, r);
    }
    uint32_t zeroP = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { const int d0 = fv[i] - pv[i]; zeroP += (uint32_t)(d0 * d0); }
    const unsigned long long zero = solo_total(zeroP);
    // ---- forward transform: a -> b -> a
    solo_pass16<false>(t.a, t.b, lane, oF, qp.s1f);
    solo_pass16<false>(t.b, t.a, lane, oF, qp.s2f);
    XH_STAMP(1);
    if (coef)
    {
        // coefficient mode (chroma: the residual's coefficients only)
        *reinterpret_cast<uint2*>(levels + e) = *reinterpret_cast<const uint2*>(t.a + e);
        if (lane == 0)
        {
            un->numSig = 0;
            un->zeroDist = zero;
            un->fwdTicks = (uint32_t)(wall_clock64() - t0);
            XH_STAMP(5);
            if (stamps) { un->reserved[0] = stamp[0] | (stamp[1] << 16); un->reserved[1] = 0; un->reserved[2] = stamp[5] << 16; }
            __hip_atomic_store(&un->readyInv, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&un->ready, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    // ---- quant: levels -> b, deltaU -> c; the coefficients stay in a
    uint32_t cnt = 0;
    {
        int cf[4], lv[4], du[4];
        const int qBits8 = qp.qBits - 8;
        load4(t.a + e, cf);
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            const int tmp = iabs(cf[i]) * qp.quantScale;
            const int l = (tmp + qp.add) >> qp.qBits;
            du[i] = (tmp - (l << qp.qBits)) >> qBits8;
            cnt += l != 0;
            lv[i] = clip3i(-32768, 32767, cf[i] < 0 ? -l : l);
        }
        store4(t.b + e, lv);
        store4(t.c + e, du);
    }
    int numSig = wave_sum((int)cnt);
    XH_STAMP(2);
    // ---- sign-bit hiding: lane = coefficient group (scan order), 16 groups; tile_chain's code for N = 16
    if (signHide && numSig >= 2)
    {
        constexpr int N = 16, CGW = 4;
        const bool mine = lane < 16;
        const int cg = lane & 15;
        const int cgPos = kDiag.s4[cg];
        const int cgx = cgPos % CGW, cgy = cgPos / CGW;
        const int base = (cgy * 4) * N + cgx * 4;
        int lv[16];
        uint32_t flags = 0;
#pragma unroll
        for (int n = 0; n < 16; n++)
        {
            const int p4 = kDiag.s4[n];
            lv[n] = t.b[base + (p4 >> 2) * N + (p4 & 3)];
            flags |= (uint32_t)(lv[n] != 0) << (15 - n);
        }
        if (!mine) flags = 0;
        const unsigned long long nz = __ballot(flags != 0);
        const int cgLast = nz ? 63 - __builtin_clzll(nz) : -1;
        int delta = 0;
        if (flags && cg <= cgLast)
        {
            const int firstNZ = 15 ^ (31 - __builtin_clz(flags));
            const int lastNZ = 15 ^ __builtin_ctz(flags);
            if (lastNZ - firstNZ >= 4)
            {
                const uint32_t signbit = lv[firstNZ] > 0 ? 0 : 1;
                int absSum = 0;
#pragma unroll
                for (int n = 0; n < 16; n++)
                    if (n >= firstNZ && n <= lastNZ) absSum += lv[n];
                if (signbit != ((uint32_t)absSum & 1))
                {
                    int minCostInc = 0x7fffffff, minN = -1, finalChange = 0, curChange = 0;
                    const int start = cg == cgLast ? lastNZ : 15;
                    uint32_t cgFlags = flags >> (15 - start);
#pragma unroll
                    for (int n = 15; n >= 0; n--)
                    {
                        if (n > start) continue;
                        const int p4 = kDiag.s4[n];
                        const int at = base + (p4 >> 2) * N + (p4 & 3);
                        const int dU = t.c[at];
                        int curCost;
                        if (cgFlags & 1)
                        {
                            if (dU > 0) { curCost = -dU; curChange = 1; }
                            else if (cgFlags == 1 && iabs(lv[n]) == 1) curCost = 0x7fffffff;
                            else { curCost = dU; curChange = -1; }
                        }
                        else if (cgFlags == 0)
                        {
                            const uint32_t thisSignBit = t.a[at] >= 0 ? 0 : 1;
                            if (thisSignBit != signbit) curCost = 0x7fffffff;
                            else { curCost = -dU; curChange = 1; }
                        }
                        else { curCost = -dU; curChange = 1; }
                        if (curCost < minCostInc) { minCostInc = curCost; finalChange = curChange; minN = n; }
                        cgFlags >>= 1;
                    }
                    if (minN >= 0)
                    {
                        const int p4 = kDiag.s4[minN];
                        const int at = base + (p4 >> 2) * N + (p4 & 3);
                        int v = t.b[at];
                        if (v == 32767 || v == -32768) finalChange = -1;
                        if (!v) delta = 1;
                        else if (finalChange == -1 && iabs(v) == 1) delta = -1;
                        const int sigMask = t.a[at] < 0 ? -1 : 0;
                        v += (finalChange ^ sigMask) - sigMask;
                        t.b[at] = (int16_t)v;
                    }
                }
            }
        }
        numSig += wave_sum(delta);
    }
    XH_STAMP(3);
    // ---- levels out, dequant_normal -> a; the forward half is published by this wave's release store behind its own stores
    {
        int lv[4], dq[4];
        const int dqAdd = 1 << (qp.dqShift - 1);
        load4(t.b + e, lv);
#pragma unroll
        for (int i = 0; i < 4; i++) dq[i] = clip3i(-32768, 32767, (lv[i] * qp.dqScale + dqAdd) >> qp.dqShift);
        *reinterpret_cast<uint2*>(levels + e) = *reinterpret_cast<const uint2*>(t.b + e);
        store4(t.a + e, dq);
    }
    if (lane == 0)
    {
        un->numSig = (uint32_t)numSig;
        un->zeroDist = zero;
        un->fwdTicks = (uint32_t)(wall_clock64() - t0);
        __hip_atomic_store(&un->ready, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (numSig == 0)
            __hip_atomic_store(&un->readyInv, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (numSig == 0)
        return;
    // ---- inverse transform: a -> b -> a
    solo_pass16<true>(t.a, t.b, lane, oI, qp.s1i);
    solo_pass16<true>(t.b, t.a, lane, oI, qp.s2i);
    XH_STAMP(4);
    uint32_t codedP = 0;
    {
        int r[4], rec[4];
        load4(t.a + e, r);
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            rec[i] = clip3i(0, qp.maxVal, pv[i] + r[i]);
            const int d1 = fv[i] - rec[i];
            codedP += (uint32_t)(d1 * d1);
        }
        *reinterpret_cast<uint2*>(resi + e) = *reinterpret_cast<const uint2*>(t.a + e);
        store4(t.b + e, rec);
    }
    const unsigned long long coded = solo_total(codedP);
    // ---- psy energies of the unit's four 8x8 blocks: lanes 0..15 the reconstruction's 4x4 tiles, lanes 16..31 the source's (a quad = one 8x8 block)
    int en = 0;
    if (lane < 32)
    {
        const int tt = lane & 15, b8 = tt >> 2, q = tt & 3;
        const int tx = (b8 & 1) * 8 + (q & 1) * 4, ty = (b8 >> 1) * 8 + (q >> 1) * 4;
        int m[16];
        if (lane < 16) tile_load(t.b + ty * 16 + tx, (int64_t)16, m);
        else tile_load(src + ty * 16 + tx, (int64_t)16, m);
        int sum = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) sum += m[i];
        hadamard4x4(m);
        const int raw = quad_sa8d_raw(m, lane);
        en = ((raw + 2) >> 2) - (quad_sum(sum) >> 2);
    }
    int energy = 0;
#pragma unroll
    for (int b = 0; b < 4; b++)
        energy += iabs(__builtin_amdgcn_readlane(en, 16 + 4 * b) - __builtin_amdgcn_readlane(en, 4 * b));
    if (lane == 0)
    {
        un->codedDist = coded;
        un->codedEnergy = (uint32_t)energy;
        XH_STAMP(5);
        if (stamps) { un->reserved[0] = stamp[0] | (stamp[1] << 16); un->reserved[1] = stamp[2] | (stamp[3] << 16); un->reserved[2] = stamp[4] | (stamp[5] << 16); }
        __hip_atomic_store(&un->readyInv, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// An INVERSE job (x265hip_cujob::coefMode == X265HIP_CUJOB_INVERSE): one 32x32 luma unit whose levels the host has made (Quant::rdoQuant) — dequant_normal ->
// cu[].idct -> reconstructed residual, sse_pp and psy energy of the reconstruction, i.e. Quant::invtransformNxN (quant.cpp:543-603) and the two measurements
// Search::estimateResidualQT takes behind it (search.cpp:3290-3300).  The levels follow the source and prediction blocks in the pixel block.
template <typename P>
__device__ __forceinline__ void team_inverse_job(SlotOut* s, JobLds& L, uint32_t seq, uint64_t t0)
{
    const x265hip_cujob& j = L.job;
    TileLds& t = L.tile[0];
    TeamLds& tm = L.team;
    const P* src = reinterpret_cast<const P*>(L.pix);
    const P* prd = src + 1024;
    const int16_t* lvIn = reinterpret_cast<const int16_t*>(prd + 1024);
    const PlaneParams qp = plane_params(j, 0, 5);
    const bool stamps = j.reserved != 0;
    uint32_t stamp[6] = { 0, 0, 0, 0, 0, 0 };
    XH_STAMP(0);
    const int tid = threadIdx.x, e = tid * 4;
    x265hip_cujob_unit* un = s->units;
    int fv[4], pv[4], lv[4], dq[4];
    load4(src + e, fv);
    load4(prd + e, pv);
    load4(lvIn + e, lv);
    uint32_t zeroP = 0, cnt = 0;
    const int dqAdd = 1 << (qp.dqShift - 1);
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        const int d0 = fv[i] - pv[i];
        zeroP += (uint32_t)(d0 * d0);
        cnt += lv[i] != 0;
        dq[i] = clip3i(-32768, 32767, (lv[i] * qp.dqScale + dqAdd) >> qp.dqShift);
    }
    store4(t.a + e, dq);
    team_part(tm.part2, tid, zeroP);
    team_part(tm.part, tid, cnt);
    team_barrier();
    const int numSig = (int)team_total(tm.part);
    if (tid == 0)
    {
        un->numSig = (uint32_t)numSig;
        un->zeroDist = team_total(tm.part2);
    }
    if (numSig)
        team_inverse32<P>(t, tm, src, 32, 0, 0, qp, fv, pv, un, s->resi, seq, t0, stamps, stamp);
    else
        team_barrier();
    if (tid == 0)
    {
        un->fwdTicks = (uint32_t)(wall_clock64() - t0);
        if (!numSig) __hip_atomic_store(&un->readyInv, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&un->ready, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// the six transform operands, once per kernel: wave w builds size 8 << w (waves 0..2)
__device__ __forceinline__ void build_operands(JobLds& L)
{
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    v4i b; int corr;
#define XH_BOP(N, K, INV) do { make_b_operand<N, INV>(lane, b, corr); BOperand& o = L.bop[K][INV ? 1 : 0][lane]; o.b[0] = b[0]; o.b[1] = b[1]; o.b[2] = b[2]; o.b[3] = b[3]; o.corr = corr; } while (0)
    if (wv == 0) { XH_BOP(8, 0, false); XH_BOP(8, 0, true); }
    else if (wv == 1) { XH_BOP(16, 1, false); XH_BOP(16, 1, true); }
    else if (wv == 2) { XH_BOP(32, 2, false); XH_BOP(32, 2, true); }
#undef XH_BOP
    make_team_operand<false>(threadIdx.x, L.team.op[0][threadIdx.x]);
    make_team_operand<true>(threadIdx.x, L.team.op[1][threadIdx.x]);
    if (wv == 3) { make_solo16_operand<false>(lane, L.team.op16[0][lane]); make_solo16_operand<true>(lane, L.team.op16[1][lane]); }
    __syncthreads();
}

// A job is served by a PAIR of workgroups (two per slot, each alone on a compute unit): role 0 takes the luma units — the 32x32 ones as a team, one after the
// other, because the submitting thread asks for them one after the other and waits for the first — role 1 the chroma units, a 32x32 tile per wave as before.
// Every unit is written by exactly one workgroup and carries its own ready words, so the two never meet.
template <typename P>
__device__ __forceinline__ void run_tiles(SlotOut* s, JobLds& L, uint32_t seq, uint64_t t0, int role, bool team)
{
    const int wv = threadIdx.x >> 6;
    const x265hip_cujob& j = L.job;
    const int N = 1 << j.log2CUSize, NC = N >> 1;
    const int lumaElems = N * N, planeElems = j.chroma ? lumaElems + lumaElems / 2 : lumaElems;
    const P* src = reinterpret_cast<const P*>(L.pix);
    const P* prd = src + planeElems;
    if (j.coefMode == X265HIP_CUJOB_INVERSE)
    {
        if (role == 0) team_inverse_job<P>(s, L, seq, t0);
        return;
    }
    int sHi, sLo;
    const int levels = x265hipi_cujob_levels(&j, &sHi, &sLo);
    // ---- tiles, largest size first; wave w takes the role's tiles w, w + 4, ... (team units are taken by all four waves together)
    int tile = 0;
    for (int lv = 0; lv < levels; lv++)
    {
        const int sz = sHi - lv;
        const int perRow = 1 << ((int)j.log2CUSize - sz), nUnits = perRow * perRow;
        for (int plane = role ? 1 : 0; plane < (role ? (j.chroma ? 3 : 1) : 1); plane++)
        {
            const int log2n = plane ? sz - 1 : sz;                          // 5, 4 (luma) or 4, 3 (chroma)
            const int G = 1 << (2 * (5 - log2n));
            const int tiles = (nUnits + G - 1) / G;
            const P* ps = plane == 0 ? src : plane == 1 ? src + lumaElems : src + lumaElems + lumaElems / 4;
            const P* pp = plane == 0 ? prd : plane == 1 ? prd + lumaElems : prd + lumaElems + lumaElems / 4;
            const int pw = plane ? NC : N;
            const PlaneParams qp = plane_params(j, plane, log2n);
            // coefficient mode: a luma tile whose source block is wanted as well is TWO work items (residual part, source part)
            const int parts = j.coefMode && j.sourceDct && plane == 0 ? 2 : 1;
            const int unitBase = x265hipi_cujob_unit_index(&j, sHi, sz, plane, 0, 0);
            const int elemBase = x265hipi_cujob_elem_offset(&j, sHi, sz, plane, 0, 0);
            if (plane == 0 && log2n == 5 && team)
            {
                // the team form: unit by unit (G == 1), both parts of a unit back to back
                for (int u = 0; u < nUnits; u++)
                    for (int part = 0; part < parts; part++)
                    {
                        const int coef = !j.coefMode ? 0 : parts == 1 ? 1 : part == 0 ? 1 | 4 : 2 | 4;
                        team_chain32<P>(L.tile[0], L.team, ps, pp, pw, u, qp, j.signHide != 0, s->units + unitBase + u, s->levels + elemBase + u * 1024,
                                        s->resi + elemBase + u * 1024, seq, t0, j.reserved != 0, coef);
                    }
                continue;
            }
            for (int kk = 0; kk < tiles * parts; kk++, tile++)
            {
                if ((tile & 3) != wv) continue;
                const int k = kk / parts;
                const int coef = !j.coefMode ? 0 : parts == 1 ? 1 : (kk % parts) == 0 ? 1 | 4 : 2 | 4;
                const int u0 = k * G, count = nUnits - u0 < G ? nUnits - u0 : G;
                if (log2n == 5) tile_chain<P, 32>(L.tile[wv], L.bop[2], ps, pp, pw, u0, count, qp, j.signHide != 0, s->units, unitBase, s->levels, s->resi, elemBase, seq, t0, j.reserved != 0, coef);
                else if (log2n == 4 && nUnits == 1 && plane && team)
                    // (the chroma units of a 32x32 CU: one 16x16 unit per plane, the wave's only work)
                    solo_chain16<P>(L.tile[wv], L.team.op16, ps, pp, qp, j.signHide != 0, s->units + unitBase, s->levels + elemBase, s->resi + elemBase, seq, t0, j.reserved != 0, coef);
                else if (log2n == 4) tile_chain<P, 16>(L.tile[wv], L.bop[1], ps, pp, pw, u0, count, qp, j.signHide != 0, s->units, unitBase, s->levels, s->resi, elemBase, seq, t0, j.reserved != 0, coef);
                else tile_chain<P, 8>(L.tile[wv], L.bop[0], ps, pp, pw, u0, count, qp, j.signHide != 0, s->units, unitBase, s->levels, s->resi, elemBase, seq, t0, j.reserved != 0, coef);
            }
        }
    }
}

// ---- SAO statistics of one CTU (x265hip_saojob; reference sao.cpp:735-917 with saoCuStatsBO / E0..E3_c :1762-1925).  The reference walks each class's
// rectangle with running sign buffers; an edge category is a function of a sample and two neighbours, so every sample is classified on its own here:
//   E0: left / right    E1: above / below    E2: above-left / below-right    E3: above-right / below-left;   category = s_eoTable[sign + sign + 2]
// Sums and counts go to LDS histograms (5 classes x 32 bins each), one plane at a time, luma first; wave 0 writes a plane's 320 numbers to the slot and
// releases units[plane].ready behind them.
static_assert(sizeof(x265hip_saojob) <= 128, "SAO job header");
__device__ __forceinline__ int sgn3(int v) { return (v > 0) - (v < 0); }
// the total of v over each row of 16 lanes, valid in lane 15 of the row: four DPP adds
__device__ __forceinline__ int row_total_lane15(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);      // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);      // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);      // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);      // row_shr:8
    return v;
}
// the wave's total of v, valid in lane 63: six DPP adds (quad swaps, half-row and row mirrors, then the row broadcasts of gfx9), no LDS
__device__ __forceinline__ int wave_total_lane63(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);       // quad_perm [1, 0, 3, 2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);       // quad_perm [2, 3, 0, 1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);      // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);      // row_mirror: every lane of a row holds the row's sum
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);      // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ void run_sao(SlotOut* s, JobLds& L, uint32_t seq, uint64_t t0)
{
    const x265hip_saojob& j = *reinterpret_cast<const x265hip_saojob*>(&L.job);
    // LDS (over tile[]): [0..159] sums of class c bin b at c * 32 + b, [160..319] counts — what goes out; [320..1599] the edge classes, [1600..3647] the band
    // class: a COLUMN per lane for every bin (edge class c, sign sum e in 0..4: 320 + ((c * 5 + e) * 64 + lane); band b: 1600 + b * 64 + lane), count in the high
    // and sum in the low half of a word.  A sample is ONE ds_add per class into its lane's column of the bin it falls in (a zero when it lies outside the class's
    // rectangle): lanes never meet on an address, the four waves share the columns (the add is atomic), and nothing is selected or compared per bin
    // (a column sees at most 64 samples: the count fits the high half-word, the sum of differences, |d| <= 255, the low one)
    int* hist = reinterpret_cast<int*>(&L.tile[0]);
    int32_t* out = reinterpret_cast<int32_t*>(s->levels);
    const int tid = threadIdx.x, lx = tid & 63, ly = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned char* at = L.pix;
    const int planes = j.planes < 3 ? (int)j.planes : 3;
    for (int p = 0; p < planes; p++)
    {
        const int w = j.plane[p].w, h = j.plane[p].h, stride = w + 1;
        const unsigned char* rec0 = at + stride + 1;
        const unsigned char* fenc0 = at + (w + 1) * (h + 1);
        at = fenc0 + w * h;
        uint32_t stamp[5] = { 0, 0, 0, 0, 0 };                         // job.reserved != 0 (tools/micro/cuserve_rt): 100 MHz ticks since the doorbell at five points
#define XH_SSTAMP(i) do { if (j.reserved) stamp[i] = (uint32_t)(wall_clock64() - t0) & 0xffffu; } while (0)
        XH_SSTAMP(0);
        {
            int4* h4 = reinterpret_cast<int4*>(hist);
            for (int i = tid; i < 3648 / 4; i += 256) h4[i] = int4{ 0, 0, 0, 0 };
        }
        __syncthreads();
        XH_SSTAMP(1);
        int x0[5], y0[5], x1[5], y1[5];
#pragma unroll
        for (int c = 0; c < 5; c++) { x0[c] = j.plane[p].x0[c]; y0[c] = j.plane[p].y0[c]; x1[c] = j.plane[p].x1[c]; y1[c] = j.plane[p].y1[c]; }
        const bool eo23 = j.eo23 != 0;
        const bool in = lx < w;
        // the rectangles' column tests do not depend on the row
        const bool cx0 = in && lx < x1[0], cx1 = in && lx >= x0[1] && lx < x1[1], cx2 = in && lx < x1[2];
        const bool cx3 = in && eo23 && lx >= x0[3] && lx < x1[3], cx4 = in && eo23 && lx >= x0[4] && lx < x1[4];
        int* edgeCol = hist + 320 + lx;
        int* bandCol = hist + 1600 + lx;
        const int xi = in ? lx : 0;                                         // a lane outside the plane reads column 0 and adds zeros
        // two rows per trip (rows ly + 8k and ly + 8k + 4): the loads of both are in flight before the first is classified — a wave is alone on its SIMD, nothing
        // else hides the LDS latency.  The second row of the last trip may lie below the plane: it is read from the first one's place and adds zeros
        auto classify = [&](int y, bool valid) __attribute__((always_inline))
        {
            const unsigned char* r = rec0 + y * stride + xi;
            const int c = r[0], d = (int)fenc0[y * w + xi] - c;
            const int one = valid ? (1 << 16) + d : 0;
            // sign(c - neighbour) = the difference clamped to [-1, 1] (v_med3_i32)
            const int sR = clip3i(-1, 1, c - (int)r[1]), sL = clip3i(-1, 1, c - (int)r[-1]), sD = clip3i(-1, 1, c - (int)r[stride]), sU = clip3i(-1, 1, c - (int)r[-stride]);
            const int sDR = clip3i(-1, 1, c - (int)r[stride + 1]), sUL = clip3i(-1, 1, c - (int)r[-stride - 1]);
            const int sDL = clip3i(-1, 1, c - (int)r[stride - 1]), sUR = clip3i(-1, 1, c - (int)r[-stride + 1]);
            // (the row tests are wave-uniform: y is)
            const bool r0 = y < y1[0], r1 = y < y1[1], r2 = y >= y0[2] && y < y1[2], r3 = y >= y0[3] && y < y1[3], r4 = y >= y0[4] && y < y1[4];
            __hip_atomic_fetch_add(bandCol + (c >> 3) * 64, cx0 && r0 ? one : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(edgeCol + (0 * 5 + sR + sL + 2) * 64, cx1 && r1 ? one : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(edgeCol + (1 * 5 + sD + sU + 2) * 64, cx2 && r2 ? one : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(edgeCol + (2 * 5 + sDR + sUL + 2) * 64, cx3 && r3 ? one : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(edgeCol + (3 * 5 + sDL + sUR + 2) * 64, cx4 && r4 ? one : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        for (int y = ly; y < h; y += 8)
        {
            const bool second = y + 4 < h;
            classify(y, true);
            classify(second ? y + 4 : y, second);
        }
        __syncthreads();
        XH_SSTAMP(2);
        // the edge classes: wave c totals class c's five columns sets (six DPP adds each for sums and counts); s_eoTable (sao.cpp:65) folds sign + sign + 2 =
        // 0..4 into the categories 1, 2, 0, 3, 4
        {
#pragma unroll
            for (int e = 0; e < 5; e++)
            {
                const int v = hist[320 + (ly * 5 + e) * 64 + lx];
                const int sumL = (int)(short)(v & 0xffff), cntL = (v - sumL) >> 16;
                const int ts = wave_total_lane63(sumL), tc = wave_total_lane63(cntL);
                const int k = e == 0 ? 1 : e == 1 ? 2 : e == 2 ? 0 : e;
                if (lx == 63)
                {
                    hist[(ly + 1) * 32 + k] = ts;
                    hist[160 + (ly + 1) * 32 + k] = tc;
                }
            }
        }
        XH_SSTAMP(3);
        // the band class: thread t adds up eight columns of band t / 8, the eight threads of a band combine by DPP
        {
            const int bnd = tid >> 3, seg = tid & 7;
            const int* col = hist + 1600 + bnd * 64 + seg * 8;
            int sumB = 0, cntB = 0;
#pragma unroll
            for (int i = 0; i < 8; i++)
            {
                const int v = col[i];
                const int sL = (int)(short)(v & 0xffff);
                sumB += sL; cntB += (v - sL) >> 16;
            }
            // eight consecutive lanes: xor 1, 2 (quad DPP), then lane + 4 of the same row
            sumB += __builtin_amdgcn_update_dpp(0, sumB, 0xB1, 0xf, 0xf, false); cntB += __builtin_amdgcn_update_dpp(0, cntB, 0xB1, 0xf, 0xf, false);
            sumB += __builtin_amdgcn_update_dpp(0, sumB, 0x4E, 0xf, 0xf, false); cntB += __builtin_amdgcn_update_dpp(0, cntB, 0x4E, 0xf, 0xf, false);
            sumB += __builtin_amdgcn_update_dpp(0, sumB, 0x104, 0xf, 0xf, false); cntB += __builtin_amdgcn_update_dpp(0, cntB, 0x104, 0xf, 0xf, false);   // row_shl:4
            if (seg == 0) { hist[bnd] = sumB; hist[160 + bnd] = cntB; }
        }
        __syncthreads();
        if (tid < 64)
        {
            for (int i = tid; i < 160; i += 64)
            {
                out[p * 160 + i] = hist[i];
                out[X265HIP_SAOJOB_STATS_ENTRIES + p * 160 + i] = hist[160 + i];
            }
            // the release store waits for this wave's own stores (s_waitcnt vmcnt(0) is per wave): lane 0 publishes after the whole wave has issued them
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_wave_barrier();
            if (tid == 0)
            {
                XH_SSTAMP(4);
                if (j.reserved) { s->units[p].reserved[0] = stamp[0] | (stamp[1] << 16); s->units[p].reserved[1] = stamp[2] | (stamp[3] << 16); s->units[p].reserved[2] = stamp[4]; }
                s->units[p].fwdTicks = (uint32_t)(wall_clock64() - t0);
                __hip_atomic_store(&s->units[p].readyInv, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&s->units[p].ready, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        __syncthreads();
#undef XH_SSTAMP
    }
}

// ---- intra mode scan jobs (x265hip_intrajob): sa8d of all 35 predictions of one block, intra.hip's intra_scan_kernel with the lines and the source block
// in the job's LDS copy.  A lane = one 4x4 tile whose 16 predicted samples are made in registers; a 32x32 block is one mode per wave and pass (9 passes of
// the workgroup), a 16x16 block four modes per wave (3 passes), an 8x8 block sixteen (1 pass).
template <typename P>
__device__ __forceinline__ void run_intra(SlotOut* s, JobLds& L, uint32_t seq, uint64_t t0)
{
    const x265hip_intrajob& j = *reinterpret_cast<const x265hip_intrajob*>(&L.job);
    const int log2n = (int)j.log2Size, n = 1 << log2n, n2 = 2 * n, depth = (int)j.bitDepth, maxv = (1 << depth) - 1;
    const int lineSamples = x265hipi_intrajob_line_samples(log2n);
    const P* raw = reinterpret_cast<const P*>(L.pix);
    const P* fenc = raw + 2 * lineSamples;
    // LDS over tile[]: [0..34] the costs (ints); from byte 256 the two lines CENTRED — c[j], j in [-2N, 2N]: 0 the corner, +j top / top-right, -j left /
    // bottom-left (intra_dev.h) — as 16-bit samples, the filtered line 320 entries behind the unfiltered one: every read below is one ds_read_u16, no select
    int* costs = reinterpret_cast<int*>(&L.tile[0]);
    uint16_t* cen = reinterpret_cast<uint16_t*>(&L.tile[0]) + 128;
    constexpr int kFlt = 320;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * (4 * n + 1); i += 256)
    {
        const int which = i >= 4 * n + 1, e = which ? i - (4 * n + 1) : i, jj = e - n2;
        cen[which * kFlt + e] = (uint16_t)raw[which * lineSamples + (jj >= 0 ? jj : n2 - jj)];
    }
    const int tiles = (n >> 2) * (n >> 2);
    const int T = tiles >= 64 ? 64 : tiles;
    const int jpw = 64 / T, sub = lane & (T - 1);
    const bool bFilter = n <= 16;
    // tile order as in pixel.hip: 16x16 blocks in raster order, inside them 8x8 blocks, inside them the four 4x4 quadrants (one DPP quad)
    const int n16x = n >= 16 ? (n >> 4) : 1;
    const int b16 = sub >> 4, b8 = (sub >> 2) & 3, q = sub & 3;
    int x = (b16 % n16x) * 16 + (b8 & 1) * 8 + (q & 1) * 4;
    int y = (b16 / n16x) * 16 + (b8 >> 1) * 8 + (q >> 1) * 4;
    if (n == 8) { x = (q & 1) * 4; y = (q >> 1) * 4; }
    int fe[16];
#pragma unroll
    for (int yy = 0; yy < 4; yy++)
        load4(fenc + (y + yy) * n + x, &fe[4 * yy]);
    __syncthreads();
    const uint16_t* cr = cen + n2;                            // the unfiltered line, centred
    int dcv = n;
    for (int i = 1; i <= n; i++)
        dcv += (int)cr[i] + (int)cr[-i];
    dcv >>= log2n + 1;
    for (int base = 0; base < 35; base += 4 * jpw)
    {
        const int want = base + wave * jpw + lane / T;
        const bool ok = want < 35;
        const int mode = ok ? want : 34;
        int m[16];
        // ---- the angular prediction of every lane's mode, branch-free (modes 0 and 1 run it as mode 2 and are overwritten below).  intra_dev.h's
        // ang_sample with the f == 0 case folded into the interpolation — ((32 - 0) * s0 + 16) >> 5 == s0 — and the line read from the centred copy
        {
            const int am = mode < 2 ? 2 : mode;
            const bool horiz = am < 18;
            const int rel = horiz ? 10 - am : am - 26;
            const int sgn = horiz ? -1 : 1;
            const int angle = kIntraAngle[8 + rel];
            const int inv = rel < 0 ? kIntraInvAngle[-rel - 1] : 0;
            const uint16_t* c = cr + (intra_uses_filtered(n, am) ? kFlt : 0);
            const bool edge = bFilter && angle == 0;
            const int corner = (int)c[0];
#pragma unroll
            for (int yy = 0; yy < 4; yy++)
#pragma unroll
                for (int xx = 0; xx < 4; xx++)
                {
                    const int px = x + xx, py = y + yy;
                    const int u = horiz ? py : px, v = horiz ? px : py;
                    const int t = (v + 1) * angle;
                    const int k = (t >> 5) + u + 1, f = t & 31;
                    const int j0 = k >= 0 ? sgn * k : -sgn * ((128 - k * inv) >> 8);
                    const int k1 = k + 1;
                    const int j1 = k1 >= 0 ? sgn * k1 : -sgn * ((128 - k1 * inv) >> 8);
                    int val = ((32 - f) * (int)c[j0] + f * (int)c[j1] + 16) >> 5;
                    // pure vertical / horizontal with edge smoothing: the first column / row follows the gradient of the other arm (intrapred.cpp:146-151)
                    const int g = (int)c[sgn] + (((int)c[-sgn * (v + 1)] - corner) >> 1);
                    const int gc = g < 0 ? 0 : (g > maxv ? maxv : g);
                    val = (edge && u == 0) ? gc : val;
                    m[4 * yy + xx] = val;
                }
        }
        // ---- planar and DC: two of the 35, taken only by the waves that hold them
        if (__builtin_amdgcn_ballot_w64(mode < 2) != 0)
        {
            const uint16_t* cp = cr + (n >= 8 ? kFlt : 0);    // planar reads the filtered line when N >= 8 (search.cpp:1363-1365)
#pragma unroll
            for (int yy = 0; yy < 4; yy++)
#pragma unroll
                for (int xx = 0; xx < 4; xx++)
                {
                    const int px = x + xx, py = y + yy;
                    const int planar = ((n - 1 - px) * (int)cp[-(py + 1)] + (px + 1) * (int)cp[n + 1] + (n - 1 - py) * (int)cp[px + 1] + (py + 1) * (int)cp[-(n + 1)] + n) >> (log2n + 1);
                    int dc = dcv;
                    if (bFilter && !(px && py))
                        dc = (px == 0 && py == 0) ? ((int)cr[1] + (int)cr[-1] + 2 * dcv + 2) >> 2
                                                  : ((py == 0 ? (int)cr[px + 1] : (int)cr[-(py + 1)]) + 3 * dcv + 2) >> 2;
                    m[4 * yy + xx] = mode == 0 ? planar : mode == 1 ? dc : m[4 * yy + xx];
                }
        }
#pragma unroll
        for (int i = 0; i < 16; i++)
            m[i] = fe[i] - m[i];
        hadamard4x4(m);
        const int raw8 = quad_sa8d_raw(m, lane);
        int acc;
        if (n >= 16)
        {
            const int s16 = group_sum((lane & 3) == 0 ? raw8 : 0, 16);
            acc = (lane & 15) == 0 ? ((s16 + 2) >> 2) : 0;     // sa8d_16x16: four raw 8x8, one rounding (pixel.cpp:341-350)
        }
        else
            acc = (lane & 3) == 0 ? ((raw8 + 2) >> 2) : 0;     // sa8d_8x8 (pixel.cpp:336)
        acc = group_sum(acc, T);
        if (ok && sub == 0)
            costs[mode] = acc;
    }
    __syncthreads();
    if (tid < 64)
    {
        int32_t* out = reinterpret_cast<int32_t*>(s->levels);
        if (tid < 35)
            out[tid] = costs[tid];
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        if (tid == 0)
        {
            s->units[0].fwdTicks = (uint32_t)(wall_clock64() - t0);
            __hip_atomic_store(&s->units[0].readyInv, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&s->units[0].ready, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __syncthreads();
}

// one job: `ticket` says how many bytes the job holds, so header and pixels arrive in one round trip.  role 0 / 1: this workgroup's half of a CU job (luma /
// chroma: each fetches the header and its own planes only); SAO statistics and intra scans are role 0's alone
__device__ __forceinline__ void run_job(const SlotIn* sin, SlotOut* s, JobLds& L, uint32_t ticket, uint64_t* busyTicks, int role, bool team)
{
    const int tid = threadIdx.x;
    const bool cuJob = (ticket & 3) != 3;
    if (role && (!cuJob || !(ticket & 4)))
        return;                                                              // nothing of this job is role 1's (uniform over the workgroup)
    const uint64_t t0 = wall_clock64();
    const uint4* in = reinterpret_cast<const uint4*>(&sin->job);
    uint4* out = reinterpret_cast<uint4*>(&L.job);
    if (!cuJob)
    {
        const int chunks = (128 + (int)ticket_bytes(ticket)) >> 4;
        for (int i = tid; i < chunks; i += 256)
            out[i] = in[i];
    }
    else
    {
        // header (8 chunks), then the role's planes of the source and of the prediction: luma n2 * B bytes at 0, chroma n2 / 2 * B bytes behind it, the
        // prediction planeBytes further on (all multiples of 16)
        const int B = (ticket & 8) ? 2 : 1, n2 = 1 << (2 * ((int)(ticket & 3) + 4));
        const int planeBytes = ((ticket & 4) ? n2 + n2 / 2 : n2) * B;
        const int first = (role ? n2 * B : 0) >> 4, count = (role ? (n2 / 2) * B : n2 * B) >> 4, pred = planeBytes >> 4;
        // (bit 4 of the ticket: an inverse job — 1 024 levels behind the two blocks)
        const int extra = (ticket & 16) ? 128 : 0;
        for (int i = tid; i < 8 + 2 * count + extra; i += 256)
        {
            const int k = i < 8 ? i : i < 8 + count ? 8 + first + (i - 8) : i < 8 + 2 * count ? 8 + pred + first + (i - 8 - count) : 8 + 2 * pred + (i - 8 - 2 * count);
            out[k] = in[k];
        }
    }
    __syncthreads();
    if (!cuJob)
    {
        const x265hip_intrajob& ij = *reinterpret_cast<const x265hip_intrajob*>(&L.job);
        if (!(ij.mark & X265HIP_INTRAJOB_MARK)) run_sao(s, L, ticket, t0);
        else if (ij.bitDepth > 8) run_intra<uint16_t>(s, L, ticket, t0);
        else run_intra<uint8_t>(s, L, ticket, t0);
    }
    else if (ticket & 8) run_tiles<uint16_t>(s, L, ticket, t0, role, team);
    else run_tiles<uint8_t>(s, L, ticket, t0, role, team);
    __syncthreads();
    if (tid == 0)
        *busyTicks += wall_clock64() - t0;
}

// mode 1: one job, one launch of the pair
__global__ __launch_bounds__(256) void cu_job_kernel(const SlotIn* sin, SlotOut* s, uint32_t ticket, uint64_t* busyTicks, uint32_t team)
{
    __shared__ JobLds L;
    build_operands(L);
    run_job(sin, s, L, ticket, busyTicks + blockIdx.x, (int)blockIdx.x, team != 0);
}

// mode 0: workgroups 2b and 2b + 1 serve slot b (roles 0 and 1, run_tiles).  The server as a whole leaves when no workgroup has taken a job for `idleTicks` (100 MHz) or the host rings
// 0xffffffff on any slot: the workgroup that notices sets ctl->quit, every workgroup leaves at its next poll, the last one tells the host.
__global__ __launch_bounds__(256) void cu_server_kernel(const SlotIn* ins, SlotOut* outs, HostCtl* hostCtl, DevCtl* ctl, uint32_t generation, uint64_t idleTicks)
{
    __shared__ JobLds L;
    const int role = (int)(blockIdx.x & 1);
    const SlotIn* sin = ins + (blockIdx.x >> 1);
    SlotOut* s = outs + (blockIdx.x >> 1);
    build_operands(L);
    uint32_t last = 0;
    if (threadIdx.x == 0)
    {
        // "the server has had work" starts now for every workgroup, whichever gets on the chip first (the word still holds the previous server's time)
        __hip_atomic_fetch_max(&ctl->lastWork, wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = __hip_atomic_load(&sin->doorbell, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        // a job rung while no server was there has not been done: the first unit of this workgroup's share is not ready (role 1: the first chroma unit of a
        // CU job with chroma — the units behind the largest level's luma units; the header of a rung job is in the slot)
        int firstUnit = 0;
        if (role && last && last != 0xffffffffu && (last & 3) != 3 && (last & 4))
        {
            const uint32_t log2cu = (last & 3) + 4, trMax = __hip_atomic_load(&sin->job.log2TrMax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const uint32_t sHi = trMax < 5 ? (trMax < log2cu ? trMax : log2cu) : (log2cu < 5 ? log2cu : 5);
            firstUnit = 1 << (2 * (log2cu - sHi));
        }
        else if (role)
            firstUnit = -1;                                                  // (not a job role 1 has a share in: nothing to look at)
        if (firstUnit >= 0 && last && last != 0xffffffffu && __hip_atomic_load(&s->units[firstUnit].ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != last)
            last = 0;
        // serverState has ONE writer, workgroup 0 (two workgroups' stores to the same host word may arrive in either order)
        if (blockIdx.x == 0)
            __hip_atomic_store(&hostCtl->serverState, generation, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (;;)
    {
        if (threadIdx.x == 0)
        {
            uint32_t v;
            int polls = 0;
            for (;;)
            {
                v = __hip_atomic_load(&sin->doorbell, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
                // `leave` lives in host memory (a PCIe read of ~1.5 us): workgroup 0 looks at it every 8th poll only, its doorbell every time
                if (v == 0xffffffffu || (blockIdx.x == 0 && (polls & 7) == 7 && __hip_atomic_load(&hostCtl->leave, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)))
                {
                    __hip_atomic_store(&ctl->quit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    v = 0xffffffffu;
                    break;
                }
                if (__hip_atomic_load(&ctl->quit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { v = 0xffffffffu; break; }
                if (v != last) break;
                if ((++polls & 15) == 0)
                {
                    // read the word first, the clock second: another workgroup may store a later time in between, never an earlier one
                    const uint64_t lw = __hip_atomic_load(&ctl->lastWork, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint64_t now = wall_clock64();
                    if (now > lw && now - lw > (idleTicks & ~(3ull << 62)))
                    {
                        __hip_atomic_store(&ctl->quit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        v = 0xffffffffu;
                        break;
                    }
                }
                // (bit 62 of idleTicks: X265HIP_CUSERVE_TEAM=0, see team_off())
                // X265HIP_CUSERVE_SPIN=1 (bit 63 of idleTicks): no sleep between polls — an experiment: does a chip that sees busy CUs clock higher?
                if (!(idleTicks >> 63)) __builtin_amdgcn_s_sleep(2);
            }
            if (v != 0xffffffffu)
                __hip_atomic_fetch_max(&ctl->lastWork, wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            L.seq = v;
        }
        __syncthreads();
        const uint32_t v = L.seq;
        __syncthreads();
        if (v == 0xffffffffu)
        {
            // everybody leaves; workgroup 0 leaves last and tells the host (a submitter that rang in between finds serverState == 0 while it waits and
            // starts the next server, which takes the pending job)
            if (threadIdx.x == 0)
            {
                if (blockIdx.x != 0)
                    __hip_atomic_fetch_add(&ctl->left, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                else
                {
                    while (__hip_atomic_load(&ctl->left, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != gridDim.x - 1)
                        __builtin_amdgcn_s_sleep(2);
                    __hip_atomic_store(&hostCtl->serverState, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            return;
        }
        last = v;
        run_job(sin, s, L, v, &ctl->busyTicks[blockIdx.x], role, !((idleTicks >> 62) & 1));
    }
}

} // namespace
} // namespace xh

using namespace xh;

// sfence: stores to write-combining memory (the BAR) become globally visible in program order across it, and the buffers are drained
static inline void store_fence()
{
#if !defined(__HIP_DEVICE_COMPILE__)
    __asm__ __volatile__("sfence" ::: "memory");
#endif
}

struct x265hip_cuserve
{
    int slots = 0, mode = 0, device = 0;
    SlotIn* in = nullptr;                         // as the host addresses it: device memory through the BAR (inDevice) or page-locked host memory
    SlotIn* inDev = nullptr;                      // as the device addresses it
    bool inDevice = false;
    SlotOut* out = nullptr;                       // page-locked, coherent
    SlotOut* outDev = nullptr;                    // the same memory as the device addresses it
    x265hip_cujob* shadow = nullptr;              // per slot: the header as the submitter fills (and reads) it; copied into the mailbox by submit
    HostCtl* hostCtl = nullptr; HostCtl* devHostCtl = nullptr;
    DevCtl* ctl = nullptr;                        // device memory
    hipStream_t serverStream = nullptr;
    hipStream_t* jobStreams = nullptr;            // mode 1
    std::atomic<uint32_t>* seq = nullptr;         // per slot
    std::atomic<uint32_t> generation{ 0 };
    std::mutex launchLock;
    std::atomic<uint64_t> jobs{ 0 }, starts{ 0 }, bytes{ 0 }, saoJobs{ 0 }, intraJobs{ 0 };
    std::atomic<int> paused{ 0 };                 // servers_pause() callers in progress: no server is started meanwhile
    uint64_t idleUs = 2000;
};
static std::mutex g_openLock;
static x265hip_cuserve* g_open[16];               // the services with a resident server (mode 0)
static int g_pauseDepth;                          // servers_pause() calls without their servers_resume() yet (under g_openLock): a service opened meanwhile joins them

// X265HIP_CUSERVE_TEAM=0: 32x32 luma units run on one wave each (rounds 4-5's form; A/B switch for the team form of round 6)
static bool team_off()
{
    static const bool off = getenv("X265HIP_CUSERVE_TEAM") && !atoi(getenv("X265HIP_CUSERVE_TEAM"));
    return off;
}

static int start_server(x265hip_cuserve* cs)
{
    std::lock_guard<std::mutex> g(cs->launchLock);
    if (__atomic_load_n(&cs->hostCtl->serverState, __ATOMIC_ACQUIRE) != 0 || cs->paused.load() > 0)
        return X265HIP_OK;                                                   // someone else has started one meanwhile / a device synchronisation is going on
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != cs->device && hipSetDevice(cs->device) != hipSuccess)
        return set_error(X265HIP_EHIP, "cuserve: hipSetDevice(%d) failed", cs->device);
    const uint32_t gen = (cs->generation.fetch_add(1) + 1) & 0x7fffffffu;
    // the device writes the word when the server runs; "being started" keeps other submitters from starting a second one
    __atomic_store_n(&cs->hostCtl->serverState, 0x80000000u | gen, __ATOMIC_RELEASE);
    // quit / left back to zero in stream order (the previous server, if any, has left: it was the one that cleared serverState)
    hipError_t e = hipMemsetAsync(cs->ctl, 0, 8, cs->serverStream);
    if (e == hipSuccess)
    {
        hipLaunchKernelGGL(cu_server_kernel, dim3(2 * cs->slots), dim3(256), 0, cs->serverStream, cs->inDev, cs->outDev, cs->devHostCtl, cs->ctl, gen ? gen : 1u,
                           cs->idleUs * 100 | (getenv("X265HIP_CUSERVE_SPIN") && atoi(getenv("X265HIP_CUSERVE_SPIN")) ? 1ull << 63 : 0ull) | (team_off() ? 1ull << 62 : 0ull));
        e = hipGetLastError();
    }
    if (cur >= 0 && cur != cs->device) (void)hipSetDevice(cur);
    if (e != hipSuccess)
    {
        __atomic_store_n(&cs->hostCtl->serverState, 0u, __ATOMIC_RELEASE);
        return check_hip(e, "cu_server_kernel");
    }
    cs->starts++;
    return X265HIP_OK;
}

namespace xh {
void servers_pause()
{
    std::lock_guard<std::mutex> g(g_openLock);
    g_pauseDepth++;
    for (x265hip_cuserve* cs : g_open)
    {
        if (!cs) continue;
        std::lock_guard<std::mutex> l(cs->launchLock);                       // no start_server between the test and the flag
        cs->paused++;
        __atomic_store_n(&cs->hostCtl->leave, 1u, __ATOMIC_RELEASE);
    }
    // the servers leave at their next poll (a job in progress is finished first): wait for that, briefly — hipFree would wait anyway
    const auto t0 = std::chrono::steady_clock::now();
    for (x265hip_cuserve* cs : g_open)
        while (cs && __atomic_load_n(&cs->hostCtl->serverState, __ATOMIC_ACQUIRE) != 0 &&
               std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(200))
            __builtin_ia32_pause();
}
void servers_resume()
{
    std::lock_guard<std::mutex> g(g_openLock);
    if (g_pauseDepth > 0) g_pauseDepth--;
    for (x265hip_cuserve* cs : g_open)
        if (cs && cs->paused.load() > 0 && --cs->paused == 0)
            __atomic_store_n(&cs->hostCtl->leave, 0u, __ATOMIC_RELEASE);
}
} // namespace xh

extern "C" {

int x265hip_cuserve_open(int slots, int mode, x265hip_cuserve** out)
{
    XH_CHECK_DEV();
    if (!out || slots < 1 || slots > 256 || mode < 0 || mode > 1)
        return set_error(X265HIP_EINVAL, "x265hip_cuserve_open: slots %d mode %d", slots, mode);
    if (mode == 0)
    {
        // every slot is a resident workgroup: all of them must fit on the chip at once, or the jobs of the ones that never get a CU are never done.  What fits is
        // bounded by the LDS a workgroup holds (sizeof(JobLds)); a quarter of the CUs stays free for the kernels that are launched and finish
        int dev = 0, cus = 0, lds = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0 &&
            hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) == hipSuccess && lds > 0)
        {
            const int perCu = lds / (int)sizeof(JobLds) > 0 ? lds / (int)sizeof(JobLds) : 1;
            const int fit = (cus - cus / 4) * perCu;
            if (2 * slots > fit)
                return set_error(X265HIP_EINVAL, "x265hip_cuserve_open: %d resident workgroups (two per slot) asked for, %d fit beside the other kernels (%d CUs, %d per CU)", 2 * slots, fit, cus, perCu);
        }
        else
            (void)hipGetLastError();
    }
    x265hip_cuserve* cs = new (std::nothrow) x265hip_cuserve;
    if (!cs) return set_error(X265HIP_ENOMEM, "x265hip_cuserve_open: out of memory");
    cs->slots = slots; cs->mode = mode;
    (void)hipGetDevice(&cs->device);
    if (const char* env = getenv("X265HIP_CUSERVE_IDLE_US")) cs->idleUs = (uint64_t)atoll(env);
    int e = check_hip(hipHostMalloc((void**)&cs->out, sizeof(SlotOut) * slots, hipHostMallocCoherent | hipHostMallocMapped), "hipHostMalloc(cuserve slots)");
    if (!e) { memset(cs->out, 0, sizeof(SlotOut) * slots); e = check_hip(hipHostGetDevicePointer((void**)&cs->outDev, cs->out, 0), "hipHostGetDevicePointer(cuserve)"); }
    if (!e)
    {
        // the host -> device half: device memory when the host can write it (large BAR), see SlotIn
        int largeBar = 0;
        (void)hipDeviceGetAttribute(&largeBar, hipDeviceAttributeIsLargeBar, cs->device);
        const char* where = getenv("X265HIP_CUSERVE_MAILBOX");
        const bool wantDevice = where ? !strcmp(where, "device") : largeBar != 0;
        bool usable = false;
        if (wantDevice && hipExtMallocWithFlags((void**)&cs->inDev, sizeof(SlotIn) * slots, hipDeviceMallocUncached) == hipSuccess &&
            hipMemset(cs->inDev, 0, sizeof(SlotIn) * slots) == hipSuccess && hipDeviceSynchronize() == hipSuccess)
        {
            // does a store through the BAR arrive?  A pattern into the last slot's pixel block, read back with a device-to-host copy, then cleared
            // (a mapping that silently drops the writes would otherwise show up as jobs that never come back)
            uint32_t pattern[16], back[16];
            for (int k = 0; k < 16; k++) pattern[k] = 0x9e3779b9u * (uint32_t)(k + 1);
            memcpy(cs->inDev[slots - 1].pixels, pattern, sizeof(pattern));
            store_fence();
            usable = hipMemcpy(back, cs->inDev[slots - 1].pixels, sizeof(back), hipMemcpyDeviceToHost) == hipSuccess && !memcmp(back, pattern, sizeof(back)) &&
                     hipMemset(cs->inDev[slots - 1].pixels, 0, sizeof(pattern)) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
        }
        if (usable)
        {
            cs->in = cs->inDev;
            cs->inDevice = true;
        }
        else
        {
            (void)hipGetLastError();
            if (cs->inDev) { (void)hipFree(cs->inDev); cs->inDev = nullptr; }
            e = check_hip(hipHostMalloc((void**)&cs->in, sizeof(SlotIn) * slots, hipHostMallocCoherent | hipHostMallocMapped), "hipHostMalloc(cuserve mailbox)");
            if (!e) { memset(cs->in, 0, sizeof(SlotIn) * slots); e = check_hip(hipHostGetDevicePointer((void**)&cs->inDev, cs->in, 0), "hipHostGetDevicePointer(cuserve mailbox)"); }
        }
    }
    if (!e)
    {
        cs->shadow = new (std::nothrow) x265hip_cujob[slots];
        if (!cs->shadow) e = set_error(X265HIP_ENOMEM, "x265hip_cuserve_open: out of memory");
        else memset(cs->shadow, 0, sizeof(x265hip_cujob) * slots);
    }
    if (!e) e = check_hip(hipHostMalloc((void**)&cs->hostCtl, 64, hipHostMallocCoherent | hipHostMallocMapped), "hipHostMalloc(cuserve control)");
    if (!e) { memset(cs->hostCtl, 0, 64); e = check_hip(hipHostGetDevicePointer((void**)&cs->devHostCtl, cs->hostCtl, 0), "hipHostGetDevicePointer(cuserve control)"); }
    if (!e) e = check_hip(hipMalloc((void**)&cs->ctl, sizeof(DevCtl)), "hipMalloc(cuserve control)");
    // the server stream gets the highest priority: HIP maps priorities to separate hardware queues, so the resident kernel never sits in front of
    // another stream's packets in the same queue
    int lo = 0, hi = 0;
    if (!e) { (void)hipDeviceGetStreamPriorityRange(&lo, &hi); e = check_hip(hipStreamCreateWithPriority(&cs->serverStream, hipStreamNonBlocking, hi), "hipStreamCreateWithPriority(cuserve)"); }
    if (!e) e = check_hip(hipMemsetAsync(cs->ctl, 0, sizeof(DevCtl), cs->serverStream), "hipMemsetAsync(cuserve control)");
    if (!e) e = check_hip(hipStreamSynchronize(cs->serverStream), "hipStreamSynchronize(cuserve)");
    if (!e)
    {
        cs->seq = new (std::nothrow) std::atomic<uint32_t>[slots];
        cs->jobStreams = new (std::nothrow) hipStream_t[slots];
        if (!cs->seq || !cs->jobStreams) e = set_error(X265HIP_ENOMEM, "x265hip_cuserve_open: out of memory");
        else
            for (int i = 0; i < slots; i++) { cs->seq[i] = 0; cs->jobStreams[i] = nullptr; }
    }
    if (!e && mode == 1)
        for (int i = 0; i < slots && !e; i++)
            e = check_hip(hipStreamCreateWithFlags(&cs->jobStreams[i], hipStreamNonBlocking), "hipStreamCreate(cuserve job)");
    if (e) { x265hip_cuserve_close(cs); return e; }
    if (mode == 0)
    {
        bool registered = false;
        {
            std::lock_guard<std::mutex> g(g_openLock);
            for (x265hip_cuserve*& slot : g_open)
                if (!slot) { slot = cs; registered = true; break; }
            // opened in the middle of somebody's device synchronisation: this service is paused with the others and resumes with them
            if (registered && g_pauseDepth > 0)
            {
                cs->paused = g_pauseDepth;
                __atomic_store_n(&cs->hostCtl->leave, 1u, __ATOMIC_RELEASE);
            }
            if (registered) resident_workgroups(cs->device, 2 * slots);
        }
        // (closed outside the lock: x265hip_cuserve_close and the device_free() it reaches take g_openLock themselves)
        if (!registered) { x265hip_cuserve_close(cs); return set_error(X265HIP_EINVAL, "x265hip_cuserve_open: more than %d services with a resident server", (int)(sizeof(g_open) / sizeof(g_open[0]))); }
    }
    *out = cs;
    return X265HIP_OK;
}

static uint64_t device_ticks(x265hip_cuserve* cs)
{
    if (!cs->ctl) return 0;
    static thread_local uint64_t buf[512];
    // a plain copy on the null stream would wait for the resident server: copy on a stream of its own
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return 0; }
    uint64_t ticks = 0;
    if (hipMemcpyAsync(buf, (char*)cs->ctl + offsetof(DevCtl, busyTicks), sizeof(uint64_t) * 2 * cs->slots, hipMemcpyDeviceToHost, st) == hipSuccess &&
        hipStreamSynchronize(st) == hipSuccess)
        for (int i = 0; i < 2 * cs->slots; i++) ticks += buf[i];
    else
        (void)hipGetLastError();
    (void)hipStreamDestroy(st);
    return ticks;
}

int x265hip_cuserve_open_at(int place, int slots, int mode, x265hip_cuserve** out)
{
    const int dev = place >= 0 ? place_device(place) : -1;
    if (dev < 0) return set_error(X265HIP_EINVAL, "x265hip_cuserve_open_at: no place %d (x265hip_places)", place);
    int cur = 0;
    const bool had = hipGetDevice(&cur) == hipSuccess;
    if (hipSetDevice(dev) != hipSuccess) return set_error(X265HIP_EHIP, "x265hip_cuserve_open_at: hipSetDevice(%d)", dev);
    const int e = x265hip_cuserve_open(slots, mode, out);
    if (had) (void)hipSetDevice(cur);
    return e;
}

int x265hip_cuserve_close(x265hip_cuserve* cs)
{
    if (!cs) return X265HIP_OK;
    {
        std::lock_guard<std::mutex> g(g_openLock);
        for (x265hip_cuserve*& slot : g_open)
            if (slot == cs) { slot = nullptr; resident_workgroups(cs->device, -2 * cs->slots); }
    }
    if (cs->in)
    {
        __atomic_store_n(&cs->in[0].doorbell, 0xffffffffu, __ATOMIC_RELEASE);             // a resident server leaves at its next poll
        store_fence();
        if (cs->serverStream) (void)hipStreamSynchronize(cs->serverStream);
        if (cs->jobStreams)
            for (int i = 0; i < cs->slots; i++)
                if (cs->jobStreams[i]) { (void)hipStreamSynchronize(cs->jobStreams[i]); (void)hipStreamDestroy(cs->jobStreams[i]); }
        clock_add(X265HIP_CLK_CUSERVE, cs->jobs.load(), device_ticks(cs) * 10, cs->bytes.load());
        if (cs->inDevice) (void)device_free(cs->in); else (void)hipHostFree(cs->in);
    }
    if (cs->out) (void)hipHostFree(cs->out);
    delete[] cs->shadow;
    if (cs->hostCtl) (void)hipHostFree(cs->hostCtl);
    if (cs->ctl) (void)device_free(cs->ctl);
    if (cs->serverStream) (void)hipStreamDestroy(cs->serverStream);
    delete[] cs->seq; delete[] cs->jobStreams;
    delete cs;
    return X265HIP_OK;
}

int x265hip_cuserve_slot(x265hip_cuserve* cs, int slot, x265hip_cujob** job, void** pixels, const x265hip_cujob_unit** units, const int16_t** levels,
                         const int16_t** resi)
{
    if (!cs || slot < 0 || slot >= cs->slots) return set_error(X265HIP_EINVAL, "x265hip_cuserve_slot: slot %d", slot);
    SlotOut* s = cs->out + slot;
    if (job) *job = cs->shadow + slot;
    if (pixels) *pixels = cs->in[slot].pixels;
    if (units) *units = s->units;
    if (levels) *levels = s->levels;
    if (resi) *resi = s->resi;
    return X265HIP_OK;
}

int x265hip_cuserve_submit(x265hip_cuserve* cs, int slot, uint32_t* seqOut)
{
    if (!cs || slot < 0 || slot >= cs->slots || !seqOut) return set_error(X265HIP_EINVAL, "x265hip_cuserve_submit: slot %d", slot);
    SlotIn* s = cs->in + slot;
    const x265hip_cujob& j = cs->shadow[slot];
    int sHi, sLo;
    if (j.log2CUSize < 4 || j.log2CUSize > 6 || x265hipi_cujob_levels(&j, &sHi, &sLo) < 1 || !valid_depth((int)j.bitDepth))
        return set_error(X265HIP_EINVAL, "x265hip_cuserve_submit: CU 2^%u, transform sizes 2^%u..2^%u, depth %u", j.log2CUSize, j.log2TrMin, j.log2TrMax, j.bitDepth);
    uint32_t run = cs->seq[slot].load(std::memory_order_relaxed) + 1;                       // a slot has one submitter at a time
    if (run >= 0xfffff0u) run = 1;
    cs->seq[slot].store(run, std::memory_order_relaxed);
    const bool inverseJob = j.coefMode == X265HIP_CUJOB_INVERSE;
    if (inverseJob && (j.log2CUSize != 5 || j.chroma || sHi != 5))
        return set_error(X265HIP_EINVAL, "x265hip_cuserve_submit: an inverse job is one 32x32 luma unit (CU 2^%u, chroma %u, transform 2^%d)", j.log2CUSize, j.chroma, sHi);
    const uint32_t seq = (run << 8) | (j.log2CUSize - 4) | (j.chroma ? 4u : 0u) | (j.bitDepth > 8 ? 8u : 0u) | (inverseJob ? 16u : 0u);
    *seqOut = seq;
    cs->jobs.fetch_add(1, std::memory_order_relaxed);
    {
        // SURVEY.md §8d, fused chain of one transform unit, as this job moves it: source + prediction in, levels + reconstructed residual out
        const uint64_t n2 = 1ull << (2 * j.log2CUSize), B = j.bitDepth > 8 ? 2 : 1;
        cs->bytes.fetch_add((uint64_t)(sHi - sLo + 1) * (j.chroma ? n2 + n2 / 2 : n2) * (2 * B + 4), std::memory_order_relaxed);
    }
    memcpy(&s->job, &j, sizeof(j));
    store_fence();                                                                           // header and pixels leave the core before the doorbell does
    if (cs->mode == 1)
    {
        std::atomic_thread_fence(std::memory_order_release);
        int cur = -1;
        (void)hipGetDevice(&cur);
        if (cur != cs->device) (void)hipSetDevice(cs->device);
        hipLaunchKernelGGL(cu_job_kernel, dim3(2), dim3(256), 0, cs->jobStreams[slot], cs->inDev + slot, cs->outDev + slot, seq, &cs->ctl->busyTicks[2 * slot], team_off() ? 0u : 1u);
        const hipError_t le = hipGetLastError();
        if (cur >= 0 && cur != cs->device) (void)hipSetDevice(cur);
        if (le != hipSuccess) return check_hip(le, "cu_job_kernel");
        return X265HIP_OK;
    }
    __atomic_store_n(&s->doorbell, seq, __ATOMIC_RELEASE);
    store_fence();                                                                           // ... and the doorbell leaves now, not when its write-combining buffer is evicted
    if (__atomic_load_n(&cs->hostCtl->serverState, __ATOMIC_ACQUIRE) == 0)
        return start_server(cs);
    return X265HIP_OK;
}

int x265hip_cuserve_submit_sao(x265hip_cuserve* cs, int slot, const x265hip_saojob* job, uint32_t* seqOut)
{
    if (!cs || slot < 0 || slot >= cs->slots || !job || !seqOut) return set_error(X265HIP_EINVAL, "x265hip_cuserve_submit_sao: slot %d", slot);
    bool ok = job->bitDepth == 8 && job->planes >= 1 && job->planes <= 3;
    for (uint32_t p = 0; ok && p < job->planes; p++)
    {
        ok = job->plane[p].w >= 1 && job->plane[p].w <= 64 && job->plane[p].h >= 1 && job->plane[p].h <= 64;
        for (int c = 0; ok && c < 5; c++)
            ok = job->plane[p].x1[c] <= job->plane[p].w && job->plane[p].y1[c] <= job->plane[p].h;
    }
    const int bytes = ok ? x265hipi_saojob_pixel_bytes(job) : 0;
    if (!ok || bytes > X265HIP_CUJOB_PIXEL_BYTES)
        return set_error(X265HIP_EINVAL, "x265hip_cuserve_submit_sao: depth %u, %u planes, luma %ux%u", job->bitDepth, job->planes, job->plane[0].w, job->plane[0].h);
    SlotIn* s = cs->in + slot;
    uint32_t run = cs->seq[slot].load(std::memory_order_relaxed) + 1;
    if (run >= 0xfffff0u) run = 1;
    cs->seq[slot].store(run, std::memory_order_relaxed);
    const uint32_t steps = (uint32_t)(128 + bytes + 511) / 512;
    const uint32_t seq = (run << 8) | 3u | (steps << 2);
    *seqOut = seq;
    cs->jobs.fetch_add(1, std::memory_order_relaxed);
    cs->saoJobs.fetch_add(1, std::memory_order_relaxed);
    // what SAO::calcSaoStatsCTU reads and writes per plane: source + reconstruction in, 2 x 5 x 32 int32 out
    cs->bytes.fetch_add((uint64_t)bytes + 2 * 160 * 4 * job->planes, std::memory_order_relaxed);
    memcpy(&s->job, job, sizeof(*job));
    store_fence();
    if (cs->mode == 1)
    {
        std::atomic_thread_fence(std::memory_order_release);
        int cur = -1;
        (void)hipGetDevice(&cur);
        if (cur != cs->device) (void)hipSetDevice(cs->device);
        hipLaunchKernelGGL(cu_job_kernel, dim3(2), dim3(256), 0, cs->jobStreams[slot], cs->inDev + slot, cs->outDev + slot, seq, &cs->ctl->busyTicks[2 * slot], team_off() ? 0u : 1u);
        const hipError_t le = hipGetLastError();
        if (cur >= 0 && cur != cs->device) (void)hipSetDevice(cur);
        if (le != hipSuccess) return check_hip(le, "cu_job_kernel (SAO statistics)");
        return X265HIP_OK;
    }
    __atomic_store_n(&s->doorbell, seq, __ATOMIC_RELEASE);
    store_fence();
    if (__atomic_load_n(&cs->hostCtl->serverState, __ATOMIC_ACQUIRE) == 0)
        return start_server(cs);
    return X265HIP_OK;
}

int x265hip_cuserve_submit_intra(x265hip_cuserve* cs, int slot, const x265hip_intrajob* job, uint32_t* seqOut)
{
    if (!cs || slot < 0 || slot >= cs->slots || !job || !seqOut) return set_error(X265HIP_EINVAL, "x265hip_cuserve_submit_intra: slot %d", slot);
    if ((job->bitDepth != 8 && job->bitDepth != 10 && job->bitDepth != 12) || job->mark != X265HIP_INTRAJOB_MARK || job->log2Size < 3 || job->log2Size > 5)
        return set_error(X265HIP_EINVAL, "x265hip_cuserve_submit_intra: depth %u, mark %#x, log2 size %u", job->bitDepth, job->mark, job->log2Size);
    const int bytes = x265hipi_intrajob_pixel_bytes(job);
    SlotIn* s = cs->in + slot;
    uint32_t run = cs->seq[slot].load(std::memory_order_relaxed) + 1;
    if (run >= 0xfffff0u) run = 1;
    cs->seq[slot].store(run, std::memory_order_relaxed);
    const uint32_t steps = (uint32_t)(128 + bytes + 511) / 512;
    const uint32_t seq = (run << 8) | 3u | (steps << 2);
    *seqOut = seq;
    cs->jobs.fetch_add(1, std::memory_order_relaxed);
    cs->intraJobs.fetch_add(1, std::memory_order_relaxed);
    // what the 35 sa8d calls of Search::checkIntraInInter read: the two lines + the source block in, 35 costs out
    cs->bytes.fetch_add((uint64_t)bytes + 35 * 4, std::memory_order_relaxed);
    memcpy(&s->job, job, sizeof(*job));
    store_fence();
    if (cs->mode == 1)
    {
        std::atomic_thread_fence(std::memory_order_release);
        int cur = -1;
        (void)hipGetDevice(&cur);
        if (cur != cs->device) (void)hipSetDevice(cs->device);
        hipLaunchKernelGGL(cu_job_kernel, dim3(2), dim3(256), 0, cs->jobStreams[slot], cs->inDev + slot, cs->outDev + slot, seq, &cs->ctl->busyTicks[2 * slot], team_off() ? 0u : 1u);
        const hipError_t le = hipGetLastError();
        if (cur >= 0 && cur != cs->device) (void)hipSetDevice(cur);
        if (le != hipSuccess) return check_hip(le, "cu_job_kernel (intra scan)");
        return X265HIP_OK;
    }
    __atomic_store_n(&s->doorbell, seq, __ATOMIC_RELEASE);
    store_fence();
    if (__atomic_load_n(&cs->hostCtl->serverState, __ATOMIC_ACQUIRE) == 0)
        return start_server(cs);
    return X265HIP_OK;
}

int x265hip_cuserve_poke(x265hip_cuserve* cs, int slot)
{
    if (!cs || slot < 0 || slot >= cs->slots) return set_error(X265HIP_EINVAL, "x265hip_cuserve_poke: slot %d", slot);
    if (__atomic_load_n(&cs->out[slot].failed, __ATOMIC_ACQUIRE)) return set_error(X265HIP_EHIP, "cuserve: the device gave up a job of slot %d", slot);
    if (cs->mode == 0)
    {
        // 1: nothing is wrong and nothing can be expected yet — a device synchronisation has the servers paused (device_free), or the server is on its way
        // onto the chip (started now, or "being started": the device clears the flag when the kernel runs).  The caller's timeout does not run meanwhile.
        if (cs->paused.load() > 0) return 1;
        const uint32_t st = __atomic_load_n(&cs->hostCtl->serverState, __ATOMIC_ACQUIRE);
        if (st == 0) { const int e = start_server(cs); return e ? e : 1; }
        if (st & 0x80000000u) return 1;
    }
    return X265HIP_OK;
}

int x265hip_cuserve_stats(x265hip_cuserve* cs, uint64_t* jobs, uint64_t* serverStarts, uint64_t* deviceNs)
{
    if (!cs) return set_error(X265HIP_EINVAL, "x265hip_cuserve_stats: null");
    if (jobs) *jobs = cs->jobs.load();
    if (serverStarts) *serverStarts = cs->starts.load();
    if (deviceNs) *deviceNs = device_ticks(cs) * 10;
    return X265HIP_OK;
}

} // extern "C"
