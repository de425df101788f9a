/* x265_oracle_rqt.c — TEST INFRASTRUCTURE ONLY (see x265_oracle.h): the transform arithmetic of an inter CU's residual quad-tree, i.e. what
 * Search::estimateResidualQT (reference source/encoder/search.cpp:3178-3560) asks of Quant::transformNxN and Quant::invtransformNxN, restated
 * on the CPU from the pinned primitives of x265_oracle.c, and the CU job built from them (include/x265hip.h, x265hip_cujob: the checker of
 * x265_amd/csrc/cuserve.hip and the engine of tests/support/la_emul.c).
 * Pinned: tests/test_oracle_vs_ref.py runs orc_transform_nxn / orc_invtransform_nxn against the real Quant class (oracle/ref_shim.cpp
 * ref_transform_nxn / ref_invtransform_nxn) over sizes, planes, QPs, slice types and sign hiding on / off. */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include "x265_oracle.h"
#include "../include/x265hip.h"

void orc_scan_order(int type, int log2, uint16_t* out);                                                  /* x265_oracle_coef.c */
int orc_scanPosLast(const uint16_t* scan, const int16_t* coeff, uint16_t* coeffSign, uint16_t* coeffFlag, uint8_t* coeffNum, int numSig);

/* common/quant.cpp:246-395 Quant::signBitHidingHDQ: per 4x4 coefficient group of the scan (last to first) whose first and last non-zero
 * levels lie at least SBH_THRESHOLD (4) scan positions apart, the parity of the sum of levels must equal the sign of the first one; where it
 * does not, the level whose change costs least (deltaU: the quantiser's rounding remainder) moves by one.  resiDct: the transform coefficients
 * (Quant::m_resiDctCoeff) — the sign a zero level would take.  Returns the new numSig. */
uint32_t orc_sign_hide_hdq(int16_t* coeff, const int32_t* deltaU, const int16_t* resiDct, uint32_t numSig, int log2TrSize, int scanType)
{
    uint16_t scan[1024];
    uint8_t coeffNum[64];
    uint16_t coeffSign[64], coeffFlag[64];
    orc_scan_order(scanType, log2TrSize, scan);
    const int lastScanPos = orc_scanPosLast(scan, coeff, coeffSign, coeffFlag, coeffNum, (int)numSig);
    const int cgLastScanPos = lastScanPos >> 4;
    const uint32_t correctOffset = 0x0F & (lastScanPos ^ 0xF);              /* :268: the last group was left-aligned only as far as it was walked */
    coeffFlag[cgLastScanPos] = (uint16_t)(coeffFlag[cgLastScanPos] << correctOffset);
    for (int cg = cgLastScanPos; cg >= 0; cg--)
    {
        const int cgStartPos = cg << 4;
        if (!coeffNum[cg])
            continue;
        /* bit (15 - n) of coeffFlag = scan position n of the group holds a non-zero level */
        int firstNZ = 0, lastNZ = 15;
        while (!(coeffFlag[cg] & (0x8000 >> firstNZ))) firstNZ++;
        while (!(coeffFlag[cg] & (0x8000 >> lastNZ))) lastNZ--;
        if (lastNZ - firstNZ < 4)
            continue;
        const uint32_t signbit = coeff[scan[cgStartPos + firstNZ]] > 0 ? 0 : 1;
        uint32_t absSum = 0;
        for (int n = firstNZ; n <= lastNZ; n++)
            absSum += (uint32_t)(int32_t)coeff[scan[n + cgStartPos]];
        if (signbit == (absSum & 1))
            continue;
        int minCostInc = INT32_MAX, minPos = -1, curCost = INT32_MAX;
        int32_t finalChange = 0, curChange = 0;
        uint32_t cgFlags = coeffFlag[cg];
        if (cg == cgLastScanPos)
            cgFlags >>= correctOffset;
        for (int n = (cg == cgLastScanPos ? lastNZ : 15); n >= 0; --n)
        {
            const uint32_t blkPos = scan[n + cgStartPos];
            if (cgFlags & 1)
            {
                if (deltaU[blkPos] > 0) { curCost = -deltaU[blkPos]; curChange = 1; }
                else if (cgFlags == 1 && abs(coeff[blkPos]) == 1) curCost = INT32_MAX;     /* the group's first level may not vanish */
                else { curCost = deltaU[blkPos]; curChange = -1; }
            }
            else if (cgFlags == 0)
            {
                /* before the first non-zero level: a new level here becomes the first one and its sign is the hidden bit */
                const uint32_t thisSignBit = resiDct[blkPos] >= 0 ? 0 : 1;
                if (thisSignBit != signbit) curCost = INT32_MAX;
                else { curCost = -deltaU[blkPos]; curChange = 1; }
            }
            else { curCost = -deltaU[blkPos]; curChange = 1; }
            if (curCost < minCostInc) { minCostInc = curCost; finalChange = curChange; minPos = (int)blkPos; }
            cgFlags >>= 1;
        }
        if (minPos < 0)
            continue;
        if (coeff[minPos] == 32767 || coeff[minPos] == -32768)
            finalChange = -1;
        if (!coeff[minPos]) numSig++;
        else if (finalChange == -1 && abs(coeff[minPos]) == 1) numSig--;
        const int16_t sigMask = (int16_t)(resiDct[minPos] >> 15);
        coeff[minPos] = (int16_t)(coeff[minPos] + (((int16_t)finalChange ^ sigMask) - sigMask));
    }
    return numSig;
}

/* common/quant.cpp:397-470 Quant::transformNxN for an inter unit without transform skip / bypass / noise reduction / RDOQ and with flat
 * quantiser matrices: cu[].dct -> quant (:455-461: qbits = 14 + per + transformShift, add = offset << (qbits - 9)) -> signBitHidingHDQ when
 * numSig >= 2 and the PPS enables it (inter units scan diagonally, cudata.cpp:2088).  resiDct (N * N) receives the transform coefficients. */
uint32_t orc_transform_nxn(const int16_t* residual, intptr_t resiStride, int16_t* coeff, int16_t* resiDct, int log2TrSize, int depth, int rem, int per,
                           int quantScale, int quantOffset, int signHide)
{
    const int n = 1 << (2 * log2TrSize), transformShift = 15 - depth - log2TrSize;
    int32_t deltaU[1024], quantCoeff[1024];
    (void)rem;
    orc_dct(log2TrSize, residual, resiDct, resiStride, depth);
    for (int i = 0; i < n; i++) quantCoeff[i] = quantScale;
    const int qbits = 14 + per + transformShift;
    const int add = quantOffset << (qbits - 9);
    uint32_t numSig = orc_quant(resiDct, quantCoeff, deltaU, coeff, qbits, add, n);
    if (numSig >= 2 && signHide)
        numSig = orc_sign_hide_hdq(coeff, deltaU, resiDct, numSig, log2TrSize, 0);
    return numSig;
}

/* common/quant.cpp:543-603 Quant::invtransformNxN, same restrictions: dequant_normal (scale = invQuantScales[rem] << per, shift = 20 - 14 -
 * transformShift) -> cu[].idct; a lone DC level takes the short cut of :586-596 (blockfill of the value the full transform would give) */
void orc_invtransform_nxn(int16_t* residual, intptr_t resiStride, const int16_t* coeff, int log2TrSize, int depth, int per, int dequantScale, uint32_t numSig)
{
    const int n = 1 << (2 * log2TrSize), size = 1 << log2TrSize, transformShift = 15 - depth - log2TrSize;
    int16_t dq[1024];
    orc_dequant_normal(coeff, dq, n, dequantScale << per, 20 - 14 - transformShift);
    if (numSig == 1 && coeff[0] != 0)
    {
        const int shift_1st = 7 - 6, add_1st = 1 << (shift_1st - 1), shift_2nd = 12 - (depth - 8) - 3, add_2nd = 1 << (shift_2nd - 1);
        const int dc_val = (((dq[0] * (64 >> 6) + add_1st) >> shift_1st) * (64 >> 3) + add_2nd) >> shift_2nd;
        orc_blockfill_s(residual, resiStride, (int16_t)dc_val, size);
        return;
    }
    orc_idct(log2TrSize, dq, residual, resiStride, depth);
}

/* ---- the CU job (include/x265hip.h): every unit of every level; returns the number of units written -------------------------------------------- */
#define ORC_CUJOB(P, SFX) \
int orc_cujob_run_##SFX(const x265hip_cujob* j, const P* pixels, x265hip_cujob_unit* units, int16_t* levels, int16_t* resi, uint32_t seq) \
{ \
    int sHi, sLo, done = 0; \
    const int nl = x265hipi_cujob_levels(j, &sHi, &sLo); \
    const int N = 1 << j->log2CUSize, N2 = N * N, planeElems = j->chroma ? N2 + N2 / 2 : N2, depth = (int)j->bitDepth; \
    const P* src = pixels; \
    const P* prd = pixels + planeElems; \
    if (j->coefMode == X265HIP_CUJOB_INVERSE) \
    { \
        /* the inverse half alone of ONE 32x32 luma unit, for levels made elsewhere (Quant::rdoQuant): Quant::invtransformNxN quant.cpp:543-603, then the \
         * reconstruction's sse_pp and psy energy (search.cpp:3290-3300); the levels follow the two blocks in the pixel block */ \
        const int16_t* lv = (const int16_t*)(pixels + 2 * 1024); \
        int16_t back[1024]; \
        P rec[1024]; \
        uint32_t ns = 0; \
        for (int i = 0; i < 1024; i++) ns += lv[i] != 0; \
        units[0].numSig = ns; \
        units[0].zeroDist = orc_sse_pp_##SFX(src, 32, prd, 32, 32, 32); \
        if (ns) \
        { \
            orc_invtransform_nxn(back, 32, lv, 5, depth, j->qpPer[0], j->dequantScale[0], ns); \
            orc_add_ps_##SFX(rec, 32, prd, back, 32, 32, 32, 32, depth); \
            units[0].codedDist = orc_sse_pp_##SFX(src, 32, rec, 32, 32, 32); \
            units[0].codedEnergy = (uint32_t)orc_psy_cost_pp_##SFX(src, 32, rec, 32, 32); \
            memcpy(resi, back, sizeof(back)); \
        } \
        units[0].fwdTicks = 0; \
        __atomic_store_n(&units[0].readyInv, seq, __ATOMIC_RELEASE); \
        __atomic_store_n(&units[0].ready, seq, __ATOMIC_RELEASE); \
        return 1; \
    } \
    for (int lv = 0; lv < nl; lv++) \
    { \
        const int s = sHi - lv, perRow = 1 << ((int)j->log2CUSize - s); \
        for (int plane = 0; plane < (j->chroma ? 3 : 1); plane++) \
        { \
            const int log2n = plane ? s - 1 : s, n = 1 << log2n, pw = plane ? N / 2 : N; \
            const P* ps = src + (plane == 0 ? 0 : plane == 1 ? N2 : N2 + N2 / 4); \
            const P* pp = prd + (plane == 0 ? 0 : plane == 1 ? N2 : N2 + N2 / 4); \
            for (int ty = 0; ty < perRow; ty++) \
                for (int tx = 0; tx < perRow; tx++) \
                { \
                    x265hip_cujob_unit* u = units + x265hipi_cujob_unit_index(j, sHi, s, plane, tx, ty); \
                    const int eo = x265hipi_cujob_elem_offset(j, sHi, s, plane, tx, ty); \
                    const P* f = ps + (ty * n) * pw + tx * n; \
                    const P* p = pp + (ty * n) * pw + tx * n; \
                    int16_t r[1024], dct[1024], back[1024]; \
                    P rec[1024]; \
                    orc_sub_ps_##SFX(r, n, f, p, pw, pw, n, n); \
                    if (j->coefMode) \
                    { \
                        /* the host quantises (Quant::rdoQuant): what Quant::transformNxN has computed by quant.cpp:432 (m_resiDctCoeff) and :436-442 \
                         * (m_fencDctCoeff: copy_ps of the source block, then the same cu[].dct) */ \
                        orc_dct(log2n, r, levels + eo, n, depth); \
                        if (j->sourceDct && plane == 0) \
                        { \
                            for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) r[y * n + x] = (int16_t)f[y * pw + x]; \
                            orc_dct(log2n, r, resi + eo, n, depth); \
                        } \
                        u->numSig = 0; \
                        u->zeroDist = orc_sse_pp_##SFX(f, pw, p, pw, n, n); \
                        u->fwdTicks = 0; \
                        __atomic_store_n(&u->readyInv, seq, __ATOMIC_RELEASE); \
                        __atomic_store_n(&u->ready, seq, __ATOMIC_RELEASE); \
                        done++; \
                        continue; \
                    } \
                    u->numSig = orc_transform_nxn(r, n, levels + eo, dct, log2n, depth, j->qpRem[plane], j->qpPer[plane], j->quantScale[plane], (int)j->quantOffset, \
                                                  (int)j->signHide); \
                    u->zeroDist = orc_sse_pp_##SFX(f, pw, p, pw, n, n); \
                    if (u->numSig) \
                    { \
                        orc_invtransform_nxn(back, n, levels + eo, log2n, depth, j->qpPer[plane], j->dequantScale[plane], u->numSig); \
                        orc_add_ps_##SFX(rec, n, p, back, pw, n, n, n, depth); \
                        u->codedDist = orc_sse_pp_##SFX(f, pw, rec, n, n, n); \
                        u->codedEnergy = (uint32_t)orc_psy_cost_pp_##SFX(f, pw, rec, n, n); \
                        memcpy(resi + eo, back, sizeof(int16_t) * n * n); \
                    } \
                    else \
                    { \
                        u->codedDist = u->zeroDist; \
                        u->codedEnergy = 0; \
                        memset(resi + eo, 0, sizeof(int16_t) * n * n); \
                    } \
                    u->fwdTicks = 0; \
                    __atomic_store_n(&u->readyInv, seq, __ATOMIC_RELEASE); \
                    __atomic_store_n(&u->ready, seq, __ATOMIC_RELEASE); \
                    done++; \
                } \
        } \
    } \
    return done; \
}
ORC_CUJOB(uint8_t, 8)
ORC_CUJOB(uint16_t, 16)


/* ---- the SAO statistics job (include/x265hip.h, x265hip_saojob): SAO::calcSaoStatsCTU (encoder/sao.cpp:735-917) for every plane of the job, the
 * primitives called in the reference's order with the reference's arguments, on the job's blocks.  diff = source - reconstruction (pitch 64, :786-806). */
void orc_saoSign_8(int8_t* dst, const uint8_t* src1, const uint8_t* src2, int endX);
void orc_saoCuStatsBO_8(const int16_t* diff, const uint8_t* rec, intptr_t stride, int endX, int endY, int32_t* stats, int32_t* count, int depth);
void orc_saoCuStatsE0_8(const int16_t* diff, const uint8_t* rec, intptr_t stride, int endX, int endY, int32_t* stats, int32_t* count);
void orc_saoCuStatsE1_8(const int16_t* diff, const uint8_t* rec, intptr_t stride, int8_t* upBuff1, int endX, int endY, int32_t* stats, int32_t* count);
void orc_saoCuStatsE2_8(const int16_t* diff, const uint8_t* rec, intptr_t stride, int8_t* upBuff1, int8_t* upBufft, int endX, int endY, int32_t* stats, int32_t* count);
void orc_saoCuStatsE3_8(const int16_t* diff, const uint8_t* rec, intptr_t stride, int8_t* upBuff1, int endX, int endY, int32_t* stats, int32_t* count);

int orc_saojob_run_8(const x265hip_saojob* j, const uint8_t* pixels, x265hip_cujob_unit* units, int32_t* out, uint32_t seq)
{
    int32_t* stats = out;
    int32_t* count = out + X265HIP_SAOJOB_STATS_ENTRIES;
    memset(out, 0, sizeof(int32_t) * 2 * X265HIP_SAOJOB_STATS_ENTRIES);
    const uint8_t* at = pixels;
    for (uint32_t p = 0; p < j->planes && p < 3; p++)
    {
        const int w = j->plane[p].w, h = j->plane[p].h, stride = w + 1;
        const uint8_t* rec0 = at + stride + 1;                 /* the CTU's first sample */
        const uint8_t* fenc0 = at + (w + 1) * (h + 1);
        at = fenc0 + w * h;
        int16_t diff[64 * 64];
        int8_t _up[2 * (64 + 16 + 16)], *upBuff1 = _up + 16, *upBufft = upBuff1 + (64 + 16 + 16);
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++)
                diff[y * 64 + x] = (int16_t)(fenc0[y * w + x] - rec0[y * stride + x]);
#define R(c, f) (j->plane[p].f[c])
        int32_t* st = stats + p * 5 * 32;
        int32_t* ct = count + p * 5 * 32;
        /* SAO_BO :810-823 */
        orc_saoCuStatsBO_8(diff, rec0, stride, R(0, x1), R(0, y1), st, ct, 8);
        /* SAO_EO_0 :826-839 */
        orc_saoCuStatsE0_8(diff + R(1, x0), rec0 + R(1, x0), stride, R(1, x1) - R(1, x0), R(1, y1), st + 32, ct + 32);
        /* SAO_EO_1 :841-861: sign of the first measured row against the row above it, then the rows */
        {
            const uint8_t* rec = rec0 + R(2, y0) * stride;
            orc_saoSign_8(upBuff1, rec, rec - stride, w);
            orc_saoCuStatsE1_8(diff + R(2, y0) * 64, rec0 + R(2, y0) * stride, stride, upBuff1, R(2, x1), R(2, y1) - R(2, y0), st + 64, ct + 64);
        }
        if (j->eo23)
        {
            /* SAO_EO_2 :865-888 */
            {
                const uint8_t* rec = rec0 + R(3, y0) * stride;
                orc_saoSign_8(upBuff1, rec + R(3, x0), rec + R(3, x0) - stride - 1, R(3, x1) - R(3, x0));
                orc_saoCuStatsE2_8(diff + R(3, x0) + R(3, y0) * 64, rec0 + R(3, x0) + R(3, y0) * stride, stride, upBuff1, upBufft, R(3, x1) - R(3, x0), R(3, y1) - R(3, y0),
                                   st + 96, ct + 96);
            }
            /* SAO_EO_3 :889-914 */
            {
                const uint8_t* rec = rec0 + R(4, y0) * stride;
                orc_saoSign_8(upBuff1, rec + R(4, x0) - 1, rec + R(4, x0) - 1 - stride + 1, R(4, x1) - R(4, x0) + 1);
                orc_saoCuStatsE3_8(diff + R(4, x0) + R(4, y0) * 64, rec0 + R(4, x0) + R(4, y0) * stride, stride, upBuff1 + 1, R(4, x1) - R(4, x0), R(4, y1) - R(4, y0),
                                   st + 128, ct + 128);
            }
        }
#undef R
        __atomic_store_n(&units[p].readyInv, seq, __ATOMIC_RELEASE);
        __atomic_store_n(&units[p].ready, seq, __ATOMIC_RELEASE);
    }
    return (int)j->planes;
}

/* ---- intra mode scan jobs (include/x265hip.h, x265hip_intrajob): what Search::checkIntraInInter (search.cpp:1291-1452) measures for one block —
 * for each of the 35 modes the prediction the reference would make (DC with edge smoothing when N <= 16, :1356; planar from the filtered line when
 * N >= 8, :1363-1367; every angle from the line g_intraFilterFlags picks, with the mode 10 / 26 edge gradient when N <= 16, :1376 / :1388) and
 * cu[].sa8d against the source block (:1357, :1368, :1383-1387).  Composed from the pinned primitives of x265_oracle_intra.c and x265_oracle.c. */
int orc_intrajob_run(const x265hip_intrajob* j, const void* pixels, x265hip_cujob_unit* units, int32_t* out, uint32_t seq)
{
    const int n = 1 << j->log2Size, line = x265hipi_intrajob_line_samples((int)j->log2Size), bFilter = n <= 16;
    if (j->bitDepth == 8)
    {
        const uint8_t* raw = (const uint8_t*)pixels;
        const uint8_t* flt = raw + line;
        const uint8_t* fenc = raw + 2 * line;
        uint8_t pred[32 * 32];
        for (int mode = 0; mode < 35; mode++)
        {
            orc_intra_pred_8(n, mode, pred, n, orc_intra_uses_filtered_8(n, mode) ? flt : raw, bFilter, 8);
            out[mode] = orc_sa8d_8(fenc, n, pred, n, n);
        }
    }
    else
    {
        const uint16_t* raw = (const uint16_t*)pixels;
        const uint16_t* flt = raw + line;
        const uint16_t* fenc = raw + 2 * line;
        uint16_t pred[32 * 32];
        for (int mode = 0; mode < 35; mode++)
        {
            orc_intra_pred_16(n, mode, pred, n, orc_intra_uses_filtered_16(n, mode) ? flt : raw, bFilter, (int)j->bitDepth);
            out[mode] = orc_sa8d_16(fenc, n, pred, n, n);
        }
    }
    units[0].readyInv = seq;
    units[0].ready = seq;
    return 35;
}
