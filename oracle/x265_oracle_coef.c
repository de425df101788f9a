/* x265_oracle_coef.c — TEST INFRASTRUCTURE ONLY (see x265_oracle.c).  CPU restatement of x265's coefficient-scan cost primitives
 * (reference source/common/dct.cpp:757-984: scanPosLast_c, findPosFirstLast_c, costCoeffNxN_c, costCoeffRemain_c, costC1C2Flag_c) and of the
 * HEVC scan orders they walk (constants.cpp:364-461 g_scan4x4 / g_scan8x8 / g_scan16x16 / g_scan32x32, generated here by the rule of the
 * standard, 6.5.3-6.5.5).  Pinned: tests/test_oracle_vs_ref.py runs every function against the real reference build (oracle/_ref) and
 * tests/golden holds digests.  The CABAC cost table (x265_entropyStateBits, constants.cpp) is data of the reference, not an algorithm:
 * it is passed in by the caller; the tests take it from tests/golden/entropy_state_bits.json, dumped from the reference by make_golden.py. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- scan orders.  type 0 = up-right diagonal, 1 = horizontal, 2 = vertical (common.h:404-407); value = y * size + x ---- */
static void scan_positions(int type, int n, int* xs, int* ys)
{
    int i = 0;
    if (type == 0)
    {
        /* 6.5.3: walk the anti-diagonals from bottom-left to top-right */
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

/* the scan of a (1 << log2) square TU: 4x4 coefficient groups visited in `type` order (the CG grid of 16x16 / 32x32 TUs always
 * diagonally: MDCS applies to 4x4 and 8x8 only, common.h:316), the sixteen positions inside a group in `type` order too */
void orc_scan_order(int type, int log2, uint16_t* out)
{
    const int size = 1 << log2, cgs = size >> 2;
    int cx[64], cy[64], px[16], py[16];
    if (log2 > 3)
        type = 0;
    scan_positions(type, 4, px, py);
    if (cgs == 1) { cx[0] = cy[0] = 0; }
    else scan_positions(type, cgs, cx, cy);
    for (int g = 0; g < cgs * cgs; g++)
        for (int k = 0; k < 16; k++)
            out[g * 16 + k] = (uint16_t)((cy[g] * 4 + py[k]) * size + cx[g] * 4 + px[k]);
}

/* dct.cpp:757-792 scanPosLast_c: walk the scan until numSig non-zero coefficients were seen; per coefficient group the count, the
 * significance bits (first-scanned coefficient in the highest used bit) and the sign bits (bit k = sign of the k-th non-zero one) */
int orc_scanPosLast(const uint16_t* scan, const int16_t* coeff, uint16_t* coeffSign, uint16_t* coeffFlag, uint8_t* coeffNum, int numSig)
{
    memset(coeffNum, 0, 64 * sizeof(*coeffNum));
    memset(coeffFlag, 0, 64 * sizeof(*coeffFlag));
    memset(coeffSign, 0, 64 * sizeof(*coeffSign));
    int pos = 0;
    do
    {
        const unsigned cg = (unsigned)pos >> 4;
        const int c = coeff[scan[pos++]];
        const unsigned nz = c != 0;
        numSig -= (int)nz;
        coeffSign[cg] = (uint16_t)(coeffSign[cg] + (uint16_t)((((uint32_t)c) >> 31) << coeffNum[cg]));
        coeffFlag[cg] = (uint16_t)((coeffFlag[cg] << 1) + nz);
        coeffNum[cg] = (uint8_t)(coeffNum[cg] + nz);
    }
    while (numSig > 0);
    return pos - 1;
}

/* dct.cpp:795-838 findPosFirstLast_c: first / last non-zero scan position inside one coefficient group and the low bit of the sum between */
uint32_t orc_findPosFirstLast(const int16_t* dstCoeff, intptr_t trSize, const uint16_t* scanTbl)
{
    int n;
    for (n = 15; n >= 0; n--)
        if (dstCoeff[(scanTbl[n] >> 2) * trSize + (scanTbl[n] & 3)])
            break;
    const uint32_t last = (uint32_t)n;
    for (n = 0; n < 16; n++)
        if (dstCoeff[(scanTbl[n] >> 2) * trSize + (scanTbl[n] & 3)])
            break;
    const uint32_t first = (uint32_t)n;
    uint32_t sum = 0;
    for (n = (int)first; n <= (int)last; n++)
        sum += (uint32_t)(int32_t)dstCoeff[(scanTbl[n] >> 2) * trSize + (scanTbl[n] & 3)];
    return (sum << 31) | (last << 8) | first;
}

/* one CABAC bin against context byte *ctx: bits from the state table, state advanced as dct.cpp:884-890 spells it
 * (== sbacNext / sbacGetEntropyBits of contexts.h, as its X265_CHECKs state) */
static uint32_t bin_cost(const uint32_t* stateBits, uint8_t* ctx, uint32_t bin)
{
    const uint32_t mstate = *ctx, mps = mstate & 1;
    const uint32_t sb = stateBits[mstate ^ bin];
    uint32_t next = (sb >> 24) + mps;
    if ((mstate ^ bin) == 1)
        next = bin;
    *ctx = (uint8_t)next;
    return sb;
}

/* dct.cpp:841-899 costCoeffNxN_c: the significance flags of one coefficient group from scanPosSigOff down to 0 */
uint32_t orc_costCoeffNxN(const uint16_t* scan, const int16_t* coeff, intptr_t trSize, uint16_t* absCoeff, const uint8_t* tabSigCtx,
                          uint32_t scanFlagMask, uint8_t* baseCtx, int offset, int scanPosSigOff, int subPosBase, const uint32_t* stateBits)
{
    uint16_t tmp[16];
    uint32_t numNonZero = scanPosSigOff < 15 ? 1 : 0, sum = 0;
    absCoeff -= numNonZero;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            tmp[i * 4 + j] = (uint16_t)abs(coeff[i * trSize + j]);
    do
    {
        const uint32_t blkPos = scan[scanPosSigOff];
        const uint32_t posZeroMask = (subPosBase + scanPosSigOff) ? ~0u : 0u;
        const uint32_t sig = scanFlagMask & 1;
        scanFlagMask >>= 1;
        if (scanPosSigOff != 0 || subPosBase == 0 || numNonZero)
        {
            const uint32_t ctxSig = (uint32_t)(tabSigCtx[blkPos] + offset) & posZeroMask;
            sum += bin_cost(stateBits, &baseCtx[ctxSig], sig);
        }
        absCoeff[numNonZero] = tmp[blkPos];
        numNonZero += sig;
        scanPosSigOff--;
    }
    while (scanPosSigOff >= 0);
    return sum & 0xFFFFFF;
}

/* dct.cpp:901-946 costCoeffRemain_c: Golomb-Rice escape lengths of the levels from idx on */
uint32_t orc_costCoeffRemain(const uint16_t* absCoeff, int numNonZero, int idx)
{
    uint32_t rice = 0, sum = 0;
    int baseLevel = 3;
    do
    {
        if (idx >= 8)
            baseLevel = 1;
        int code = absCoeff[idx] - baseLevel;
        if (code >= 0)
        {
            uint32_t length = 0;
            code = (int)((uint32_t)code >> rice) - 3;
            if (code >= 0)
            {
                uint32_t v = (uint32_t)code + 1;
                while (v >>= 1) length++;                    /* CLZ(cidx, codeNumber + 1): index of the highest set bit */
                code = (int)(length + length);
            }
            sum += 3 + 1 + rice + (uint32_t)code;
            if (absCoeff[idx] > (3u << rice))
                rice = (rice + 1) - (rice >> 2);
        }
        baseLevel = 2;
        idx++;
    }
    while (idx < numNonZero);
    return sum;
}

/* dct.cpp:949-1006 costC1C2Flag_c: greater-than-1 flags of up to eight levels and the first greater-than-2 flag */
uint32_t orc_costC1C2Flag(const uint16_t* absCoeff, intptr_t numC1Flag, uint8_t* baseCtxMod, intptr_t ctxOffset, const uint32_t* stateBits)
{
    uint32_t sum = 0, c1 = 1, firstC2Idx = 8, firstC2Flag = 2, c1Next = 0xFFFFFFFEu;
    int idx = 0;
    do
    {
        const uint32_t s1 = absCoeff[idx] > 1, s2 = absCoeff[idx] > 2;
        sum += bin_cost(stateBits, &baseCtxMod[c1], s1) & 0xFFFFFF;
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
    while (idx < numC1Flag);
    if (!c1)
        sum += bin_cost(stateBits, &baseCtxMod[ctxOffset], firstC2Flag) & 0xFFFFFF;
    return (sum & 0x00FFFFFF) + (c1 << 26) + (firstC2Idx << 28);
}
