/* x265hip.h — C ABI of libx265hip.so: MI355X (gfx950) implementations of x265's data-parallel encode primitives.
 *
 * This is the drop-in boundary.  Every entry point is the BATCHED form of one (family of) slot(s) of x265's
 * `EncoderPrimitives` function-pointer table (reference: source/common/primitives.h:237-429); the typedef each
 * one replaces is cited next to it.  A slot call `p.pu[part].sad(fenc, s0, ref, s1)` becomes one job
 * `{offA, offB}` of `x265hip_pixcmp_batch(X265HIP_CMP_SAD, depth, w, h, planeA, s0, planeB, s1, offA[], offB[], n, out[])`.
 * The per-call shims that fill the reference's table (setupAssemblyPrimitives, primitives.h:470) live in
 * x265_amd/host/x265_hip_primitives.cpp and only use the `x265hip_call_*` entry points at the end of this file.
 *
 * Conventions
 *  - plain C: pointers + sizes, no C++ / torch types.  Every function returns 0 on success or a negative
 *    X265HIP_E* code; x265hip_last_error() gives the text.  Nothing here falls back to a CPU implementation:
 *    without a usable GPU every call fails with X265HIP_ENODEV.
 *  - `depth` is x265's internal bit depth (X265_DEPTH): 8 -> pixel = uint8_t; 10 or 12 -> pixel = uint16_t
 *    (reference: source/common/common.h:126-142).
 *  - planes, job arrays and outputs are DEVICE pointers unless the name says `host`.  Strides and offsets are in
 *    ELEMENTS (as in x265), offsets are relative to the plane pointer and may be negative (picture margins).
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls are asynchronous on it.
 *  - block sizes (w, h) must be one of x265's 25 luma PU shapes or their 4:2:0 chroma halves
 *    (primitives.h:41-55, :80-90); transform sizes are 4, 8, 16, 32.
 */
#ifndef X265HIP_H
#define X265HIP_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define X265HIP_OK        0
#define X265HIP_EINVAL   -1   /* bad size / depth / argument */
#define X265HIP_ENODEV   -2   /* no usable HIP device */
#define X265HIP_EHIP     -3   /* a HIP runtime call failed (see x265hip_last_error) */
#define X265HIP_ENOMEM   -4

/* ---------------------------------------------------------------- runtime ---------------------------------- */
int  x265hip_init(int device);                 /* select the device for the calling thread; 0 on success */
int  x265hip_device_count(void);
const char* x265hip_last_error(void);          /* thread-local text of the last failure */
const char* x265hip_version(void);
int  x265hip_malloc(void** dptr, size_t bytes);
int  x265hip_free(void* dptr);
int  x265hip_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream);
int  x265hip_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream);
int  x265hip_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream);
int  x265hip_memset(void* dst, int value, size_t bytes, void* stream);
int  x265hip_stream_create(void** stream);
int  x265hip_stream_destroy(void* stream);
int  x265hip_stream_sync(void* stream);
/* timing helper used by bench.py: HIP events on `stream` (torch.cuda.Event only sees torch's own stream) */
int  x265hip_event_create(void** ev);
int  x265hip_event_destroy(void* ev);
int  x265hip_event_record(void* ev, void* stream);
int  x265hip_event_elapsed_ms(void* start, void* stop, float* ms);   /* synchronises on `stop` */

/* Several GPUs in one encoder (x265's frame-thread model, SURVEY.md §8e: frames <-> GPUs).  A PLACE is where a picture's device copy lives: place p
 * is on HIP device devices[p] (several places may share a device — that is how the exchange is exercised on a one-GPU box).  Reference-picture
 * mirrors and source pictures are created at a place (x265hip_refpic_create_at, x265hip_srcpic_create_at); a SAD surface is built at its SOURCE
 * picture's place, and when the reference picture's mirror lives elsewhere the library keeps a replica of the reconstructed picture at the source's
 * place: every band of rows the owner uploads is pushed device to device (hipMemcpyPeerAsync — xGMI between the GPUs of a node; no host round trip) —
 * the reconstructed-reference exchange of frame-parallel encoding (reference frameencoder.cpp:848-861 waits on the rows, framefilter.cpp:654-664
 * publishes them).  Objects created without a place (x265hip_refpic_create, x265hip_srcpic_create) live on the calling thread's device. */
int  x265hip_places(int n, const int* devices);                        /* places can be added later, not moved */
int  x265hip_peer_stats(uint64_t* replicas, uint64_t* bands, uint64_t* bytes);   /* per process: replicas created, bands pushed, bytes pushed */

/* Device-time ledger of the modules the bound encoder uses (lookahead session, reference-picture mirrors, SAD surfaces, source energy planes): every
 * launch group of those modules runs between two HIP events of its own stream; after the stream synchronisation that the module performs anyway the
 * elapsed time is added to its clock.  Per process.  bench.py derives the encode's device duty cycle and the live launch durations of the
 * roofline block from it (x265_amd/host prints the clocks at exit under X265HIP_VERBOSE). */
#define X265HIP_CLK_LA_SEARCH  0   /* lookahead_p_kernel launches of the session (x265hip_la_estimate_batch*)                         */
#define X265HIP_CLK_LA_OTHER   1   /* the session's other kernels: P cost, bidir, scatter, weight analysis, lowres planes + intra      */
#define X265HIP_CLK_PLANES     2   /* sub-pel plane bands of the reference-picture mirrors                                             */
#define X265HIP_CLK_SADSURF    3   /* search-window kernel of the SAD surfaces                                                         */
#define X265HIP_CLK_ENERGY     4   /* source energy planes                                                                             */
#define X265HIP_CLK_CUSERVE    5   /* CU residual quad-tree jobs (x265hip_cuserve_*)                                                   */
#define X265HIP_CLK_SUBPEL     6   /* sub-pel SATD tables of the SAD surfaces                                                          */
#define X265HIP_CLK_COUNT      7
/* algorithmicBytes: SURVEY.md §8d bytes of the work inside the spans where the module can state them itself — SAD surfaces: per block of a built CTU
 * the exhaustive search's unique footprint W H B + (W + R - 1)(H + R - 1) B + 4 R^2 (R = 2 searchRange); plane bands: rows x (padded width) x
 * (1 picture + 15 phase planes) x B; 0 for the other clocks (bench.py prices the lookahead searches per 8x8 block from its own count). */
int  x265hip_device_time(int clock, uint64_t* spans, uint64_t* nanoseconds, uint64_t* algorithmicBytes);

/* ---------------------------------------------------------------- pixel comparisons ------------------------- */
/* pixelcmp_t  (primitives.h:133): pu[].sad, pu[].satd, cu[].sa8d, chroma sa8d; out[i] = cmp(A+offA[i], B+offB[i]) */
#define X265HIP_CMP_SAD    0   /* pixel.cpp:40   sad<lx,ly>                                  */
#define X265HIP_CMP_SATD   1   /* pixel.cpp:210-297 satd_4x4 / satd_8x4 tilers               */
#define X265HIP_CMP_SA8D   2   /* pixel.cpp:336-376 cu[].sa8d: square 4..64, one rounding per 16x16 */
#define X265HIP_CMP_SA8D8  3   /* pixel.cpp:352 sa8d8<w,h>: every 8x8 rounded (chroma tables) */
#define X265HIP_CMP_PSY    4   /* pixelcmp_t-shaped psy_cost_pp (primitives.h:224, pixel.cpp:726), square 4..64 */
int x265hip_pixcmp_batch(int op, int depth, int w, int h,
                         const void* planeA, int64_t strideA, const void* planeB, int64_t strideB,
                         const int32_t* offA, const int32_t* offB, int n, int32_t* out, void* stream);
/* pixelcmp_x3_t / pixelcmp_x4_t (primitives.h:139-140, pixel.cpp:74-119): K = 3 or 4 reference candidates per
 * fenc block; offRef is [n][K], out is [n][K].  (x265 fixes the fenc stride at FENC_STRIDE = 64; here it is free.) */
int x265hip_sad_xn_batch(int K, int depth, int w, int h,
                         const void* fenc, int64_t strideF, const void* ref, int64_t strideR,
                         const int32_t* offF, const int32_t* offRef, int n, int32_t* out, void* stream);
/* pixel_sse_t (primitives.h:135, pixel.cpp:167): out is sse_t widened to u64 */
int x265hip_sse_pp_batch(int depth, int w, int h, const void* planeA, int64_t strideA, const void* planeB, int64_t strideB,
                         const int32_t* offA, const int32_t* offB, int n, uint64_t* out, void* stream);
/* pixel_sse_ss_t (primitives.h:136) on int16 planes; planeB == NULL gives pixel_ssd_s_t (primitives.h:137, pixel.cpp:379) */
int x265hip_sse_ss_batch(int w, int h, const int16_t* planeA, int64_t strideA, const int16_t* planeB, int64_t strideB,
                         const int32_t* offA, const int32_t* offB, int n, uint64_t* out, void* stream);

/* ---------------------------------------------------------------- block arithmetic -------------------------- */
/* pixel_sub_ps_t (primitives.h:147, pixel.cpp:815): resi = a - b */
int x265hip_sub_ps_batch(int depth, int w, int h, int16_t* resi, int64_t strideD, const void* planeA, int64_t strideA,
                         const void* planeB, int64_t strideB, const int32_t* offD, const int32_t* offA, const int32_t* offB,
                         int n, void* stream);
/* pixel_add_ps_t (primitives.h:148, pixel.cpp:829): recon = clip(pred + resi) */
int x265hip_add_ps_batch(int depth, int w, int h, void* recon, int64_t strideD, const void* pred, int64_t strideA,
                         const int16_t* resi, int64_t strideR, const int32_t* offD, const int32_t* offA, const int32_t* offR,
                         int n, void* stream);
/* addAvg_t (primitives.h:173, pixel.cpp:842) */
int x265hip_addavg_batch(int depth, int w, int h, const int16_t* src0, int64_t stride0, const int16_t* src1, int64_t stride1,
                         void* dst, int64_t strideD, const int32_t* off0, const int32_t* off1, const int32_t* offD,
                         int n, void* stream);
/* pixelavg_pp_t (primitives.h:149, pixel.cpp:545) */
int x265hip_pixelavg_pp_batch(int depth, int w, int h, void* dst, int64_t strideD, const void* src0, int64_t stride0,
                              const void* src1, int64_t stride1, const int32_t* offD, const int32_t* off0, const int32_t* off1,
                              int n, void* stream);
/* copy_pp_t / copy_sp_t / copy_ps_t / copy_ss_t (primitives.h:143-146, pixel.cpp:759-812): kind = "pp","sp","ps","ss" as 0..3 */
int x265hip_copy_batch(int kind, int depth, int w, int h, void* dst, int64_t strideD, const void* src, int64_t strideS,
                       const int32_t* offD, const int32_t* offS, int n, void* stream);
/* filter_p2s_t (primitives.h:185, ipfilter.cpp:40) */
int x265hip_p2s_batch(int depth, int w, int h, const void* src, int64_t strideS, int16_t* dst, int64_t strideD,
                      const int32_t* offS, const int32_t* offD, int n, void* stream);

/* ---------------------------------------------------------------- transforms -------------------------------- */
/* dct_t (primitives.h:153; dct.cpp:442-525): src blocks of `size` x `size` int16 at src+offS[i] with row stride
 * strideS -> dst + i*size*size (contiguous).  dst4 != 0 selects the 4x4 DST (cu[].dct vs dst4x4 slot). */
int x265hip_dct_batch(int size, int dst4, int depth, const int16_t* src, int64_t strideS, const int32_t* offS,
                      int16_t* dst, int n, void* stream);
/* idct_t (primitives.h:154; dct.cpp:527-610): src + i*size*size (contiguous) -> dst+offD[i] with row stride strideD */
int x265hip_idct_batch(int size, int dst4, int depth, const int16_t* src, int16_t* dst, int64_t strideD,
                       const int32_t* offD, int n, void* stream);
/* quant_t (primitives.h:159; dct.cpp:664): n TUs of numCoeff coefficients, all contiguous.  quantCoeff is
 * numCoeff entries shared by every TU (x265: ScalingList::m_quantCoef[size][list][rem]).  deltaU may be NULL.
 * numSig[i] receives the return value of the reference call. */
int x265hip_quant_batch(const int16_t* coef, const int32_t* quantCoeff, int32_t* deltaU, int16_t* qCoef,
                        int qBits, int add, int numCoeff, int n, uint32_t* numSig, void* stream);
/* nquant_t (primitives.h:160; dct.cpp:688) */
int x265hip_nquant_batch(const int16_t* coef, const int32_t* quantCoeff, int16_t* qCoef,
                         int qBits, int add, int numCoeff, int n, uint32_t* numSig, void* stream);
/* dequant_normal_t (primitives.h:162; dct.cpp:612): elementwise over `num` coefficients (any number of TUs) */
int x265hip_dequant_normal(const int16_t* quantCoef, int16_t* coef, int64_t num, int scale, int shift, void* stream);
/* dequant_scaling_t (primitives.h:161; dct.cpp:636): deQuantCoef has numCoeff entries shared by the n TUs */
int x265hip_dequant_scaling_batch(const int16_t* quantCoef, const int32_t* deQuantCoef, int16_t* coef,
                                  int numCoeff, int n, int per, int shift, void* stream);
/* count_nonzero_t (primitives.h:163; dct.cpp:714) per TU */
int x265hip_count_nonzero_batch(const int16_t* qCoef, int numCoeff, int n, uint32_t* out, void* stream);

/* cpy2Dto1D_shl_t / cpy2Dto1D_shr_t / cpy1Dto2D_shl_t / cpy1Dto2D_shr_t (primitives.h:147-150; pixel.cpp:401-467): kind 0..3 in
 * that order.  The 2-D side is a block at plane + off[i] with row stride `stride`, the 1-D side is dense [i][size*size]. */
int x265hip_cpy_shift_batch(int kind, int size, int16_t* dst, const int16_t* src, int64_t stride, const int32_t* off, int shift, int n, void* stream);
/* copy_cnt_t (primitives.h:151; dct.cpp:728): coeff[i] = residual block i (dense), numSig[i] = its non-zero count */
int x265hip_copy_cnt_batch(int size, int16_t* coeff, const int16_t* resi, int64_t stride, const int32_t* off, int n, uint32_t* numSig, void* stream);
/* blockfill_s_t (primitives.h:141; pixel.cpp:393): block i at dst + off[i] filled with val[i] */
int x265hip_blockfill_s_batch(int size, int16_t* dst, int64_t stride, const int32_t* off, const int16_t* val, int n, void* stream);
/* denoiseDct_t (primitives.h:155; dct.cpp:743): in place on dctCoef [n][numCoeff]; resSum / offset [numCoeff] are shared */
int x265hip_denoise_dct_batch(int16_t* dctCoef, uint32_t* resSum, const uint16_t* offset, int numCoeff, int n, void* stream);
/* nonPsyRdoQuant_t / psyRdoQuant_t / psyRdoQuant_t1 / psyRdoQuant_t2 (primitives.h:229-232; dct.cpp:985-1069): kind 0..3.  Job i is
 * the 4x4 coefficient group at blkPos[i] of TU tu[i] (resiDct / fencDct / costUncoded are dense [numTU][size*size]); cgUncoded[i]
 * and cgRd[i] are what the reference adds to *totalUncodedCost and *totalRdCost for that group. */
int x265hip_rdoq_cost_batch(int kind, int size, int depth, const int16_t* resiDct, const int16_t* fencDct, const int64_t* psyScale,
                            const int32_t* tu, const int32_t* blkPos, int n, int64_t* costUncoded, int64_t* cgUncoded, int64_t* cgRd, void* stream);

/* ---- coefficient-scan cost primitives of the RDOQ / bit-estimation loop (dct.cpp:757-1006; callers quant.cpp:610-1420, entropy.cpp
 * codeCoeffNxN).  scanType: SCAN_DIAG 0, SCAN_HOR 1, SCAN_VER 2 (common.h:404-407; 16x16 and 32x32 units are always scanned diagonally).
 * The scan orders (constants.cpp g_scanOrder / g_scan4x4) are built in the library by the rule of the standard.  The CABAC cost table is
 * the encoder's data: upload x265_entropyStateBits[128] (constants.cpp) once before the two cost calls that read it (host pointer). */
int x265hip_set_entropy_state_bits(const uint32_t* bits128);
/* scanPosLast_t (primitives.h:217; dct.cpp:757): TU i = coeff[i << (2 log2TrSize) ...] (dense), every TU must hold at least one non-zero
 * coefficient for lastPos to mean what the reference's does (an all-zero TU gives 0); numSig of the reference is implied by the data.
 * Outputs per TU: coeffSign / coeffFlag / coeffNum [64] (MLS_GRP_NUM) and the last significant scan position. */
int x265hip_scan_pos_last_batch(int log2TrSize, int scanType, const int16_t* coeff, int n, uint16_t* coeffSign, uint16_t* coeffFlag,
                                uint8_t* coeffNum, int32_t* lastPos, void* stream);
/* findPosFirstLast_t (primitives.h:218; dct.cpp:795): job i is the 4x4 group at coeff + cgOffsets[i] of a unit with row pitch trSize;
 * out = (absSumSign << 31) | (lastNZPosInCG << 8) | firstNZPosInCG.  Undefined for an all-zero group, as in the reference. */
int x265hip_find_pos_first_last_batch(const int16_t* coeff, const int64_t* cgOffsets, int64_t trSize, int scanType, int n, uint32_t* out,
                                      void* stream);
/* costCoeffNxN_t (primitives.h:220; dct.cpp:841): the significance flags of one 4x4 group from scanPosSigOff down to 0.  Job fields are
 * the reference's arguments (scanType selects its `scan` = g_scan4x4[type]); baseCtx + i * ctxStride is job i's private copy of the
 * significance contexts (read and advanced), absCoeff + 16 i receives the levels as the reference's absCoeff does, bits[i] the sum. */
typedef struct x265hip_coeff_group_job
{
    int64_t coeffOffset;         /* the group's top-left coefficient, in elements from `coeff` */
    int32_t trSize;              /* row pitch of the unit */
    int32_t scanType;
    uint32_t scanFlagMask;
    int32_t offset;
    int32_t scanPosSigOff;
    int32_t subPosBase;
    uint8_t tabSigCtx[16];
} x265hip_coeff_group_job;
int x265hip_cost_coeff_nxn_batch(const int16_t* coeff, const x265hip_coeff_group_job* jobs, int n, uint8_t* baseCtx, int ctxStride,
                                 uint16_t* absCoeff, uint32_t* bits, void* stream);
/* costCoeffRemain_t (primitives.h:221; dct.cpp:901): job i = levels absCoeff[16 i ...], numNonZero[i], first index firstIdx[i] */
int x265hip_cost_coeff_remain_batch(const uint16_t* absCoeff, const int32_t* numNonZero, const int32_t* firstIdx, int n, uint32_t* bits,
                                    void* stream);
/* costC1C2Flag_t (primitives.h:222; dct.cpp:949): job i = levels absCoeff[16 i ...], numC1Flag[i] (1..8); baseCtxMod + i * ctxStride is
 * the job's greater-than-1 context set (4 bytes, advanced) with its greater-than-2 context at [ctxOffset]. */
int x265hip_cost_c1c2_flag_batch(const uint16_t* absCoeff, const int32_t* numC1Flag, uint8_t* baseCtxMod, int ctxStride, int ctxOffset, int n,
                                 uint32_t* out, void* stream);

/* ---- in-loop filter primitives.  One job = one call of the reference's primitive.
 * pelFilterLumaStrong_t / pelFilterChroma_t (primitives.h:224-225; loopfilter.cpp:139-185): job i filters the four lines starting at
 * plane + off[i] (line step srcStep, sample step across the edge `offset`: EDGE_VER = (stride, 1), EDGE_HOR = (1, stride)). */
int x265hip_pel_filter_luma_strong_batch(int depth, void* plane, const int64_t* off, int64_t srcStep, int64_t offset, const int32_t* tcP,
                                         const int32_t* tcQ, int n, void* stream);
int x265hip_pel_filter_chroma_batch(int depth, void* plane, const int64_t* off, int64_t srcStep, int64_t offset, const int32_t* tc,
                                    const int32_t* maskP, const int32_t* maskQ, int n, void* stream);
/* Deblock::edgeFilterLuma / edgeFilterChroma (deblock.cpp:317-513), the whole per-unit work: job i is one 4-line unit of an edge whose first
 * Q-side sample sits at (xy[2i], xy[2i+1]) of the plane (chroma: of the chroma planes, 4:2:0); dir 0 = vertical edge (EDGE_VER), 1 =
 * horizontal.  bs = the unit's boundary strength (0 skips; chroma filters bs 2 only), qpP / qpQ the QPs of the two sides, bypass [n][2]
 * the cu_transquant_bypass flags of P and Q or NULL (pps bTransquantBypassEnabled off).  beta / tc from Table 8-12 with the slice's
 * offsets, the decisions from lines 0 and 3, then pelFilterLumaStrong or the normal filter (deblock.cpp:276-314); chroma: tc from the
 * mapped chroma QP, then pelFilterChroma on Cb and Cr.  All vertical edges of a picture may go in one launch, then all horizontal ones. */
int x265hip_deblock_luma_batch(int depth, void* plane, int64_t stride, int dir, const int32_t* xy, const uint8_t* bs, const int8_t* qpP,
                               const int8_t* qpQ, const uint8_t* bypass, int betaOffsetDiv2, int tcOffsetDiv2, int n, void* stream);
int x265hip_deblock_chroma_batch(int depth, void* cb, void* cr, int64_t strideC, int dir, const int32_t* xy, const uint8_t* bs,
                                 const int8_t* qpP, const int8_t* qpQ, const uint8_t* bypass, int tcOffsetDiv2, int cbQpOffset, int crQpOffset,
                                 int n, void* stream);
/* sign_t (primitives.h:206; loopfilter.cpp:38): dst[i] = sign(src1[i] - src2[i]) */
int x265hip_sao_sign(int depth, int8_t* dst, const void* src1, const void* src2, int n, void* stream);
/* saoCuOrgE0 / E1 / E1_2Rows / E2 / E3 / B0 (primitives.h:194-198; loopfilter.cpp:44-137).  kind 0..5 in that order.  Sign buffers live in
 * the int8 array `aux` at element offsets aux0 / aux1: E1, E1_2Rows: aux0 = upBuff1 (read and updated); E2: aux0 = bufft (written at
 * [1 .. width]), aux1 = buff1 (read); E3: aux0 = upBuff1 (read at [startX + 1, endX), written one to the left), width = endX.
 * offsets = offsetEo[5] (or the 32 band offsets of B0); signLeft for E0; height for B0 (E0 and E1_2Rows cover two rows, the rest one).
 * Jobs of one launch must not touch each other's samples; width <= 256. */
typedef struct x265hip_sao_job
{
    int64_t recOff, aux0, aux1;
    int32_t width, height, startX;
    int8_t offsets[32];
    int8_t signLeft[2];
    int8_t reserved[2];
} x265hip_sao_job;
int x265hip_sao_apply_batch(int depth, int kind, void* plane, int64_t stride, int8_t* aux, const x265hip_sao_job* jobs, int n, void* stream);
/* saoCuStatsBO / E0 / E1 / E2 / E3 (primitives.h:200-204; sao.cpp:1762-1925): kind 0..4.  Job i: diff + diffOff (pitch 64 = MAX_CU_SIZE),
 * plane + recOff, endX x endY samples; aux0 = upBuff1 (E1, E2, E3: read for the first row, left as the reference leaves it), aux1 =
 * upBufft (E2).  stats / count: [n][32] int32, ADDED to (BO uses 32 classes, the edge kinds entries 0..4 folded by SAO::s_eoTable). */
typedef struct x265hip_sao_stats_job { int64_t diffOff, recOff, aux0, aux1; int32_t endX, endY; } x265hip_sao_stats_job;
int x265hip_sao_stats_batch(int depth, int kind, const int16_t* diff, const void* plane, int64_t stride, int8_t* aux,
                            const x265hip_sao_stats_job* jobs, int n, int32_t* stats, int32_t* count, void* stream);

/* ---------------------------------------------------------------- interpolation ----------------------------- */
/* filter_pp_t / filter_hps_t / filter_ps_t / filter_sp_t / filter_ss_t / filter_hv_pp_t (primitives.h:176-183;
 * ipfilter.cpp:79-369).  taps = 8 (luma) or 4 (chroma).  One job = one W x H block:
 *   src + offS[i] is the block origin in the source plane (the filter reads 3 (luma) / 1 (chroma) elements before
 *   it and 4 / 2 after, horizontally and/or vertically, exactly as the reference does);
 *   coeff[i] = coeffIdx; for HV: coeff[i] = idxX | (idxY << 4).
 * src element type: pixel for HPP/HPS/VPP/VPS/HV, int16 for VSP/VSS; dst: pixel for *PP/VSP/HV, int16 for *PS/VSS. */
#define X265HIP_IF_HPP 0
#define X265HIP_IF_HPS 1   /* isRowExt = (flags & 1) : taps-1 extra rows starting taps/2-1 rows above */
#define X265HIP_IF_VPP 2
#define X265HIP_IF_VPS 3
#define X265HIP_IF_VSP 4
#define X265HIP_IF_VSS 5
#define X265HIP_IF_HVPP 6
int x265hip_interp_batch(int kind, int taps, int depth, int w, int h,
                         const void* src, int64_t strideS, void* dst, int64_t strideD,
                         const int32_t* offS, const int32_t* offD, const int32_t* coeff, int flags, int n, void* stream);

/* ---------------------------------------------------------------- fused hot loops --------------------------- */
/* MotionEstimate::motionEstimate (reference: source/encoder/motion.cpp:739-1569) for n PUs of one shape in one
 * launch: predictor / zero / candidate tests, integer search (searchMethod X265_DIA_SEARCH 0, X265_HEX_SEARCH 1, X265_UMH_SEARCH 2,
 * X265_STAR_SEARCH 3, X265_FULL_SEARCH 5; x265.h), then the sub-pel refine of workload[subme] (motion.cpp:48-58) with luma_hpp/vpp/hvpp
 * + sad/satd (subpelCompare, motion.cpp:1571).  Luma only (bChromaSATD == false, i.e. subme <= 2 semantics for chroma).
 *   fencPlane/refPlane: source and (padded) reconstructed reference luma planes; PU i sits at pu_xy[2i], pu_xy[2i+1]
 *   in both planes' coordinates (same origin);  mvmin/mvmax: full-pel search bounds [n][2] (Search::setSearchRange,
 *   search.cpp:2724);  qmvp: quarter-pel predictor [n][2];  numCand candidates per PU in mvc [n][numCand][2];
 *   mvcost: the lambda-scaled u16 MVD cost row of BitCost::setQP (bitcost.cpp:32), pointer to the entry of MVD 0,
 *   valid for indices [-mvcostHalf, mvcostHalf];  outMv [n][2] quarter-pel;  outCost [n]. */
int x265hip_motion_estimate_batch(int depth, int w, int h,
                                  const void* fencPlane, int64_t strideF, const void* refPlane, int64_t strideR,
                                  const int32_t* pu_xy, const int32_t* mvmin, const int32_t* mvmax, const int32_t* qmvp,
                                  int numCand, const int32_t* mvc, int merange, int searchMethod, int subme,
                                  const uint16_t* mvcost, int mvcostHalf,
                                  int n, int32_t* outMv, int32_t* outCost, void* stream);

/* --me sea (X265_SEA, motion.cpp:1242-1395).  x265hip_build_integral_planes: the twelve window-sum planes FrameFilter::computeMEIntegral
 * keeps per reference picture (framefilter.cpp:684-830; FrameData::m_meIntegral, framedata.h:171: 32x32 32x24 32x8 24x32 16x16 16x12 16x4 12x16 8x32
 * 8x8 4x16 4x4) over a whole padded picture buffer (bufBase = first sample of the buffer, rows x stride): planes[k * planeElems + y * stride + x]
 * = sum of the w_k x h_k window at (x, y), 0 where the window leaves the buffer and in row 0.  scratch: 6 * planeElems uint32.
 * x265hip_motion_estimate_sea_batch: motionEstimate with the SEA search; refPlane / integralPlanes address the same sample (the picture
 * origin or the buffer start) so that pu_xy applies to both; merange <= 126.  Everything else as x265hip_motion_estimate_batch. */
int x265hip_build_integral_planes(int depth, const void* bufBase, int64_t stride, int rows, uint32_t* planes, int64_t planeElems, uint32_t* scratch,
                                  void* stream);
int x265hip_motion_estimate_sea_batch(int depth, int w, int h, const void* fencPlane, int64_t strideF, const void* refPlane, int64_t strideR,
                                      const uint32_t* integralPlanes, int64_t planeElems, const int32_t* pu_xy, const int32_t* mvmin,
                                      const int32_t* mvmax, const int32_t* qmvp, int numCand, const int32_t* mvc, int merange, int subme,
                                      const uint16_t* mvcost, int mvcostHalf, int n, int32_t* outMv, int32_t* outCost, void* stream);

/* All 16 quarter-pel phases of a whole (padded) reference picture, computed once per reference: plane[yFrac*4 + xFrac](x,y)
 * is exactly what luma_hpp / luma_vpp / luma_hvpp (ipfilter.cpp:79-369) produce for that pixel, plane 0 is the picture.
 * planesOrigin addresses pixel (0,0) of plane 0; plane p starts planeElems elements later; same stride and margins as the
 * reference.  Everything except the outermost 4 rows / columns of the padded area is written. */
int x265hip_build_subpel_planes(int depth, const void* refOrigin, int64_t stride, int picW, int picH, int marginX, int marginY,
                                void* planesOrigin, int64_t planeElems, void* stream);
/* x265hip_motion_estimate_batch with the sub-pel candidates read from pre-filtered planes instead of being filtered per
 * candidate (identical results; subpelPlanes == NULL falls back to filtering).  refPlane and subpelPlanes share strideR. */
int x265hip_motion_estimate_planes_batch(int depth, int w, int h,
                                         const void* fencPlane, int64_t strideF, const void* refPlane, int64_t strideR,
                                         const void* subpelPlanes, int64_t planeElems,
                                         const int32_t* pu_xy, const int32_t* mvmin, const int32_t* mvmax, const int32_t* qmvp,
                                         int numCand, const int32_t* mvc, int merange, int searchMethod, int subme,
                                         const uint16_t* mvcost, int mvcostHalf,
                                         int n, int32_t* outMv, int32_t* outCost, void* stream);
/* The same search as Search::predInterSearch runs it on a 4:2:0 picture (setSourcePU with bChroma, motion.cpp:196-224): for
 * subme > 2 and chroma blocks of whole 4x4 tiles every subpelCompare call adds the SATD of the Cb and Cr blocks predicted at
 * the candidate vector (motion.cpp:1601-1660; 4-tap filters, eighth-pel).  Chroma planes are picture origins at half
 * resolution; pu_xy stays in luma samples (even).  Other shapes / subme <= 2 fall through to the luma-only search. */
int x265hip_motion_estimate_chroma_batch(int depth, int w, int h, const void* fencPlane, int64_t strideF, const void* fencCb,
                                         const void* fencCr, int64_t strideFC, const void* refPlane, int64_t strideR,
                                         const void* refCb, const void* refCr, int64_t strideRC,
                                         const int32_t* pu_xy, const int32_t* mvmin, const int32_t* mvmax, const int32_t* qmvp,
                                         int numCand, const int32_t* mvc, int merange, int searchMethod, int subme,
                                         const uint16_t* mvcost, int mvcostHalf, int n, int32_t* outMv, int32_t* outCost, void* stream);

/* Search::setSearchRange (reference: source/encoder/search.cpp:2724-2770) with CUData::clipMv (cudata.cpp:1915-1928) for n
 * CUs at cu_xy [n][2]: qmvp[i] = mvSrc[srcIdx[i]] (quarter-pel; (0,0) when mvSrc is NULL or srcIdx[i] < 0), then
 * [mvmin, mvmax] = clip(qmvp -/+ merange) >> 2 with the frame-parallel vertical bound refLagPixels (search.cpp:92).
 * Intra-refresh and multi-slice restrictions are at their x265 defaults (off). */
int x265hip_set_search_range_batch(int picW, int picH, int maxCUSize, int merange, int refLagPixels,
                                   const int32_t* cu_xy, const int32_t* mvSrc, const int32_t* srcIdx, int n,
                                   int32_t* qmvp, int32_t* mvmin, int32_t* mvmax, void* stream);
/* Predict::predInterLumaPixel (reference: source/common/predict.cpp:245-266) for n PUs of one shape: the prediction
 * of PU i (at pu_xy[i] in both planes) with quarter-pel vector qmv[i] is written into dst at the PU position:
 * copy_pp / luma_hpp / luma_vpp / luma_hvpp selected by the fractional parts. */
int x265hip_pred_inter_luma_batch(int depth, int w, int h, const void* refPlane, int64_t strideR, void* dst, int64_t strideD,
                                  const int32_t* pu_xy, const int32_t* qmv, int n, void* stream);
/* Predict::predInterChromaPixel (reference: source/common/predict.cpp:306-352) for n 4:2:0 PUs of luma size lumaW x lumaH (a
 * multiple of 8 wide): Cb and Cr prediction of PU i (luma position pu_xy[i], quarter-pel vector qmv[i] = eighth-pel chroma
 * vector) written at the PU's chroma position: copy / filter_hpp / filter_vpp / filter_hps + filter_vsp by the fractions. */
int x265hip_pred_inter_chroma_batch(int depth, int lumaW, int lumaH, const void* refCb, const void* refCr, int64_t strideR,
                                    void* dstCb, void* dstCr, int64_t strideD, const int32_t* pu_xy, const int32_t* qmv, int n,
                                    void* stream);
/* Picture border extension (reference: source/common/pixel.cpp:1027-1041 extendPicBorder, margins picyuv.cpp:87-115):
 * replicates the edge pixels of the picW x picH picture at picOrigin into marginX / marginY pixels all around. */
int x265hip_extend_border(int depth, void* picOrigin, int64_t stride, int picW, int picH, int marginX, int marginY, void* stream);
/* BitCost::setQP (reference: source/encoder/bitcost.cpp:32-60, CalculateLogs :108-125): HOST function, fills the u16
 * MVD cost row for `qp`: table[half + i] = table[half - i] = cost of |MVD| = i quarter-pels, i = 0..half (x265: half =
 * 2 * BC_MAX_MV = 65536).  lambda = x265_lambda_tab[qp] of the given bit depth (constants.cpp:34-152). */
int x265hip_mvcost_table(int qp, int depth, uint16_t* hostTable, int half);

/* The residual chain of Search::estimateResidualQT for n TUs of one size (reference: search.cpp:3178-3330 ->
 * quant.cpp:397 transformNxN, :543 invtransformNxN): resi = fenc - pred (sub_ps) -> dct -> quant (flat quantCoeff,
 * add = rounding offset) -> numSig; if numSig: dequant_normal -> idct -> recon = clip(pred + resi') (add_ps)
 * else recon = pred; dist = sse_pp(fenc, recon).  Every intermediate equals the corresponding primitive's output.
 *   TU i: fenc block at fenc+offF[i], prediction at pred+offP[i], recon written at recon+offR[i];
 *   level: [n][size*size] quantised coefficients (quant_t's qCoef); numSig[n]; dist[n]. */
int x265hip_residual_chain_batch(int size, int depth,
                                 const void* fenc, int64_t strideF, const void* pred, int64_t strideP,
                                 void* recon, int64_t strideR,
                                 const int32_t* offF, const int32_t* offP, const int32_t* offR,
                                 const int32_t* quantCoeff, int qBits, int add, int dqScale, int dqShift,
                                 int16_t* level, uint32_t* numSig, uint64_t* dist, int n, void* stream);

/* ---------------------------------------------------------------- frame pass ------------------------------------ */
/* One P-frame worth of the hot path as a fixed pipeline of the kernels above, all on one stream, no host round trip
 * (what bench.py times).  For a width x height luma picture (multiples of 8):
 *   1. motion search, top-down over CU sizes 64, 32, 16, 8 (every 2Nx2N PU that lies inside the picture): setSearchRange
 *      + motionEstimate; the 64x64 level searches around (0,0), every smaller PU around its parent's result;
 *   2. prediction of the picture from the 8x8 vectors (predInterLumaPixel);
 *   3. the residual chain with 32x32 TUs over the 32-aligned area and 8x8 TUs over the rest -> levels, recon, SSE;
 *   4. sa8d(src, pred) mode costs for every CU of the four sizes (analysis.cpp:3065);
 *   5. border extension of the reconstructed picture so it can serve as the next frame's reference.
 * Planes are passed by their picture ORIGIN pointer (pixel 0,0) with margins of at least maxCUSize + 32 = 96 pixels
 * around them (PicYuv, picyuv.cpp:87-89). */
typedef struct x265hip_framepass x265hip_framepass;
int x265hip_framepass_create(int width, int height, int depth, int qp, int merange, int searchMethod, int subme,
                             x265hip_framepass** out);
int x265hip_framepass_destroy(x265hip_framepass* fp);
int x265hip_framepass_run(x265hip_framepass* fp, const void* src, int64_t strideS, const void* ref, int64_t strideR,
                          void* pred, int64_t strideP, void* recon, int64_t strideRec, int marginX, int marginY, void* stream);
/* The same pass on a 4:2:0 picture: additionally Predict::predInterChromaPixel (predict.cpp:306-352) from the 8x8 vectors and
 * the residual chain on Cb and Cr (16x16 TUs under the 32x32 luma TUs, 4x4 under the 8x8 ones) with the chroma QpParam of
 * Quant::setChromaQP (quant.cpp:233-243), and border extension of the chroma reconstruction.  Plane pointers address pixel
 * (0,0); chroma margins are marginX/2, marginY/2.  The Cb and Cr planes of one picture must lie within 2^31 elements of each
 * other (one launch covers both, Cr is addressed from the Cb pointer with 32-bit offsets): allocate a picture's planes in one
 * buffer, as x265's PicYuv does; otherwise X265HIP_EINVAL. */
typedef struct x265hip_yuv { void* y; void* cb; void* cr; int64_t strideY; int64_t strideC; } x265hip_yuv;
/* Predict::motionCompensation for an unweighted bi-predicted PU (predict.cpp:131-199): predInterLumaShort / predInterChromaShort of
 * both references (:268-306, :364-420: convert_p2s / hps / vps / hps + vss by the vector's fractions, 14-bit) combined by Yuv::addAvg
 * (yuv.cpp:189-211, pixel.cpp:842-862) — luma, Cb and Cr of n PUs of one shape (w multiple of 4) in one launch, written at the PU
 * positions of `dst`.  pu_xy in luma samples; mv0 / mv1 quarter-pel luma vectors (eighth-pel for the 4:2:0 chroma). */
int x265hip_pred_inter_bi_batch(int depth, int w, int h, const x265hip_yuv* ref0, const x265hip_yuv* ref1, const x265hip_yuv* dst,
                                const int32_t* pu_xy, const int32_t* mv0, const int32_t* mv1, int n, void* stream);
/* Predict::motionCompensation, every branch (predict.cpp:77-266), for n PUs of one shape of a 4:2:0 picture:
 *   ref1 == NULL  uni-prediction from ref0 with mv0 (a P slice :84-119, or a B-slice PU that uses one list :201-265): weighted when
 *                 wp0 && wp0[0].wtPresent -> predInter*Short + addWeightUni (:525-576, weight_sp); else predInterLumaPixel / ChromaPixel
 *   both given    bi-prediction (:176-199): weighted when wp0 && wp1 && (wp0[0].wtPresent || wp1[0].wtPresent) -> addWeightBi (:411-522);
 *                 else Yuv::addAvg (the same result as x265hip_pred_inter_bi_batch)
 * wp0 / wp1: the slice's WeightParam entries (slice.h:295) of the reference in each list, [Y, Cb, Cr], or NULL when the PPS has
 * weighted (bi-)prediction off.  Vectors are used as given (the caller applies CUData::clipMv).  Host pointers for wp*. */
typedef struct x265hip_weight_param { int32_t inputWeight; int32_t inputOffset; int32_t log2WeightDenom; int32_t wtPresent; } x265hip_weight_param;
int x265hip_motion_compensation_batch(int depth, int w, int h, const x265hip_yuv* ref0, const x265hip_yuv* ref1, const x265hip_yuv* dst,
                                      const int32_t* pu_xy, const int32_t* mv0, const int32_t* mv1, int n,
                                      const x265hip_weight_param* wp0, const x265hip_weight_param* wp1, void* stream);
/* ---- the lookahead's weighted-prediction analysis.
 * LookaheadTLD::weightCostLuma (slicetype.cpp:807-840) for n candidate weights in one launch: fencPlane / refPlane are the lowres planes of
 * the frame and of its reference (plane origins, same stride, padded as Lowres pads them), width x lines the lowres size, intraCost the
 * frame's per-8x8 intra costs; costs[i] (device) = sum over blocks of min(satd8x8(weighted reference, frame), intraCost).  wp: host array;
 * an entry with wtPresent == 0 measures the unweighted reference. */
int x265hip_lookahead_weight_cost_batch(int depth, const void* fencPlane, const void* refPlane, int64_t stride, int width, int lines,
                                        const int32_t* intraCost, const x265hip_weight_param* wp, int n, uint32_t* costs, void* stream);
/* LookaheadTLD::weightsAnalyse (slicetype.cpp:860-960): the scale / offset guess from the frames' statistics (wp_ssd[0], wp_sum[0] of the
 * two Lowres), its two cost evaluations, the 0.2 % test, and the weighting of the reference's four lowres planes.  refBuffers /
 * weightedBuffers: four padded buffers of planeElems elements each, one after the other (Lowres::buffer[0..3], LookaheadTLD::wbuffer);
 * padOffset = plane origin inside a buffer (Lowres::lowresPlane[0] - buffer[0]).  Blocks until done: *isWeighted and *chosen (host) are
 * ReferencePlanes::isWeighted and the weight the reference keeps only inside the weighted planes.  Re-entrant: scratch is per call / per thread. */
int x265hip_lookahead_weights_analyse(int depth, const void* fencPlane, const void* refBuffers, int64_t planeElems, int64_t stride, int64_t padOffset,
                                      int paddedLines, int width, int lines, const int32_t* intraCost, uint64_t fencSsd, uint64_t fencSum,
                                      uint64_t refSsd, uint64_t refSum, void* weightedBuffers, x265hip_weight_param* chosen, int* isWeighted,
                                      void* stream);

/* ---- adaptive quantisation of the lookahead: LookaheadTLD::calcAdaptiveQuantFrame (slicetype.cpp:444-700), 4:2:0, aqMode 0..3 (X265_AQ_NONE,
 * VARIANCE, AUTO_VARIANCE, AUTO_VARIANCE_BIASED; no hevc-aq / HDR10 / user offsets / edge mode).
 * x265hip_aq_block_energy: the device part — energy[i] (device) = acEnergyCu of the i-th qgSize x qgSize block in raster order
 * (ceil(width / qgSize) per row; blocks past the edge read the picture's padding), sums[0..2] / [3..5] (device, ADDED to) the frame's
 * pixel sums and squared sums per plane.  x265hip_lookahead_aq_frame: the whole function — energies on the device, the per-block offsets
 * in double precision on the host side in the reference's order; outputs are host arrays sized like Lowres allocates them (blockCount
 * entries; invQscaleFactor8x8 only for qgSize 8): qpAqOffset (= qpCuTreeOffset), invQscaleFactor, wpStats = { wp_sum[3], wp_ssd[3] }
 * (final, as weightsAnalyse reads them; only meaningful with weightp).  Blocks until done. */
int x265hip_aq_block_energy(int depth, const x265hip_yuv* pic, int width, int height, int qgSize, uint32_t* energy, uint64_t* sums, void* stream);
int x265hip_lookahead_aq_frame(int depth, const x265hip_yuv* pic, int width, int height, int qgSize, int aqMode, double aqStrength, int weightp,
                               double* qpAqOffset, int32_t* invQscaleFactor, int32_t* invQscaleFactor8x8, uint64_t* wpStats, int* blockCount,
                               void* stream);
/* CU-tree: Lookahead::estimateCUPropagate (slicetype.cpp:2641-2750) with the propagateCost primitive (pixel.cpp:914-940) for frame b between
 * p0 and p1 (a P frame: p1MinusP0 == bMinusP0, list 1 unused).  All arrays are device arrays per 8x8 lowres block of frame b: its own
 * propagateCost (ignored when !referenced), intraCost, lowresCosts[b-p0][p1-b], invQscaleFactor (invQscaleFactor8x8 for qgSize 8), the two
 * lists' vectors; refCosts0 / refCosts1 = the reference frames' propagateCost, updated (saturating at 65535).  scratch: 2 * blocks uint64.
 * cuTreeFinish (the log2 of the ratio, :2889-2937) stays host logic. */
int x265hip_cutree_propagate(int widthInCU, int heightInCU, int fpsNum, int fpsDenom, double averageDuration, int bMinusP0, int p1MinusP0,
                             int referenced, int weightedBiPred, const uint16_t* propagateIn, const int32_t* intraCost, const uint16_t* lowresCosts,
                             const int32_t* invQscale, const int32_t* mvs0, const int32_t* mvs1, uint16_t* refCosts0, uint16_t* refCosts1,
                             uint64_t* scratch, void* stream);
int x265hip_framepass_run_yuv(x265hip_framepass* fp, const x265hip_yuv* src, const x265hip_yuv* ref, const x265hip_yuv* pred,
                              const x265hip_yuv* recon, int marginX, int marginY, void* stream);
/* The B-frame variant: a second (future) reference of the same geometry.  Both lists are searched at every level (list 1's vectors
 * and costs: x265hip_framepass_output levels 4..7 of X265HIP_FP_MV / X265HIP_FP_MECOST), and the prediction is the unweighted
 * bi-predictive average of the two lists' 8x8 vectors (x265hip_pred_inter_bi_batch, Predict::motionCompensation B-slice branch)
 * for luma and chroma; the residual chains, sa8d costs and borders follow as in the P pass. */
int x265hip_framepass_run_yuv_b(x265hip_framepass* fp, const x265hip_yuv* src, const x265hip_yuv* ref0, const x265hip_yuv* ref1,
                                const x265hip_yuv* pred, const x265hip_yuv* recon, int marginX, int marginY, void* stream);
/* device pointers to the results of the last run (owned by fp).  `level`: 0..3 = CU size 64, 32, 16, 8 for the ME
 * outputs; for the transform outputs 0..1 = luma TU size 32, 8 and (after run_yuv) 2..3 = Cb 16x16, 4x4, 4..5 = Cr 16x16, 4x4.
 * *count = number of PUs / TUs. */
#define X265HIP_FP_PU_XY     0   /* int32 [n][2]                         */
#define X265HIP_FP_MV        1   /* int32 [n][2] quarter-pel             */
#define X265HIP_FP_MECOST    2   /* int32 [n]                            */
#define X265HIP_FP_SA8D      3   /* int32 [n]   sa8d(src, pred) per CU   */
#define X265HIP_FP_TU_OFF    4   /* int32 [n]   y*stride+x is NOT stored; (x, y) pairs: int32 [n][2] */
#define X265HIP_FP_LEVEL     5   /* int16 [n][size*size]                 */
#define X265HIP_FP_NUMSIG    6   /* uint32 [n]                           */
#define X265HIP_FP_DIST      7   /* uint64 [n]                           */
int x265hip_framepass_output(x265hip_framepass* fp, int which, int level, void** devPtr, int* count);
/* Stage timing with HIP events on the run's own stream.  After set_profiling(fp, 1) every run records an event at each
 * stage boundary; stage_ms() waits for the last run and returns the 11 stage durations in ms:
 * [0] quarter-pel planes, [1..4] motion search of CU size 64/32/16/8 (setSearchRange + motionEstimate), [5] luma prediction,
 * [6] 32x32 residual chain, [7] 8x8 residual chain, [8] sa8d of the four CU sizes, [9] chroma prediction + chroma chains
 * (0 for the luma-only pass), [10] border extension. */
int x265hip_framepass_set_profiling(x265hip_framepass* fp, int enable);
int x265hip_framepass_stage_ms(x265hip_framepass* fp, float* ms11);

/* ---------------------------------------------------------------- small primitives around the path ----------- */
/* var_t (primitives.h:173, pixel_var pixel.cpp:704): out[i] = sum | (sum of squares) << 32 of the size x size block, 32-bit wrap as in the reference */
int x265hip_var_batch(int depth, int size, const void* plane, int64_t stride, const int32_t* off, int n, uint64_t* out, void* stream);
/* weightp_pp_t / weightp_sp_t (primitives.h:164-165, pixel.cpp:518 / :493): explicit weighted prediction of a width x height region —
 * from pixels (reference-plane weighting, MotionReference::applyWeight) or from the 14-bit intermediate (Predict::addWeightUni) */
int x265hip_weight_pp(int depth, const void* src, void* dst, int64_t stride, int width, int height, int w0, int round, int shift, int offset, void* stream);
int x265hip_weight_sp(int depth, const int16_t* src, void* dst, int64_t srcStride, int64_t dstStride, int width, int height, int w0, int round, int shift,
                      int offset, void* stream);
/* scale1D_t / scale2D_t (primitives.h:166-167, pixel.cpp:559 / :585): the 2:1 downscales of the 64x64 intra mode scan (search.cpp:1327-1345):
 * n neighbour lines of 128 + 128 samples -> 64 + 64; n 64x64 blocks -> dense 32x32 */
int x265hip_scale1d_128to64_batch(int depth, const void* src, void* dst, int n, void* stream);
int x265hip_scale2d_64to32_batch(int depth, const void* plane, int64_t stride, const int32_t* off, void* dst, int n, void* stream);
/* transpose_t (primitives.h:158, pixel.cpp:485): n size x size blocks -> dense transposed blocks */
int x265hip_transpose_batch(int depth, int size, const void* plane, int64_t stride, const int32_t* off, void* dst, int n, void* stream);

/* ---------------------------------------------------------------- intra prediction + lookahead lowres ------- */
/* The L-shaped neighbour line of an N x N block is x265's array: line[0] = top-left, line[1..2N] = top + top-right,
 * line[2N+1..4N] = left + bottom-left (Predict::initAdiPattern, predict.cpp; intrapred.cpp).  N = 4, 8, 16, 32.
 *
 * intra_pred_t (primitives.h:143; cu[].intra_pred[35]: planar_pred_c intrapred.cpp:88, intra_pred_dc_c :71,
 * intra_pred_ang_c :106): job i predicts mode (modes[i] & 255) with bFilter = (modes[i] >> 8) & 1 from the line at
 * lines + lineOff[i] into dst + dstOff[i] (row pitch dstStride). */
int x265hip_intra_pred_batch(int depth, int n, const void* lines, const int32_t* lineOff, const int32_t* modes,
                             void* dst, const int32_t* dstOff, int64_t dstStride, int count, void* stream);
/* intra_allangs_t (primitives.h:144, all_angs_pred_c intrapred.cpp:224): per block the 33 angular modes back to back
 * (dest[(i*33 + mode-2) * N*N], horizontal modes stored transposed), each from the raw or the filtered line as
 * g_intraFilterFlags says (constants.cpp:561). */
int x265hip_intra_allangs_batch(int depth, int n, const void* lines, const int32_t* lineOff, const int32_t* filteredOff,
                                int bLuma, void* dest, int count, void* stream);
/* intra_filter_t (primitives.h:145, intraFilter<N> intrapred.cpp:32): [1 2 1]/4 along the line, both ends kept */
int x265hip_intra_filter_batch(int depth, int n, const void* in, const int32_t* inOff, void* out, const int32_t* outOff,
                               int count, void* stream);
/* The distortion half of the intra mode decision (Search::checkIntraInInter search.cpp:1291-1452, estIntraPredQT :1509-1696): for
 * each block the cu[].sa8d cost (satd_4x4 at N = 4) of all 35 predictions against the source block fenc + fencOff[i] (row pitch
 * fencStride): costs[i*35 + mode].  DC is edge-smoothed when N <= 16, planar and the angles take the raw or the filtered line as
 * g_intraFilterFlags says, the mode 10 / 26 edge gradient is on when N <= 16 — exactly the calls the reference makes; the mode
 * bits and the argmin stay with the caller (they need the entropy coder's state).  64x64 CUs are scanned at 32x32 by the
 * reference after a 2:1 downscale (:1327-1345), which is the caller's business too. */
int x265hip_intra_scan_batch(int depth, int n, const void* lines, const int32_t* lineOff, const int32_t* filteredOff, const void* fenc,
                             int64_t fencStride, const int32_t* fencOff, int count, int32_t* costs, void* stream);
/* downscale_t (primitives.h:168; frameInitLowres = frame_init_lowres_core pixel.cpp:604): the half-resolution picture and
 * its H / V / diagonal half-pel companions; reads src rows 0..2*height and columns 0..2*width (the source margins). */
int x265hip_frame_init_lowres(int depth, const void* src, int64_t srcStride, void* dst0, void* dstH, void* dstV, void* dstC,
                              int64_t dstStride, int width, int height, void* stream);
/* Lowres::init (lowres.cpp:297-305): frameInitLowres + extendPicBorder of the four planes (planes[i] = picture origins) */
int x265hip_lowres_init(int depth, const void* src, int64_t srcStride, void* const planes[4], int64_t dstStride,
                        int width, int height, int marginX, int marginY, void* stream);
/* LookaheadTLD::lowresIntraEstimate (slicetype.cpp:696-802) over a border-extended lowres plane: for every 8x8 block the
 * cheapest of DC, planar and the coarse-to-fine angular scan by SATD, + 5*lambda(lookahead QP) + 4.  intraCost /
 * intraMode are [heightInCU*widthInCU]; rowSatd[heightInCU] and costEst[1] (sum over the non-edge blocks) are optional
 * (both NULL to skip).  No AQ scaling (invQscaleFactor == NULL in the reference). */
int x265hip_lowres_intra_estimate(int depth, const void* plane, int64_t stride, int widthInCU, int heightInCU,
                                  int32_t* intraCost, uint8_t* intraMode, int32_t* rowSatd, int32_t* costEst, void* stream);

/* The lookahead's P-frame cost pass for a batch of (frame, reference) pairs — CostEstimateGroup::estimateCUCost
 * (slicetype.cpp:3218-3385) for b == p1 over every 8x8 block of the lowres frame, i.e. per block: SATD of the already
 * known neighbour vectors (right, below, below-left, below-right; :3271-3307) -> start vector, the lowres flavour of
 * MotionEstimate::motionEstimate (HEX, merange 16, subpel refine 1: motion.cpp:775, :855-944, :1471-1501 with
 * Lowres::lowresQPelCost, lowres.h:94), + 4 against the block's intra cost, frame score over the non-edge blocks, and the
 * AQ-scaled flavour of the score and of the row sums when `invQscale` is given (:3353-3384).
 * Rows are walked bottom-up in `numSlices` independent slices of `numRowsPerSlice` rows (the reference's cooperative
 * lookahead slices, :3092-3104; 1 slice = the serial loop :3170-3179).  No HME.  A weighted reference (weightp) is the same
 * pass with `ref` pointing at the weighted planes (x265hip_lookahead_weights_analyse).
 * All pairs share the geometry; planes are border-extended lowres planes (x265hip_lowres_init), `ref` = hpel plane 0 of
 * the reference with planes 1..3 `planeElems` elements apart.  `sync` is ncu u64 of scratch per pair, zeroed once when
 * allocated; `epoch` must be non-zero and differ from every earlier call that used the same scratch (the caller owns that
 * rule: a reused epoch is NOT detected — stale handshake words would be taken for this call's; x265hip_la_* below owns its
 * scratch and epochs).  `pairs` is a DEVICE array.
 * est[4 i ..] = { costEst, costEstAq, intraMbs, status } of pair i (int64 each); status != 0 means the row handshake timed
 * out (a `sync` scratch that was not zeroed) and the pair's outputs are invalid. */
typedef struct x265hip_lookahead_pair
{
    const void*    fenc;         /* lowresPlane[0] origin of the frame being costed */
    const void*    ref;          /* reference frame: hpel plane 0 origin */
    const int32_t* intraCost;    /* [ncu] of the frame being costed (x265hip_lowres_intra_estimate) */
    int32_t*       mvs;          /* out [ncu][2]  lowresMvs[0][b-p0], quarter-pel */
    int32_t*       mvCosts;      /* out [ncu]     lowresMvCosts[0][b-p0] */
    uint16_t*      lowresCosts;  /* out [ncu]     min(cost, 16383) | listused << 14 */
    int32_t*       rowSatds;     /* out [heightInCU] */
    uint64_t*      sync;         /* scratch [ncu] */
    const int32_t* invQscale;    /* [ncu] Lowres::invQscaleFactor (invQscaleFactor8x8 for qg-size 8) of the frame being costed, or NULL: no AQ */
    int32_t        bidirList;    /* 0: P frame (everything above).  1: one list of a B frame: the search applies the bidir skip rule
                                    (slicetype.cpp:3303-3317) and only mvs / mvCosts are written; finish with x265hip_lookahead_bidir_batch */
    int32_t        sliceGeom;    /* 0: the launch's numRowsPerSlice / numSlices.  Otherwise numRowsPerSlice | numSlices << 16 of THIS pair: estimates of
                                    different cooperative-slice geometries (batch mode = 1 slice, single estimates = m_numCoopSlices) share a launch */
} x265hip_lookahead_pair;
int x265hip_lookahead_cost_p_batch(int depth, const x265hip_lookahead_pair* pairs, int nPairs, int64_t stride, int64_t planeElems,
                                   int widthInCU, int heightInCU, int numRowsPerSlice, int numSlices,
                                   const uint16_t* mvcost, uint32_t epoch, int64_t* est, void* stream);
/* The same bookkeeping for a P estimate whose list-0 search was done before (bDoSearch[0] == false, slicetype.cpp:3260-3264): cost from the
 * stored mvCosts (an INPUT here; mvs / sync / ref unused) + 4 against intra, packed lowresCosts, rowSatds, est as above. */
int x265hip_lookahead_pcost_batch(const x265hip_lookahead_pair* pairs, int nPairs, int widthInCU, int heightInCU, int64_t* est, void* stream);

/* B frames (p0 < b < p1): run the two list searches as pairs with bidirList = 1 (a list that was searched before keeps its stored
 * vectors and costs, slicetype.cpp:3126-3127, :3260-3264), then this pass over every block: cheapest of the two lists, the average
 * of the two motion-compensated blocks and the co-located average by SATD (:3320-3338), + 4, packed lowresCosts, rowSatds, and
 * est[2 i ..] = { raw sum over the non-edge blocks, its AQ-scaled flavour } (int64; the caller scales costEst by 100 / (130 + bFrameBias),
 * :3183-3186).  `frames` is a DEVICE array. */
typedef struct x265hip_lookahead_bframe
{
    const void*    fenc;         /* lowresPlane[0] origin of the B frame */
    const void*    ref0;         /* past reference, hpel plane 0 origin */
    const void*    ref1;         /* future reference, hpel plane 0 origin */
    const int32_t* mvs0;         /* [ncu][2] lowresMvs[0][b-p0] */
    const int32_t* mvs1;         /* [ncu][2] lowresMvs[1][p1-b] */
    const int32_t* mvCosts0;     /* [ncu] */
    const int32_t* mvCosts1;     /* [ncu] */
    uint16_t*      lowresCosts;  /* out [ncu] */
    int32_t*       rowSatds;     /* out [heightInCU] */
    const int32_t* invQscale;    /* [ncu] or NULL, as in x265hip_lookahead_pair */
} x265hip_lookahead_bframe;
int x265hip_lookahead_bidir_batch(int depth, const x265hip_lookahead_bframe* frames, int nFrames, int64_t stride, int64_t planeElems,
                                  int widthInCU, int heightInCU, int64_t* est, void* stream);

/* ---------------------------------------------------------------- the lookahead session (x265's batching seam) ---- */
/* What an x265 build binds where the reference itself batches lookahead work: CostEstimateGroup::add / finishBatch
 * (slicetype.cpp:3027-3048, up to 512 estimates per batch) and estimateFrameCost (:3115-3214).  The session mirrors x265's Lowres
 * (common/lowres.h:152) in HBM: a frame SLOT holds the four padded half-resolution planes, the per-8x8 intra costs and AQ factors,
 * and per (list, distance) the vectors and costs of every motion search done so far.  One estimate = slot indices + which lists to
 * search; a batch = one launch of every list search, one of every P bookkeeping pass, one of every B pass, one copy back.
 * x265_amd/host/x265_hip_lookahead.cpp is the x265-side binding.  All pointers are HOST pointers;
 * calls block until their results are in host memory; a session serialises its callers.  The weighted plane sets are a two-call protocol:
 * x265hip_la_weights_analyse hands out a weightedId that stays reserved until the NEXT x265hip_la_estimate_batch* call of the session returns
 * (whatever it returns) — so the thread that analysed must be the one that sends the batch, with no other batch of the session in between: use one
 * thread per session (the x265 binding holds its lock across both calls).  After a batch the session keeps at most 8 plane sets for reuse. */
typedef struct x265hip_la x265hip_la;
typedef struct x265hip_la_config
{
    int32_t depth;                   /* 8 / 10 / 12 */
    int32_t width, lines;            /* Lowres::width, Lowres::lines (lowres picture, pixels) */
    int64_t stride;                  /* Lowres::lumaStride */
    int64_t planeElems;              /* Lowres::buffer[1] - buffer[0] (padded plane, elements) */
    int64_t padOffset;               /* Lowres::lowresPlane[0] - buffer[0] */
    int32_t widthInCU, heightInCU;   /* Lookahead::m_8x8Width, m_8x8Height */
    int32_t maxDist;                 /* bframes + 2: (list, distance) entries per frame (lowres.cpp:135-150) */
    int32_t numSlots;                /* frames resident at once (lookahead depth + bframes + the reference's slack) */
} x265hip_la_config;
x265hip_la* x265hip_la_create(const x265hip_la_config* cfg);          /* NULL on failure: x265hip_last_error() */
/* the same session on the device of place `place` (x265hip_places): the sessions of one process take the places in turn (round 5; every call of the
 * session switches the calling thread to the session's device, so it may be used from any thread) */
x265hip_la* x265hip_la_create_at(int place, const x265hip_la_config* cfg);
void x265hip_la_destroy(x265hip_la* la);
/* Lowres::init + lowresIntraEstimate + calcAdaptiveQuantFrame results of a frame entering the lookahead: `buffers` = the four padded
 * planes, contiguous (Lowres::buffer[0], 4 * planeElems pixels); intraCost [ncu]; invQscale [ncu] (invQscaleFactor, or
 * invQscaleFactor8x8 for qg-size 8) or NULL when the encoder allocates none.  Forgets every vector stored for the slot. */
int x265hip_la_set_frame(x265hip_la* la, int slot, const void* buffers, const int32_t* intraCost, const int32_t* invQscale);
/* vectors of a search the session has not seen (done by another path): mvs [ncu][2] quarter-pel, mvCosts [ncu] */
int x265hip_la_put_vectors(x265hip_la* la, int slot, int list, int dist, const int32_t* mvs, const int32_t* mvCosts);
int x265hip_la_has_vectors(x265hip_la* la, int slot, int list, int dist);
/* LookaheadTLD::weightsAnalyse (slicetype.cpp:860-960) of frame slotB against slotRef from the frames' wp_ssd[0] / wp_sum[0]: *isWeighted,
 * *chosen, and — when weighted — *weightedId naming the weighted planes for the NEXT x265hip_la_estimate_batch (they live until that call
 * returns, like the reference's per-thread wbuffer lives until the next analysis). */
int x265hip_la_weights_analyse(x265hip_la* la, int slotB, int slotRef, uint64_t fencSsd, uint64_t fencSum, uint64_t refSsd, uint64_t refSum,
                               x265hip_weight_param* chosen, int* isWeighted, int* weightedId);
typedef struct x265hip_la_estimate
{
    int32_t   b, p0, p1;             /* slots; p1 == b: P estimate */
    int32_t   dist0, dist1;          /* b - p0, p1 - b in frames */
    int32_t   search0, search1;      /* bDoSearch[] (slicetype.cpp:3125-3127); a list that is not searched uses the session's stored vectors */
    int32_t   weightedId;            /* -1, or x265hip_la_weights_analyse's id: list 0 searches the weighted planes */
    int32_t*  mvs0;                  /* out when search0: Lowres::lowresMvs[0][dist0] ([ncu] MV = 2 x int32) */
    int32_t*  mvCosts0;              /* out when search0: lowresMvCosts[0][dist0] */
    int32_t*  mvs1;                  /* out when search1 */
    int32_t*  mvCosts1;
    uint16_t* lowresCosts;           /* out: lowresCosts[dist0][dist1] */
    int32_t*  rowSatds;              /* out: rowSatds[dist0][dist1] */
    int64_t   costEst, costEstAq;    /* out: sums over the non-edge blocks, before the B-frame scaling (:3183-3186) */
    int32_t   intraMbs, reserved;    /* out (P estimates) */
} x265hip_la_estimate;
/* numRowsPerSlice / numSlices: the cooperative-slice geometry the reference would use for these estimates (1 slice of heightInCU rows in
 * batch mode, m_numCoopSlices otherwise, :3141-3180) */
int x265hip_la_estimate_batch(x265hip_la* la, x265hip_la_estimate* est, int n, int numRowsPerSlice, int numSlices);
/* Searches AHEAD of the reference's request.  One list search of estimateCUCost — the vectors and vector costs of (frame b, list, dist) —
 * is a function of the two frames (plus the list-0 weights), of whether it runs inside a P or a B estimate (`bidir`: the zero-vector skip rule,
 * slicetype.cpp:3303-3317) and of the cooperative-slice geometry (a slice's last row has no row below, :3274-3281).  x265 asks for most of
 * them one at a time (slicetypePathCost / scenecut / cuTree -> singleCost), which leaves the device > 90 % empty; a binding that can see
 * which searches are still missing in the lookahead window hands them over WITH a batch that has to run anyway, each tagged with the variant
 * the reference will ask for.  They ride in the same lookahead_p_kernel launch; their results stay in the session, keyed by
 * (slot, list, dist, bidir, geometry).  A later estimate that needs exactly that search (search0 / search1 set, same variant, and — list 0 —
 * the same weighting decision, which is a function of the frame pair too) is served from there: no search launch, the vectors are copied
 * to the estimate's outputs as if it had searched.  Anything never asked for is simply dropped with the slot.  Exact by construction.
 * weightedId: as in x265hip_la_estimate (list 0 only). */
typedef struct x265hip_la_search
{
    int32_t b, ref;                  /* slots: the frame and the reference of this list */
    int32_t list, dist;              /* 0 / 1; |b - ref| in frames */
    int32_t bidir;                   /* 0: as inside a P estimate, 1: as inside a B estimate */
    int32_t weightedId;              /* -1, or x265hip_la_weights_analyse's id (list 0) */
    int32_t numRowsPerSlice, numSlices;
} x265hip_la_search;
int x265hip_la_estimate_batch_ahead(x265hip_la* la, x265hip_la_estimate* est, int n, int numRowsPerSlice, int numSlices,
                                    const x265hip_la_search* ahead, int nAhead);
/* 1 when the session holds that search ahead of its request (so the binding can skip the weights analysis that would precede it) */
int x265hip_la_has_ahead(x265hip_la* la, int slot, int list, int dist, int bidir, int numRowsPerSlice, int numSlices);
int x265hip_la_stats(x265hip_la* la, uint64_t* batches, uint64_t* estimates, uint64_t* searches);
/* searches launched ahead of their request / of those, how many an estimate later used / lookahead_p_kernel launches / pairs in them */
int x265hip_la_stats_ahead(x265hip_la* la, uint64_t* launchedAhead, uint64_t* usedAhead, uint64_t* searchLaunches, uint64_t* searchPairs);

/* ---------------------------------------------------------------- reference-picture mirrors (lookup face) --------- */
/* The sub-pel filters the encoder applies to a reference picture (luma_hpp / luma_vpp / luma_hvpp from MotionEstimate::subpelCompare,
 * motion.cpp:1571-1600, and Predict::predInterLumaPixel, predict.cpp:245-266) are per-pixel functions of that picture, so a mirrored
 * picture gets all 15 fractional planes computed once, as its CTU rows are published, and the table slots serve a filter call as a block
 * copy (x265_amd/host/x265_hip_refplanes.cpp).  Everything here is asynchronous to the encoder: rows_final() queues work
 * for a worker thread and returns; readers use only what rows_ready() has published and compute the rest themselves.
 * Geometry = PicYuv's (picyuv.cpp:87-128): hostBase = m_picBuf[0], bufRows rows of `stride` elements, picture origin at
 * (marginY, marginX).  The buffer stays the caller's and must outlive the refpic. */
typedef struct x265hip_refpic x265hip_refpic;
x265hip_refpic* x265hip_refpic_create(int depth, int picW, int picH, int64_t stride, int marginX, int marginY, int bufRows, const void* hostBase);
x265hip_refpic* x265hip_refpic_create_at(int place, int depth, int picW, int picH, int64_t stride, int marginX, int marginY, int bufRows, const void* hostBase);   /* x265hip_places */
void x265hip_refpic_destroy(x265hip_refpic* rp);
/* a new picture is about to be reconstructed into the buffer: nothing is valid any more; queued work of the old picture is dropped */
int x265hip_refpic_reset(x265hip_refpic* rp);
/* picture rows [0, rowsFinal) are final in the buffer together with their left / right margins and the top margin (rowsFinal >= picH:
 * the whole padded picture) — what FrameFilter::processPostRow guarantees when it sets m_reconRowFlag (framefilter.cpp:664) */
int x265hip_refpic_rows_final(x265hip_refpic* rp, int rowsFinal);
/* host plane of phase p = yFrac * 4 + xFrac (1..15): same layout as the buffer — element offset of a pixel in hostBase == in the plane */
const void* x265hip_refpic_plane(x265hip_refpic* rp, int phase);
/* r: the planes hold picture rows [-(marginY - 4), r) (r <= picH + marginY - 4 once the picture is complete); columns
 * [-(marginX - 4), picW + marginX - 4).  *_ptr: the same int for readers that poll it per call (load it with acquire semantics). */
int x265hip_refpic_rows_ready(x265hip_refpic* rp);
const int* x265hip_refpic_rows_ready_ptr(x265hip_refpic* rp);
int x265hip_refpic_wait(x265hip_refpic* rp);        /* blocks until the worker has nothing queued for rp; reports a worker failure */

/* ---------------------------------------------------------------- source-picture energy planes (lookup face) ------- */
/* The source half of psyCost_pp (pixel.cpp:726-757) for every aligned block of a SOURCE plane: e8[by * (width / 8) + bx] =
 * sa8d_8x8(block, 0) - (sum(block) >> 2) of the 8x8 block at (8 bx, 8 by); e4[...] the same with satd_4x4 for every 4x4 block
 * (row pitch width / 8 * 2).  Host pointers, blocks until the planes are in host memory; one call per plane of a picture entering the
 * encoder.  x265_amd/host/x265_hip_srcplanes.cpp serves cu[].psy_cost_pp from them. */
int x265hip_source_energy(int depth, const void* hostPlane, int64_t stride, int width, int height, int32_t* hostE8, int32_t* hostE4);

/* ---------------------------------------------------------------- SAD surfaces (lookup face) ----------------------- */
/* The integer-pel candidates of MotionEstimate::motionEstimate are sad(fenc, FENC_STRIDE, fref + mx + my * stride, stride) (motion.cpp:246-330
 * macros; HEX :770-944, STAR :1132-1240) with fenc a copy of the SOURCE picture's PU (setSourcePU, :194-222) and fref the finished reference
 * picture at the PU's position (:752-756).  Which candidates a search visits depends on the decisions before it; what a candidate costs does
 * not — it is a function of (source picture, reference picture, position, vector).  A sadsurf holds those values for every aligned N x N block
 * (N = 8 << level) of one (source picture, reference picture) pair over a WIN x WIN window of vectors per block, built on the device as the
 * reference picture's rows become final, and mirrored into page-locked host memory; x265_amd/host/x265_hip_sadplanes.cpp swaps a
 * MotionEstimate's sad / sad_x3 / sad_x4 pointers for table loads while the search of a covered PU runs.
 * Where a block's window lies is decided on the device, per 64 x 64 region top down over the block sizes 64, 32, 16: every vector of
 * [-searchRange, searchRange)^2 that keeps the block inside the padded reference picture is measured (the search-window kernel: source CTU and
 * reference window staged in LDS, 16x16 SADs of all candidates by v_qsad_pk_u16_u8 kept in LDS, folded to 32 / 64), the cheapest by
 * SAD + (lambda20 * (bits(4 |vx - px|) + bits(4 |vy - py|)) + 10) / 20 (bits(d) = 2 floor(log2(d + 1)) + 1; p = the parent block's best vector,
 * (0, 0) at the top; ties: smaller vy, then smaller vx) is the block's best, and its window starts at best - WIN / 2, clamped into the range
 * and into the padded picture.  That choice only decides how many lookups hit — every entry is the exact SAD, and a vector outside the window is
 * computed by the C function as before.  lambda20 = 20 x the encoder's lambda (SAD units per bit of vector cost).
 * 8-bit pictures: searchRange up to 32, v_qsad_pk_u16_u8, u16 surface in LDS; 16-bit pictures (depth 10 / 12): searchRange up to 16 (the u32 surface of a
 * CTU must fit LDS), v_sad_u16, every table u32, no 8 x 8 level.  Strides count samples. */
#define X265HIP_SADSURF_WIN 16
#define X265HIP_SADSURF_LEVELS 4
typedef struct x265hip_srcpic x265hip_srcpic;        /* the luma plane of a source picture, resident on the device */
x265hip_srcpic* x265hip_srcpic_create(int depth, int width, int height);
x265hip_srcpic* x265hip_srcpic_create_at(int place, int depth, int width, int height);       /* x265hip_places */
int x265hip_srcpic_upload(x265hip_srcpic* sp, const void* hostLuma, int64_t stride);          /* blocks until the copy is on the device */
void x265hip_srcpic_destroy(x265hip_srcpic* sp);
typedef struct x265hip_sadsurf x265hip_sadsurf;
typedef struct x265hip_sadsurf_level
{
    int32_t        blocksX, blocksY;     /* width / N, height / N (blocks that lie inside the picture) */
    int32_t        entryBytes;           /* 2 (uint16: N * N * pixel max < 65536) or 4 (uint32) */
    int32_t        blocksPerCtuRow;      /* 64 / N block rows per row of 64 picture lines */
    /* Results are laid out per row of 64 picture lines (one device-to-host copy per band): block (bx, by) lives in chunk r = by / blocksPerCtuRow
     * at index k = (by % blocksPerCtuRow) * blocksX + bx:
     *   origin  (const int16_t*)((const char*)origin + r * ctuRowPitch) + 2 k      (ox, oy) = the vector of window entry (0, 0), full-pel
     *   table   (const char*)table + r * ctuRowPitch + k * WIN * WIN * entryBytes   entry j * WIN + i = SAD at vector (ox + i, oy + j)
     * origin == NULL: level not built */
    const int16_t* origin;
    const void*    table;
    /* Sub-pel SATDs around the window's centre c = (ox + WIN / 2, oy + WIN / 2) (round 4; levels 1..3, requested with bit 4 of `levels`, built
     * where the reference picture's sub-pel planes live): what MotionEstimate::subpelCompare (reference encoder/motion.cpp:1571-1600) measures with
     * the satd comparison at the 7 x 7 quarter-pel vectors q = 4 c + (dx, dy), dx, dy in -3..3 — every position the sub-pel refinement of a
     * search that ends its integer stage at c can visit:
     *   entries (const uint32_t*)((const char*)subpel + r * ctuRowPitch) + k * X265HIP_SADSURF_SUBPEL, entry (dy + 3) * 7 + (dx + 3)
     * NULL: not built */
    const uint32_t* subpel;
} x265hip_sadsurf_level;
#define X265HIP_SADSURF_SUBPEL 49
typedef struct x265hip_sadsurf_view
{
    x265hip_sadsurf_level level[X265HIP_SADSURF_LEVELS];
    int64_t    ctuRowPitch;              /* bytes between the chunks of consecutive rows of 64 picture lines */
    const int* ctuRowsReady;             /* origins and tables of every block above picture line 64 * (*ctuRowsReady) are in host memory (load with
                                            acquire semantics); grows as the reference picture's rows become final */
} x265hip_sadsurf_view;
/* Levels 1..3 (N = 16, 32, 64) are searched; level 0 (N = 8; x265hip_sadsurf_attach_levels, origin == NULL when it is not built) takes its window from the parent 16 x 16 block (same origin; its SADs are measured after
 * the parent's window is known); an 8 x 8 block whose parent does not lie inside the picture has no window: origin (-32768, -32768), never look it up.  searchRange: 8..32, a multiple of 4.  The surface follows `ref`'s progress (x265hip_refpic_rows_final) by itself: rows that are
 * final already are built at once, the others as they arrive; x265hip_refpic_reset / _destroy of `ref` ends it (no further rows are published;
 * the handle stays valid until it is released).  `src` must stay unchanged and alive until the release.  NULL on failure. */
x265hip_sadsurf* x265hip_sadsurf_attach(x265hip_srcpic* src, x265hip_refpic* ref, int searchRange, int lambda20);          /* levels 1..3 */
/* levels: bit l = level l is built; bits 1..3 must be set, bit 4 adds the sub-pel SATD tables of levels 1..3, bit 0 adds the 8 x 8 windows (2.5 times the table bytes: on the 1080p bench clip only a
 * third of the 8 x 8 searches stay inside their parent's window, so x265_amd/host leaves it off unless X265HIP_SADPLANES_LEVELS says otherwise) */
x265hip_sadsurf* x265hip_sadsurf_attach_levels(x265hip_srcpic* src, x265hip_refpic* ref, int searchRange, int lambda20, int levels);
const x265hip_sadsurf_view* x265hip_sadsurf_get_view(x265hip_sadsurf* ss);
void x265hip_sadsurf_release(x265hip_sadsurf* ss);
/* per process: surfaces attached, CTU rows built, kernel launches that built them (rows of several surfaces of one reference picture share a launch)
 * and the device time of those launches (HIP events around each launch on the stream it runs on), nanoseconds */
int x265hip_sadsurf_stats(uint64_t* attached, uint64_t* ctuRows, uint64_t* launches, uint64_t* kernelNs);

/* ---------------------------------------------------------------- per-call entry points (host pointers) ----- */
/* What the reference-side table shims bind (x265_amd/host/x265_hip_primitives.cpp).  Arguments are the slot's own
 * arguments (HOST pointers, caller-owned, valid only during the call — primitives.h:133-234); each call stages the
 * operands into pinned memory, launches the batched kernel with n = 1 on the calling thread's stream and waits.
 * On any failure they return a negative code and leave the outputs untouched so the shim can call the C slot. */
unsigned long long x265hip_call_count(void);   /* per-call launches served so far (process-wide) */
int x265hip_call_pixcmp(int op, int depth, int w, int h, const void* a, int64_t sa, const void* b, int64_t sb, int32_t* result);
int x265hip_call_sad_xn(int K, int depth, int w, int h, const void* fenc, const void* const* refs, int64_t strideR, int32_t* res);
int x265hip_call_sse_pp(int depth, int w, int h, const void* a, int64_t sa, const void* b, int64_t sb, uint64_t* result);
int x265hip_call_sse_ss(int w, int h, const int16_t* a, int64_t sa, const int16_t* b, int64_t sb, uint64_t* result);
int x265hip_call_dct(int size, int dst4, int depth, const int16_t* src, int16_t* dst, int64_t srcStride);
int x265hip_call_idct(int size, int dst4, int depth, const int16_t* src, int16_t* dst, int64_t dstStride);
int x265hip_call_quant(const int16_t* coef, const int32_t* quantCoeff, int32_t* deltaU, int16_t* qCoef,
                       int qBits, int add, int numCoeff, uint32_t* numSig);
int x265hip_call_nquant(const int16_t* coef, const int32_t* quantCoeff, int16_t* qCoef, int qBits, int add, int numCoeff,
                        uint32_t* numSig);
int x265hip_call_dequant_normal(const int16_t* quantCoef, int16_t* coef, int num, int scale, int shift);
int x265hip_call_dequant_scaling(const int16_t* quantCoef, const int32_t* deQuantCoef, int16_t* coef, int num, int per, int shift);
int x265hip_call_interp(int kind, int taps, int depth, int w, int h, const void* src, int64_t strideS,
                        void* dst, int64_t strideD, int coeffIdx, int coeffIdy, int isRowExt);
int x265hip_call_sub_ps(int depth, int w, int h, int16_t* dst, int64_t ds, const void* a, const void* b, int64_t sa, int64_t sb);
int x265hip_call_add_ps(int depth, int w, int h, void* dst, int64_t ds, const void* a, const int16_t* r, int64_t sa, int64_t sr);
int x265hip_call_addavg(int depth, int w, int h, const int16_t* s0, const int16_t* s1, void* dst, int64_t st0, int64_t st1, int64_t ds);
int x265hip_call_pixelavg_pp(int depth, int w, int h, void* dst, int64_t ds, const void* s0, int64_t st0, const void* s1, int64_t st1);
int x265hip_call_copy(int kind, int depth, int w, int h, void* dst, int64_t ds, const void* src, int64_t ss);
int x265hip_call_p2s(int depth, int w, int h, const void* src, int64_t ss, int16_t* dst, int64_t ds);
int x265hip_call_cpy_shift(int kind, int size, int16_t* dst, const int16_t* src, int64_t stride, int shift);
int x265hip_call_copy_cnt(int size, int16_t* coeff, const int16_t* resi, int64_t stride, uint32_t* numSig);
int x265hip_call_count_nonzero(int size, const int16_t* qCoef, int* count);
int x265hip_call_blockfill_s(int size, int16_t* dst, int64_t ds, int16_t val);
int x265hip_call_denoise_dct(int16_t* dctCoef, uint32_t* resSum, const uint16_t* offset, int numCoeff);
int x265hip_call_rdoq_cost(int kind, int size, int depth, const int16_t* resiDct, const int16_t* fencDct, int64_t* costUncoded,
                           int64_t* totalUncoded, int64_t* totalRd, const int64_t* psyScale, uint32_t blkPos);

int x265hip_call_var(int depth, int size, const void* pix, int64_t stride, uint64_t* result);
int x265hip_call_weight_pp(int depth, const void* src, void* dst, int64_t stride, int width, int height, int w0, int round, int shift, int offset);
int x265hip_call_weight_sp(int depth, const int16_t* src, void* dst, int64_t srcStride, int64_t dstStride, int width, int height, int w0, int round, int shift, int offset);
int x265hip_call_scale1d_128to64(int depth, void* dst, const void* src);
int x265hip_call_scale2d_64to32(int depth, void* dst, const void* src, int64_t stride);
int x265hip_call_transpose(int depth, int size, void* dst, const void* src, int64_t stride);
/* scanPosLast_t / findPosFirstLast_t / costCoeffNxN_t / costCoeffRemain_t / costC1C2Flag_t (primitives.h:217-222) on host buffers;
 * `scanType` is what the caller's scan-table pointer identifies (the shim compares it with g_scanOrder / g_scan4x4). */
int x265hip_call_scan_pos_last(int log2TrSize, int scanType, const int16_t* coeff, uint16_t* coeffSign, uint16_t* coeffFlag, uint8_t* coeffNum,
                               int numSig, int* lastPos);
int x265hip_call_find_pos_first_last(const int16_t* dstCoeff, int64_t trSize, int scanType, uint32_t* result);
int x265hip_call_cost_coeff_nxn(int scanType, const int16_t* coeff, int64_t trSize, uint16_t* absCoeff, const uint8_t* tabSigCtx,
                                uint32_t scanFlagMask, uint8_t* baseCtx, int offset, int scanPosSigOff, int subPosBase, uint32_t* result);
int x265hip_call_cost_coeff_remain(const uint16_t* absCoeff, int numNonZero, int idx, uint32_t* result);
int x265hip_call_cost_c1c2_flag(const uint16_t* absCoeff, int64_t numC1Flag, uint8_t* baseCtxMod, int64_t ctxOffset, uint32_t* result);
/* pelFilterLumaStrong_t / pelFilterChroma_t / sign_t / saoCuOrg* / saoCuStats* (primitives.h:194-206, 224-225) on host buffers.
 * sao_apply: kind 0..5 = E0, E1, E1_2Rows, E2, E3, B0; a = width (endX for E3), b = height (B0) or startX (E3); aux0 / aux1 = the
 * primitive's sign buffers in argument order.  sao_stats: kind 0..4 = BO, E0..E3; up1 / upt = upBuff1 / upBufft. */
int x265hip_call_pel_filter_luma_strong(int depth, void* src, int64_t srcStep, int64_t offset, int32_t tcP, int32_t tcQ);
int x265hip_call_pel_filter_chroma(int depth, void* src, int64_t srcStep, int64_t offset, int32_t tc, int32_t maskP, int32_t maskQ);
int x265hip_call_sao_sign(int depth, int8_t* dst, const void* src1, const void* src2, int endX);
int x265hip_call_sao_apply(int depth, int kind, void* rec, int64_t stride, int a, int b, int8_t* aux0, int8_t* aux1, const int8_t* offsets,
                           const int8_t* signLeft);
int x265hip_call_sao_stats(int depth, int kind, const int16_t* diff, const void* rec, int64_t stride, int8_t* up1, int8_t* upt, int endX, int endY,
                           int32_t* stats, int32_t* count);
int x265hip_call_intra_pred(int depth, int n, int mode, int bFilter, void* dst, int64_t dstStride, const void* line);
int x265hip_call_intra_allangs(int depth, int n, void* dest, const void* line, const void* filtered, int bLuma);
int x265hip_call_intra_filter(int depth, int n, const void* line, void* filtered);
int x265hip_call_frame_init_lowres(int depth, const void* src, int64_t srcStride, void* dst0, void* dstH, void* dstV, void* dstC,
                                   int64_t dstStride, int width, int height);

/* ---------------------------------------------------------------- CU residual quad-tree jobs ----------------------- */
/* One job = the transform arithmetic of one inter CU's residual quad-tree, everything Search::estimateResidualQT (reference
 * source/encoder/search.cpp:3178-3560) asks of Quant::transformNxN (common/quant.cpp:397-470: cu[].dct -> quant -> signBitHidingHDQ
 * :246-395) and Quant::invtransformNxN (:543-603: dequant_normal -> cu[].idct) for the luma transform sizes 32 and 16 the tree may
 * try (4:2:0 chroma 16 and 8), plus the two sse_pp distortions of every unit (search.cpp:3269, :3295).  The caller (x265_amd/host/
 * x265_hip_cuserve.cpp) keeps the entropy coder and every decision.  Jobs travel through mailbox SLOTS in page-locked host memory
 * that the device reads and writes directly: the caller fills the slot's job header and pixel block, submits, and polls the
 * per-unit `ready` / `readyInv` words (each unit's forward half — numSig and levels — is published as soon as it is done, luma first;
 * its inverse half follows).  Flat quantiser only (no scaling lists), no
 * transform skip, no transquant bypass, no noise reduction, no RDOQ: the caller does not submit such CUs.
 *
 * Pixel block: source Y (N x N, N = 1 << log2CUSize), source Cb, Cr (N/2 x N/2 each, 4:2:0; absent when chroma == 0), then the
 * prediction in the same order; rows contiguous; elements uint8_t (bitDepth 8) or uint16_t.
 * Levels: luma transform sizes s from sHi = min(5, log2TrMax, log2CUSize) down to sLo = max(4, log2TrMin); no level when sHi < sLo.
 * Units of a level: plane 0 (Y), then 1 (Cb), 2 (Cr); within a plane raster order of the (N >> s)^2 units. */
typedef struct x265hip_cujob
{
    uint32_t log2CUSize;          /* 4..6 */
    uint32_t log2TrMax, log2TrMin;/* the luma transform sizes the tree may try (Search::estimateResidualQT's depthRange, largest first) */
    uint32_t chroma;              /* 1: 4:2:0 Cb and Cr blocks are part of the job */
    uint32_t bitDepth;            /* 8, 10, 12 */
    uint32_t quantOffset;         /* 171 (I slice) or 85: quant.cpp:466 `add = offset << (qbits - 9)` */
    uint32_t signHide;            /* pps->bSignHideEnabled: signBitHidingHDQ runs on units with numSig >= 2 */
    uint32_t reserved;            /* 0; non-zero: diagnostic stage stamps in the units' reserved words (tools/micro/cuserve_rt) */
    int32_t  qpRem[3], qpPer[3];  /* Quant::m_qpParam[Y, Cb, Cr] */
    int32_t  quantScale[3];       /* the flat m_quantCoef entry of each plane's rem (scalinglist.cpp:386) */
    int32_t  dequantScale[3];     /* s_invQuantScales[rem] (scalinglist.cpp:130); dequant_normal's scale = dequantScale << per */
    uint32_t coefMode;            /* 1: the host quantises (Quant::rdoQuant, quant.cpp:610-1420 — presets slow and slower; its decisions read the entropy coder's
                                   * state).  A unit's `levels` block then receives the TRANSFORM COEFFICIENTS of its residual — cu[].dct's output, what
                                   * Quant::transformNxN leaves in m_resiDctCoeff (quant.cpp:432) — numSig is 0, zeroDist as always, and there is no inverse half:
                                   * codedDist / codedEnergy / the `resi` block of chroma units are not written.  `ready`: the coefficients are in place; `readyInv`:
                                   * so is everything else the unit will get (the luma units' source transform with sourceDct: it may arrive before or after) */
    uint32_t sourceDct;           /* with coefMode: 1 = a LUMA unit's `resi` block receives cu[].dct of the unit's SOURCE pixels — m_fencDctCoeff, which psy-rdoq
                                   * compares the levels' reconstruction against (quant.cpp:436-442) */
} x265hip_cujob;
/* coefMode == X265HIP_CUJOB_INVERSE: the INVERSE half alone, for levels the host has made (Quant::rdoQuant): Quant::invtransformNxN (quant.cpp:543-603:
 * dequant_normal -> cu[].idct) of ONE 32x32 luma unit and the measurements Search::estimateResidualQT takes behind it.  log2CUSize = log2TrMax = 5, chroma = 0;
 * pixel block: the unit's source block, its prediction, then its 1 024 levels (int16, rows contiguous).  Results in units[0]: numSig (non-zero levels counted),
 * zeroDist, codedDist, codedEnergy and the unit's `resi` block; readyInv, then ready. */
#define X265HIP_CUJOB_INVERSE 8u
typedef struct x265hip_cujob_unit
{
    uint32_t ready;               /* == the job's ticket once this unit's numSig, zeroDist and levels are in place (the forward half) */
    uint32_t numSig;              /* transformNxN's return value (after sign-bit hiding) */
    uint64_t zeroDist;            /* sse_pp(source, prediction) */
    uint64_t codedDist;           /* sse_pp(source, clip(prediction + reconstructed residual)); defined when numSig != 0 */
    uint32_t readyInv;            /* == the job's ticket once codedDist and the reconstructed residual are in place (the inverse half; coefMode: set with ready) */
    uint32_t fwdTicks;            /* diagnostic: 100 MHz device ticks from the job's start to this unit's forward half */
    uint32_t codedEnergy;         /* psy_cost_pp(source, clip(prediction + reconstructed residual)) (reference common/pixel.cpp:726-748); with readyInv, when numSig != 0 */
    uint32_t reserved[3];         /* diagnostic (job.reserved != 0): 16-bit 100 MHz ticks since the job's start, low | high half: [0] chain starts | forward transform
                                   * done, [1] quantised | sign hiding done, [2] inverse transform done | readyInv issued */
} x265hip_cujob_unit;
#define X265HIP_CUJOB_MAX_UNITS   60                       /* 64x64, sizes 32 + 16: 3 * (4 + 16) */
#define X265HIP_CUJOB_MAX_ELEMS   (2 * 6144)               /* int16 entries of `levels` (and of `resi`) of the largest job */
#define X265HIP_CUJOB_PIXEL_BYTES (2 * 6144 * 2)           /* source + prediction of a 64x64 4:2:0 CU at 16 bit */
/* inline layout helpers (x265hipi_*: not exported symbols) shared by the library, the bindings and the checkers */
#if defined(__HIPCC__)
#define X265HIP_HD __host__ __device__
#else
#define X265HIP_HD
#endif
X265HIP_HD static inline int x265hipi_cujob_levels(const x265hip_cujob* j, int* sHi, int* sLo)
{
    int hi = j->log2TrMax < 5 ? (int)j->log2TrMax : 5, lo = j->log2TrMin > 4 ? (int)j->log2TrMin : 4;
    if (hi > (int)j->log2CUSize) hi = (int)j->log2CUSize;
    *sHi = hi; *sLo = lo;
    return hi - lo + 1;
}
/* index of unit (s, plane, tuX, tuY) in the slot's `units` array; s = log2 of the LUMA transform size of the level */
X265HIP_HD static inline int x265hipi_cujob_unit_index(const x265hip_cujob* j, int sHi, int s, int plane, int tuX, int tuY)
{
    const int planes = j->chroma ? 3 : 1;
    int base = 0;
    for (int k = sHi; k > s; k--)
        base += planes << (2 * ((int)j->log2CUSize - k));
    const int n = 1 << ((int)j->log2CUSize - s);
    return base + plane * n * n + tuY * n + tuX;
}
/* offset (int16 entries) of the same unit's block in the slot's `levels` and `resi` arrays: (size of the unit)^2 entries, rows contiguous */
X265HIP_HD static inline int x265hipi_cujob_elem_offset(const x265hip_cujob* j, int sHi, int s, int plane, int tuX, int tuY)
{
    const int N2 = 1 << (2 * (int)j->log2CUSize), n = 1 << ((int)j->log2CUSize - s);
    const int perLevel = j->chroma ? N2 + N2 / 2 : N2;
    const int t = tuY * n + tuX;
    int off = (sHi - s) * perLevel;
    if (plane == 0) return off + (t << (2 * s));
    off += N2 + (plane == 2 ? N2 / 4 : 0);
    return off + (t << (2 * (s - 1)));
}
typedef struct x265hip_cuserve x265hip_cuserve;
/* mode 0: a resident server kernel per slot group polls the slots' doorbells (started on demand, ends by itself after `idle
 * microseconds` without work so that nothing in the process ever waits on it for long); mode 1: one launch per job on the slot's own
 * stream.  Either way completion is signalled through the units' `ready` words, never through a stream synchronisation. */
int x265hip_cuserve_open(int slots, int mode, x265hip_cuserve** out);
/* the same service at place `place` (x265hip_places): its slots are served by that place's device */
int x265hip_cuserve_open_at(int place, int slots, int mode, x265hip_cuserve** out);
int x265hip_cuserve_close(x265hip_cuserve* cs);
/* the slot's memory: the caller fills *job (ordinary memory: it may read it back; submit copies it into the mailbox) and WRITES *pixels, front to
 * back — when the device has a large BAR that block is device memory mapped write-combined into the process (the server then reads the job from its
 * own HBM instead of over PCIe; X265HIP_CUSERVE_MAILBOX=host|device overrides the choice), so never read it — and reads units / levels / resi
 * (page-locked host memory the device writes) */
int x265hip_cuserve_slot(x265hip_cuserve* cs, int slot, x265hip_cujob** job, void** pixels, const x265hip_cujob_unit** units,
                         const int16_t** levels, const int16_t** resi);
/* hands the slot's job to the device; *seq = the ticket: the value the units' `ready` / `readyInv` words take (unique among the slot's recent jobs) */
int x265hip_cuserve_submit(x265hip_cuserve* cs, int slot, uint32_t* seq);
/* to be called now and then by a caller that is still waiting (mode 0: restarts a server that has gone idle meanwhile).  Returns 0: wait on;
 * 1: wait on, and do not count this time against your timeout — the servers are paused for a device synchronisation or the server is still on its
 * way onto the chip; X265HIP_EHIP (negative): the device reported a failure, the caller gives up the job and computes on the host */
int x265hip_cuserve_poke(x265hip_cuserve* cs, int slot);
int x265hip_cuserve_stats(x265hip_cuserve* cs, uint64_t* jobs, uint64_t* serverStarts, uint64_t* deviceNs);

/* ---- SAO statistics of one CTU as a job of the same service (round 5) ---------------------------------------------------------------------
 * SAO::calcSaoStatsCTU (reference source/encoder/sao.cpp:735-917; called per plane from rdoSaoUnitCu :1293-1305) measures, for the deblocked CTU, the
 * sums and counts of (source - reconstruction) per band (SAO_BO, saoCuStatsBO_c :1762) and per edge category of the four edge classes (saoCuStatsE0..E3_c
 * :1780-1925): a function of the CTU's deblocked samples (with one row above and one column to the left), its source samples and, per class, the
 * rectangle of samples the reference measures (what is not deblocked yet at the right / bottom is left out, :778-899).  The caller computes the
 * rectangles exactly as the reference does and hands them over; the device measures all classes of all planes in one job, luma first.
 *
 * The slot's pixel block, per plane p (0 .. planes-1) in turn: the reconstruction block, (h + 1) rows of (w + 1) samples starting at the sample ABOVE-LEFT
 * of the CTU's first one (row pitch w + 1), then the source block, h rows of w samples (row pitch w).  8-bit samples only (bitDepth 8).
 * Results: the slot's `levels` block read as int32: stats[plane][class][32] (3 * 5 * 32 entries) followed by count[plane][class][32]; class order
 * BO, EO_0, EO_1, EO_2, EO_3; an edge class fills entries 0..4 (the category order of SAO::s_eoTable: the values saoCuStatsE*_c add into their `stats`).
 * units[p].ready == the ticket when plane p's numbers are in place. */
typedef struct x265hip_saojob
{
    uint32_t bitDepth;            /* 8 */
    uint32_t planes;              /* 1 or 3 */
    uint32_t eo23;                /* 0: EO_2 and EO_3 are not measured (--limit-sao, sao.cpp:853-854): their entries are zero */
    uint32_t reserved;
    struct
    {
        uint16_t w, h;            /* the CTU's part of the plane */
        uint8_t  x0[5], y0[5], x1[5], y1[5];   /* per class: samples [x0, x1) x [y0, y1) are measured */
    } plane[3];
} x265hip_saojob;
#define X265HIP_SAOJOB_STATS_ENTRIES (3 * 5 * 32)
/* bytes of the pixel block the job describes */
X265HIP_HD static inline int x265hipi_saojob_pixel_bytes(const x265hip_saojob* j)
{
    int n = 0;
    for (uint32_t p = 0; p < j->planes && p < 3; p++)
        n += (j->plane[p].w + 1) * (j->plane[p].h + 1) + j->plane[p].w * j->plane[p].h;
    return n;
}
/* hands the job to the device: `job` is copied into the mailbox in front of the pixel block the caller has written (x265hip_cuserve_slot's *pixels);
 * *seq = the ticket */
int x265hip_cuserve_submit_sao(x265hip_cuserve* cs, int slot, const x265hip_saojob* job, uint32_t* seq);

/* Intra mode scan as a job of the same service: the distortion half of Search::checkIntraInInter (search.cpp:1291-1452) for ONE block —
 * sa8d(source block, prediction of mode m) for all 35 modes, exactly the calls the reference makes (DC edge-smoothed when N <= 16, planar from the
 * filtered line when N >= 8, each angle from the raw or the filtered line as g_intraFilterFlags says, the mode 10 / 26 edge gradient when N <= 16;
 * cu[].sa8d: sa8d_8x8, one rounding per 16x16 above).  The arithmetic of x265hip_intra_scan_batch, one block per job, the round trip of a CU job.
 * Pixel block (samples: bytes at 8 bit, uint16_t above): the unfiltered neighbour line in the reference's array layout (Predict::initAdiPattern:
 * [0] corner, [1..2N] top + top-right, [2N+1..4N] left + bottom-left) at sample 0, the [1 2 1] filtered line at sample 4N + 16, the N x N source
 * block (stride N) at sample 2 * (4N + 16).  Results: units[0].ready = units[0].readyInv = the ticket; the 35 costs as int32 at the start of
 * the slot's `levels` block (cost of mode m at index m). */
#define X265HIP_INTRAJOB_MARK 0x100u
typedef struct x265hip_intrajob
{
    uint32_t bitDepth;      /* 8, 10 or 12 */
    uint32_t mark;          /* X265HIP_INTRAJOB_MARK: where an SAO statistics job holds its plane count (both kinds travel under the same ticket class) */
    uint32_t log2Size;      /* 3, 4 or 5 */
    uint32_t reserved;
} x265hip_intrajob;
X265HIP_HD static inline int x265hipi_intrajob_line_samples(int log2Size) { return (4 << log2Size) + 16; }
X265HIP_HD static inline int x265hipi_intrajob_pixel_bytes(const x265hip_intrajob* j)
{
    return (2 * x265hipi_intrajob_line_samples((int)j->log2Size) + (1 << (2 * j->log2Size))) * (j->bitDepth > 8 ? 2 : 1);
}
int x265hip_cuserve_submit_intra(x265hip_cuserve* cs, int slot, const x265hip_intrajob* job, uint32_t* seq);

#ifdef __cplusplus
}
#endif
#endif /* X265HIP_H */
