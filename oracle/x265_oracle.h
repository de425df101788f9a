/* x265_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the x265 C reference primitives on the hot path
 * plus MotionEstimate::motionEstimate.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library; the product path (x265_amd/, libx265hip.so) never does.
 *
 * Parity status: PINNED — every function here is checked bit-for-bit against the real reference
 * (oracle/_ref/libx265ref{8,10,12}.so, compiled from /root/reference/source by oracle/Makefile) in
 * tests/test_oracle_vs_ref.py, and against digests of reference outputs committed under tests/golden/.
 *
 * Conventions: strides are in ELEMENTS (as in x265). `depth` is the internal bit depth (8, 10 or 12);
 * pixel storage is uint8_t when depth == 8 (functions suffixed _8) and uint16_t otherwise (_16).
 * Each declaration cites the reference function it restates (paths relative to /root/reference/source).
 */
#ifndef X265_ORACLE_H
#define X265_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_FENC_STRIDE 64 /* common/common.h:70 */

/* ---- pixel comparisons -------------------------------------------------------------------------------- */
#define ORC_DECL_PIXEL(P, SFX) \
/* common/pixel.cpp:40 sad<lx,ly> */ \
int orc_sad_##SFX(const P* a, intptr_t sa, const P* b, intptr_t sb, int w, int h); \
/* common/pixel.cpp:74 sad_x3<lx,ly>: fenc stride fixed at FENC_STRIDE */ \
void orc_sad_x3_##SFX(const P* fenc, const P* r0, const P* r1, const P* r2, intptr_t rs, int w, int h, int32_t* res); \
/* common/pixel.cpp:96 sad_x4<lx,ly> */ \
void orc_sad_x4_##SFX(const P* fenc, const P* r0, const P* r1, const P* r2, const P* r3, intptr_t rs, int w, int h, int32_t* res); \
/* common/pixel.cpp:210 satd_4x4, :239 satd_8x4, :263-297 satd4/satd8 tilers, table :1131-1155 */ \
int orc_satd_##SFX(const P* a, intptr_t sa, const P* b, intptr_t sb, int w, int h); \
/* common/pixel.cpp:299-376: cu[].sa8d — 4x4=satd_4x4, 8x8=sa8d_8x8, >=16 sum of sa8d_16x16 (table :1163-1167) */ \
int orc_sa8d_##SFX(const P* a, intptr_t sa, const P* b, intptr_t sb, int size); \
/* common/pixel.cpp:352 sa8d8<w,h> (chroma tables :1258, :1343): each 8x8 rounded separately */ \
int orc_sa8d8_##SFX(const P* a, intptr_t sa, const P* b, intptr_t sb, int w, int h); \
/* common/pixel.cpp:167 sse<lx,ly,pixel,pixel> */ \
uint64_t orc_sse_pp_##SFX(const P* a, intptr_t sa, const P* b, intptr_t sb, int w, int h); \
/* common/pixel.cpp:726 psyCost_pp<size> (size = log2(dim)-2) */ \
int orc_psy_cost_pp_##SFX(const P* src, intptr_t ss, const P* rec, intptr_t rs, int dim); \
/* common/pixel.cpp:704 pixel_var<size> */ \
uint64_t orc_var_##SFX(const P* p, intptr_t s, int size); \
/* common/pixel.cpp:518 weight_pp_c, :493 weight_sp_c */ \
void orc_weight_pp_##SFX(const P* src, P* dst, intptr_t stride, int width, int height, int w0, int round, int shift, int offset, int depth); \
void orc_weight_sp_##SFX(const int16_t* src, P* dst, intptr_t ss, intptr_t ds, int width, int height, int w0, int round, int shift, int offset, int depth); \
/* common/pixel.cpp:559 scale1D_128to64, :585 scale2D_64to32, :485 transpose<size> */ \
void orc_scale1d_128to64_##SFX(P* dst, const P* src); \
void orc_scale2d_64to32_##SFX(P* dst, const P* src, intptr_t stride); \
void orc_transpose_##SFX(P* dst, const P* src, intptr_t stride, int size); \
/* common/pixel.cpp:815 pixel_sub_ps_c */ \
void orc_sub_ps_##SFX(int16_t* d, intptr_t ds, const P* a, const P* b, intptr_t sa, intptr_t sb, int w, int h); \
/* common/pixel.cpp:829 pixel_add_ps_c */ \
void orc_add_ps_##SFX(P* d, intptr_t ds, const P* a, const int16_t* r, intptr_t sa, intptr_t sr, int w, int h, int depth); \
/* common/pixel.cpp:469 getResidual<blockSize>: one stride for all three */ \
void orc_calcresidual_##SFX(const P* fenc, const P* pred, int16_t* resi, intptr_t stride, int size); \
/* common/pixel.cpp:842 addAvg<bx,by> */ \
void orc_addAvg_##SFX(const int16_t* a, const int16_t* b, P* d, intptr_t sa, intptr_t sb, intptr_t ds, int w, int h, int depth); \
/* common/pixel.cpp:545 pixelavg_pp<lx,ly> */ \
void orc_pixelavg_pp_##SFX(P* d, intptr_t ds, const P* a, intptr_t sa, const P* b, intptr_t sb, int w, int h); \
/* common/pixel.cpp:759 blockcopy_pp_c, :785 blockcopy_sp_c, :802 blockcopy_ps_c */ \
void orc_copy_pp_##SFX(P* d, intptr_t ds, const P* s, intptr_t ss, int w, int h); \
void orc_copy_sp_##SFX(P* d, intptr_t ds, const int16_t* s, intptr_t ss, int w, int h); \
void orc_copy_ps_##SFX(int16_t* d, intptr_t ds, const P* s, intptr_t ss, int w, int h); \
/* ---- interpolation (N = 8 luma / 4 chroma taps) ---- */ \
/* common/ipfilter.cpp:79 interp_horiz_pp_c */ \
void orc_interp_horiz_pp_##SFX(int N, const P* src, intptr_t ss, P* dst, intptr_t ds, int w, int h, int coeffIdx, int depth); \
/* common/ipfilter.cpp:120 interp_horiz_ps_c */ \
void orc_interp_horiz_ps_##SFX(int N, const P* src, intptr_t ss, int16_t* dst, intptr_t ds, int w, int h, int coeffIdx, int isRowExt, int depth); \
/* common/ipfilter.cpp:164 interp_vert_pp_c */ \
void orc_interp_vert_pp_##SFX(int N, const P* src, intptr_t ss, P* dst, intptr_t ds, int w, int h, int coeffIdx, int depth); \
/* common/ipfilter.cpp:205 interp_vert_ps_c */ \
void orc_interp_vert_ps_##SFX(int N, const P* src, intptr_t ss, int16_t* dst, intptr_t ds, int w, int h, int coeffIdx, int depth); \
/* common/ipfilter.cpp:241 interp_vert_sp_c */ \
void orc_interp_vert_sp_##SFX(int N, const int16_t* src, intptr_t ss, P* dst, intptr_t ds, int w, int h, int coeffIdx, int depth); \
/* common/ipfilter.cpp:362 interp_hv_pp_c (hps with row extension, then vertical sp) */ \
void orc_interp_hv_pp_##SFX(int N, const P* src, intptr_t ss, P* dst, intptr_t ds, int w, int h, int idxX, int idxY, int depth); \
/* common/ipfilter.cpp:40 filterPixelToShort_c */ \
void orc_p2s_##SFX(const P* src, intptr_t ss, int16_t* dst, intptr_t ds, int w, int h, int depth); \
/* ---- motion estimation: encoder/motion.cpp:739 MotionEstimate::motionEstimate (non-lowres, no chroma SATD) \
 * plane/stride: full-pel reference luma plane (must have >= merange+8 valid margin around the block); \
 * fenc: PU pixels at stride FENC_STRIDE; (bx,by): PU position in the plane; cost: u16 table centred on MVD 0 \
 * (orc_mvcost_table); method: 0 DIA, 1 HEX, 2 UMH(unsupported), 3 STAR, 5 FULL (x265.h X265_*_SEARCH). \
 * Returns bcost, writes the quarter-pel MV. */ \
int orc_motion_estimate_##SFX(const P* plane, intptr_t stride, int bx, int by, const P* fenc, int w, int h, \
                              const int32_t mvmin[2], const int32_t mvmax[2], const int32_t qmvp[2], \
                              int numCand, const int32_t* mvc, int merange, int method, int subme, \
                              const uint16_t* cost, int depth, int32_t outQMv[2]); \
/* encoder/motion.cpp:739 with the chroma SATD term of subpelCompare (:1601-1660) as Search::predInterSearch runs it on 4:2:0 */ \
int orc_motion_estimate_chroma_##SFX(const P* plane, intptr_t stride, const P* cbPlane, const P* crPlane, intptr_t strideC, int bx, int by, \
                            const P* fenc, const P* fencCb, const P* fencCr, int w, int h, \
                            const int32_t mvmin[2], const int32_t mvmax[2], const int32_t qmvp[2], \
                            int numCand, const int32_t* mvc, int merange, int method, int subme, \
                            const uint16_t* cost, int depth, int32_t outQMv[2]); \
/* encoder/motion.cpp:1571 MotionEstimate::subpelCompare (luma only); cmp: 0 = sad, 1 = satd */ \
int orc_subpel_compare_##SFX(const P* plane, intptr_t stride, int bx, int by, const P* fenc, int w, int h, \
                             int qmvx, int qmvy, int cmp, int depth);

ORC_DECL_PIXEL(uint8_t, 8)
ORC_DECL_PIXEL(uint16_t, 16)

/* ---- frame pass (x265_oracle_frame.c): the pipeline libx265hip's x265hip_framepass_run executes, restated on the CPU
 * from the pinned primitives above.  Checker for the GPU frame pass and bench.py's cpu_baseline ("port"). */
/* encoder/search.cpp:2724 Search::setSearchRange + common/cudata.cpp:1915 CUData::clipMv */
void orc_set_search_range(int picW, int picH, int maxCUSize, int merange, int refLagPixels, int cuX, int cuY,
                          const int32_t qmvp[2], int32_t mvmin[2], int32_t mvmax[2]);
#define ORC_DECL_LOOKAHEAD(P, SFX) \
/* encoder/slicetype.cpp:3218 CostEstimateGroup::estimateCUCost over a whole P frame (lowres motionEstimate: motion.cpp:775, :1471) */ \
int64_t orc_lookahead_cost_p_##SFX(const P* fencPlane, const P* const* ref, intptr_t stride, int widthInCU, int heightInCU, \
                                   int numRowsPerSlice, int numSlices, int depth, const int32_t* intraCost, const uint16_t* mvcost, \
                                   int32_t* mvs, int32_t* mvCosts, uint16_t* lowresCosts, int32_t* rowSatds, int32_t* intraMbs); \
/* --me sea (motion.cpp:1242-1395): window-sum planes (framefilter.cpp:684-830) + the search */ \
void orc_integral_plane_##SFX(const P* buf, intptr_t stride, int rows, int w, int h, uint32_t* out); \
int orc_motion_estimate_sea_##SFX(const P* plane, intptr_t stride, const uint32_t* const* integral, int bx, int by, const P* fenc, int w, int h, \
                                  const int32_t mvmin[2], const int32_t mvmax[2], const int32_t qmvp[2], int numCand, const int32_t* mvc, \
                                  int merange, int subme, const uint16_t* cost, int depth, int32_t outQMv[2]); \
/* the two passes with the AQ-scaled sums (slicetype.cpp:3362-3384) and, for P, the reuse of a stored search (:3260-3264) */ \
int64_t orc_lookahead_cost_p_aq_##SFX(const P* fencPlane, const P* const* ref, intptr_t stride, int widthInCU, int heightInCU, \
                                      int numRowsPerSlice, int numSlices, int depth, const int32_t* intraCost, const uint16_t* mvcost, \
                                      int32_t* mvs, int32_t* mvCosts, uint16_t* lowresCosts, int32_t* rowSatds, int32_t* intraMbs, \
                                      const int32_t* invQscale, int doSearch, int64_t* costEstAq); \
int64_t orc_lookahead_cost_b_aq_##SFX(const P* fencPlane, const P* const* ref0, const P* const* ref1, intptr_t stride, int widthInCU, int heightInCU, \
                                      int numRowsPerSlice, int numSlices, int depth, const uint16_t* mvcost, const int32_t doSearch[2], \
                                      int32_t* mvs0, int32_t* mvCosts0, int32_t* mvs1, int32_t* mvCosts1, uint16_t* lowresCosts, int32_t* rowSatds, \
                                      const int32_t* invQscale, int64_t* costEstAq); \
/* the same for a B frame (p0 < b < p1): two lists + the two bi-predictive candidates (slicetype.cpp:3254-3350) */ \
int64_t orc_lookahead_cost_b_##SFX(const P* fencPlane, const P* const* ref0, const P* const* ref1, intptr_t stride, int widthInCU, int heightInCU, \
                                   int numRowsPerSlice, int numSlices, int depth, const uint16_t* mvcost, const int32_t doSearch[2], \
                                   int32_t* mvs0, int32_t* mvCosts0, int32_t* mvs1, int32_t* mvCosts1, uint16_t* lowresCosts, int32_t* rowSatds);
ORC_DECL_LOOKAHEAD(uint8_t, 8)
ORC_DECL_LOOKAHEAD(uint16_t, 16)
#define ORC_DECL_FRAME(P, SFX) \
/* common/predict.cpp:245 Predict::predInterLumaPixel */ \
void orc_pred_inter_luma_##SFX(const P* ref, intptr_t rs, P* dst, intptr_t ds, int bx, int by, int w, int h, int qx, int qy, int depth); \
/* common/pixel.cpp:1027 extendPicBorder */ \
void orc_extend_border_##SFX(P* pic, intptr_t stride, int picW, int picH, int mx, int my); \
void orc_frame_pass_##SFX(int width, int height, int depth, int qp, int merange, int method, int subme, \
                          const P* src, intptr_t ss, const P* ref, intptr_t rs, P* pred, intptr_t ps, P* recon, intptr_t cs, \
                          int marginX, int marginY, int32_t* mv[4], int32_t* mecost[4], int32_t* sa8d[4], \
                          int16_t* level[2], uint32_t* numSig[2], uint64_t* dist[2], \
                          const P* const srcC[2], const P* const refC[2], P* const predC[2], P* const reconC[2], \
                          intptr_t ssC, intptr_t rsC, intptr_t psC, intptr_t csC, \
                          int16_t* clevel[4], uint32_t* cnumSig[4], uint64_t* cdist[4], \
                          const P* ref1, const P* const ref1C[2], int32_t* mv1[4], int32_t* mecost1[4]); \
/* common/predict.cpp:131-199 bi-predictive motion compensation: predInterLumaShort / predInterChromaShort of both lists + addAvg */ \
void orc_pred_inter_bi_##SFX(const P* const ref0[3], const P* const ref1[3], intptr_t rs, intptr_t rsC, int bx, int by, int w, int h, \
                             const int32_t mv0[2], const int32_t mv1[2], P* dstY, intptr_t dsY, P* dstCb, P* dstCr, intptr_t dsC, int depth); \
/* common/predict.cpp:306 Predict::predInterChromaPixel (4:2:0) */ \
void orc_pred_inter_chroma_##SFX(const P* ref, intptr_t rs, P* dst, intptr_t ds, int bx, int by, int lumaW, int lumaH, int qx, int qy, int depth);
ORC_DECL_FRAME(uint8_t, 8)
ORC_DECL_FRAME(uint16_t, 16)

/* ---- intra prediction, lowres init, lookahead intra estimate (x265_oracle_intra.c) ---------------------------- */
#define ORC_DECL_INTRA(P, SFX) \
/* common/intrapred.cpp:32 intraFilter<N> */ \
void orc_intra_filter_##SFX(int n, const P* nb, P* out); \
/* common/intrapred.cpp:57-222 intra_pred[35]: planar (0), DC (1), angular 2..34 */ \
void orc_intra_pred_##SFX(int n, int mode, P* dst, intptr_t ds, const P* nb, int bFilter, int depth); \
/* common/constants.cpp:561 g_intraFilterFlags[mode] & n */ \
int orc_intra_uses_filtered_##SFX(int n, int mode); \
/* common/intrapred.cpp:224 all_angs_pred_c */ \
void orc_intra_allangs_##SFX(int n, P* dest, const P* nb, const P* nbFiltered, int bLuma, int depth); \
/* common/pixel.cpp:604 frame_init_lowres_core */ \
void orc_frame_init_lowres_##SFX(const P* src, P* d0, P* dh, P* dv, P* dc, intptr_t ss, intptr_t dstStride, int width, int height); \
/* encoder/slicetype.cpp:696 LookaheadTLD::lowresIntraEstimate */ \
int orc_lowres_intra_estimate_##SFX(const P* plane, intptr_t stride, int widthInCU, int heightInCU, int depth, \
                                    int32_t* intraCost, uint8_t* intraMode, int32_t* rowSatd);
ORC_DECL_INTRA(uint8_t, 8)
ORC_DECL_INTRA(uint16_t, 16)

/* ---- int16 block helpers (pixel-type independent) ------------------------------------------------------ */
/* common/pixel.cpp:167 sse<..,int16_t,int16_t> */
uint64_t orc_sse_ss(const int16_t* a, intptr_t sa, const int16_t* b, intptr_t sb, int w, int h);
/* common/pixel.cpp:379 pixel_ssd_s_c */
uint64_t orc_ssd_s(const int16_t* a, intptr_t sa, int size);
/* common/pixel.cpp:393 blockfill_s_c */
void orc_blockfill_s(int16_t* d, intptr_t ds, int16_t val, int size);
/* common/pixel.cpp:401-467 cpy2Dto1D_shl/shr, cpy1Dto2D_shl/shr */
void orc_cpy2Dto1D_shl(int16_t* d, const int16_t* s, intptr_t ss, int shift, int size);
void orc_cpy2Dto1D_shr(int16_t* d, const int16_t* s, intptr_t ss, int shift, int size);
void orc_cpy1Dto2D_shl(int16_t* d, const int16_t* s, intptr_t ds, int shift, int size);
void orc_cpy1Dto2D_shr(int16_t* d, const int16_t* s, intptr_t ds, int shift, int size);
/* common/pixel.cpp:772 blockcopy_ss_c */
void orc_copy_ss(int16_t* d, intptr_t ds, const int16_t* s, intptr_t ss, int w, int h);
/* common/dct.cpp:714 count_nonzero_c, :728 copy_count */
int orc_count_nonzero(const int16_t* q, int size);
uint32_t orc_copy_cnt(int16_t* coeff, const int16_t* resi, intptr_t rs, int size);
/* common/ipfilter.cpp:284 interp_vert_ss_c */
void orc_interp_vert_ss(int N, const int16_t* src, intptr_t ss, int16_t* dst, intptr_t ds, int w, int h, int coeffIdx);

/* ---- transforms ---------------------------------------------------------------------------------------- */
/* common/dct.cpp:459-525 dct4_c..dct32_c (log2n = 2..5); src has stride, dst is contiguous N*N */
void orc_dct(int log2n, const int16_t* src, int16_t* dst, intptr_t srcStride, int depth);
/* common/dct.cpp:544-610 idct4_c..idct32_c; src contiguous N*N, dst has stride */
void orc_idct(int log2n, const int16_t* src, int16_t* dst, intptr_t dstStride, int depth);
/* common/dct.cpp:442 dst4_c / :527 idst4_c */
void orc_dst4(const int16_t* src, int16_t* dst, intptr_t srcStride, int depth);
void orc_idst4(const int16_t* src, int16_t* dst, intptr_t dstStride, int depth);
/* common/dct.cpp:664 quant_c */
uint32_t orc_quant(const int16_t* coef, const int32_t* quantCoeff, int32_t* deltaU, int16_t* qCoef, int qBits, int add, int numCoeff);
/* common/dct.cpp:688 nquant_c */
uint32_t orc_nquant(const int16_t* coef, const int32_t* quantCoeff, int16_t* qCoef, int qBits, int add, int numCoeff);
/* common/dct.cpp:612 dequant_normal_c */
void orc_dequant_normal(const int16_t* q, int16_t* coef, int num, int scale, int shift);
/* common/dct.cpp:636 dequant_scaling_c */
void orc_dequant_scaling(const int16_t* q, const int32_t* deQuantCoef, int16_t* coef, int num, int per, int shift);
/* common/dct.cpp:743 denoiseDct_c */
void orc_denoise_dct(int16_t* dctCoef, uint32_t* resSum, const uint16_t* offset, int numCoeff);
/* common/dct.cpp:985 nonPsyRdoQuant_c, :1005 psyRdoQuant_c, :1030 psyRdoQuant_c_1, :1049 psyRdoQuant_c_2 */
void orc_nonpsy_rdoquant(int log2n, const int16_t* resiDct, int64_t* costUncoded, int64_t* totalUncoded, int64_t* totalRd, uint32_t blkPos, int depth);
void orc_psy_rdoquant(int log2n, const int16_t* resiDct, const int16_t* fencDct, int64_t* costUncoded, int64_t* totalUncoded, int64_t* totalRd, const int64_t* psyScale, uint32_t blkPos, int depth);
void orc_psy_rdoquant_1p(int log2n, const int16_t* resiDct, int64_t* costUncoded, int64_t* totalUncoded, int64_t* totalRd, uint32_t blkPos, int depth);
void orc_psy_rdoquant_2p(int log2n, const int16_t* resiDct, const int16_t* fencDct, int64_t* costUncoded, int64_t* totalUncoded, int64_t* totalRd, const int64_t* psyScale, uint32_t blkPos, int depth);

/* ---- data tables ---------------------------------------------------------------------------------------- */
/* common/constants.cpp:250 g_lumaFilter, :258 g_chromaFilter, :270-344 g_t4..g_t32 (regenerated, not pasted) */
const int16_t* orc_luma_filter(int idx);     /* 8 taps */
const int16_t* orc_chroma_filter(int idx);   /* 4 taps */
const int16_t* orc_dct_matrix(int log2n);    /* N*N row-major */
/* encoder/bitcost.cpp:32 BitCost::setQP + :108 CalculateLogs: table[i + 2*32768] = cost of MVD component i (qpel),
 * i in [-65536, 65536]; lambda = x265_lambda_tab[qp] for the given depth (common/constants.cpp:34/74/114).
 * `table` must hold 4*32768+1 entries. */
void orc_mvcost_table(int qp, int depth, uint16_t* table);
/* common/primitives.cpp lumaPartitionMapTable via partitionFromSizes (primitives.h:435): LumaPU enum or -1 */
int orc_partition_from_sizes(int w, int h);
/* per-call traffic accounting of orc_motion_estimate (SURVEY.md §8d figures); out = { bytes, primitive calls } */
void orc_me_stats_reset(void);
void orc_me_stats(uint64_t out[2]);
/* per motion-search level of the last orc_frame_pass_*: { bytes, primitive calls, PUs } x 4 */
void orc_frame_pass_me_stats(uint64_t out[12]);

#ifdef __cplusplus
}
#endif
#endif
