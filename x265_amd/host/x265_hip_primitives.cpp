// x265_hip_primitives.cpp — the ONE translation unit a maintainer adds to an x265 build to run the hot-path slots of
// `EncoderPrimitives` on an MI355X through libx265hip.so (C ABI: include/x265hip.h).
//
// It is compiled INSIDE the x265 tree (it includes x265's own common.h / primitives.h, so it always sees the exact
// struct layout, pixel type and X265_NS of that build) and defines the two symbols x265 expects from its assembly
// directory (reference: source/common/primitives.h:469-470, called from primitives.cpp:260-265 and from the TestBench,
// test/testbench.cpp:191,211):
//     void setupInstrinsicPrimitives(EncoderPrimitives& p, int cpuMask);     // (sic) — no-op here
//     void setupAssemblyPrimitives(EncoderPrimitives& p, int cpuMask);       // overwrites the slots listed below
// plus the cpu-a.asm helpers the ENABLE_ASSEMBLY x86 build references (primitives.cpp:288-303).
//
// Every shim is a synchronous call on caller-owned host memory, exactly the slot's signature; it forwards to the
// per-call entry point x265hip_call_* and, if the GPU path reports an error, to the reference's own C implementation
// (a private table filled by setupCPrimitives) — never garbage, never abort (SURVEY.md §8b "Errors").
// `cpuMask` is ignored: the GPU path is selected by the environment variables X265HIP (default on, "0" disables everything) and
// X265HIP_TABLE (see setupAssemblyPrimitives).
#include "common.h"
#include "primitives.h"
#include "constants.h"
#include "contexts.h"
#include "x265hip.h"
#include "x265_hip_debug.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <atomic>
#include <mutex>
#include <thread>
#include <unistd.h>

namespace X265_NS {

static EncoderPrimitives g_c;          // the C reference table, for fallback
static bool g_cReady = false;

static void ensure_c_table()
{
    static std::mutex once;                         // encoders opened at the same time both come through x265_setup_primitives
    std::lock_guard<std::mutex> g(once);
    if (!g_cReady)
    {
        memset(&g_c, 0, sizeof(g_c));
        setupCPrimitives(g_c);
        setupAliasPrimitives(g_c);
        g_cReady = true;
    }
}

// the reference's C table, built here (setupCPrimitives + setupAliasPrimitives) rather than copied from whatever table is being set up: the other
// binding TUs take their "what the slot did before" functions from it, complete and free of anybody's wrappers
const EncoderPrimitives& x265hip_c_table()
{
    ensure_c_table();
    return g_c;
}

#define D X265_DEPTH

// ---- pixel comparisons (primitives.h:133-140) ----------------------------------------------------------------------
template <int W, int H, int PART> static int sad_hip(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb)
{
    int32_t r;
    if (x265hip_call_pixcmp(X265HIP_CMP_SAD, D, W, H, a, sa, b, sb, &r)) return g_c.pu[PART].sad(a, sa, b, sb);
    return r;
}
template <int W, int H, int PART> static int satd_hip(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb)
{
    int32_t r;
    if (x265hip_call_pixcmp(X265HIP_CMP_SATD, D, W, H, a, sa, b, sb, &r)) return g_c.pu[PART].satd(a, sa, b, sb);
    return r;
}
template <int W, int H, int PART> static void sad_x3_hip(const pixel* f, const pixel* r0, const pixel* r1, const pixel* r2, intptr_t rs, int32_t* res)
{
    const void* refs[3] = { r0, r1, r2 };
    if (x265hip_call_sad_xn(3, D, W, H, f, refs, rs, res)) g_c.pu[PART].sad_x3(f, r0, r1, r2, rs, res);
}
template <int W, int H, int PART> static void sad_x4_hip(const pixel* f, const pixel* r0, const pixel* r1, const pixel* r2, const pixel* r3, intptr_t rs, int32_t* res)
{
    const void* refs[4] = { r0, r1, r2, r3 };
    if (x265hip_call_sad_xn(4, D, W, H, f, refs, rs, res)) g_c.pu[PART].sad_x4(f, r0, r1, r2, r3, rs, res);
}
template <int N, int CU> static int sa8d_hip(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb)
{
    int32_t r;
    if (x265hip_call_pixcmp(X265HIP_CMP_SA8D, D, N, N, a, sa, b, sb, &r)) return g_c.cu[CU].sa8d(a, sa, b, sb);
    return r;
}
template <int N, int CU> static int psy_hip(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb)
{
    int32_t r;
    if (x265hip_call_pixcmp(X265HIP_CMP_PSY, D, N, N, a, sa, b, sb, &r)) return g_c.cu[CU].psy_cost_pp(a, sa, b, sb);
    return r;
}
template <int N, int CU> static sse_t sse_pp_hip(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb)
{
    uint64_t r;
    if (x265hip_call_sse_pp(D, N, N, a, sa, b, sb, &r)) return g_c.cu[CU].sse_pp(a, sa, b, sb);
    return (sse_t)r;
}
template <int N, int CU> static sse_t sse_ss_hip(const int16_t* a, intptr_t sa, const int16_t* b, intptr_t sb)
{
    uint64_t r;
    if (x265hip_call_sse_ss(N, N, a, sa, b, sb, &r)) return g_c.cu[CU].sse_ss(a, sa, b, sb);
    return (sse_t)r;
}
template <int N, int CU> static sse_t ssd_s_hip(const int16_t* a, intptr_t sa)
{
    uint64_t r;
    if (x265hip_call_sse_ss(N, N, a, sa, NULL, 0, &r)) return g_c.cu[CU].ssd_s[NONALIGNED](a, sa);
    return (sse_t)r;
}

// ---- transforms (primitives.h:153-163) --------------------------------------------------------------------------------
template <int N, int CU> static void dct_hip(const int16_t* src, int16_t* dst, intptr_t stride)
{
    if (x265hip_call_dct(N, 0, D, src, dst, stride)) g_c.cu[CU].dct(src, dst, stride);
}
template <int N, int CU> static void idct_hip(const int16_t* src, int16_t* dst, intptr_t stride)
{
    if (x265hip_call_idct(N, 0, D, src, dst, stride)) g_c.cu[CU].idct(src, dst, stride);
}
static void dst4_hip(const int16_t* src, int16_t* dst, intptr_t stride)
{
    if (x265hip_call_dct(4, 1, D, src, dst, stride)) g_c.dst4x4(src, dst, stride);
}
static void idst4_hip(const int16_t* src, int16_t* dst, intptr_t stride)
{
    if (x265hip_call_idct(4, 1, D, src, dst, stride)) g_c.idst4x4(src, dst, stride);
}
static bool tu_count(int n) { return n == 16 || n == 64 || n == 256 || n == 1024; }
static uint32_t quant_hip(const int16_t* coef, const int32_t* quantCoeff, int32_t* deltaU, int16_t* qCoef, int qBits, int add, int numCoeff)
{
    uint32_t ns;
    if (!tu_count(numCoeff) || x265hip_call_quant(coef, quantCoeff, deltaU, qCoef, qBits, add, numCoeff, &ns))
        return g_c.quant(coef, quantCoeff, deltaU, qCoef, qBits, add, numCoeff);
    return ns;
}
static uint32_t nquant_hip(const int16_t* coef, const int32_t* quantCoeff, int16_t* qCoef, int qBits, int add, int numCoeff)
{
    uint32_t ns;
    if (!tu_count(numCoeff) || x265hip_call_nquant(coef, quantCoeff, qCoef, qBits, add, numCoeff, &ns))
        return g_c.nquant(coef, quantCoeff, qCoef, qBits, add, numCoeff);
    return ns;
}
static void dequant_normal_hip(const int16_t* q, int16_t* coef, int num, int scale, int shift)
{
    if ((num & 3) || x265hip_call_dequant_normal(q, coef, num, scale, shift)) g_c.dequant_normal(q, coef, num, scale, shift);
}
static void dequant_scaling_hip(const int16_t* q, const int32_t* dq, int16_t* coef, int num, int per, int shift)
{
    if (!tu_count(num) || x265hip_call_dequant_scaling(q, dq, coef, num, per, shift)) g_c.dequant_scaling(q, dq, coef, num, per, shift);
}

// ---- interpolation (primitives.h:176-183) -------------------------------------------------------------------------------
template <int W, int H, int PART> static void hpp_hip(const pixel* s, intptr_t ss, pixel* d, intptr_t ds, int c)
{
    if (x265hip_call_interp(X265HIP_IF_HPP, 8, D, W, H, s, ss, d, ds, c, 0, 0)) g_c.pu[PART].luma_hpp(s, ss, d, ds, c);
}
template <int W, int H, int PART> static void hps_hip(const pixel* s, intptr_t ss, int16_t* d, intptr_t ds, int c, int ext)
{
    if (x265hip_call_interp(X265HIP_IF_HPS, 8, D, W, H, s, ss, d, ds, c, 0, ext)) g_c.pu[PART].luma_hps(s, ss, d, ds, c, ext);
}
template <int W, int H, int PART> static void vpp_hip(const pixel* s, intptr_t ss, pixel* d, intptr_t ds, int c)
{
    if (x265hip_call_interp(X265HIP_IF_VPP, 8, D, W, H, s, ss, d, ds, c, 0, 0)) g_c.pu[PART].luma_vpp(s, ss, d, ds, c);
}
template <int W, int H, int PART> static void vps_hip(const pixel* s, intptr_t ss, int16_t* d, intptr_t ds, int c)
{
    if (x265hip_call_interp(X265HIP_IF_VPS, 8, D, W, H, s, ss, d, ds, c, 0, 0)) g_c.pu[PART].luma_vps(s, ss, d, ds, c);
}
template <int W, int H, int PART> static void vsp_hip(const int16_t* s, intptr_t ss, pixel* d, intptr_t ds, int c)
{
    if (x265hip_call_interp(X265HIP_IF_VSP, 8, D, W, H, s, ss, d, ds, c, 0, 0)) g_c.pu[PART].luma_vsp(s, ss, d, ds, c);
}
template <int W, int H, int PART> static void vss_hip(const int16_t* s, intptr_t ss, int16_t* d, intptr_t ds, int c)
{
    if (x265hip_call_interp(X265HIP_IF_VSS, 8, D, W, H, s, ss, d, ds, c, 0, 0)) g_c.pu[PART].luma_vss(s, ss, d, ds, c);
}
template <int W, int H, int PART> static void hvpp_hip(const pixel* s, intptr_t ss, pixel* d, intptr_t ds, int cx, int cy)
{
    if (x265hip_call_interp(X265HIP_IF_HVPP, 8, D, W, H, s, ss, d, ds, cx, cy, 0)) g_c.pu[PART].luma_hvpp(s, ss, d, ds, cx, cy);
}

// ---- block arithmetic / layout (primitives.h:141-151, :163, :173, :185) -------------------------------------------------------
template <int W, int H, int PART> static void copy_pp_hip(pixel* d, intptr_t ds, const pixel* s, intptr_t ss)
{
    if (x265hip_call_copy(0, D, W, H, d, ds, s, ss)) g_c.pu[PART].copy_pp(d, ds, s, ss);
}
template <int W, int H, int PART> static void addAvg_hip(const int16_t* a, const int16_t* b, pixel* d, intptr_t sa, intptr_t sb, intptr_t ds)
{
    if (x265hip_call_addavg(D, W, H, a, b, d, sa, sb, ds)) g_c.pu[PART].addAvg[NONALIGNED](a, b, d, sa, sb, ds);
}
template <int W, int H, int PART> static void pixelavg_hip(pixel* d, intptr_t ds, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb, int w)
{
    if (x265hip_call_pixelavg_pp(D, W, H, d, ds, a, sa, b, sb)) g_c.pu[PART].pixelavg_pp[NONALIGNED](d, ds, a, sa, b, sb, w);
}
template <int W, int H, int PART> static void p2s_hip(const pixel* s, intptr_t ss, int16_t* d, intptr_t ds)
{
    if (x265hip_call_p2s(D, W, H, s, ss, d, ds)) g_c.pu[PART].convert_p2s[NONALIGNED](s, ss, d, ds);
}
template <int W, int H, int PART> static void c_p2s_hip(const pixel* s, intptr_t ss, int16_t* d, intptr_t ds)
{
    if (x265hip_call_p2s(D, W / 2, H / 2, s, ss, d, ds)) g_c.chroma[X265_CSP_I420].pu[PART].p2s[NONALIGNED](s, ss, d, ds);
}
template <int N, int CU> static void sub_ps_hip(int16_t* d, intptr_t ds, const pixel* a, const pixel* b, intptr_t sa, intptr_t sb)
{
    if (x265hip_call_sub_ps(D, N, N, d, ds, a, b, sa, sb)) g_c.cu[CU].sub_ps(d, ds, a, b, sa, sb);
}
/* calcresidual_t (primitives.h:157, getResidual pixel.cpp:472): sub_ps with one stride for all three blocks */
template <int N, int CU> static void calcres_hip(const pixel* fenc, const pixel* pred, int16_t* resi, intptr_t stride)
{
    if (x265hip_call_sub_ps(D, N, N, resi, stride, fenc, pred, stride, stride)) g_c.cu[CU].calcresidual[NONALIGNED](fenc, pred, resi, stride);
}
template <int N, int CU> static void add_ps_hip(pixel* d, intptr_t ds, const pixel* a, const int16_t* r, intptr_t sa, intptr_t sr)
{
    if (x265hip_call_add_ps(D, N, N, d, ds, a, r, sa, sr)) g_c.cu[CU].add_ps[NONALIGNED](d, ds, a, r, sa, sr);
}
template <int N, int CU> static void copy_sp_hip(pixel* d, intptr_t ds, const int16_t* s, intptr_t ss)
{
    if (x265hip_call_copy(1, D, N, N, d, ds, s, ss)) g_c.cu[CU].copy_sp(d, ds, s, ss);
}
template <int N, int CU> static void copy_ps_hip(int16_t* d, intptr_t ds, const pixel* s, intptr_t ss)
{
    if (x265hip_call_copy(2, D, N, N, d, ds, s, ss)) g_c.cu[CU].copy_ps(d, ds, s, ss);
}
template <int N, int CU> static void copy_ss_hip(int16_t* d, intptr_t ds, const int16_t* s, intptr_t ss)
{
    if (x265hip_call_copy(3, D, N, N, d, ds, s, ss)) g_c.cu[CU].copy_ss(d, ds, s, ss);
}
template <int N, int CU> static void blockfill_hip(int16_t* d, intptr_t ds, int16_t v)
{
    if (x265hip_call_blockfill_s(N, d, ds, v)) g_c.cu[CU].blockfill_s[NONALIGNED](d, ds, v);
}
template <int N, int CU> static void cpy2Dto1D_shl_hip(int16_t* d, const int16_t* s, intptr_t ss, int shift)
{
    if (x265hip_call_cpy_shift(0, N, d, s, ss, shift)) g_c.cu[CU].cpy2Dto1D_shl(d, s, ss, shift);
}
template <int N, int CU> static void cpy2Dto1D_shr_hip(int16_t* d, const int16_t* s, intptr_t ss, int shift)
{
    if (shift < 1 || x265hip_call_cpy_shift(1, N, d, s, ss, shift)) g_c.cu[CU].cpy2Dto1D_shr(d, s, ss, shift);
}
template <int N, int CU> static void cpy1Dto2D_shl_hip(int16_t* d, const int16_t* s, intptr_t ds, int shift)
{
    if (x265hip_call_cpy_shift(2, N, d, s, ds, shift)) g_c.cu[CU].cpy1Dto2D_shl[NONALIGNED](d, s, ds, shift);
}
template <int N, int CU> static void cpy1Dto2D_shr_hip(int16_t* d, const int16_t* s, intptr_t ds, int shift)
{
    if (shift < 1 || x265hip_call_cpy_shift(3, N, d, s, ds, shift)) g_c.cu[CU].cpy1Dto2D_shr(d, s, ds, shift);
}
template <int N, int CU> static uint32_t copy_cnt_hip(int16_t* coeff, const int16_t* resi, intptr_t stride)
{
    uint32_t ns;
    if (x265hip_call_copy_cnt(N, coeff, resi, stride, &ns)) return g_c.cu[CU].copy_cnt(coeff, resi, stride);
    return ns;
}
template <int N, int CU> static int count_nonzero_hip(const int16_t* q)
{
    int c;
    if (x265hip_call_count_nonzero(N, q, &c)) return g_c.cu[CU].count_nonzero(q);
    return c;
}
template <int N, int CU> static void nonpsy_rdoq_hip(int16_t* resi, int64_t* cu_, int64_t* tu, int64_t* tr, uint32_t blkPos)
{
    if (x265hip_call_rdoq_cost(0, N, D, resi, NULL, cu_, tu, tr, NULL, blkPos)) g_c.cu[CU].nonPsyRdoQuant(resi, cu_, tu, tr, blkPos);
}
template <int N, int CU> static void psy_rdoq_hip(int16_t* resi, int16_t* fenc, int64_t* cu_, int64_t* tu, int64_t* tr, int64_t* psy, uint32_t blkPos)
{
    if (x265hip_call_rdoq_cost(1, N, D, resi, fenc, cu_, tu, tr, psy, blkPos)) g_c.cu[CU].psyRdoQuant(resi, fenc, cu_, tu, tr, psy, blkPos);
}
template <int N, int CU> static void psy_rdoq_1p_hip(int16_t* resi, int64_t* cu_, int64_t* tu, int64_t* tr, uint32_t blkPos)
{
    if (x265hip_call_rdoq_cost(2, N, D, resi, NULL, cu_, tu, tr, NULL, blkPos)) g_c.cu[CU].psyRdoQuant_1p(resi, cu_, tu, tr, blkPos);
}
template <int N, int CU> static void psy_rdoq_2p_hip(int16_t* resi, int16_t* fenc, int64_t* cu_, int64_t* tu, int64_t* tr, int64_t* psy, uint32_t blkPos)
{
    if (x265hip_call_rdoq_cost(3, N, D, resi, fenc, cu_, tu, tr, psy, blkPos)) g_c.cu[CU].psyRdoQuant_2p(resi, fenc, cu_, tu, tr, psy, blkPos);
}
static void denoise_hip(int16_t* coef, uint32_t* resSum, const uint16_t* offset, int numCoeff)
{
    if (x265hip_call_denoise_dct(coef, resSum, offset, numCoeff)) g_c.denoiseDct(coef, resSum, offset, numCoeff);
}

// ---- 4:2:0 chroma interpolation (primitives.h:399-404): chroma block of luma partition W x H is (W/2) x (H/2), 4 taps ----
template <int W, int H, int PART> static void c_hpp_hip(const pixel* s, intptr_t ss, pixel* d, intptr_t ds, int c)
{
    if (x265hip_call_interp(X265HIP_IF_HPP, 4, D, W / 2, H / 2, s, ss, d, ds, c, 0, 0)) g_c.chroma[X265_CSP_I420].pu[PART].filter_hpp(s, ss, d, ds, c);
}
template <int W, int H, int PART> static void c_hps_hip(const pixel* s, intptr_t ss, int16_t* d, intptr_t ds, int c, int ext)
{
    if (x265hip_call_interp(X265HIP_IF_HPS, 4, D, W / 2, H / 2, s, ss, d, ds, c, 0, ext)) g_c.chroma[X265_CSP_I420].pu[PART].filter_hps(s, ss, d, ds, c, ext);
}
template <int W, int H, int PART> static void c_vpp_hip(const pixel* s, intptr_t ss, pixel* d, intptr_t ds, int c)
{
    if (x265hip_call_interp(X265HIP_IF_VPP, 4, D, W / 2, H / 2, s, ss, d, ds, c, 0, 0)) g_c.chroma[X265_CSP_I420].pu[PART].filter_vpp(s, ss, d, ds, c);
}
template <int W, int H, int PART> static void c_vps_hip(const pixel* s, intptr_t ss, int16_t* d, intptr_t ds, int c)
{
    if (x265hip_call_interp(X265HIP_IF_VPS, 4, D, W / 2, H / 2, s, ss, d, ds, c, 0, 0)) g_c.chroma[X265_CSP_I420].pu[PART].filter_vps(s, ss, d, ds, c);
}
template <int W, int H, int PART> static void c_vsp_hip(const int16_t* s, intptr_t ss, pixel* d, intptr_t ds, int c)
{
    if (x265hip_call_interp(X265HIP_IF_VSP, 4, D, W / 2, H / 2, s, ss, d, ds, c, 0, 0)) g_c.chroma[X265_CSP_I420].pu[PART].filter_vsp(s, ss, d, ds, c);
}
template <int W, int H, int PART> static void c_vss_hip(const int16_t* s, intptr_t ss, int16_t* d, intptr_t ds, int c)
{
    if (x265hip_call_interp(X265HIP_IF_VSS, 4, D, W / 2, H / 2, s, ss, d, ds, c, 0, 0)) g_c.chroma[X265_CSP_I420].pu[PART].filter_vss(s, ss, d, ds, c);
}

#undef D

#define HIP_PU(W, H) do { \
        const int part = LUMA_ ## W ## x ## H; \
        p.pu[part].sad = sad_hip<W, H, LUMA_ ## W ## x ## H>; \
        p.pu[part].sad_x3 = sad_x3_hip<W, H, LUMA_ ## W ## x ## H>; \
        p.pu[part].sad_x4 = sad_x4_hip<W, H, LUMA_ ## W ## x ## H>; \
        p.pu[part].satd = satd_hip<W, H, LUMA_ ## W ## x ## H>; \
        p.pu[part].luma_hpp = hpp_hip<W, H, LUMA_ ## W ## x ## H>; \
        p.pu[part].luma_hps = hps_hip<W, H, LUMA_ ## W ## x ## H>; \
        p.pu[part].luma_vpp = vpp_hip<W, H, LUMA_ ## W ## x ## H>; \
        p.pu[part].luma_vps = vps_hip<W, H, LUMA_ ## W ## x ## H>; \
        p.pu[part].luma_vsp = vsp_hip<W, H, LUMA_ ## W ## x ## H>; \
        p.pu[part].luma_vss = vss_hip<W, H, LUMA_ ## W ## x ## H>; \
        p.pu[part].luma_hvpp = hvpp_hip<W, H, LUMA_ ## W ## x ## H>; \
        p.pu[part].copy_pp = copy_pp_hip<W, H, LUMA_ ## W ## x ## H>; \
        p.pu[part].addAvg[NONALIGNED] = addAvg_hip<W, H, LUMA_ ## W ## x ## H>; \
        p.pu[part].addAvg[ALIGNED] = addAvg_hip<W, H, LUMA_ ## W ## x ## H>; \
        p.pu[part].pixelavg_pp[NONALIGNED] = pixelavg_hip<W, H, LUMA_ ## W ## x ## H>; \
        p.pu[part].pixelavg_pp[ALIGNED] = pixelavg_hip<W, H, LUMA_ ## W ## x ## H>; \
        p.pu[part].convert_p2s[NONALIGNED] = p2s_hip<W, H, LUMA_ ## W ## x ## H>; \
        p.pu[part].convert_p2s[ALIGNED] = p2s_hip<W, H, LUMA_ ## W ## x ## H>; \
    } while (0)

#define HIP_CHROMA420(W, H) do { \
        const int part = LUMA_ ## W ## x ## H; \
        p.chroma[X265_CSP_I420].pu[part].filter_hpp = c_hpp_hip<W, H, LUMA_ ## W ## x ## H>; \
        p.chroma[X265_CSP_I420].pu[part].filter_hps = c_hps_hip<W, H, LUMA_ ## W ## x ## H>; \
        p.chroma[X265_CSP_I420].pu[part].filter_vpp = c_vpp_hip<W, H, LUMA_ ## W ## x ## H>; \
        p.chroma[X265_CSP_I420].pu[part].filter_vps = c_vps_hip<W, H, LUMA_ ## W ## x ## H>; \
        p.chroma[X265_CSP_I420].pu[part].filter_vsp = c_vsp_hip<W, H, LUMA_ ## W ## x ## H>; \
        p.chroma[X265_CSP_I420].pu[part].filter_vss = c_vss_hip<W, H, LUMA_ ## W ## x ## H>; \
        p.chroma[X265_CSP_I420].pu[part].p2s[NONALIGNED] = c_p2s_hip<W, H, LUMA_ ## W ## x ## H>; \
        p.chroma[X265_CSP_I420].pu[part].p2s[ALIGNED] = c_p2s_hip<W, H, LUMA_ ## W ## x ## H>; \
    } while (0)

#define HIP_CU(N) do { \
        const int cu = BLOCK_ ## N ## x ## N; \
        p.cu[cu].sa8d = sa8d_hip<N, BLOCK_ ## N ## x ## N>; \
        p.cu[cu].psy_cost_pp = psy_hip<N, BLOCK_ ## N ## x ## N>; \
        p.cu[cu].sse_pp = sse_pp_hip<N, BLOCK_ ## N ## x ## N>; \
        p.cu[cu].sse_ss = sse_ss_hip<N, BLOCK_ ## N ## x ## N>; \
        p.cu[cu].ssd_s[NONALIGNED] = ssd_s_hip<N, BLOCK_ ## N ## x ## N>; \
        p.cu[cu].ssd_s[ALIGNED] = ssd_s_hip<N, BLOCK_ ## N ## x ## N>; \
        p.cu[cu].sub_ps = sub_ps_hip<N, BLOCK_ ## N ## x ## N>; \
        p.cu[cu].calcresidual[NONALIGNED] = calcres_hip<N, BLOCK_ ## N ## x ## N>; \
        p.cu[cu].calcresidual[ALIGNED] = calcres_hip<N, BLOCK_ ## N ## x ## N>; \
        p.cu[cu].add_ps[NONALIGNED] = add_ps_hip<N, BLOCK_ ## N ## x ## N>; \
        p.cu[cu].add_ps[ALIGNED] = add_ps_hip<N, BLOCK_ ## N ## x ## N>; \
        p.cu[cu].copy_sp = copy_sp_hip<N, BLOCK_ ## N ## x ## N>; \
        p.cu[cu].copy_ps = copy_ps_hip<N, BLOCK_ ## N ## x ## N>; \
        p.cu[cu].copy_ss = copy_ss_hip<N, BLOCK_ ## N ## x ## N>; \
        p.cu[cu].blockfill_s[NONALIGNED] = blockfill_hip<N, BLOCK_ ## N ## x ## N>; \
        p.cu[cu].blockfill_s[ALIGNED] = blockfill_hip<N, BLOCK_ ## N ## x ## N>; \
    } while (0)

#define HIP_TU(N) do { \
        p.cu[BLOCK_ ## N ## x ## N].dct = dct_hip<N, BLOCK_ ## N ## x ## N>; \
        p.cu[BLOCK_ ## N ## x ## N].idct = idct_hip<N, BLOCK_ ## N ## x ## N>; \
        p.cu[BLOCK_ ## N ## x ## N].cpy2Dto1D_shl = cpy2Dto1D_shl_hip<N, BLOCK_ ## N ## x ## N>; \
        p.cu[BLOCK_ ## N ## x ## N].cpy2Dto1D_shr = cpy2Dto1D_shr_hip<N, BLOCK_ ## N ## x ## N>; \
        p.cu[BLOCK_ ## N ## x ## N].cpy1Dto2D_shl[NONALIGNED] = cpy1Dto2D_shl_hip<N, BLOCK_ ## N ## x ## N>; \
        p.cu[BLOCK_ ## N ## x ## N].cpy1Dto2D_shl[ALIGNED] = cpy1Dto2D_shl_hip<N, BLOCK_ ## N ## x ## N>; \
        p.cu[BLOCK_ ## N ## x ## N].cpy1Dto2D_shr = cpy1Dto2D_shr_hip<N, BLOCK_ ## N ## x ## N>; \
        p.cu[BLOCK_ ## N ## x ## N].copy_cnt = copy_cnt_hip<N, BLOCK_ ## N ## x ## N>; \
        p.cu[BLOCK_ ## N ## x ## N].count_nonzero = count_nonzero_hip<N, BLOCK_ ## N ## x ## N>; \
        p.cu[BLOCK_ ## N ## x ## N].nonPsyRdoQuant = nonpsy_rdoq_hip<N, BLOCK_ ## N ## x ## N>; \
        p.cu[BLOCK_ ## N ## x ## N].psyRdoQuant = psy_rdoq_hip<N, BLOCK_ ## N ## x ## N>; \
        p.cu[BLOCK_ ## N ## x ## N].psyRdoQuant_1p = psy_rdoq_1p_hip<N, BLOCK_ ## N ## x ## N>; \
        p.cu[BLOCK_ ## N ## x ## N].psyRdoQuant_2p = psy_rdoq_2p_hip<N, BLOCK_ ## N ## x ## N>; \
    } while (0)

/* ---- intra prediction (primitives.h:143-145, intrapred.cpp) and the lookahead downscale (primitives.h:168) ---- */
template <int N, int CU> static void intra_pred_hip(pixel* dst, intptr_t ds, const pixel* line, int mode, int bFilter)
{
    if (x265hip_call_intra_pred(X265_DEPTH, N, mode, bFilter, dst, ds, line)) g_c.cu[CU].intra_pred[mode](dst, ds, line, mode, bFilter);
}
template <int N, int CU> static void intra_planar_hip(pixel* dst, intptr_t ds, const pixel* line, int, int bFilter)
{
    if (x265hip_call_intra_pred(X265_DEPTH, N, PLANAR_IDX, bFilter, dst, ds, line)) g_c.cu[CU].intra_pred[PLANAR_IDX](dst, ds, line, PLANAR_IDX, bFilter);
}
template <int N, int CU> static void intra_dc_hip(pixel* dst, intptr_t ds, const pixel* line, int, int bFilter)
{
    if (x265hip_call_intra_pred(X265_DEPTH, N, DC_IDX, bFilter, dst, ds, line)) g_c.cu[CU].intra_pred[DC_IDX](dst, ds, line, DC_IDX, bFilter);
}
template <int N, int CU> static void intra_allangs_hip(pixel* dest, pixel* line, pixel* filtered, int bLuma)
{
    if (x265hip_call_intra_allangs(X265_DEPTH, N, dest, line, filtered, bLuma)) g_c.cu[CU].intra_pred_allangs(dest, line, filtered, bLuma);
}
template <int N, int CU> static void intra_filter_hip(const pixel* line, pixel* filtered)
{
    if (x265hip_call_intra_filter(X265_DEPTH, N, line, filtered)) g_c.cu[CU].intra_filter(line, filtered);
}
static void frame_init_lowres_hip(const pixel* src, pixel* d0, pixel* dh, pixel* dv, pixel* dc, intptr_t ss, intptr_t ds, int w, int h)
{
    if (x265hip_call_frame_init_lowres(X265_DEPTH, src, ss, d0, dh, dv, dc, ds, w, h)) g_c.frameInitLowres(src, d0, dh, dv, dc, ss, ds, w, h);
}

#define HIP_INTRA(N) do { \
        const int cu = BLOCK_ ## N ## x ## N; \
        p.cu[cu].intra_pred[PLANAR_IDX] = intra_planar_hip<N, BLOCK_ ## N ## x ## N>; \
        p.cu[cu].intra_pred[DC_IDX] = intra_dc_hip<N, BLOCK_ ## N ## x ## N>; \
        for (int m = 2; m < NUM_INTRA_MODE; m++) \
            p.cu[cu].intra_pred[m] = intra_pred_hip<N, BLOCK_ ## N ## x ## N>; \
        p.cu[cu].intra_pred_allangs = intra_allangs_hip<N, BLOCK_ ## N ## x ## N>; \
        p.cu[cu].intra_filter = intra_filter_hip<N, BLOCK_ ## N ## x ## N>; \
    } while (0)

/* ---- small primitives around the path: var, explicit weighting, the 64x64 intra-scan downscales, transpose ---- */
template <int N, int CU> static uint64_t var_hip(const pixel* pix, intptr_t stride)
{
    uint64_t r = 0;
    if (x265hip_call_var(X265_DEPTH, N, pix, stride, &r)) return g_c.cu[CU].var(pix, stride);
    return r;
}
template <int N, int CU> static void transpose_hip(pixel* dst, const pixel* src, intptr_t stride)
{
    if (x265hip_call_transpose(X265_DEPTH, N, dst, src, stride)) g_c.cu[CU].transpose(dst, src, stride);
}
static void weight_pp_hip(const pixel* src, pixel* dst, intptr_t stride, int width, int height, int w0, int round, int shift, int offset)
{
    if (x265hip_call_weight_pp(X265_DEPTH, src, dst, stride, width, height, w0, round, shift, offset)) g_c.weight_pp(src, dst, stride, width, height, w0, round, shift, offset);
}
static void weight_sp_hip(const int16_t* src, pixel* dst, intptr_t ss, intptr_t ds, int width, int height, int w0, int round, int shift, int offset)
{
    if (x265hip_call_weight_sp(X265_DEPTH, src, dst, ss, ds, width, height, w0, round, shift, offset)) g_c.weight_sp(src, dst, ss, ds, width, height, w0, round, shift, offset);
}
static void scale1d_hip(pixel* dst, const pixel* src)
{
    if (x265hip_call_scale1d_128to64(X265_DEPTH, dst, src)) g_c.scale1D_128to64[NONALIGNED](dst, src);
}
static void scale2d_hip(pixel* dst, const pixel* src, intptr_t stride)
{
    if (x265hip_call_scale2d_64to32(X265_DEPTH, dst, src, stride)) g_c.scale2D_64to32(dst, src, stride);
}

// ---- coefficient-scan cost primitives (dct.cpp:757-1006).  The table passes scan orders as pointers into constants.cpp; the library
// builds the same orders itself, so the pointer only has to be recognised.
static int scan_type_of(const uint16_t* scan, int sizeIdx)
{
    for (int t = 0; t < NUM_SCAN_TYPE; t++)
        if (scan == g_scanOrder[t][sizeIdx]) return t;
    return -1;
}
static int scan4_type_of(const uint16_t* scan)
{
    for (int t = 0; t < NUM_SCAN_TYPE; t++)
        if (scan == g_scan4x4[t]) return t;
    return -1;
}
static int scanPosLast_hip(const uint16_t* scan, const coeff_t* coeff, uint16_t* coeffSign, uint16_t* coeffFlag, uint8_t* coeffNum, int numSig,
                           const uint16_t* scanCG4x4, const int trSize)
{
    const int log2 = trSize == 4 ? 2 : trSize == 8 ? 3 : trSize == 16 ? 4 : 5;
    const int t = scan_type_of(scan, log2 - 2);
    int last = 0;
    if (t < 0 || x265hip_call_scan_pos_last(log2, t, coeff, coeffSign, coeffFlag, coeffNum, numSig, &last))
        return g_c.scanPosLast(scan, coeff, coeffSign, coeffFlag, coeffNum, numSig, scanCG4x4, trSize);
    return last;
}
static uint32_t findPosFirstLast_hip(const int16_t* dstCoeff, const intptr_t trSize, const uint16_t scanTbl[16])
{
    const int t = scan4_type_of(scanTbl);
    uint32_t r = 0;
    if (t < 0 || x265hip_call_find_pos_first_last(dstCoeff, trSize, t, &r))
        return g_c.findPosFirstLast(dstCoeff, trSize, scanTbl);
    return r;
}
static uint32_t costCoeffNxN_hip(const uint16_t* scan, const coeff_t* coeff, intptr_t trSize, uint16_t* absCoeff, const uint8_t* tabSigCtx,
                                 uint32_t scanFlagMask, uint8_t* baseCtx, int offset, int scanPosSigOff, int subPosBase)
{
    const int t = scan4_type_of(scan);
    uint32_t r = 0;
    if (t < 0 || x265hip_call_cost_coeff_nxn(t, coeff, trSize, absCoeff, tabSigCtx, scanFlagMask, baseCtx, offset, scanPosSigOff, subPosBase, &r))
        return g_c.costCoeffNxN(scan, coeff, trSize, absCoeff, tabSigCtx, scanFlagMask, baseCtx, offset, scanPosSigOff, subPosBase);
    return r;
}
static uint32_t costCoeffRemain_hip(uint16_t* absCoeff, int numNonZero, int idx)
{
    uint32_t r = 0;
    if (x265hip_call_cost_coeff_remain(absCoeff, numNonZero, idx, &r)) return g_c.costCoeffRemain(absCoeff, numNonZero, idx);
    return r;
}
static uint32_t costC1C2Flag_hip(uint16_t* absCoeff, intptr_t numC1Flag, uint8_t* baseCtxMod, intptr_t ctxOffset)
{
    uint32_t r = 0;
    if (x265hip_call_cost_c1c2_flag(absCoeff, numC1Flag, baseCtxMod, ctxOffset, &r)) return g_c.costC1C2Flag(absCoeff, numC1Flag, baseCtxMod, ctxOffset);
    return r;
}

// ---- in-loop filter primitives (loopfilter.cpp, sao.cpp:1762-1925)
static void pelFilterLumaStrong_hip(pixel* src, intptr_t srcStep, intptr_t offset, int32_t tcP, int32_t tcQ)
{
    if (x265hip_call_pel_filter_luma_strong(X265_DEPTH, src, srcStep, offset, tcP, tcQ)) g_c.pelFilterLumaStrong[0](src, srcStep, offset, tcP, tcQ);
}
static void pelFilterChroma_hip(pixel* src, intptr_t srcStep, intptr_t offset, int32_t tc, int32_t maskP, int32_t maskQ)
{
    if (x265hip_call_pel_filter_chroma(X265_DEPTH, src, srcStep, offset, tc, maskP, maskQ)) g_c.pelFilterChroma[0](src, srcStep, offset, tc, maskP, maskQ);
}
static void saoSign_hip(int8_t* dst, const pixel* src1, const pixel* src2, const int endX)
{
    if (x265hip_call_sao_sign(X265_DEPTH, dst, src1, src2, endX)) g_c.sign(dst, src1, src2, endX);
}
static void saoCuOrgE0_hip(pixel* rec, int8_t* offsetEo, int width, int8_t* signLeft, intptr_t stride)
{
    if (x265hip_call_sao_apply(X265_DEPTH, 0, rec, stride, width, 0, NULL, NULL, offsetEo, signLeft)) g_c.saoCuOrgE0(rec, offsetEo, width, signLeft, stride);
}
static void saoCuOrgE1_hip(pixel* rec, int8_t* upBuff1, int8_t* offsetEo, intptr_t stride, int width)
{
    if (x265hip_call_sao_apply(X265_DEPTH, 1, rec, stride, width, 0, upBuff1, NULL, offsetEo, NULL)) g_c.saoCuOrgE1(rec, upBuff1, offsetEo, stride, width);
}
static void saoCuOrgE1_2Rows_hip(pixel* rec, int8_t* upBuff1, int8_t* offsetEo, intptr_t stride, int width)
{
    if (x265hip_call_sao_apply(X265_DEPTH, 2, rec, stride, width, 0, upBuff1, NULL, offsetEo, NULL)) g_c.saoCuOrgE1_2Rows(rec, upBuff1, offsetEo, stride, width);
}
static void saoCuOrgE2_hip(pixel* rec, int8_t* bufft, int8_t* buff1, int8_t* offsetEo, int width, intptr_t stride)
{
    if (x265hip_call_sao_apply(X265_DEPTH, 3, rec, stride, width, 0, bufft, buff1, offsetEo, NULL)) g_c.saoCuOrgE2[0](rec, bufft, buff1, offsetEo, width, stride);
}
static void saoCuOrgE3_hip(pixel* rec, int8_t* upBuff1, int8_t* offsetEo, intptr_t stride, int startX, int endX)
{
    if (x265hip_call_sao_apply(X265_DEPTH, 4, rec, stride, endX, startX, upBuff1, NULL, offsetEo, NULL)) g_c.saoCuOrgE3[0](rec, upBuff1, offsetEo, stride, startX, endX);
}
static void saoCuOrgB0_hip(pixel* rec, const int8_t* offset, int ctuWidth, int ctuHeight, intptr_t stride)
{
    if (x265hip_call_sao_apply(X265_DEPTH, 5, rec, stride, ctuWidth, ctuHeight, NULL, NULL, offset, NULL)) g_c.saoCuOrgB0(rec, offset, ctuWidth, ctuHeight, stride);
}
static void saoCuStatsBO_hip(const int16_t* diff, const pixel* rec, intptr_t stride, int endX, int endY, int32_t* stats, int32_t* count)
{
    if (x265hip_call_sao_stats(X265_DEPTH, 0, diff, rec, stride, NULL, NULL, endX, endY, stats, count)) g_c.saoCuStatsBO(diff, rec, stride, endX, endY, stats, count);
}
static void saoCuStatsE0_hip(const int16_t* diff, const pixel* rec, intptr_t stride, int endX, int endY, int32_t* stats, int32_t* count)
{
    if (x265hip_call_sao_stats(X265_DEPTH, 1, diff, rec, stride, NULL, NULL, endX, endY, stats, count)) g_c.saoCuStatsE0(diff, rec, stride, endX, endY, stats, count);
}
static void saoCuStatsE1_hip(const int16_t* diff, const pixel* rec, intptr_t stride, int8_t* upBuff1, int endX, int endY, int32_t* stats, int32_t* count)
{
    if (x265hip_call_sao_stats(X265_DEPTH, 2, diff, rec, stride, upBuff1, NULL, endX, endY, stats, count))
        g_c.saoCuStatsE1(diff, rec, stride, upBuff1, endX, endY, stats, count);
}
static void saoCuStatsE2_hip(const int16_t* diff, const pixel* rec, intptr_t stride, int8_t* upBuff1, int8_t* upBufft, int endX, int endY, int32_t* stats,
                             int32_t* count)
{
    if (x265hip_call_sao_stats(X265_DEPTH, 3, diff, rec, stride, upBuff1, upBufft, endX, endY, stats, count))
        g_c.saoCuStatsE2(diff, rec, stride, upBuff1, upBufft, endX, endY, stats, count);
}
static void saoCuStatsE3_hip(const int16_t* diff, const pixel* rec, intptr_t stride, int8_t* upBuff1, int endX, int endY, int32_t* stats, int32_t* count)
{
    if (x265hip_call_sao_stats(X265_DEPTH, 4, diff, rec, stride, upBuff1, NULL, endX, endY, stats, count))
        g_c.saoCuStatsE3(diff, rec, stride, upBuff1, endX, endY, stats, count);
}

#define HIP_SMALL(N) do { \
        p.cu[BLOCK_ ## N ## x ## N].var = var_hip<N, BLOCK_ ## N ## x ## N>; \
        p.cu[BLOCK_ ## N ## x ## N].transpose = transpose_hip<N, BLOCK_ ## N ## x ## N>; \
    } while (0)

void x265hip_install_lookup_slots(EncoderPrimitives& p);        // x265_hip_refplanes.cpp
void x265hip_install_psy_slots(EncoderPrimitives& p);           // x265_hip_srcplanes.cpp
void x265hip_install_cuserve_slots(EncoderPrimitives& p);       // x265_hip_cuserve.cpp

static void report_calls()
{
    fprintf(stderr, "x265hip: %llu primitive calls served by the GPU\n", x265hip_call_count());
}

// X265HIP_VERBOSE: the device-time ledger of the bound modules (include/x265hip.h, x265hip_device_time) — what the GPU was busy with, in total
static void report_device_time()
{
    static const char* const names[X265HIP_CLK_COUNT] = { "lookahead searches", "other lookahead kernels", "sub-pel plane bands", "SAD surfaces", "source energy planes", "CU residual quad-tree jobs", "sub-pel SATD tables" };
    uint64_t total = 0;
    char line[1024];
    int n = 0;
    for (int c = 0; c < X265HIP_CLK_COUNT; c++)
    {
        uint64_t spans = 0, ns = 0, bytes = 0;
        x265hip_device_time(c, &spans, &ns, &bytes);
        total += ns;
        n += snprintf(line + n, sizeof(line) - n, "%s%s %.3f ms in %llu launch groups (%llu algorithmic bytes)", c ? ", " : "", names[c], ns * 1e-6,
                      (unsigned long long)spans, (unsigned long long)bytes);
    }
    if (total)
        fprintf(stderr, "x265hip: device time (HIP events around every launch group): %s; total %.3f ms\n", line, total * 1e-6);
    x265hip_debug_mark("last exit handler of the bindings");
}

void setupInstrinsicPrimitives(EncoderPrimitives&, int) {}

void setupAssemblyPrimitives(EncoderPrimitives& p, int /* cpuMask: CPU ISA bits, meaningless for a GPU path */)
{
    const char* env = getenv("X265HIP");
    if (env && !strcmp(env, "0"))
        return;
    x265hip_debug_mark("setupAssemblyPrimitives enters");
    struct MarkExit { ~MarkExit() { x265hip_debug_mark("setupAssemblyPrimitives returns"); } } markExit;
    // x265_setup_primitives (primitives.cpp) is not serialised: an encoder opened on another thread sees `primitives.pu[0].sad` set, skips the
    // set-up and starts using the table while this call is still running — and this call can take long (the first one initialises the HIP
    // runtime).  The reference closes the table with setupAliasPrimitives AFTER this function; do it first as well, so that the table is complete
    // (every slot a C function) for the whole time spent here.  Without it a second encoder called through null alias slots on the GPU box.
    setupAliasPrimitives(p);
    // No device: never silent.  The bindings switch themselves off and the reference's own host code runs (an encoder must still encode), but it says
    // so; X265HIP=require turns that into an error (bench.py and the GPU tests run with it: a number measured on a silent fallback is worthless).
    // The probe runs beside the rest of x265_encoder_open (thread pools, frame encoders, lookahead: ~0.1 s at 1080p) instead of in front of it: the
    // first HIP call of a process initialises the runtime and the device (50-160 ms on the MI355X box), and nothing needs the answer before the
    // first picture arrives — the seams ask x265hip_device_count() themselves, which waits for the same initialisation if it is still running.
    static std::once_flag probeOnce;
    const bool require = env && !strcmp(env, "require");
    const bool percall = getenv("X265HIP_TABLE") && !strcmp(getenv("X265HIP_TABLE"), "percall");
    static std::atomic<bool> probeDone(false);
    auto probe = [require]
    {
        struct Done { ~Done() { probeDone = true; } } done;
        const int devices = x265hip_device_count();
        x265hip_debug_mark("device count known (HIP runtime initialised)");
        if (devices < 1)
        {
            fprintf(stderr, "x265hip: no HIP device visible: GPU bindings are OFF, the encoder runs the reference's host code only%s\n",
                    require ? "" : " (X265HIP=0 silences this, X265HIP=require makes it fatal)");
            if (require)
                abort();
            return;
        }
        // the device's context and its first stream are the other slow first-time steps: take them here as well
        void* st = NULL;
        if (!x265hip_init(0) && !x265hip_stream_create(&st))
            x265hip_stream_destroy(st);
        x265hip_debug_mark("device context warm");
    };
    std::call_once(probeOnce, [&]
    {
        if (percall) { probe(); return; }
        std::thread(probe).detach();
        // a process that ends while the runtime is still initialising must not run the runtime's exit handlers underneath it
        atexit([] { while (!probeDone) usleep(1000); });
    });
    // X265HIP_TABLE selects what the table holds:
    //   percall   every slot below becomes its per-call shim (one slot call = one 1-job launch + sync): the bit-exactness proof of each
    //             kernel under the reference's own TestBench and encoder, ~1000x slower than the C code
    //   (default) the C slots stay, except the luma sub-pel filters, which look their result up in planes the GPU built once per reference
    //             picture (x265_hip_refplanes.cpp); beside the table the GPU serves the lookahead's batched cost estimates
    //             (x265_hip_lookahead.cpp).  This is the configuration encode fps is measured on
    const char* mode = getenv("X265HIP_TABLE");
    if (!mode || strcmp(mode, "percall"))
    {
        static bool timeReport = false;
        if (!timeReport && getenv("X265HIP_VERBOSE"))
        {
            timeReport = true;
            atexit(report_device_time);
        }
        x265hip_install_lookup_slots(p);            // x265_hip_refplanes.cpp: luma sub-pel filters served from GPU-built planes
        x265hip_install_psy_slots(p);               // x265_hip_srcplanes.cpp: the source half of psy_cost_pp from GPU-built energy planes
        x265hip_install_cuserve_slots(p);           // x265_hip_cuserve.cpp: sse_pp / psy_cost_pp answers out of the CU residual quad-tree jobs
        return;
    }
    if (x265hip_device_count() < 1 || x265hip_init(0))
        return;                                     // no usable GPU: leave the table alone (C path stays)
    ensure_c_table();
    static bool registered = false;
    if (!registered && getenv("X265HIP_VERBOSE"))
    {
        registered = true;
        atexit(report_calls);
    }

    HIP_PU(4, 4);   HIP_PU(8, 8);   HIP_PU(16, 16); HIP_PU(32, 32); HIP_PU(64, 64);
    HIP_PU(8, 4);   HIP_PU(4, 8);   HIP_PU(16, 8);  HIP_PU(8, 16);  HIP_PU(32, 16); HIP_PU(16, 32);
    HIP_PU(64, 32); HIP_PU(32, 64); HIP_PU(16, 12); HIP_PU(12, 16); HIP_PU(16, 4);  HIP_PU(4, 16);
    HIP_PU(32, 24); HIP_PU(24, 32); HIP_PU(32, 8);  HIP_PU(8, 32);  HIP_PU(64, 48); HIP_PU(48, 64);
    HIP_PU(64, 16); HIP_PU(16, 64);

    // 4:2:0 chroma filters for every luma partition whose chroma block is at least 4 wide and 2 high... the reference has no
    // 2xN / Nx2 filter kernels worth a launch, keep those on the C path
    HIP_CHROMA420(8, 8);   HIP_CHROMA420(16, 16); HIP_CHROMA420(32, 32); HIP_CHROMA420(64, 64);
    HIP_CHROMA420(16, 8);  HIP_CHROMA420(8, 16);  HIP_CHROMA420(32, 16); HIP_CHROMA420(16, 32);
    HIP_CHROMA420(64, 32); HIP_CHROMA420(32, 64); HIP_CHROMA420(32, 24); HIP_CHROMA420(24, 32);
    HIP_CHROMA420(32, 8);  HIP_CHROMA420(8, 32);  HIP_CHROMA420(64, 48); HIP_CHROMA420(48, 64);
    HIP_CHROMA420(64, 16); HIP_CHROMA420(16, 64); HIP_CHROMA420(16, 12); HIP_CHROMA420(12, 16);
    HIP_CHROMA420(8, 4);   HIP_CHROMA420(4, 8);   HIP_CHROMA420(16, 4);  HIP_CHROMA420(4, 16);

    HIP_CU(4); HIP_CU(8); HIP_CU(16); HIP_CU(32); HIP_CU(64);
    HIP_TU(4); HIP_TU(8); HIP_TU(16); HIP_TU(32);
    p.dst4x4 = dst4_hip;
    p.idst4x4 = idst4_hip;
    p.quant = quant_hip;
    p.nquant = nquant_hip;
    p.dequant_normal = dequant_normal_hip;
    p.dequant_scaling = dequant_scaling_hip;
    p.denoiseDct = denoise_hip;
    HIP_INTRA(4); HIP_INTRA(8); HIP_INTRA(16); HIP_INTRA(32);
    p.frameInitLowres = frame_init_lowres_hip;
    HIP_SMALL(8); HIP_SMALL(16); HIP_SMALL(32); HIP_SMALL(64);
    p.cu[BLOCK_4x4].transpose = transpose_hip<4, BLOCK_4x4>;
    p.weight_pp = weight_pp_hip;
    p.weight_sp = weight_sp_hip;
    p.scale1D_128to64[NONALIGNED] = scale1d_hip;
    p.scale1D_128to64[ALIGNED] = scale1d_hip;
    p.scale2D_64to32 = scale2d_hip;
    // in-loop filters: deblocking edge filters, SAO offset application and statistics (the C code is the same for both edge directions
    // and both width classes, loopfilter.cpp:187-205)
    p.pelFilterLumaStrong[0] = p.pelFilterLumaStrong[1] = pelFilterLumaStrong_hip;
    p.pelFilterChroma[0] = p.pelFilterChroma[1] = pelFilterChroma_hip;
    p.sign = saoSign_hip;
    p.saoCuOrgE0 = saoCuOrgE0_hip;
    p.saoCuOrgE1 = saoCuOrgE1_hip;
    p.saoCuOrgE1_2Rows = saoCuOrgE1_2Rows_hip;
    p.saoCuOrgE2[0] = p.saoCuOrgE2[1] = saoCuOrgE2_hip;
    p.saoCuOrgE3[0] = p.saoCuOrgE3[1] = saoCuOrgE3_hip;
    p.saoCuOrgB0 = saoCuOrgB0_hip;
    p.saoCuStatsBO = saoCuStatsBO_hip;
    p.saoCuStatsE0 = saoCuStatsE0_hip;
    p.saoCuStatsE1 = saoCuStatsE1_hip;
    p.saoCuStatsE2 = saoCuStatsE2_hip;
    p.saoCuStatsE3 = saoCuStatsE3_hip;
    if (!x265hip_set_entropy_state_bits(PFX(entropyStateBits)))
    {
        // the coefficient-scan cost helpers of RDOQ and of the bit estimation (the CABAC cost table is the encoder's own data)
        p.scanPosLast = scanPosLast_hip;
        p.findPosFirstLast = findPosFirstLast_hip;
        p.costCoeffNxN = costCoeffNxN_hip;
        p.costCoeffRemain = costCoeffRemain_hip;
        p.costC1C2Flag = costC1C2Flag_hip;
    }
}

} // namespace X265_NS

// cpu-a.asm stand-ins for an ENABLE_ASSEMBLY build without nasm (reference primitives.cpp:288-303 defines these only
// in the non-assembly configuration)
extern "C" {
int PFX(cpu_cpuid_test)(void) { return 0; }
void PFX(cpu_emms)(void) {}
void PFX(cpu_cpuid)(uint32_t, uint32_t* eax, uint32_t* ebx, uint32_t* ecx, uint32_t* edx) { *eax = *ebx = *ecx = *edx = 0; }
void PFX(cpu_xgetbv)(uint32_t, uint32_t* eax, uint32_t* edx) { *eax = *edx = 0; }
#if X265_ARCH_ARM == 0
void PFX(cpu_neon_test)(void) {}
int PFX(cpu_fast_neon_mrc_test)(void) { return 0; }
#endif
}
