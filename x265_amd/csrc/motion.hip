// motion.hip — MotionEstimate::motionEstimate for a whole batch of PUs in one launch (gfx950).
//
// Reference semantics (bit-exact MV and cost): source/encoder/motion.cpp — motionEstimate :739-1569 (predictor,
// zero and candidate tests :761-812; DIA :831-852; HEX :855-944; FULL :1397-1441; bestpre/bmv merge :1449-1455;
// sub-pel refine :1504-1561 driven by workload[] :48-58), subpelCompare :1571 (luma_hpp / luma_vpp / luma_hvpp +
// sad / satd), COPY*_IF_LT tie-breaking (common.h:183-204), MV::clipped / checkRange (mv.h:88-100),
// BitCost::mvcost (bitcost.h:42-45: u16 sum of two table entries).
//
// Mapping: ONE WAVE PER PU, 4 independent waves per workgroup.  The search is a serial chain of decisions, but every
// decision variable (bmv, bcost, dir ...) is wave-uniform, so the wave runs the reference's control flow in lockstep
// while the 64 lanes split the pixels of each SAD / interpolation / SATD:
//   * the PU's source pixels live in LDS (loaded once per PU, coalesced);
//   * integer-pel SADs read the reference plane straight from L1/L2 with unaligned packed loads + v_sad_u8/u16;
//     small PUs (<= 16 quads) evaluate the 3 / 4 candidates of a sad_x3 / sad_x4 step in ONE pass, 16 (or fewer) lanes
//     per candidate, the hex / square pattern costs come back through DPP + readlane;
//   * sub-pel candidates are interpolated into an LDS block (hv through a 14-bit LDS intermediate) and compared there
//     with the 4x4-tile Hadamard of tiles.h.
// HBM traffic per PU is its source block once plus whatever part of the search window the pattern touches (L2 absorbs
// the overlap between neighbouring PUs); results are 12 bytes.
#include "common.h"
#include "tiles.h"
#include "filters.h"
#include "searchrange.h"
#include "mestar.h"
#include "meumh.h"
#include <cstdlib>

namespace xh {

struct Mv { int x, y; };

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <typename P> struct Packed;
template <> struct Packed<uint8_t>
{
    typedef uint32_t T;
    static __device__ __forceinline__ unsigned sad(T a, T b, unsigned acc) { return __builtin_amdgcn_sad_u8(a, b, acc); }
};
template <> struct Packed<uint16_t>
{
    typedef uint2 T;
    static __device__ __forceinline__ unsigned sad(T a, T b, unsigned acc)
    {
        acc = __builtin_amdgcn_sad_u16(a.x, b.x, acc);
        return __builtin_amdgcn_sad_u16(a.y, b.y, acc);
    }
};

template <typename P>
struct MeCtx
{
    typedef typename Packed<P>::T Q;     // 4 packed pixels
    const P* fref;                       // reference plane at the PU origin (full-pel mv (0,0))
    int64_t stride;
    P* fenc;                             // LDS, w x h, stride w
    P* pred;                             // LDS, w x h, stride w (interpolated candidate)
    int16_t* tmp;                        // LDS, (h + 7) x w, hv intermediate
    const uint16_t* cost;                // mvcost row, entry of MVD 0
    Mv qmvp;
    int w, h, depth, lane;
    int quadsX, quads;                   // quads = (w/4) * h groups of 4 horizontally adjacent pixels
    int gs, ngroups;                     // lanes per candidate group, groups per wave
    bool chroma;                         // bChromaSATD (motion.cpp:212): subme > 2 and the 4:2:0 chroma block is whole 4x4 tiles
    const P* crefCb;                     // reference chroma planes at the PU's chroma origin
    const P* crefCr;
    int64_t strideC;
    P* fencCb;                           // LDS, (w/2) x (h/2) each
    P* fencCr;

    __device__ __forceinline__ int mvcost(int qx, int qy) const
    {
        return uni((int)(uint16_t)(cost[qx - qmvp.x] + cost[qy - qmvp.y]));
    }

    // SAD of the PU against the reference at full-pel (mx, my); whole wave on one candidate
    __device__ __forceinline__ int sad_at(int mx, int my) const
    {
        const P* r = fref + (int64_t)my * stride + mx;
        unsigned acc = 0;
        for (int q = lane; q < quads; q += 64)
        {
            const int row = q / quadsX, c4 = (q - row * quadsX) * 4;
            acc = Packed<P>::sad(ld_unaligned<Q>(r + (int64_t)row * stride + c4), *reinterpret_cast<const Q*>(fenc + row * w + c4), acc);
        }
        return uni(wave_sum((int)acc));
    }

    // mestar.h contract
    __device__ __forceinline__ int fullpel_cost(int mx, int my, int shift) const { return sad_at(mx, my) + mvcost(mx << shift, my << shift); }
    template <int K>
    __device__ __forceinline__ void fullpel_costs(const int (&mx)[K], const int (&my)[K], int (&out)[K]) const
    {
#pragma unroll 1
        for (int k = 0; k < K; k++) out[k] = fullpel_cost(mx[k], my[k], 2);
    }

    // K (<= 4) candidates; costs[k] = SAD only.  Small PUs run all candidates side by side in lane groups.
    __device__ __forceinline__ void sad_multi(int K, const Mv* mvs, int* costs) const
    {
        if (ngroups == 1)
        {
            for (int k = 0; k < K; k++)
                costs[k] = sad_at(mvs[k].x, mvs[k].y);
            return;
        }
        const int g = lane / gs, s = lane - g * gs;
        for (int k0 = 0; k0 < K; k0 += ngroups)
        {
            int k = k0 + g;
            if (k >= K) k = K - 1;                     // idle groups recompute the last candidate; result unused
            Mv m = mvs[0];
#pragma unroll
            for (int i = 1; i < 4; i++)
                if (i == k) m = mvs[i];
            const P* r = fref + (int64_t)m.y * stride + m.x;
            unsigned acc = 0;
            for (int q = s; q < quads; q += gs)
            {
                const int row = q / quadsX, c4 = (q - row * quadsX) * 4;
                acc = Packed<P>::sad(ld_unaligned<Q>(r + (int64_t)row * stride + c4), *reinterpret_cast<const Q*>(fenc + row * w + c4), acc);
            }
            const int sum = group_sum((int)acc, gs);
            for (int j = 0; j < ngroups && k0 + j < K; j++)
                costs[k0 + j] = __builtin_amdgcn_readlane(sum, j * gs);
        }
    }

    // ---- bw x bh block at `r` (integer position in a reference plane of pitch rs) filtered to phase (xFrac, yFrac) into dst:
    // copy / hpp / vpp / hps + vsp exactly as subpelCompare (motion.cpp:1571-1600 luma 8-tap, :1626-1658 chroma 4-tap) and
    // Predict::predInterLumaPixel (predict.cpp:245-266) select them.  dst may be LDS or global.
    template <int NT>
    __device__ __forceinline__ void interp_blk(const P* r, int64_t rs, int xFrac, int yFrac, P* dst, int64_t ds, int bw, int bh) const
    {
        constexpr int HALF = NT / 2 - 1, SPAN = 4 + NT - 1;
        const int qX = bw >> 2, nq = qX * bh;
        if (!(xFrac | yFrac))
        {
            for (int q = lane; q < nq; q += 64)
            {
                const int row = q / qX, c4 = (q - row * qX) * 4;
                st_unaligned<Q>(dst + row * ds + c4, ld_unaligned<Q>(r + (int64_t)row * rs + c4));
            }
        }
        else if (!yFrac)
        {
            const Stage st = stage_for(IF_HPP, depth);
            int c[NT];
#pragma unroll
            for (int i = 0; i < NT; i++) c[i] = filter_tap<NT>(xFrac, i);
            for (int q = lane; q < nq; q += 64)
            {
                const int row = q / qX, c4 = (q - row * qX) * 4;
                int v[SPAN], out[4];
                load_span<SPAN>(r + (int64_t)row * rs + c4 - HALF, v);
#pragma unroll
                for (int o = 0; o < 4; o++)
                {
                    int sum = 0;
#pragma unroll
                    for (int i = 0; i < NT; i++) sum += v[o + i] * c[i];
                    out[o] = finish(sum, st);
                }
                store4(dst + row * ds + c4, out);
            }
        }
        else if (!xFrac)
        {
            const Stage st = stage_for(IF_VPP, depth);
            int c[NT];
#pragma unroll
            for (int i = 0; i < NT; i++) c[i] = filter_tap<NT>(yFrac, i);
            for (int q = lane; q < nq; q += 64)
            {
                const int row = q / qX, c4 = (q - row * qX) * 4;
                int sum[4] = { 0, 0, 0, 0 }, out[4];
#pragma unroll
                for (int i = 0; i < NT; i++)
                {
                    int v[4];
                    load4(r + (int64_t)(row + i - HALF) * rs + c4, v);
#pragma unroll
                    for (int o = 0; o < 4; o++) sum[o] += v[o] * c[i];
                }
#pragma unroll
                for (int o = 0; o < 4; o++) out[o] = finish(sum[o], st);
                store4(dst + row * ds + c4, out);
            }
        }
        else
        {
            const Stage s1 = stage_for(IF_HPS, depth), s2 = stage_for(IF_VSP, depth);
            int c1[NT], c2[NT];
#pragma unroll
            for (int i = 0; i < NT; i++) { c1[i] = filter_tap<NT>(xFrac, i); c2[i] = filter_tap<NT>(yFrac, i); }
            const int quads1 = qX * (bh + NT - 1);
            for (int q = lane; q < quads1; q += 64)
            {
                const int row = q / qX, c4 = (q - row * qX) * 4;
                int v[SPAN], out[4];
                load_span<SPAN>(r + (int64_t)(row - HALF) * rs + c4 - HALF, v);
#pragma unroll
                for (int o = 0; o < 4; o++)
                {
                    int sum = 0;
#pragma unroll
                    for (int i = 0; i < NT; i++) sum += v[o + i] * c1[i];
                    out[o] = finish(sum, s1);
                }
                store4(tmp + row * bw + c4, out);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            for (int q = lane; q < nq; q += 64)
            {
                const int row = q / qX, c4 = (q - row * qX) * 4;
                int sum[4] = { 0, 0, 0, 0 }, out[4];
#pragma unroll
                for (int i = 0; i < NT; i++)
                {
                    int v[4];
                    load4(tmp + (row + i) * bw + c4, v);
#pragma unroll
                    for (int o = 0; o < 4; o++) sum[o] += v[o] * c2[i];
                }
#pragma unroll
                for (int o = 0; o < 4; o++) out[o] = finish(sum[o], s2);
                store4(dst + row * ds + c4, out);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ void interp_block(const P* r, int xFrac, int yFrac, P* dst, int64_t ds) const
    {
        interp_blk<8>(r, stride, xFrac, yFrac, dst, ds, w, h);
    }

    __device__ __forceinline__ void build_pred(int qx, int qy) const
    {
        interp_block(fref + (int64_t)(qy >> 2) * stride + (qx >> 2), qx & 3, qy & 3, pred, (int64_t)w);
    }

    __device__ __forceinline__ int sad_pred() const
    {
        unsigned acc = 0;
        for (int q = lane; q < quads; q += 64)
            acc = Packed<P>::sad(*reinterpret_cast<const Q*>(pred + q * 4), *reinterpret_cast<const Q*>(fenc + q * 4), acc);
        return uni(wave_sum((int)acc));
    }

    // pixel.cpp:263-297 satd tilers, see pixel.hip for the per-tile >> 1 argument
    __device__ __forceinline__ int satd_blocks(const P* a, int sa, const P* b, int sb, int bw, int bh) const
    {
        const int tilesX = bw >> 2, tiles = tilesX * (bh >> 2);
        int acc = 0;
        for (int t = lane; t < tiles; t += 64)
        {
            const int ty = t / tilesX, tx = t - ty * tilesX;
            int m[16];
            tile_diff(a + ty * 4 * sa + tx * 4, (int64_t)sa, b + ty * 4 * sb + tx * 4, (int64_t)sb, m);
            hadamard4x4(m);
            acc += abs_sum16(m) >> 1;
        }
        return uni(wave_sum(acc));
    }
    __device__ __forceinline__ int satd_pred() const { return satd_blocks(fenc, w, pred, w, w, h); }

    // the chroma part of subpelCompare (motion.cpp:1601-1660, 4:2:0): SATD of the Cb and Cr blocks predicted at the same
    // vector, eighth-pel in chroma samples; `pred` is free again once the luma cost has been summed
    __device__ __forceinline__ int chroma_term(Mv q) const
    {
        const int cw = w >> 1, ch = h >> 1;
        const int64_t off = (int64_t)(q.y >> 3) * strideC + (q.x >> 3);
        int cost = 0;
#pragma unroll 1
        for (int pl = 0; pl < 2; pl++)
        {
            interp_blk<4>((pl ? crefCr : crefCb) + off, strideC, q.x & 7, q.y & 7, pred, (int64_t)cw, cw, ch);
            cost += satd_blocks(pl ? fencCr : fencCb, cw, pred, cw, cw, ch);
            __builtin_amdgcn_wave_barrier();
        }
        return cost;
    }

    // MotionEstimate::subpelCompare: cmp 0 = sad, 1 = satd (luma), + the chroma SATD term when bChromaSATD
    __device__ __forceinline__ int subpel(Mv q, int cmp) const
    {
        int v;
        if (!cmp && !((q.x | q.y) & 3))
            v = sad_at(q.x >> 2, q.y >> 2);
        else
        {
            build_pred(q.x, q.y);
            v = cmp ? satd_pred() : sad_pred();
            __builtin_amdgcn_wave_barrier();
        }
        if (chroma)
            v += chroma_term(q);
        return v;
    }
};

__device__ __forceinline__ Mv mv_clip(Mv v, Mv lo, Mv hi)
{
    Mv r = { v.x > hi.x ? hi.x : v.x, v.y > hi.y ? hi.y : v.y };
    r.x = r.x < lo.x ? lo.x : r.x;
    r.y = r.y < lo.y ? lo.y : r.y;
    return r;
}
__device__ __forceinline__ bool mv_in_range(Mv v, Mv lo, Mv hi) { return v.x >= lo.x && v.x <= hi.x && v.y >= lo.y && v.y <= hi.y; }
__device__ __forceinline__ int sext2(int v) { return (v & 2) ? (v | ~3) : v; }

__device__ __constant__ const int8_t kHex2[8][2] = { {-1,-2}, {-2,0}, {-1,2}, {1,2}, {2,0}, {1,-2}, {-1,-2}, {-2,0} };   // motion.cpp:63
__device__ __constant__ const uint8_t kMod6m1[8] = { 5, 0, 1, 2, 3, 4, 5, 0 };                                             // motion.cpp:64
__device__ __constant__ const int8_t kSquare1[9][2] = { {0,0}, {0,-1}, {0,1}, {-1,0}, {1,0}, {-1,-1}, {-1,1}, {1,-1}, {1,1} }; // motion.cpp:65
// motion.cpp:48-58 workload[subme] = { hpel_iters, hpel_dirs, qpel_iters, qpel_dirs, hpel_satd }
__device__ __constant__ const uint8_t kWorkload[8][5] = { {1,4,0,4,0}, {1,4,1,4,0}, {1,4,1,4,1}, {2,4,1,4,1}, {2,4,2,4,1}, {1,8,1,8,1}, {2,8,1,8,1}, {2,8,2,8,1} };

template <typename P>
__global__ __launch_bounds__(256) void motion_kernel(const P* __restrict__ fencPlane, int64_t strideF,
                                                     const P* __restrict__ refPlane, int64_t strideR,
                                                     const int32_t* __restrict__ pu_xy, const int32_t* __restrict__ mvminA,
                                                     const int32_t* __restrict__ mvmaxA, const int32_t* __restrict__ qmvpA,
                                                     int numCand, const int32_t* __restrict__ mvcA, int merange, int method, int subme,
                                                     const uint16_t* __restrict__ mvcost, int w, int h, int depth, int n,
                                                     int perWaveBytes, int32_t* __restrict__ outMv, int32_t* __restrict__ outCost, ChromaPlanes cp, SeaPlanes sea)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wv = threadIdx.x >> 6;
    MeCtx<P> c;
    c.lane = threadIdx.x & 63;
    c.w = w; c.h = h; c.depth = depth;
    c.quadsX = w >> 2; c.quads = c.quadsX * h;
    c.gs = c.quads >= 64 ? 64 : pow2_ceil(c.quads);
    c.ngroups = 64 / c.gs;
    unsigned char* base = smem + (size_t)wv * perWaveBytes;
    c.fenc = reinterpret_cast<P*>(base);
    c.pred = reinterpret_cast<P*>(base + (size_t)w * h * sizeof(P));
    c.tmp = reinterpret_cast<int16_t*>(base + 2 * (size_t)w * h * sizeof(P));
    c.chroma = cp.enable != 0;
    c.strideC = cp.strideRC;
    c.fencCb = reinterpret_cast<P*>(base + 2 * (size_t)w * h * sizeof(P) + (size_t)(h + 7) * w * 2);
    c.fencCr = c.fencCb + (w >> 1) * (h >> 1);
    c.stride = strideR;
    c.cost = mvcost;
    typedef typename MeCtx<P>::Q Q;

    const int wavesPerWg = blockDim.x >> 6;
    const int wavesTotal = gridDim.x * wavesPerWg;
    for (int pu = blockIdx.x * wavesPerWg + wv; pu < n; pu += wavesTotal)
    {
        const int bx = pu_xy[2 * pu], by = pu_xy[2 * pu + 1];
        const Mv mvmin = { mvminA[2 * pu], mvminA[2 * pu + 1] }, mvmax = { mvmaxA[2 * pu], mvmaxA[2 * pu + 1] };
        const Mv qmvp = { qmvpA[2 * pu], qmvpA[2 * pu + 1] };
        const Mv qmvmin = { mvmin.x * 4, mvmin.y * 4 }, qmvmax = { mvmax.x * 4, mvmax.y * 4 };
        c.qmvp = qmvp;
        c.fref = refPlane + (int64_t)by * strideR + bx;
        // source block -> LDS
        {
            const P* f = fencPlane + (int64_t)by * strideF + bx;
            for (int q = c.lane; q < c.quads; q += 64)
            {
                const int row = q / c.quadsX, c4 = (q - row * c.quadsX) * 4;
                *reinterpret_cast<Q*>(c.fenc + row * w + c4) = ld_unaligned<Q>(f + (int64_t)row * strideF + c4);
            }
            if (c.chroma)
            {
                const int64_t coff = (int64_t)(by >> 1) * cp.strideFC + (bx >> 1);
                const int cw = w >> 1, cq = (cw >> 2) * (h >> 1);
                c.crefCb = (const P*)cp.refCb + (int64_t)(by >> 1) * cp.strideRC + (bx >> 1);
                c.crefCr = (const P*)cp.refCr + (int64_t)(by >> 1) * cp.strideRC + (bx >> 1);
                for (int q = c.lane; q < cq; q += 64)
                {
                    const int row = q / (cw >> 2), c4 = (q - row * (cw >> 2)) * 4;
                    *reinterpret_cast<Q*>(c.fencCb + row * cw + c4) = ld_unaligned<Q>((const P*)cp.fencCb + coff + (int64_t)row * cp.strideFC + c4);
                    *reinterpret_cast<Q*>(c.fencCr + row * cw + c4) = ld_unaligned<Q>((const P*)cp.fencCr + coff + (int64_t)row * cp.strideFC + c4);
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        }

#define YOK(yy) (((yy) >= mvmin.y) & ((yy) <= mvmax.y))
#define LT1(v) do { const int v_ = (v); if (v_ < bcost) bcost = v_; } while (0)
        // ---- predictor, zero and candidates (motion.cpp:761-812)
        const Mv pmv = mv_clip(qmvp, qmvmin, qmvmax);
        Mv bestpre = pmv;
        int bprecost = c.subpel(pmv, 0);
        Mv bmv = { (pmv.x + 2) >> 2, (pmv.y + 2) >> 2 };
        int bcost = bprecost;
        if ((pmv.x & 3) | (pmv.y & 3))
            bcost = c.sad_at(bmv.x, bmv.y) + c.mvcost(bmv.x * 4, bmv.y * 4);
        if (pmv.x | pmv.y)
        {
            const int cst = c.sad_at(0, 0) + c.mvcost(0, 0);
            if (cst < bcost)
            {
                bcost = cst;
                bmv.x = 0;
                bmv.y = max(min(0, mvmax.y), mvmin.y);
            }
        }
        for (int i = 0; i < numCand; i++)
        {
            const Mv raw = { mvcA[((int64_t)pu * numCand + i) * 2], mvcA[((int64_t)pu * numCand + i) * 2 + 1] };
            const Mv m = mv_clip(raw, qmvmin, qmvmax);
            if ((m.x | m.y) && !(m.x == pmv.x && m.y == pmv.y) && !(m.x == bestpre.x && m.y == bestpre.y))
            {
                const int cst = c.subpel(m, 0) + c.mvcost(m.x, m.y);
                if (cst < bprecost)
                {
                    bprecost = cst;
                    bestpre = m;
                }
            }
        }

        int costs[4];
        Mv cand[4];
#define DIRS3(ax, ay, bx_, by_, cx, cy) do { \
            cand[0] = Mv{ bmv.x + (ax), bmv.y + (ay) }; cand[1] = Mv{ bmv.x + (bx_), bmv.y + (by_) }; cand[2] = Mv{ bmv.x + (cx), bmv.y + (cy) }; \
            c.sad_multi(3, cand, costs); \
            for (int k_ = 0; k_ < 3; k_++) costs[k_] += c.mvcost(cand[k_].x * 4, cand[k_].y * 4); } while (0)
#define DIRS4(ax, ay, bx_, by_, cx, cy, dx, dy) do { \
            cand[0] = Mv{ bmv.x + (ax), bmv.y + (ay) }; cand[1] = Mv{ bmv.x + (bx_), bmv.y + (by_) }; \
            cand[2] = Mv{ bmv.x + (cx), bmv.y + (cy) }; cand[3] = Mv{ bmv.x + (dx), bmv.y + (dy) }; \
            c.sad_multi(4, cand, costs); \
            for (int k_ = 0; k_ < 4; k_++) costs[k_] += c.mvcost(cand[k_].x * 4, cand[k_].y * 4); } while (0)

        // X265_UMH_SEARCH (meumh.h) ends either for good or in the hexagon refine of X265_HEX_SEARCH (goto me_hex2, motion.cpp:1127)
        int meth = method, hexRange = merange;       // UMH scales the range the hexagon refine then runs with (motion.cpp:1039)
        if (meth == 2)
            meth = umh_search(c, mvmin.x, mvmin.y, mvmax.x, mvmax.y, hexRange, bmv.x, bmv.y, bcost, (pmv.x + 2) >> 2, (pmv.y + 2) >> 2, numCand,
                              mvcA + (int64_t)pu * numCand * 2, qmvp.x, qmvp.y, w, h) ? 1 : -1;
        if (meth == 0)
        {
            // X265_DIA_SEARCH, motion.cpp:831-852
            bcost <<= 4;
            int i = merange;
            do
            {
                DIRS4(0, -1, 0, 1, -1, 0, 1, 0);
                if (YOK(bmv.y - 1)) LT1((costs[0] << 4) + 1);
                if (YOK(bmv.y + 1)) LT1((costs[1] << 4) + 3);
                LT1((costs[2] << 4) + 4);
                LT1((costs[3] << 4) + 12);
                if (!(bcost & 15))
                    break;
                bmv.x -= sext2((bcost >> 2) & 3);
                bmv.y -= sext2(bcost & 3);
                bcost &= ~15;
            }
            while (--i && mv_in_range(bmv, mvmin, mvmax));
            bcost >>= 4;
        }
        else if (meth == 1)
        {
            // X265_HEX_SEARCH, motion.cpp:855-944
            DIRS3(-2, 0, -1, 2, 1, 2);
            bcost <<= 3;
            if (YOK(bmv.y)) LT1((costs[0] << 3) + 2);
            if (YOK(bmv.y + 2))
            {
                LT1((costs[1] << 3) + 3);
                LT1((costs[2] << 3) + 4);
            }
            DIRS3(2, 0, 1, -2, -1, -2);
            if (YOK(bmv.y)) LT1((costs[0] << 3) + 5);
            if (YOK(bmv.y - 2))
            {
                LT1((costs[1] << 3) + 6);
                LT1((costs[2] << 3) + 7);
            }
            if (bcost & 7)
            {
                int dir = (bcost & 7) - 2;
                if (YOK(bmv.y + kHex2[dir + 1][1]))
                {
                    bmv.x += kHex2[dir + 1][0];
                    bmv.y += kHex2[dir + 1][1];
                    for (int i = (hexRange >> 1) - 1; i > 0 && mv_in_range(bmv, mvmin, mvmax); i--)
                    {
                        DIRS3(kHex2[dir + 0][0], kHex2[dir + 0][1], kHex2[dir + 1][0], kHex2[dir + 1][1], kHex2[dir + 2][0], kHex2[dir + 2][1]);
                        bcost &= ~7;
                        if (YOK(bmv.y + kHex2[dir + 0][1])) LT1((costs[0] << 3) + 1);
                        if (YOK(bmv.y + kHex2[dir + 1][1])) LT1((costs[1] << 3) + 2);
                        if (YOK(bmv.y + kHex2[dir + 2][1])) LT1((costs[2] << 3) + 3);
                        if (!(bcost & 7))
                            break;
                        dir += (bcost & 7) - 2;
                        dir = kMod6m1[dir + 1];
                        bmv.x += kHex2[dir + 1][0];
                        bmv.y += kHex2[dir + 1][1];
                    }
                }
            }
            bcost >>= 3;
            // square refine, motion.cpp:918-942
            int dir = 0;
            DIRS4(0, -1, 0, 1, -1, 0, 1, 0);
            if (YOK(bmv.y - 1) && costs[0] < bcost) { bcost = costs[0]; dir = 1; }
            if (YOK(bmv.y + 1) && costs[1] < bcost) { bcost = costs[1]; dir = 2; }
            if (costs[2] < bcost) { bcost = costs[2]; dir = 3; }
            if (costs[3] < bcost) { bcost = costs[3]; dir = 4; }
            DIRS4(-1, -1, -1, 1, 1, -1, 1, 1);
            if (YOK(bmv.y - 1) && costs[0] < bcost) { bcost = costs[0]; dir = 5; }
            if (YOK(bmv.y + 1) && costs[1] < bcost) { bcost = costs[1]; dir = 6; }
            if (YOK(bmv.y - 1) && costs[2] < bcost) { bcost = costs[2]; dir = 7; }
            if (YOK(bmv.y + 1) && costs[3] < bcost) { bcost = costs[3]; dir = 8; }
            bmv.x += kSquare1[dir][0];
            bmv.y += kSquare1[dir][1];
        }
        else if (meth == 3)
            star_search(c, mvmin.x, mvmin.y, mvmax.x, mvmax.y, merange, bmv.x, bmv.y, bcost);  // X265_STAR_SEARCH (mestar.h)
        else if (meth == 4)
        {
            // X265_SEA, motion.cpp:1242-1395.  A row of the window is one step: every lane forms the DC lower bound of its position (ads_x4 / x2 / x1,
            // pixel.cpp:121-166: |sum of the PU's quarters - window sums of the reference| + the position's x cost) against the row-start best;
            // a ballot is the list of survivors in x order; survivors are SAD-ed three at a time and compared in order (COST_MV_X3_ABS), the
            // last one or two of the row with the ordinary cost (COST_MV).  The arithmetic is the reference's as it stands (see the oracle).
            const int minX = max(bmv.x - merange, mvmin.x), minY = max(bmv.y - merange, mvmin.y);
            const int maxX = min(bmv.x + merange, mvmax.x), maxY = min(bmv.y + merange, mvmax.y);
            const int meRangeWidth = (maxX - minX + 3) & ~3;
            int deltaX = w <= 8 ? w : w >> 1;
            int64_t deltaY = h <= 8 ? h : h >> 1;
            const int wh = (w << 8) | h;
            auto is = [wh](int a, int b) { return wh == ((a << 8) | b); };
            const bool isSmall = is(4, 4) || is(16, 12) || is(12, 16) || is(16, 4) || is(4, 16);
            const bool isV = is(32, 64) || is(16, 32) || is(8, 16) || is(4, 8), isH = is(64, 32) || is(32, 16) || is(16, 8) || is(8, 4);
            const bool isAsym = is(12, 16) || is(4, 16) || is(24, 32) || is(8, 32) || is(48, 64) || is(16, 64) || is(16, 12) || is(16, 4) || is(32, 24) ||
                                is(32, 8) || is(64, 48) || is(64, 16);
            int tw, th;
            if (isV) { tw = w; th = h >> 1; }
            else if (isH) { tw = w >> 1; th = h; }
            else if (isAsym) { tw = isSmall ? w : w >> 1; th = isSmall ? h : h >> 1; }
            else { tw = w <= 8 ? w : w >> 1; th = w <= 8 ? h : h >> 1; }
            // sums of the four sub-blocks of the source PU (sad_x4 against zeros, :1306-1312); blocks that leave the PU (shapes for which the
            // reference reads whatever its 64x64 source cache holds there) are clamped to the PU: those shapes have no defined result
            int encDC[4];
#pragma unroll 1
            for (int k = 0; k < 4; k++)
            {
                const int ox = (k & 1) ? deltaX : 0, oy = (k & 2) ? (int)deltaY : 0;
                int sum = 0;
                for (int e = c.lane; e < tw * th; e += 64)
                {
                    const int yy = min(oy + e / tw, h - 1), xx = min(ox + e % tw, w - 1);
                    sum += (int)c.fenc[yy * w + xx];
                }
                encDC[k] = uni(wave_sum(sum));
            }
            int plane;
            switch (deltaX)
            {
            case 32: plane = (deltaY % 24 == 0) ? 1 : (deltaY == 8 ? 2 : 0); break;
            case 24: plane = 3; break;
            case 16: plane = (deltaY % 12 == 0) ? 5 : (deltaY == 4 ? 6 : 4); break;
            case 12: plane = 7; break;
            case 8: plane = deltaY == 32 ? 8 : 9; break;
            case 4: plane = deltaY == 16 ? 10 : 11; break;
            default: plane = 11; break;
            }
            const uint32_t* sumsBase = sea.base + (int64_t)plane * sea.planeElems + (int64_t)by * strideR + bx;
            const bool strided = is(64, 64) || is(32, 32) || is(16, 16) || is(32, 64) || is(16, 32) || is(8, 16) || is(4, 8) || is(12, 16) || is(4, 16) ||
                                 is(24, 32) || is(8, 32) || is(48, 64) || is(16, 64);
            if (strided) deltaY *= strideR;
            if (isV) encDC[1] = encDC[2];
            if (isH) deltaY = deltaX;
            const int kind = (is(4, 4) || is(8, 8) || is(16, 12) || is(12, 16) || is(16, 4) || is(4, 16)) ? 1
                           : ((is(8, 4) || is(4, 8) || is(16, 8) || is(8, 16) || is(32, 16) || is(16, 32) || is(64, 32) || is(32, 64)) ? 2 : 4);
            const int chunks = (meRangeWidth + 63) >> 6;                 // <= 4: merange <= 126 (checked at the entry point)
#pragma unroll 1
            for (int ty = minY; ty <= maxY; ty++)
            {
                const int ycost = uni((int)c.cost[ty - 2 * qmvp.y]) << 2;
                if (bcost <= ycost)
                    continue;
                bcost -= ycost;
                unsigned long long keep[4] = { 0, 0, 0, 0 };
                const uint32_t* sums = sumsBase + minX + (int64_t)ty * strideR;
#pragma unroll
                for (int ch = 0; ch < 4; ch++)
                {
                    const int i = ch * 64 + c.lane;
                    bool k = false;
                    if (ch < chunks && i < meRangeWidth)
                    {
                        long long a = llabs((long long)encDC[0] - (long long)sums[i]);
                        if (kind == 4)
                            a += llabs((long long)encDC[1] - (long long)sums[i + (w >> 1)]) + llabs((long long)encDC[2] - (long long)sums[i + deltaY]) +
                                 llabs((long long)encDC[3] - (long long)sums[i + deltaY + (w >> 1)]);
                        else if (kind == 2)
                            a += llabs((long long)encDC[1] - (long long)sums[i + deltaY]);
                        k = (int)(a + c.cost[4 * (minX + i) - qmvp.x]) < bcost;
                    }
                    keep[ch] = __ballot(k);
                }
                // survivors in x order, three at a time
                int px[3], np = 0;
#pragma unroll 1
                for (int ch = 0; ch < chunks; ch++)
                {
                    unsigned long long m = keep[ch];
                    while (m)
                    {
                        const int bit = __builtin_ctzll(m);
                        m &= m - 1;
                        px[np++] = minX + ch * 64 + bit;
                        if (np == 3)
                        {
                            cand[0] = Mv{ px[0], ty }; cand[1] = Mv{ px[1], ty }; cand[2] = Mv{ px[2], ty }; cand[3] = cand[2];
                            c.sad_multi(3, cand, costs);
#pragma unroll
                            for (int k3 = 0; k3 < 3; k3++)
                            {
                                const int cst = costs[k3] + uni((int)c.cost[4 * px[k3] - 2 * qmvp.x]);      // x cost only, predictor taken off twice (:307-309)
                                if (cst < bcost) { bcost = cst; bmv.x = px[k3]; bmv.y = ty; }
                            }
                            np = 0;
                        }
                    }
                }
                bcost += ycost;
                for (int k3 = 0; k3 < np; k3++)                                                              // COST_MV
                {
                    const int cst = c.sad_at(px[k3], ty) + c.mvcost(px[k3] * 4, ty * 4);
                    if (cst < bcost) { bcost = cst; bmv.x = px[k3]; bmv.y = ty; }
                }
            }
        }
        else if (meth == 5)
        {
            // X265_FULL_SEARCH, motion.cpp:1397-1441: raster order, strict '<' keeps the first minimum
            for (int ty = mvmin.y; ty <= mvmax.y; ty++)
                for (int tx = mvmin.x; tx <= mvmax.x; tx += 4)
                {
                    const int K = min(4, mvmax.x - tx + 1);
                    for (int k = 0; k < 4; k++)
                        cand[k] = Mv{ tx + min(k, K - 1), ty };
                    c.sad_multi(K, cand, costs);
                    for (int k = 0; k < K; k++)
                    {
                        const int cst = costs[k] + c.mvcost((tx + k) * 4, ty * 4);
                        if (cst < bcost)
                        {
                            bcost = cst;
                            bmv.x = tx + k;
                            bmv.y = ty;
                        }
                    }
                }
        }

        // motion.cpp:1449-1455
        if (bprecost < bcost)
        {
            bmv = bestpre;
            bcost = bprecost;
        }
        else
        {
            bmv.x *= 4;
            bmv.y *= 4;
        }

        if (!bcost)
            bcost = c.mvcost(bmv.x, bmv.y);            // motion.cpp:1466-1471
        else
        {
            // motion.cpp:1504-1561
            const int hpelIters = kWorkload[subme][0], hpelDirs = kWorkload[subme][1];
            const int qpelIters = kWorkload[subme][2], qpelDirs = kWorkload[subme][3], hpelSatd = kWorkload[subme][4];
            int hpelcomp = 0;
            if (hpelSatd)
            {
                bcost = c.subpel(bmv, 1) + c.mvcost(bmv.x, bmv.y);
                hpelcomp = 1;
            }
            for (int iter = 0; iter < hpelIters; iter++)
            {
                int bdir = 0;
                for (int i = 1; i <= hpelDirs; i++)
                {
                    const Mv q = { bmv.x + kSquare1[i][0] * 2, bmv.y + kSquare1[i][1] * 2 };
                    if ((q.y < qmvmin.y) | (q.y > qmvmax.y))
                        continue;
                    const int cst = c.subpel(q, hpelcomp) + c.mvcost(q.x, q.y);
                    if (cst < bcost) { bcost = cst; bdir = i; }
                }
                if (bdir)
                {
                    bmv.x += kSquare1[bdir][0] * 2;
                    bmv.y += kSquare1[bdir][1] * 2;
                }
                else
                    break;
            }
            if (!hpelSatd)
                bcost = c.subpel(bmv, 1) + c.mvcost(bmv.x, bmv.y);
            for (int iter = 0; iter < qpelIters; iter++)
            {
                int bdir = 0;
                for (int i = 1; i <= qpelDirs; i++)
                {
                    const Mv q = { bmv.x + kSquare1[i][0], bmv.y + kSquare1[i][1] };
                    if ((q.y < qmvmin.y) | (q.y > qmvmax.y))
                        continue;
                    const int cst = c.subpel(q, 1) + c.mvcost(q.x, q.y);
                    if (cst < bcost) { bcost = cst; bdir = i; }
                }
                if (bdir)
                {
                    bmv.x += kSquare1[bdir][0];
                    bmv.y += kSquare1[bdir][1];
                }
                else
                    break;
            }
        }
#undef YOK
#undef LT1
#undef DIRS3
#undef DIRS4
        if (c.lane == 0)
        {
            outMv[2 * pu] = bmv.x;
            outMv[2 * pu + 1] = bmv.y;
            outCost[pu] = bcost;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
}

} // namespace xh

using namespace xh;

namespace xh {
int motion2_dispatch(int depth, int w, int h, const void* fencPlane, int64_t strideF, const void* refPlane, int64_t strideR,
                     const int32_t* pu_xy, const int32_t* mvmin, const int32_t* mvmax, const int32_t* qmvp, int numCand,
                     const int32_t* mvc, int merange, int method, int subme, const uint16_t* mvcost, int n, const void* planes,
                     int64_t planeElems, int32_t* outMv, int32_t* outCost, hipStream_t st, int* rc, const DeriveRange* drp = nullptr);
}

extern "C" int x265hip_motion_estimate_batch(int depth, int w, int h, const void* fencPlane, int64_t strideF,
                                             const void* refPlane, int64_t strideR, const int32_t* pu_xy,
                                             const int32_t* mvmin, const int32_t* mvmax, const int32_t* qmvp,
                                             int numCand, const int32_t* mvc, int merange, int searchMethod, int subme,
                                             const uint16_t* mvcost, int mvcostHalf, int n, int32_t* outMv,
                                             int32_t* outCost, void* stream)
{
    return x265hip_motion_estimate_planes_batch(depth, w, h, fencPlane, strideF, refPlane, strideR, nullptr, 0, pu_xy, mvmin, mvmax, qmvp,
                                                numCand, mvc, merange, searchMethod, subme, mvcost, mvcostHalf, n, outMv, outCost, stream);
}

namespace xh {
static int launch_motion_v1(int depth, int w, int h, const void* fencPlane, int64_t strideF, const void* refPlane, int64_t strideR,
                            const int32_t* pu_xy, const int32_t* mvmin, const int32_t* mvmax, const int32_t* qmvp, int numCand, const int32_t* mvc,
                            int merange, int searchMethod, int subme, const uint16_t* mvcost, int n, int32_t* outMv, int32_t* outCost,
                            const ChromaPlanes& cp, hipStream_t st, const SeaPlanes& sea = SeaPlanes{})
{
    const int B = depth == 8 ? 1 : 2;
    int perWave = 2 * w * h * B + (h + 7) * w * 2 + (cp.enable ? 2 * (w >> 1) * (h >> 1) * B : 0);
    perWave = (perWave + 15) & ~15;
    // dynamic LDS is kept under 64 KiB per workgroup: big PUs at 16-bit run 2 waves per workgroup instead of 4
    const int wpg = perWave * 4 <= 65536 ? 4 : (perWave * 2 <= 65536 ? 2 : 1);
    dim3 grid(grid_for((n + wpg - 1) / wpg, 256 * 8)), block(64 * wpg);
    if (depth == 8)
        hipLaunchKernelGGL((motion_kernel<uint8_t>), grid, block, wpg * perWave, st, (const uint8_t*)fencPlane, strideF,
                           (const uint8_t*)refPlane, strideR, pu_xy, mvmin, mvmax, qmvp, numCand, mvc, merange, searchMethod, subme,
                           mvcost, w, h, depth, n, perWave, outMv, outCost, cp, sea);
    else
        hipLaunchKernelGGL((motion_kernel<uint16_t>), grid, block, wpg * perWave, st, (const uint16_t*)fencPlane, strideF,
                           (const uint16_t*)refPlane, strideR, pu_xy, mvmin, mvmax, qmvp, numCand, mvc, merange, searchMethod, subme,
                           mvcost, w, h, depth, n, perWave, outMv, outCost, cp, sea);
    XH_LAUNCH_CHECK("motion_kernel");
    return X265HIP_OK;
}
static int check_me_args(const char* who, int depth, int w, int h, int n, int searchMethod, int subme, int numCand, int merange, int mvcostHalf)
{
    if (!valid_depth(depth) || !valid_block(w, h) || (w & 3) || (h & 3) || (w == 4 && h == 4) || n < 0)
        return set_error(X265HIP_EINVAL, "%s: depth %d PU %dx%d n %d", who, depth, w, h, n);
    if (searchMethod != 0 && searchMethod != 1 && searchMethod != 2 && searchMethod != 3 && searchMethod != 5)
        return set_error(X265HIP_EINVAL, "%s: searchMethod %d not available here (DIA 0, HEX 1, UMH 2, STAR 3, FULL 5; SEA 4 needs the window-sum planes: "
                         "x265hip_motion_estimate_sea_batch)", who, searchMethod);
    if (subme < 0 || subme > 7 || numCand < 0 || merange < 1 || mvcostHalf < 4 * (merange + 64))
        return set_error(X265HIP_EINVAL, "%s: subme %d numCand %d merange %d mvcostHalf %d", who, subme, numCand, merange, mvcostHalf);
    return X265HIP_OK;
}
} // namespace xh

extern "C" int x265hip_motion_estimate_planes_batch(int depth, int w, int h, const void* fencPlane, int64_t strideF,
                                                    const void* refPlane, int64_t strideR, const void* subpelPlanes, int64_t planeElems,
                                                    const int32_t* pu_xy, const int32_t* mvmin, const int32_t* mvmax, const int32_t* qmvp,
                                                    int numCand, const int32_t* mvc, int merange, int searchMethod, int subme,
                                                    const uint16_t* mvcost, int mvcostHalf, int n, int32_t* outMv,
                                                    int32_t* outCost, void* stream)
{
    XH_CHECK_DEV();
    int e = check_me_args("motion_estimate", depth, w, h, n, searchMethod, subme, numCand, merange, mvcostHalf);
    if (e) return e;
    if (!n) return X265HIP_OK;
    // square 8..64 PUs run on the team kernel of motion2.hip; everything else (and X265HIP_ME_V1=1) on the generic one
    static const bool forceV1 = getenv("X265HIP_ME_V1") != nullptr;
    int rc2 = 0;
    if (!forceV1 && motion2_dispatch(depth, w, h, fencPlane, strideF, refPlane, strideR, pu_xy, mvmin, mvmax, qmvp, numCand, mvc, merange,
                                     searchMethod, subme, mvcost, n, subpelPlanes, planeElems, outMv, outCost, as_stream(stream), &rc2))
        return rc2;
    const ChromaPlanes none{};
    return launch_motion_v1(depth, w, h, fencPlane, strideF, refPlane, strideR, pu_xy, mvmin, mvmax, qmvp, numCand, mvc, merange, searchMethod,
                            subme, mvcost, n, outMv, outCost, none, as_stream(stream));
}

extern "C" int x265hip_motion_estimate_chroma_batch(int depth, int w, int h, const void* fencPlane, int64_t strideF, const void* fencCb,
                                                    const void* fencCr, int64_t strideFC, const void* refPlane, int64_t strideR,
                                                    const void* refCb, const void* refCr, int64_t strideRC,
                                                    const int32_t* pu_xy, const int32_t* mvmin, const int32_t* mvmax, const int32_t* qmvp,
                                                    int numCand, const int32_t* mvc, int merange, int searchMethod, int subme,
                                                    const uint16_t* mvcost, int mvcostHalf, int n, int32_t* outMv, int32_t* outCost, void* stream)
{
    XH_CHECK_DEV();
    int e = check_me_args("motion_estimate_chroma", depth, w, h, n, searchMethod, subme, numCand, merange, mvcostHalf);
    if (e) return e;
    if (!n) return X265HIP_OK;
    // motion.cpp:212: bChromaSATD = subpelRefine > 2 && chromaSatd != NULL (the 4:2:0 block must be whole 4x4 tiles, pixel.cpp:1200-1226)
    const bool on = subme > 2 && !((w >> 1) & 3) && !((h >> 1) & 3);
    if (!on)
        return x265hip_motion_estimate_planes_batch(depth, w, h, fencPlane, strideF, refPlane, strideR, nullptr, 0, pu_xy, mvmin, mvmax, qmvp,
                                                    numCand, mvc, merange, searchMethod, subme, mvcost, mvcostHalf, n, outMv, outCost, stream);
    if (!fencCb || !fencCr || !refCb || !refCr)
        return set_error(X265HIP_EINVAL, "motion_estimate_chroma: chroma planes missing");
    const ChromaPlanes cp{ fencCb, fencCr, strideFC, refCb, refCr, strideRC, 1 };
    return launch_motion_v1(depth, w, h, fencPlane, strideF, refPlane, strideR, pu_xy, mvmin, mvmax, qmvp, numCand, mvc, merange, searchMethod,
                            subme, mvcost, n, outMv, outCost, cp, as_stream(stream));
}

// ---- --me sea: window-sum planes of a reference picture + the search --------------------------------------------------------------------
namespace xh {
// horizontal window sums of widths 4, 8, 12, 16, 24, 32 for every position of a padded picture buffer (one pass over 32 pixels gives all six)
template <typename P>
__global__ __launch_bounds__(256) void integral_h_kernel(const P* __restrict__ buf, int64_t stride, int rows, uint32_t* __restrict__ hsum, int64_t planeElems)
{
    const int64_t total = stride * (int64_t)rows;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256)
    {
        const int x = (int)(idx % stride);
        const P* p = buf + idx;
        uint32_t s = 0, out[6];
#pragma unroll
        for (int i = 0; i < 32; i++)
        {
            s += (x + i < stride) ? (uint32_t)p[i] : 0u;
            if (i == 3) out[0] = s;
            if (i == 7) out[1] = s;
            if (i == 11) out[2] = s;
            if (i == 15) out[3] = s;
            if (i == 23) out[4] = s;
            if (i == 31) out[5] = s;
        }
#pragma unroll
        for (int k = 0; k < 6; k++)
            hsum[(int64_t)k * planeElems + idx] = out[k];
    }
}
// plane k (w_k x h_k): vertical sum of h_k rows of the width-w_k horizontal sums; zero where the window leaves the buffer and in row 0, the
// base row of the reference's running sums (framefilter.cpp:768-790)
__global__ __launch_bounds__(256) void integral_v_kernel(const uint32_t* __restrict__ hsum, int64_t stride, int rows, int64_t planeElems, uint32_t* __restrict__ planes)
{
    const int kW[12] = { 32, 32, 32, 24, 16, 16, 16, 12, 8, 8, 4, 4 }, kH[12] = { 32, 24, 8, 32, 16, 12, 4, 16, 32, 8, 16, 4 };
    const int k = blockIdx.y, w = kW[k], h = kH[k];
    const int hw = w == 4 ? 0 : w == 8 ? 1 : w == 12 ? 2 : w == 16 ? 3 : w == 24 ? 4 : 5;
    const uint32_t* src = hsum + (int64_t)hw * planeElems;
    uint32_t* dst = planes + (int64_t)k * planeElems;
    const int64_t total = stride * (int64_t)rows;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256)
    {
        const int x = (int)(idx % stride), y = (int)(idx / stride);
        uint32_t s = 0;
        if (y >= 1 && y + h <= rows && x + w <= stride)
            for (int j = 0; j < h; j++)
                s += src[idx + (int64_t)j * stride];
        dst[idx] = s;
    }
}
} // namespace xh

extern "C" int x265hip_build_integral_planes(int depth, const void* bufBase, int64_t stride, int rows, uint32_t* planes, int64_t planeElems, uint32_t* scratch,
                                             void* stream)
{
    XH_CHECK_DEV();
    if (!valid_depth(depth) || !bufBase || !planes || !scratch || stride < 64 || rows < 64 || planeElems < stride * rows)
        return set_error(X265HIP_EINVAL, "build_integral_planes: depth %d stride %lld rows %d", depth, (long long)stride, rows);
    const int64_t total = stride * (int64_t)rows;
    dim3 block(256), grid(grid_for((total + 255) / 256, 256 * 32));
    if (depth == 8)
        hipLaunchKernelGGL((integral_h_kernel<uint8_t>), grid, block, 0, as_stream(stream), (const uint8_t*)bufBase, stride, rows, scratch, planeElems);
    else
        hipLaunchKernelGGL((integral_h_kernel<uint16_t>), grid, block, 0, as_stream(stream), (const uint16_t*)bufBase, stride, rows, scratch, planeElems);
    XH_LAUNCH_CHECK("integral_h_kernel");
    hipLaunchKernelGGL(integral_v_kernel, dim3(grid.x, 12), block, 0, as_stream(stream), scratch, stride, rows, planeElems, planes);
    XH_LAUNCH_CHECK("integral_v_kernel");
    return X265HIP_OK;
}

extern "C" int x265hip_motion_estimate_sea_batch(int depth, int w, int h, const void* fencPlane, int64_t strideF, const void* refPlane, int64_t strideR,
                                                 const uint32_t* integralPlanes, int64_t planeElems, const int32_t* pu_xy, const int32_t* mvmin,
                                                 const int32_t* mvmax, const int32_t* qmvp, int numCand, const int32_t* mvc, int merange, int subme,
                                                 const uint16_t* mvcost, int mvcostHalf, int n, int32_t* outMv, int32_t* outCost, void* stream)
{
    XH_CHECK_DEV();
    int e = check_me_args("motion_estimate_sea", depth, w, h, n, 5, subme, numCand, merange, mvcostHalf);
    if (e) return e;
    if (!integralPlanes || merange > 126)
        return set_error(X265HIP_EINVAL, "motion_estimate_sea: window-sum planes missing or merange %d > 126", merange);
    if (!n) return X265HIP_OK;
    const ChromaPlanes none{};
    const SeaPlanes sea{ integralPlanes, planeElems, 1 };
    return launch_motion_v1(depth, w, h, fencPlane, strideF, refPlane, strideR, pu_xy, mvmin, mvmax, qmvp, numCand, mvc, merange, 4, subme, mvcost, n, outMv,
                            outCost, none, as_stream(stream), sea);
}

// ======================================================================================================================
// Callers' helpers either side of motionEstimate: Search::setSearchRange and Predict::predInterLumaPixel
// ======================================================================================================================
namespace xh {

// Search::setSearchRange (search.cpp:2724-2770) with CUData::clipMv (cudata.cpp:1915-1928); intra-refresh and slice
// restrictions are off (their x265 defaults).  qmvp[i] = mvSrc[srcIdx[i]] (quarter-pel) or (0,0) when there is none.
__global__ __launch_bounds__(256) void search_range_kernel(int picW, int picH, int maxCUSize, int merange, int refLagPixels,
                                                           const int32_t* __restrict__ cu_xy, const int32_t* __restrict__ mvSrc,
                                                           const int32_t* __restrict__ srcIdx, int n,
                                                           int32_t* __restrict__ qmvp, int32_t* __restrict__ mvminO, int32_t* __restrict__ mvmaxO)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    {
        int px = 0, py = 0;
        if (mvSrc && srcIdx && srcIdx[i] >= 0)
        {
            px = mvSrc[2 * srcIdx[i]];
            py = mvSrc[2 * srcIdx[i] + 1];
        }
        const SearchRange r = search_range(picW, picH, maxCUSize, merange, refLagPixels, cu_xy[2 * i], cu_xy[2 * i + 1], px, py);
        const int minx = r.minx, miny = r.miny, maxx = r.maxx, maxy = r.maxy;
        qmvp[2 * i] = px; qmvp[2 * i + 1] = py;
        mvminO[2 * i] = minx; mvminO[2 * i + 1] = miny;
        mvmaxO[2 * i] = maxx; mvmaxO[2 * i + 1] = maxy;
    }
}

// Predict::predInterLumaPixel (predict.cpp:245-266): one wave per PU, prediction written into the destination plane
template <typename P>
__global__ __launch_bounds__(256) void pred_inter_luma_kernel(const P* __restrict__ refPlane, int64_t strideR, P* __restrict__ dst, int64_t strideD,
                                                              const int32_t* __restrict__ pu_xy, const int32_t* __restrict__ qmv,
                                                              int w, int h, int depth, int n, int perWaveBytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wv = threadIdx.x >> 6;
    MeCtx<P> c;
    c.lane = threadIdx.x & 63;
    c.w = w; c.h = h; c.depth = depth;
    c.quadsX = w >> 2; c.quads = c.quadsX * h;
    c.stride = strideR;
    c.chroma = false;
    c.tmp = reinterpret_cast<int16_t*>(smem + (size_t)wv * perWaveBytes);
    const int wavesPerWg = blockDim.x >> 6;
    for (int pu = blockIdx.x * wavesPerWg + wv; pu < n; pu += gridDim.x * wavesPerWg)
    {
        const int bx = pu_xy[2 * pu], by = pu_xy[2 * pu + 1];
        const int qx = qmv[2 * pu], qy = qmv[2 * pu + 1];
        const P* r = refPlane + (int64_t)(by + (qy >> 2)) * strideR + bx + (qx >> 2);
        c.interp_block(r, qx & 3, qy & 3, dst + (int64_t)by * strideD + bx, strideD);
    }
}

} // namespace xh

extern "C" int x265hip_set_search_range_batch(int picW, int picH, int maxCUSize, int merange, int refLagPixels,
                                              const int32_t* cu_xy, const int32_t* mvSrc, const int32_t* srcIdx, int n,
                                              int32_t* qmvp, int32_t* mvmin, int32_t* mvmax, void* stream)
{
    XH_CHECK_DEV();
    if (picW < 8 || picH < 8 || maxCUSize < 8 || merange < 1 || n < 0)
        return set_error(X265HIP_EINVAL, "set_search_range: pic %dx%d maxCU %d merange %d n %d", picW, picH, maxCUSize, merange, n);
    if (!n) return X265HIP_OK;
    hipLaunchKernelGGL(search_range_kernel, dim3(grid_for((n + 255) / 256)), dim3(256), 0, as_stream(stream), picW, picH, maxCUSize, merange,
                       refLagPixels, cu_xy, mvSrc, srcIdx, n, qmvp, mvmin, mvmax);
    XH_LAUNCH_CHECK("search_range_kernel");
    return X265HIP_OK;
}

extern "C" int x265hip_pred_inter_luma_batch(int depth, int w, int h, const void* refPlane, int64_t strideR, void* dst, int64_t strideD,
                                             const int32_t* pu_xy, const int32_t* qmv, int n, void* stream)
{
    XH_CHECK_DEV();
    if (!valid_depth(depth) || !valid_block(w, h) || (w & 3) || n < 0)
        return set_error(X265HIP_EINVAL, "pred_inter_luma: depth %d PU %dx%d n %d", depth, w, h, n);
    if (!n) return X265HIP_OK;
    int perWave = ((h + 7) * w * 2 + 15) & ~15;
    dim3 grid(grid_for((n + 3) / 4)), block(256);
    if (depth == 8)
        hipLaunchKernelGGL((pred_inter_luma_kernel<uint8_t>), grid, block, 4 * perWave, as_stream(stream), (const uint8_t*)refPlane, strideR,
                           (uint8_t*)dst, strideD, pu_xy, qmv, w, h, depth, n, perWave);
    else
        hipLaunchKernelGGL((pred_inter_luma_kernel<uint16_t>), grid, block, 4 * perWave, as_stream(stream), (const uint16_t*)refPlane, strideR,
                           (uint16_t*)dst, strideD, pu_xy, qmv, w, h, depth, n, perWave);
    XH_LAUNCH_CHECK("pred_inter_luma_kernel");
    return X265HIP_OK;
}
