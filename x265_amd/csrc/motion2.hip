// motion2.hip — MotionEstimate::motionEstimate, second-generation kernel for the square PU sizes (8, 16, 32, 64).
//
// Same reference semantics as motion.hip (source/encoder/motion.cpp:739-1569, subpelCompare :1571) and the same bit-exact
// outputs; what changes is how a PU's serial decision chain is executed, because that chain — not bandwidth — bounds the
// kernel:
//   * compile-time PU shape: every loop over pixels unrolls, all loads of a search step are in flight together;
//   * a TEAM per PU: one wave for 8x8 / 16x16, four waves (one workgroup) for 32x32 / 64x64, so the small-count levels
//     still fill the chip; team results meet in LDS with one barrier per step;
//   * candidate-parallel evaluation everywhere: the 3 / 4 SADs of a sad_x3 / sad_x4 step AND the 4 (or 8) directions of a
//     half/quarter-pel iteration are computed side by side (lane groups of 16 inside a wave, or one wave per candidate
//     inside a workgroup); the reference's sequential "if (cost < bcost)" updates are then replayed in reference order —
//     costs do not depend on bcost, so the result is identical;
//   * sub-pel candidates are never materialised: each lane filters the 4x4 tiles it needs straight from the reference
//     picture into registers (luma_hpp / luma_vpp / luma_hvpp arithmetic of ipfilter.cpp:79-369) and feeds the tile
//     Hadamard; no LDS block, no barrier inside a candidate;
//   * wave reductions use the DPP row_shr / row_bcast ladder (7 VALU ops) instead of ds_bpermute chains; the MVD cost of
//     candidate k is fetched by lane k while the pixel loads are in flight;
//   * the source block sits in registers for the SAD steps (same lane -> same quad for every candidate) and in LDS for the
//     tile stage; candidate / cost arrays are registers (v1 spilt them: 4-8 MB of scratch writes per launch).
#include "common.h"
#include "tiles.h"
#include "filters.h"
#include "searchrange.h"
#include "mestar.h"
#include "meumh.h"
#include <cstdlib>

#ifndef ME2_MIN_WAVES
#define ME2_MIN_WAVES 2
#endif
namespace xh {

struct Mv2 { int x, y; };

__device__ __forceinline__ int uni2(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ---- DPP reductions ---------------------------------------------------------------------------------------------------
template <int CTRL, int ROWMASK, int BANKMASK>
__device__ __forceinline__ int dpp0(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROWMASK, BANKMASK, false); }
// sum over each row of 16 lanes; lane 15 of every row holds its row's total
__device__ __forceinline__ int row16_sum(int v)
{
    int s = v + dpp0<0x111, 0xf, 0xf>(v) + dpp0<0x112, 0xf, 0xf>(v) + dpp0<0x113, 0xf, 0xf>(v);   // row_shr:1,2,3
    s += dpp0<0x114, 0xf, 0xe>(s);                                                                   // row_shr:4, lanes 4..15
    s += dpp0<0x118, 0xf, 0xc>(s);                                                                   // row_shr:8, lanes 8..15
    return s;
}
// sum over aligned groups of 8 lanes; lanes 7 and 15 of every row hold their group's total
__device__ __forceinline__ int row8_sum(int v)
{
    int s = v + dpp0<0x111, 0xf, 0xf>(v) + dpp0<0x112, 0xf, 0xf>(v) + dpp0<0x113, 0xf, 0xf>(v);
    s += dpp0<0x114, 0xf, 0xa>(s);                                                                   // row_shr:4 into lanes 4..7 and 12..15
    return s;
}
// sum over the wave; valid in lane 63
__device__ __forceinline__ int wave64_sum_l63(int v)
{
    int s = row16_sum(v);
    s += dpp0<0x142, 0xa, 0xf>(s);                                                                   // row_bcast:15 into rows 1,3
    s += dpp0<0x143, 0xc, 0xf>(s);                                                                   // row_bcast:31 into rows 2,3
    return s;
}

template <typename P> struct Pk;
template <> struct Pk<uint8_t>
{
    typedef uint32_t T;
    static __device__ __forceinline__ unsigned sad(T a, T b, unsigned acc) { return __builtin_amdgcn_sad_u8(a, b, acc); }
};
template <> struct Pk<uint16_t>
{
    typedef uint2 T;
    static __device__ __forceinline__ unsigned sad(T a, T b, unsigned acc)
    {
        acc = __builtin_amdgcn_sad_u16(a.x, b.x, acc);
        return __builtin_amdgcn_sad_u16(a.y, b.y, acc);
    }
};

// ---- one 4x4 tile of the sub-pel prediction, in registers ----------------------------------------------------------------
// r: reference picture at the tile's integer-pel origin.  Selects copy / luma_hpp / luma_vpp / luma_hvpp exactly as
// subpelCompare (motion.cpp:1583-1597) does.
template <typename P>
__device__ __forceinline__ void tile_pred(const P* r, int64_t stride, int xFrac, int yFrac, int depth, int out[16])
{
    if (!(xFrac | yFrac))
    {
#pragma unroll
        for (int y = 0; y < 4; y++)
            load4(r + y * stride, &out[4 * y]);
    }
    else if (!yFrac)
    {
        const Stage st = stage_for(IF_HPP, depth);
        int c[8];
#pragma unroll
        for (int i = 0; i < 8; i++) c[i] = kLumaFilter[xFrac][i];
#pragma unroll
        for (int y = 0; y < 4; y++)
        {
            int v[11];
            load_span<11>(r + y * stride - 3, v);
#pragma unroll
            for (int o = 0; o < 4; o++)
            {
                int sum = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) sum += v[o + i] * c[i];
                out[4 * y + o] = finish(sum, st);
            }
        }
    }
    else if (!xFrac)
    {
        const Stage st = stage_for(IF_VPP, depth);
        int c[8];
#pragma unroll
        for (int i = 0; i < 8; i++) c[i] = kLumaFilter[yFrac][i];
        int sum[16];
#pragma unroll
        for (int i = 0; i < 16; i++) sum[i] = 0;
#pragma unroll
        for (int i = 0; i < 11; i++)
        {
            int v[4];
            load4(r + (int64_t)(i - 3) * stride, v);
#pragma unroll
            for (int y = 0; y < 4; y++)
                if (i - y >= 0 && i - y < 8)
                {
#pragma unroll
                    for (int o = 0; o < 4; o++) sum[4 * y + o] += v[o] * c[i - y];
                }
        }
#pragma unroll
        for (int i = 0; i < 16; i++) out[i] = finish(sum[i], st);
    }
    else
    {
        const Stage s1 = stage_for(IF_HPS, depth), s2 = stage_for(IF_VSP, depth);
        int c1[8], c2[8];
#pragma unroll
        for (int i = 0; i < 8; i++) { c1[i] = kLumaFilter[xFrac][i]; c2[i] = kLumaFilter[yFrac][i]; }
        int sum[16];
#pragma unroll
        for (int i = 0; i < 16; i++) sum[i] = 0;
#pragma unroll
        for (int i = 0; i < 11; i++)
        {
            int v[11], t[4];
            load_span<11>(r + (int64_t)(i - 3) * stride - 3, v);
#pragma unroll
            for (int o = 0; o < 4; o++)
            {
                int s = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) s += v[o + k] * c1[k];
                t[o] = finish(s, s1);
            }
#pragma unroll
            for (int y = 0; y < 4; y++)
                if (i - y >= 0 && i - y < 8)
                {
#pragma unroll
                    for (int o = 0; o < 4; o++) sum[4 * y + o] += t[o] * c2[i - y];
                }
        }
#pragma unroll
        for (int i = 0; i < 16; i++) out[i] = finish(sum[i], s2);
    }
}

__device__ __forceinline__ Mv2 mv_clip2(Mv2 v, Mv2 lo, Mv2 hi)
{
    Mv2 r = { v.x > hi.x ? hi.x : v.x, v.y > hi.y ? hi.y : v.y };
    r.x = r.x < lo.x ? lo.x : r.x;
    r.y = r.y < lo.y ? lo.y : r.y;
    return r;
}
__device__ __forceinline__ bool mv_in_range2(Mv2 v, Mv2 lo, Mv2 hi) { return v.x >= lo.x && v.x <= hi.x && v.y >= lo.y && v.y <= hi.y; }
__device__ __forceinline__ int sext2b(int v) { return (v & 2) ? (v | ~3) : v; }

__device__ __constant__ const uint8_t kWorkloadB[8][5] = { {1,4,0,4,0}, {1,4,1,4,0}, {1,4,1,4,1}, {2,4,1,4,1}, {2,4,2,4,1}, {1,8,1,8,1}, {2,8,1,8,1}, {2,8,2,8,1} }; // motion.cpp:48-58

// The search-pattern tables of motion.cpp:63-65 as packed nibbles (value + 8): a lookup is two VALU ops on a literal instead of
// a dependent constant-memory load sitting on the serial chain (per lane in the SIMT row-team kernel).
__device__ __forceinline__ int hex2xB(int i) { return (int)((0x679A9767u >> (4 * i)) & 15) - 8; }      // {-1,-2,-1,1,2,1,-1,-2}
__device__ __forceinline__ int hex2yB(int i) { return (int)((0x8668AA86u >> (4 * i)) & 15) - 8; }      // {-2,0,2,2,0,-2,-2,0}
__device__ __forceinline__ int mod6m1B(int i) { return (int)((0x05432105u >> (4 * i)) & 15); }          // {5,0,1,2,3,4,5,0}
__device__ __forceinline__ int sq1xB(int i) { return (int)((0x997797888ull >> (4 * i)) & 15) - 8; }     // {0,0,0,-1,1,-1,-1,1,1}
__device__ __forceinline__ int sq1yB(int i) { return (int)((0x979788978ull >> (4 * i)) & 15) - 8; }     // {0,-1,1,0,0,-1,1,-1,1}

// ---- the team context ------------------------------------------------------------------------------------------------------
template <typename P, int N, int WAVES, bool PLANES, bool CHROMA = false>
struct Team
{
    typedef typename Pk<P>::T Q;
    static constexpr int T = 64 * WAVES;
    static constexpr int QX = N / 4, QUADS = QX * N;                 // 16, 64, 256, 1024
    static constexpr int GS = (WAVES > 1 || QUADS >= 64) ? T : QUADS; // threads per integer-SAD candidate
    static constexpr int NG = T / GS;                                // 4 for 8x8, else 1
    static constexpr int IPT = QUADS / GS;                           // quads per thread per candidate
    static constexpr int TX = N / 4, TILES = TX * TX;

    const P* fref;
    const P* plane0;                // sub-pel planes at the PU origin (plane p at plane0 + p * planeElems), or NULL
    // candidate SADs and MVD costs address memory as (uniform base) + (32-bit unsigned byte offset), see motion3.hip: scalar-base loads
    static constexpr uint32_t kBias = 1u << 22, kCostBias = 1u << 16;
    const char* baseF;              // (const char*)(refPlane - kBias)
    const char* baseP;              // (const char*)(planes - kBias)
    const char* costBase;           // (const char*)(mvcost centre - kCostBias)
    uint32_t org;                   // byte offset of the PU origin from either base (both planes share the geometry)
    uint32_t qoffB[IPT];            // byte offset of this thread's j-th quad inside the block
    int64_t planeElems;
    int64_t stride;
    const uint16_t* cost;
    Mv2 qmvp;
    int depth, tid, lane, wv;
    P* fencL;                       // LDS copy of the source block, stride N
    int* part;                      // LDS [2][8][WAVES] team partial sums (WAVES > 1)
    int phase;                      // alternates the partial buffer
    Q fq[IPT];                      // this thread's quads of the source block
    // bChromaSATD (64x64 PU, 4:2:0): the 32x32 chroma block of a plane is 256 quads = one per thread; tile-major (4 threads = 4 rows of a tile)
    const P* cref[2];
    int strideC;
    int fuc[2][4];

    // the chroma part of subpelCompare (motion.cpp:1601-1660): SATD of the Cb and Cr blocks predicted at vector q, over the team
    __device__ __forceinline__ int chroma_cost(Mv2 q)
    {
        int acc = 0;
#pragma unroll
        for (int j = 0; j < 2; j++)
        {
            int p[4];
            chroma_quad_hv(cref[j], strideC, q.x, q.y, depth, p);
            const int d[4] = { fuc[j][0] - p[0], fuc[j][1] - p[1], fuc[j][2] - p[2], fuc[j][3] - p[3] };
            acc += quad_row_hadamard_abs(d, lane & 1, lane & 2);
        }
        int v[1] = { wave64_sum_l63(acc) };
        team_combine<1>(v);
        return v[0] >> 1;                               // every tile sum is even (pixel.hip): one shift = the per-tile >> 1 of satd_4x4
    }

    __device__ __forceinline__ void team_barrier() const { if (WAVES > 1) __syncthreads(); }

    __device__ __forceinline__ uint16_t cost_at(int i) const { return *reinterpret_cast<const uint16_t*>(costBase + (size_t)(((uint32_t)i + kCostBias) * 2u)); }
    __device__ __forceinline__ int mvcost_lane(int qx, int qy) const { return (int)(uint16_t)(cost_at(qx - qmvp.x) + cost_at(qy - qmvp.y)); }
    template <bool QPEL>
    __device__ __forceinline__ uint32_t cand_off(Mv2 m) const
    {
        if (QPEL)
        {
            const int ph = (m.y & 3) * 4 + (m.x & 3);
            const uint32_t po = planeElems < (1 << 24) ? (uint32_t)__umul24(ph, (int)planeElems) : (uint32_t)ph * (uint32_t)planeElems;
            return org + (po + (uint32_t)(__mul24(m.y >> 2, (int)stride) + (m.x >> 2))) * (uint32_t)sizeof(P);
        }
        return org + (uint32_t)(__mul24(m.y, (int)stride) + m.x) * (uint32_t)sizeof(P);
    }
    __device__ __forceinline__ int mvcost(int qx, int qy) const { return uni2(mvcost_lane(qx, qy)); }

    template <bool QPEL>
    __device__ __forceinline__ const P* cand_ptr(Mv2 m) const
    {
        if (QPEL)
        {
            // 24-bit full-rate multiplies on the candidate chain (stride < 2^23 is checked at dispatch; planes below 2^24 elements up to 4K)
            const int ph = (m.y & 3) * 4 + (m.x & 3);
            const int64_t po = planeElems < (1 << 24) ? (int64_t)__umul24(ph, (int)planeElems) : (int64_t)ph * planeElems;
            return plane0 + po + (__mul24(m.y >> 2, (int)stride) + (m.x >> 2));
        }
        return fref + (__mul24(m.y, (int)stride) + m.x);
    }

    // combine K per-wave values (valid in lane 63 of each wave) over the team; every thread gets the K totals
    template <int K>
    __device__ __forceinline__ void team_combine(int (&v)[K])
    {
        if (WAVES == 1)
        {
#pragma unroll
            for (int k = 0; k < K; k++) v[k] = __builtin_amdgcn_readlane(v[k], 63);
        }
        else
        {
            int* p = part + phase * 8 * WAVES;
            phase ^= 1;
            if (lane == 63)
            {
#pragma unroll
                for (int k = 0; k < K; k++) p[k * WAVES + wv] = v[k];
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < K; k++)
            {
                int s = 0;
#pragma unroll
                for (int w = 0; w < WAVES; w++) s += p[k * WAVES + w];
                v[k] = uni2(s);
            }
        }
    }

    // integer-pel SAD + MVD cost of K (<= 4) candidates (sad / sad_x3 / sad_x4 of the reference)
    // QPEL: candidates are quarter-pel vectors read from the pre-filtered planes (SAD of subpelCompare, motion.cpp:1571)
    template <int K, bool QPEL = false>
    __device__ __forceinline__ void eval_sad(const Mv2 (&c)[K], int (&costs)[K])
    {
        // MVD cost of candidate k is fetched by lane k while the pixel loads fly
        // lane k <- candidate k (a plain compare/select chain on the lane id gets "optimised" into a scratch-memory array)
        Mv2 mine = c[0];
#pragma unroll
        for (int k = 1; k < K; k++)
        {
            const int is = lane == k ? -1 : 0;          // arithmetic select: keeps the compiler from building a lane-indexed array
            mine.x += is & (c[k].x - c[0].x);
            mine.y += is & (c[k].y - c[0].y);
        }
        const int mvc = QPEL ? mvcost_lane(mine.x, mine.y) : mvcost_lane(mine.x * 4, mine.y * 4);
        if (NG == 1)
        {
            unsigned acc[K];
#pragma unroll
            for (int k = 0; k < K; k++)
            {
                acc[k] = 0;
                const uint32_t r = cand_off<QPEL>(c[k]);
                const char* b = QPEL ? baseP : baseF;
#pragma unroll
                for (int j = 0; j < IPT; j++)
                    acc[k] = Pk<P>::sad(ld_unaligned<Q>(b + (size_t)(r + qoffB[j])), fq[j], acc[k]);
            }
#pragma unroll
            for (int k = 0; k < K; k++) costs[k] = wave64_sum_l63((int)acc[k]);
            team_combine<K>(costs);
        }
        else
        {
            // 8x8: four lane groups of 16, one candidate each per pass (idle groups redo the last candidate)
            const int g = lane >> 4, s = lane & 15;
            const int row = s / QX, c4 = (s % QX) * 4;
            constexpr int PASSES = (K + 3) / 4;
            unsigned a[PASSES];
#pragma unroll
            for (int ps = 0; ps < PASSES; ps++)
            {
                Mv2 m = c[K - 1];
#pragma unroll
                for (int k = 4 * ps; k < K - 1 && k < 4 * ps + 4; k++)
                    if (g == k - 4 * ps) m = c[k];
                const P* r = cand_ptr<QPEL>(m);
                a[ps] = Pk<P>::sad(ld_unaligned<Q>(r + (int64_t)row * stride + c4), fq[0], 0u);
            }
#pragma unroll
            for (int ps = 0; ps < PASSES; ps++)
            {
                const int rs = row16_sum((int)a[ps]);
#pragma unroll
                for (int k = 4 * ps; k < K && k < 4 * ps + 4; k++) costs[k] = __builtin_amdgcn_readlane(rs, 16 * (k - 4 * ps) + 15);
            }
        }
#pragma unroll
        for (int k = 0; k < K; k++) costs[k] += __builtin_amdgcn_readlane(mvc, k);
    }

    // mestar.h contract: K points at once on the candidate-parallel path (shift 2)
    template <int K>
    __device__ __forceinline__ void fullpel_costs(const int (&mx)[K], const int (&my)[K], int (&out)[K])
    {
        Mv2 cd[K];
#pragma unroll
        for (int k = 0; k < K; k++) cd[k] = Mv2{ mx[k], my[k] };
        eval_sad<K>(cd, out);
    }
    // mestar.h contract: sad + mvcost((mx, my) << shift)
    __device__ __forceinline__ int fullpel_cost(int mx, int my, int shift)
    {
        const int v = sad_one(mx, my);                       // sad + mvcost(<< 2)
        return shift == 2 ? v : v - mvcost(mx * 4, my * 4) + mvcost(mx << shift, my << shift);
    }
    __device__ __forceinline__ int sad_one(int mx, int my)
    {
        const Mv2 c[1] = { { mx, my } };
        int v[1];
        eval_sad<1>(c, v);
        return v[0];           // includes mvcost(mx*4, my*4)
    }

    // partial sub-pel cost of candidate q over tiles s, s+G, ... : sad (cmp 0) or satd (cmp 1) against the source block
    template <int G>
    __device__ __forceinline__ int subpel_partial(Mv2 q, int cmp, int s) const
    {
        const int xFrac = q.x & 3, yFrac = q.y & 3;
        const P* r = fref + (int64_t)(q.y >> 2) * stride + (q.x >> 2);
        int acc = 0;
#pragma unroll 1
        for (int t = s; t < TILES; t += G)
        {
            const int ty = t / TX, tx = t % TX;
            if (PLANES && sizeof(P) == 1 && cmp)
            {
                // 8-bit SATD of one 4x4 tile on packed 16-bit pairs (see common.h): rows as ((c0, c1), (c2, c3)); the vertical butterflies
                // are plain packed adds across the four row registers, the horizontal ones rotate a register by 16 bits
                const P* pp = plane0 + (int64_t)(yFrac * 4 + xFrac) * planeElems + (int64_t)((q.y >> 2) + ty * 4) * stride + (q.x >> 2) + tx * 4;
                s2v a[4], b[4];
#pragma unroll
                for (int y = 0; y < 4; y++)
                {
                    s2v p01, p23, f01, f23;
                    Pk16<uint8_t>::split(ld_unaligned<uint32_t>(pp + y * stride), p01, p23);
                    Pk16<uint8_t>::split(*reinterpret_cast<const uint32_t*>(fencL + (ty * 4 + y) * N + tx * 4), f01, f23);
                    a[y] = f01 - p01;
                    b[y] = f23 - p23;
                }
                const s2v one = { 1, 1 }, kh = { 1, -1 };
                int tile = 0;
#pragma unroll
                for (int h = 0; h < 2; h++)
                {
                    s2v* v = h ? b : a;
                    const s2v s0 = v[0] + v[1], e0 = v[0] - v[1], s1 = v[2] + v[3], e1 = v[2] - v[3];
                    v[0] = s0 + s1; v[1] = e0 + e1; v[2] = s0 - s1; v[3] = e0 - e1;
                }
#pragma unroll
                for (int y = 0; y < 4; y++)
                {
                    const s2v A = a[y] + b[y], B = a[y] - b[y];
                    const s2v Ar = as_s2(__builtin_amdgcn_alignbit(as_u(A), as_u(A), 16)), Br = as_s2(__builtin_amdgcn_alignbit(as_u(B), as_u(B), 16));
                    const s2v m01 = Ar * kh + A, m23 = Br * kh + B;
                    tile = __builtin_amdgcn_sdot2(__builtin_elementwise_max(m01, -m01), one, tile, false);
                    tile = __builtin_amdgcn_sdot2(__builtin_elementwise_max(m23, -m23), one, tile, false);
                }
                acc += tile >> 1;
                continue;
            }
            int p[16], f[16];
            if (PLANES)
            {
                // pre-filtered phase plane (x265hip_build_subpel_planes): the candidate is a plain block
                const P* pp = plane0 + (int64_t)(yFrac * 4 + xFrac) * planeElems + (int64_t)((q.y >> 2) + ty * 4) * stride + (q.x >> 2) + tx * 4;
#pragma unroll
                for (int y = 0; y < 4; y++)
                    load4(pp + y * stride, &p[4 * y]);
            }
            else
                tile_pred(r + (int64_t)(ty * 4) * stride + tx * 4, stride, xFrac, yFrac, depth, p);
#pragma unroll
            for (int y = 0; y < 4; y++)
                load4(fencL + (ty * 4 + y) * N + tx * 4, &f[4 * y]);
            if (cmp)
            {
#pragma unroll
                for (int i = 0; i < 16; i++) f[i] -= p[i];
                hadamard4x4(f);
                acc += abs_sum16(f) >> 1;
            }
            else
            {
#pragma unroll
                for (int i = 0; i < 16; i++) acc += iabs(f[i] - p[i]);
            }
        }
        return acc;
    }

    // K (<= 4) sub-pel candidates side by side; costs[k] = subpelCompare + mvcost (only meaningful where ok[k])
    template <int K>
    __device__ __forceinline__ void eval_subpel(const Mv2 (&q)[K], const bool (&ok)[K], int cmp, int (&costs)[K])
    {
        Mv2 mine = q[0];
#pragma unroll
        for (int k = 1; k < K; k++)
        {
            const int is = lane == k ? -1 : 0;          // arithmetic select: keeps the compiler from building a lane-indexed array
            mine.x += is & (q[k].x - q[0].x);
            mine.y += is & (q[k].y - q[0].y);
        }
        const int mvc = mvcost_lane(mine.x, mine.y);
        if (WAVES == 1)
        {
            if (K == 1)
            {
                costs[0] = __builtin_amdgcn_readlane(wave64_sum_l63(subpel_partial<64>(q[0], cmp, lane)), 63);
            }
            else
            {
                constexpr int G = K > 4 ? 8 : 16;       // lanes per candidate
                const int g = lane / G, s = lane % G;
                Mv2 m = q[0];
                bool live = false;                      // lane groups beyond K stay idle
#pragma unroll
                for (int k = 0; k < K; k++)
                    if (g == k) { m = q[k]; live = ok[k]; }
                int a = 0;
                if (live)
                    a = subpel_partial<G>(m, cmp, s);
                const int rs = G == 8 ? row8_sum(a) : row16_sum(a);
#pragma unroll
                for (int k = 0; k < K; k++) costs[k] = __builtin_amdgcn_readlane(rs, G * k + G - 1);
            }
        }
        else
        {
            // one wave per candidate (two rounds when K > WAVES).  Lane k of `mine` already holds candidate k (see above), so a
            // wave picks its candidate with v_readlane on its (uniform) wave index — a register array indexed by wv would be
            // demoted to scratch memory by the compiler.
            int okmask = 0;
#pragma unroll
            for (int k = 0; k < K; k++) okmask |= (ok[k] ? 1 : 0) << k;
            int* p = part + phase * 8 * WAVES;
            phase ^= 1;
#pragma unroll
            for (int k0 = 0; k0 < K; k0 += WAVES)
            {
                const int sel = k0 + wv;
                const Mv2 m = { __builtin_amdgcn_readlane(mine.x, sel & 63), __builtin_amdgcn_readlane(mine.y, sel & 63) };
                const bool live = sel < K && ((okmask >> sel) & 1);
                int a = 0;
                if (live)
                    a = subpel_partial<64>(m, cmp, lane);
                a = wave64_sum_l63(a);
                if (lane == 63 && sel < 8) p[sel] = a;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < K; k++) costs[k] = uni2(p[k]);
        }
#pragma unroll
        for (int k = 0; k < K; k++) costs[k] += __builtin_amdgcn_readlane(mvc, k);
    }

    __device__ __forceinline__ int subpel_one(Mv2 q, int cmp)
    {
        const Mv2 c[1] = { q };
        const bool ok[1] = { true };
        int v[1];
        eval_subpel<1>(c, ok, cmp, v);
        return v[0];           // includes mvcost(q)
    }
};

template <typename P, int N, int WAVES, bool PLANES, bool CHROMA = false>
__global__ __launch_bounds__(256, ME2_MIN_WAVES) void motion2_kernel(const P* __restrict__ fencPlane, int64_t strideF,
                                                             const P* __restrict__ refPlane, int64_t strideR,
                                                             const int32_t* __restrict__ pu_xy, const int32_t* __restrict__ mvminA,
                                                             const int32_t* __restrict__ mvmaxA, const int32_t* __restrict__ qmvpA,
                                                             int numCand, const int32_t* __restrict__ mvcA, int merange, int method, int subme,
                                                             const uint16_t* __restrict__ mvcostTab, int depth, int n,
                                                             const P* __restrict__ planes, int64_t planeElems, DeriveRange dr,
                                                             int32_t* __restrict__ outMv, int32_t* __restrict__ outCost, ChromaPlanes cp)
{
    static_assert(!CHROMA || (N == 64 && WAVES == 4 && PLANES), "the chroma SATD term lives in the 64x64 team configuration (smaller PUs: motion3.hip)");
    typedef Team<P, N, WAVES, PLANES, CHROMA> TM;
    typedef typename TM::Q Q;
    constexpr int TPB = (WAVES > 1) ? 1 : 4;                           // teams per workgroup
    __shared__ __attribute__((aligned(16))) P fencS[TPB][N * N];
    __shared__ int partS[TPB][2 * 8 * WAVES];
    const int team = (WAVES > 1) ? 0 : (threadIdx.x >> 6);
    TM c;
    c.tid = (WAVES > 1) ? threadIdx.x : (threadIdx.x & 63);
    c.lane = threadIdx.x & 63;
    c.wv = (WAVES > 1) ? (threadIdx.x >> 6) : 0;
    c.depth = depth;
    c.stride = strideR;
    c.cost = mvcostTab;
    c.costBase = reinterpret_cast<const char*>(mvcostTab) - (size_t)TM::kCostBias * 2;
    c.baseF = reinterpret_cast<const char*>(refPlane) - (size_t)TM::kBias * sizeof(P);
    c.baseP = reinterpret_cast<const char*>(planes) - (size_t)TM::kBias * sizeof(P);
    c.fencL = fencS[team];
    c.part = partS[team];
    c.phase = 0;

    // XCD-aware block order: hardware places workgroup b on XCD b % 8 (MI355X_MICROARCH.md), and each XCD has its own 4 MiB L2.
    // PUs are listed in raster order, so giving XCD x the x-th contiguous eighth of the list keeps every XCD's reads inside one
    // horizontal stripe of the reference picture / planes instead of spraying the whole 43 MB over all eight L2s.
    const int chunk = gridDim.x >> 3;                       // the launcher pads the grid to a multiple of 8
    const int lblock = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    const int teamsTotal = gridDim.x * TPB;
    for (int pu = lblock * TPB + team; pu < n; pu += teamsTotal)
    {
        const int bx = pu_xy[2 * pu], by = pu_xy[2 * pu + 1];
        Mv2 mvmin, mvmax, qmvp;
        if (dr.enable)
        {
            // Search::setSearchRange fused into the launch (searchrange.h); the arrays are still written for the caller
            qmvp = Mv2{ 0, 0 };
            if (dr.mvSrc && dr.srcIdx[pu] >= 0)
                qmvp = Mv2{ dr.mvSrc[2 * dr.srcIdx[pu]], dr.mvSrc[2 * dr.srcIdx[pu] + 1] };
            const SearchRange sr = search_range(dr.picW, dr.picH, dr.maxCUSize, merange, dr.refLagPixels, bx, by, qmvp.x, qmvp.y);
            mvmin = Mv2{ sr.minx, sr.miny };
            mvmax = Mv2{ sr.maxx, sr.maxy };
            if (threadIdx.x == ((WAVES > 1) ? 0 : (team << 6)))
            {
                dr.qmvpO[2 * pu] = qmvp.x; dr.qmvpO[2 * pu + 1] = qmvp.y;
                dr.mvminO[2 * pu] = mvmin.x; dr.mvminO[2 * pu + 1] = mvmin.y;
                dr.mvmaxO[2 * pu] = mvmax.x; dr.mvmaxO[2 * pu + 1] = mvmax.y;
            }
        }
        else
        {
            mvmin = Mv2{ mvminA[2 * pu], mvminA[2 * pu + 1] };
            mvmax = Mv2{ mvmaxA[2 * pu], mvmaxA[2 * pu + 1] };
            qmvp = Mv2{ qmvpA[2 * pu], qmvpA[2 * pu + 1] };
        }
        const Mv2 qmvmin = { mvmin.x * 4, mvmin.y * 4 }, qmvmax = { mvmax.x * 4, mvmax.y * 4 };
        c.qmvp = qmvp;
        c.fref = refPlane + (int64_t)by * strideR + bx;
        c.plane0 = PLANES ? planes + (int64_t)by * strideR + bx : nullptr;
        c.planeElems = planeElems;
        c.org = (TM::kBias + (uint32_t)(by * (int)strideR + bx)) * (uint32_t)sizeof(P);
        // source block: registers (SAD mapping) + LDS (tile stage)
        c.team_barrier();                                               // previous PU's tile reads are done
        {
            const P* f = fencPlane + (int64_t)by * strideF + bx;
#pragma unroll
            for (int j = 0; j < TM::IPT; j++)
            {
                const int q = (TM::NG == 1 ? c.tid : (c.lane & 15)) + j * TM::GS, row = q / TM::QX, c4 = (q % TM::QX) * 4;
                c.fq[j] = ld_unaligned<Q>(f + (int64_t)row * strideF + c4);
                c.qoffB[j] = (uint32_t)(row * (int)strideR + c4) * (uint32_t)sizeof(P);
                if (TM::NG == 1 || c.lane < 16)
                    *reinterpret_cast<Q*>(c.fencL + row * N + c4) = c.fq[j];
            }
        }
        if (CHROMA)
        {
            const int t = c.tid >> 2, r = c.tid & 3;                    // 64 tiles of the 32x32 chroma block, 4 threads = the 4 rows of a tile
            const int row = (t >> 3) * 4 + r, col = (t & 7) * 4;
            const int64_t offF = (int64_t)((by >> 1) + row) * cp.strideFC + (bx >> 1) + col;
            const int64_t offR = (int64_t)((by >> 1) + row) * cp.strideRC + (bx >> 1) + col;
            c.strideC = (int)cp.strideRC;
            load4((const P*)cp.fencCb + offF, c.fuc[0]);
            load4((const P*)cp.fencCr + offF, c.fuc[1]);
            c.cref[0] = (const P*)cp.refCb + offR;
            c.cref[1] = (const P*)cp.refCr + offR;
        }
        if (WAVES > 1)
            __syncthreads();
        else
        {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        }

#define YOK(yy) (((yy) >= mvmin.y) & ((yy) <= mvmax.y))
#define LT1(v) do { const int v_ = (v); if (v_ < bcost) bcost = v_; } while (0)
        // ---- predictor, zero and candidates (motion.cpp:761-812)
        const Mv2 pmv = mv_clip2(qmvp, qmvmin, qmvmax);
        Mv2 bestpre = pmv;
        // bprecost = subpelCompare(pmv, sad) WITHOUT mv cost (motion.cpp:771); subpel_one adds mvcost(pmv), take it out again
        int bprecost, bcost;
        Mv2 bmv = { (pmv.x + 2) >> 2, (pmv.y + 2) >> 2 };
        if (PLANES)
        {
            // the three opening measurements (predictor, its full-pel rounding, zero) are independent: one 3-wide step
            const Mv2 q3[3] = { pmv, { bmv.x * 4, bmv.y * 4 }, { 0, 0 } };
            int cs[3];
            c.template eval_sad<3, true>(q3, cs);
            bprecost = cs[0] - c.mvcost(pmv.x, pmv.y);
            if (CHROMA) bprecost += c.chroma_cost(pmv);
            bcost = bprecost;
            if ((pmv.x & 3) | (pmv.y & 3))
                bcost = cs[1];
            if ((pmv.x | pmv.y) && cs[2] < bcost)
            {
                bcost = cs[2];
                bmv.x = 0;
                bmv.y = max(min(0, mvmax.y), mvmin.y);
            }
        }
        else
        {
            bprecost = c.subpel_one(pmv, 0) - c.mvcost(pmv.x, pmv.y);
            bcost = bprecost;
            if ((pmv.x & 3) | (pmv.y & 3))
                bcost = c.sad_one(bmv.x, bmv.y);
            if (pmv.x | pmv.y)
            {
                const int cst = c.sad_one(0, 0);
                if (cst < bcost)
                {
                    bcost = cst;
                    bmv.x = 0;
                    bmv.y = max(min(0, mvmax.y), mvmin.y);
                }
            }
        }
        for (int i = 0; i < numCand; i++)
        {
            const Mv2 raw = { mvcA[((int64_t)pu * numCand + i) * 2], mvcA[((int64_t)pu * numCand + i) * 2 + 1] };
            const Mv2 m = mv_clip2(raw, qmvmin, qmvmax);
            if ((m.x | m.y) && !(m.x == pmv.x && m.y == pmv.y) && !(m.x == bestpre.x && m.y == bestpre.y))
            {
                int cst = c.subpel_one(m, 0);
                if (CHROMA) cst += c.chroma_cost(m);
                if (cst < bprecost)
                {
                    bprecost = cst;
                    bestpre = m;
                }
            }
        }

        // X265_UMH_SEARCH (meumh.h) ends either for good or in the hexagon refine of X265_HEX_SEARCH (goto me_hex2, motion.cpp:1127)
        int meth = method, hexRange = merange;       // UMH scales the range the hexagon refine then runs with (motion.cpp:1039)
        if (meth == 2)
            meth = umh_search(c, mvmin.x, mvmin.y, mvmax.x, mvmax.y, hexRange, bmv.x, bmv.y, bcost, (pmv.x + 2) >> 2, (pmv.y + 2) >> 2, numCand,
                              mvcA + (int64_t)pu * numCand * 2, qmvp.x, qmvp.y, N, N) ? 1 : -1;
        if (meth == 0)
        {
            // X265_DIA_SEARCH, motion.cpp:831-852
            bcost <<= 4;
            int i = merange;
            do
            {
                const Mv2 cd[4] = { { bmv.x, bmv.y - 1 }, { bmv.x, bmv.y + 1 }, { bmv.x - 1, bmv.y }, { bmv.x + 1, bmv.y } };
                int costs[4];
                c.template eval_sad<4>(cd, costs);
                if (YOK(bmv.y - 1)) LT1((costs[0] << 4) + 1);
                if (YOK(bmv.y + 1)) LT1((costs[1] << 4) + 3);
                LT1((costs[2] << 4) + 4);
                LT1((costs[3] << 4) + 12);
                if (!(bcost & 15))
                    break;
                bmv.x -= sext2b((bcost >> 2) & 3);
                bmv.y -= sext2b(bcost & 3);
                bcost &= ~15;
            }
            while (--i && mv_in_range2(bmv, mvmin, mvmax));
            bcost >>= 4;
        }
        else if (meth == 1)
        {
            // X265_HEX_SEARCH, motion.cpp:855-944
            {
                // the two sad_x3 calls of motion.cpp:857-873 are independent: one 6-wide evaluation, replayed in order
                const Mv2 cd[6] = { { bmv.x - 2, bmv.y }, { bmv.x - 1, bmv.y + 2 }, { bmv.x + 1, bmv.y + 2 },
                                    { bmv.x + 2, bmv.y }, { bmv.x + 1, bmv.y - 2 }, { bmv.x - 1, bmv.y - 2 } };
                int cs[6];
                c.template eval_sad<6>(cd, cs);
                bcost <<= 3;
                if (YOK(bmv.y)) LT1((cs[0] << 3) + 2);
                if (YOK(bmv.y + 2))
                {
                    LT1((cs[1] << 3) + 3);
                    LT1((cs[2] << 3) + 4);
                }
                if (YOK(bmv.y)) LT1((cs[3] << 3) + 5);
                if (YOK(bmv.y - 2))
                {
                    LT1((cs[4] << 3) + 6);
                    LT1((cs[5] << 3) + 7);
                }
            }
            if (bcost & 7)
            {
                int dir = (bcost & 7) - 2;
                if (YOK(bmv.y + hex2yB(dir + 1)))
                {
                    bmv.x += hex2xB(dir + 1);
                    bmv.y += hex2yB(dir + 1);
                    for (int i = (hexRange >> 1) - 1; i > 0 && mv_in_range2(bmv, mvmin, mvmax); i--)
                    {
                        const Mv2 cd[3] = { { bmv.x + hex2xB(dir + 0), bmv.y + hex2yB(dir + 0) },
                                            { bmv.x + hex2xB(dir + 1), bmv.y + hex2yB(dir + 1) },
                                            { bmv.x + hex2xB(dir + 2), bmv.y + hex2yB(dir + 2) } };
                        int cs[3];
                        c.template eval_sad<3>(cd, cs);
                        bcost &= ~7;
                        if (YOK(cd[0].y)) LT1((cs[0] << 3) + 1);
                        if (YOK(cd[1].y)) LT1((cs[1] << 3) + 2);
                        if (YOK(cd[2].y)) LT1((cs[2] << 3) + 3);
                        if (!(bcost & 7))
                            break;
                        dir += (bcost & 7) - 2;
                        dir = mod6m1B(dir + 1);
                        bmv.x += hex2xB(dir + 1);
                        bmv.y += hex2yB(dir + 1);
                    }
                }
            }
            bcost >>= 3;
            // square refine, motion.cpp:918-942
            int dir = 0;
            {
                // both sad_x4 calls (motion.cpp:920-937) are centred on the same bmv: one 8-wide evaluation
                const Mv2 cd[8] = { { bmv.x, bmv.y - 1 }, { bmv.x, bmv.y + 1 }, { bmv.x - 1, bmv.y }, { bmv.x + 1, bmv.y },
                                    { bmv.x - 1, bmv.y - 1 }, { bmv.x - 1, bmv.y + 1 }, { bmv.x + 1, bmv.y - 1 }, { bmv.x + 1, bmv.y + 1 } };
                int costs[8];
                c.template eval_sad<8>(cd, costs);
                if (YOK(bmv.y - 1) && costs[0] < bcost) { bcost = costs[0]; dir = 1; }
                if (YOK(bmv.y + 1) && costs[1] < bcost) { bcost = costs[1]; dir = 2; }
                if (costs[2] < bcost) { bcost = costs[2]; dir = 3; }
                if (costs[3] < bcost) { bcost = costs[3]; dir = 4; }
                if (YOK(bmv.y - 1) && costs[4] < bcost) { bcost = costs[4]; dir = 5; }
                if (YOK(bmv.y + 1) && costs[5] < bcost) { bcost = costs[5]; dir = 6; }
                if (YOK(bmv.y - 1) && costs[6] < bcost) { bcost = costs[6]; dir = 7; }
                if (YOK(bmv.y + 1) && costs[7] < bcost) { bcost = costs[7]; dir = 8; }
            }
            bmv.x += sq1xB(dir);
            bmv.y += sq1yB(dir);
        }
        else if (meth == 3)
            star_search(c, mvmin.x, mvmin.y, mvmax.x, mvmax.y, merange, bmv.x, bmv.y, bcost);  // X265_STAR_SEARCH (mestar.h)
        else if (meth == 5)
        {
            // X265_FULL_SEARCH, motion.cpp:1397-1441: raster order, strict '<' keeps the first minimum
            for (int ty = mvmin.y; ty <= mvmax.y; ty++)
                for (int tx = mvmin.x; tx <= mvmax.x; tx += 4)
                {
                    const int K = min(4, mvmax.x - tx + 1);
                    const Mv2 cd[4] = { { tx, ty }, { tx + min(1, K - 1), ty }, { tx + min(2, K - 1), ty }, { tx + min(3, K - 1), ty } };
                    int costs[4];
                    c.template eval_sad<4>(cd, costs);
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (k < K && costs[k] < bcost)
                        {
                            bcost = costs[k];
                            bmv.x = tx + k;
                            bmv.y = ty;
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
            // motion.cpp:1504-1561; the directions of one iteration are evaluated together and replayed in order
            const int hpelIters = kWorkloadB[subme][0], hpelDirs = kWorkloadB[subme][1];
            const int qpelIters = kWorkloadB[subme][2], qpelDirs = kWorkloadB[subme][3], hpelSatd = kWorkloadB[subme][4];
            int hpelcomp = 0;
            bool firstDone = false;
            if (hpelSatd)
            {
                hpelcomp = 1;
                if (hpelDirs == 4)
                {
                    // bcost = satd(bmv) (motion.cpp:1507) and the first half-pel iteration are independent: 5-wide evaluation
                    Mv2 q[5]; bool ok[5]; int cs[5];
                    q[0] = bmv; ok[0] = true;
#pragma unroll
                    for (int k = 0; k < 4; k++)
                    {
                        q[k + 1] = Mv2{ bmv.x + sq1xB(1 + k) * 2, bmv.y + sq1yB(1 + k) * 2 };
                        ok[k + 1] = !((q[k + 1].y < qmvmin.y) | (q[k + 1].y > qmvmax.y));
                    }
                    c.template eval_subpel<5>(q, ok, 1, cs);
                    if (CHROMA)
                    {
#pragma unroll
                        for (int k = 0; k < 5; k++)
                            if (ok[k]) cs[k] += c.chroma_cost(q[k]);
                    }
                    bcost = cs[0];
                    int bdir = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (ok[k + 1] && cs[k + 1] < bcost) { bcost = cs[k + 1]; bdir = 1 + k; }
                    if (bdir)
                    {
                        bmv.x += sq1xB(bdir) * 2;
                        bmv.y += sq1yB(bdir) * 2;
                    }
                    firstDone = true;
                    if (!bdir)
                        hpelcomp = 3;                   // satd + "first iteration found nothing: stop half-pel iterations"
                }
                else
                {
                    bcost = c.subpel_one(bmv, 1);
                    if (CHROMA) bcost += c.chroma_cost(bmv);
                }
            }
#define REFINE(ITERS, DIRS, STEP, CMP) \
            for (int iter = 0; iter < (ITERS); iter++) \
            { \
                int bdir = 0; \
                for (int d0 = 1; d0 <= (DIRS); d0 += 4) \
                { \
                    Mv2 q[4]; bool ok[4]; int cs[4]; \
                    _Pragma("unroll") for (int k = 0; k < 4; k++) \
                    { \
                        q[k] = Mv2{ bmv.x + sq1xB(d0 + k) * (STEP), bmv.y + sq1yB(d0 + k) * (STEP) }; \
                        ok[k] = !((q[k].y < qmvmin.y) | (q[k].y > qmvmax.y)); \
                    } \
                    c.template eval_subpel<4>(q, ok, (CMP), cs); \
                    if (CHROMA) { _Pragma("unroll") for (int k = 0; k < 4; k++) if (ok[k]) cs[k] += c.chroma_cost(q[k]); } \
                    _Pragma("unroll") for (int k = 0; k < 4; k++) \
                        if (ok[k] && cs[k] < bcost) { bcost = cs[k]; bdir = d0 + k; } \
                } \
                if (bdir) \
                { \
                    bmv.x += sq1xB(bdir) * (STEP); \
                    bmv.y += sq1yB(bdir) * (STEP); \
                } \
                else \
                    break; \
            }
            {
                const int itersLeft = (hpelcomp == 3) ? 0 : hpelIters - (firstDone ? 1 : 0);
                const int cmpH = hpelcomp ? 1 : 0;
                REFINE(itersLeft, hpelDirs, 2, cmpH)
            }
            if (!hpelSatd)
            {
                bcost = c.subpel_one(bmv, 1);
                if (CHROMA) bcost += c.chroma_cost(bmv);
            }
            REFINE(qpelIters, qpelDirs, 1, 1)
#undef REFINE
        }
#undef YOK
#undef LT1
        if (threadIdx.x == ((WAVES > 1) ? 0 : (team << 6)))
        {
            outMv[2 * pu] = bmv.x;
            outMv[2 * pu + 1] = bmv.y;
            outCost[pu] = bcost;
        }
    }
}

template <typename P, int N, int WAVES>
static int launch_motion2(const void* fencPlane, int64_t strideF, const void* refPlane, int64_t strideR, const int32_t* pu_xy,
                          const int32_t* mvmin, const int32_t* mvmax, const int32_t* qmvp, int numCand, const int32_t* mvc,
                          int merange, int method, int subme, const uint16_t* mvcost, int depth, int n, const void* planes,
                          int64_t planeElems, const DeriveRange& dr, int32_t* outMv, int32_t* outCost, hipStream_t st)
{
    constexpr int TPB = (WAVES > 1) ? 1 : 4;
    const int blocks = (grid_for((n + TPB - 1) / TPB, 256 * 32) + 7) & ~7;      // multiple of 8 for the XCD-aware order
    dim3 grid(blocks), block(64 * (WAVES > 1 ? WAVES : 4));
    if (planes)
        hipLaunchKernelGGL((motion2_kernel<P, N, WAVES, true>), grid, block, 0, st, (const P*)fencPlane, strideF, (const P*)refPlane, strideR, pu_xy,
                           mvmin, mvmax, qmvp, numCand, mvc, merange, method, subme, mvcost, depth, n, (const P*)planes, planeElems, dr, outMv, outCost, ChromaPlanes{});
    else
        hipLaunchKernelGGL((motion2_kernel<P, N, WAVES, false>), grid, block, 0, st, (const P*)fencPlane, strideF, (const P*)refPlane, strideR, pu_xy,
                           mvmin, mvmax, qmvp, numCand, mvc, merange, method, subme, mvcost, depth, n, (const P*)planes, planeElems, dr, outMv, outCost, ChromaPlanes{});
    XH_LAUNCH_CHECK("motion2_kernel");
    return X265HIP_OK;
}

// 64x64 PUs with the chroma SATD term (bChromaSATD): the 4-wave team kernel on planes
template <typename P>
static int launch_motion2_chroma64(const void* fencPlane, int64_t strideF, int64_t strideR, const int32_t* pu_xy, int merange, int method, int subme,
                                   const uint16_t* mvcost, int depth, int n, const void* planes, int64_t planeElems, const DeriveRange& dr,
                                   const ChromaPlanes& cp, int32_t* outMv, int32_t* outCost, hipStream_t st)
{
    const int blocks = (grid_for(n, 256 * 32) + 7) & ~7;
    hipLaunchKernelGGL((motion2_kernel<P, 64, 4, true, true>), dim3(blocks), dim3(256), 0, st, (const P*)fencPlane, strideF, (const P*)planes, strideR, pu_xy,
                       dr.mvminO, dr.mvmaxO, dr.qmvpO, 0, (const int32_t*)nullptr, merange, method, subme, mvcost, depth, n, (const P*)planes, planeElems, dr,
                       outMv, outCost, cp);
    XH_LAUNCH_CHECK("motion2_kernel(chroma)");
    return X265HIP_OK;
}

int motion3_dispatch(int depth, int size, const void* fencPlane, int64_t strideF, int64_t strideR, const int32_t* pu_xy,
                     const int32_t* mvmin, const int32_t* mvmax, const int32_t* qmvp, int numCand, const int32_t* mvc, int merange,
                     int method, int subme, const uint16_t* mvcost, int n, const void* planes, int64_t planeElems, const DeriveRange* drp,
                     int32_t* outMv, int32_t* outCost, hipStream_t st, int* rc, const ChromaPlanes* cpp = nullptr);

// returns 1 when the shape is handled here, 0 when the caller should use the generic kernel of motion.hip
int motion2_dispatch(int depth, int w, int h, const void* fencPlane, int64_t strideF, const void* refPlane, int64_t strideR,
                     const int32_t* pu_xy, const int32_t* mvmin, const int32_t* mvmax, const int32_t* qmvp, int numCand,
                     const int32_t* mvc, int merange, int method, int subme, const uint16_t* mvcost, int n, const void* planes,
                     int64_t planeElems, int32_t* outMv, int32_t* outCost, hipStream_t st, int* rc, const DeriveRange* drp = nullptr)
{
    // 8x8 / 16x16 with planes: four PUs per wave (motion3.hip) unless X265HIP_ME_TEAM=1 asks for the wave-per-PU kernel
    static const bool forceTeam = getenv("X265HIP_ME_TEAM") != nullptr;
    if (!forceTeam && w == h && motion3_dispatch(depth, w, fencPlane, strideF, strideR, pu_xy, mvmin, mvmax, qmvp, numCand, mvc, merange, method,
                                                 subme, mvcost, n, planes, planeElems, drp, outMv, outCost, st, rc))
        return 1;
    DeriveRange dr{};
    if (drp) dr = *drp;
    if (w != h || !(w == 8 || w == 16 || w == 32 || w == 64) || strideR >= (1 << 23))
        return 0;
#define M2(P, N, WV) *rc = launch_motion2<P, N, WV>(fencPlane, strideF, refPlane, strideR, pu_xy, mvmin, mvmax, qmvp, numCand, mvc, merange, \
                                                    method, subme, mvcost, depth, n, planes, planeElems, dr, outMv, outCost, st)
    if (depth == 8)
    {
        if (w == 8) M2(uint8_t, 8, 1); else if (w == 16) M2(uint8_t, 16, 1); else if (w == 32) M2(uint8_t, 32, 4); else M2(uint8_t, 64, 4);
    }
    else
    {
        if (w == 8) M2(uint16_t, 8, 1); else if (w == 16) M2(uint16_t, 16, 1); else if (w == 32) M2(uint16_t, 32, 4); else M2(uint16_t, 64, 4);
    }
#undef M2
    return 1;
}

int motion_estimate_fused(int depth, int size, const void* fencPlane, int64_t strideF, const void* refPlane, int64_t strideR,
                          const void* planes, int64_t planeElems, const int32_t* pu_xy, const DeriveRange& dr, int merange, int method,
                          int subme, const uint16_t* mvcost, int n, int32_t* outMv, int32_t* outCost, hipStream_t st)
{
    int rc = X265HIP_OK;
    if (!n) return rc;
    if (!motion2_dispatch(depth, size, size, fencPlane, strideF, refPlane, strideR, pu_xy, dr.mvminO, dr.mvmaxO, dr.qmvpO, 0, nullptr, merange,
                          method, subme, mvcost, n, planes, planeElems, outMv, outCost, st, &rc, &dr))
        return set_error(X265HIP_EINVAL, "motion_estimate_fused: PU size %d is not a team-kernel shape", size);
    return rc;
}

int motion_estimate_fused_chroma(int depth, int size, const void* fencPlane, int64_t strideF, int64_t strideR, const void* planes, int64_t planeElems,
                                 const ChromaPlanes& cp, const int32_t* pu_xy, const DeriveRange& dr, int merange, int method, int subme,
                                 const uint16_t* mvcost, int n, int32_t* outMv, int32_t* outCost, hipStream_t st, int* rc)
{
    *rc = X265HIP_OK;
    if (!n) return 1;
    if (size == 64 && planes && dr.enable)
    {
        *rc = depth == 8 ? launch_motion2_chroma64<uint8_t>(fencPlane, strideF, strideR, pu_xy, merange, method, subme, mvcost, depth, n, planes, planeElems, dr, cp,
                                                            outMv, outCost, st)
                         : launch_motion2_chroma64<uint16_t>(fencPlane, strideF, strideR, pu_xy, merange, method, subme, mvcost, depth, n, planes, planeElems, dr, cp,
                                                             outMv, outCost, st);
        return 1;
    }
    return motion3_dispatch(depth, size, fencPlane, strideF, strideR, pu_xy, dr.mvminO, dr.mvmaxO, dr.qmvpO, 0, nullptr, merange, method, subme,
                            mvcost, n, planes, planeElems, &dr, outMv, outCost, st, rc, &cp);
}

} // namespace xh
