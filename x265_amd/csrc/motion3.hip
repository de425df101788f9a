// motion3.hip — MotionEstimate::motionEstimate for 8x8 and 16x16 PUs, "row team" kernel: FOUR PUs per wave.
//
// Same reference semantics and bit-exact outputs as motion.hip / motion2.hip (source/encoder/motion.cpp:739-1569).
// For small PUs a whole wave per PU wastes lanes: the team kernel of motion2.hip spends ~1200 VALU instructions per 8x8 PU
// with 4-20 useful lanes (profiles/r01_v2_pmc_sq.txt).  Here one PU is owned by one DPP ROW of 16 lanes, so a wave64
// carries four independent searches in SIMT fashion:
//   * every decision variable (bmv, bcost, dir ...) is an ordinary per-thread value, identical across the 16 lanes of a row;
//     the four rows of a wave diverge freely (the hardware exec mask handles it);
//   * a candidate costs each lane one packed quad load per 64 pixels of PU + v_sad_u8, then a 4-step `row_ror` DPP all-reduce
//     leaves the block sum in all 16 lanes — no readlane, no LDS, no barrier;
//   * lanes are tile-major (4 consecutive lanes = the 4 rows of one 4x4 tile), so the SATD's vertical Hadamard runs across
//     DPP quads (tiles.h idea) and its horizontal one in registers;
//   * every candidate, integer or sub-pel, is a block read from the pre-filtered quarter-pel planes
//     (x265hip_build_subpel_planes; plane 0 is the picture), so the serial chain contains no filtering.
#include "common.h"
#include <cstdlib>
#include "searchrange.h"
#include "mestar.h"
#include "meumh.h"
#include "filters.h"

namespace xh {

struct Mv3 { int x, y; };

template <typename P> struct Pk3;
template <> struct Pk3<uint8_t>
{
    typedef uint32_t T;
    static __device__ __forceinline__ unsigned sad(T a, T b, unsigned acc) { return __builtin_amdgcn_sad_u8(a, b, acc); }
    static __device__ __forceinline__ void unpack(T a, int v[4]) { v[0] = a & 255; v[1] = (a >> 8) & 255; v[2] = (a >> 16) & 255; v[3] = a >> 24; }
};
template <> struct Pk3<uint16_t>
{
    typedef uint2 T;
    static __device__ __forceinline__ unsigned sad(T a, T b, unsigned acc)
    {
        acc = __builtin_amdgcn_sad_u16(a.x, b.x, acc);
        return __builtin_amdgcn_sad_u16(a.y, b.y, acc);
    }
    static __device__ __forceinline__ void unpack(T a, int v[4]) { v[0] = a.x & 0xffff; v[1] = a.x >> 16; v[2] = a.y & 0xffff; v[3] = a.y >> 16; }
};

// packed 16-bit SATD helpers (s2v, Pk16) live in common.h
// dpp_all / row_allsum (16-lane DPP all-reduce) live in common.h

__device__ __constant__ const uint8_t kWorkloadC[8][5] = { {1,4,0,4,0}, {1,4,1,4,0}, {1,4,1,4,1}, {2,4,1,4,1}, {2,4,2,4,1}, {1,8,1,8,1}, {2,8,1,8,1}, {2,8,2,8,1} }; // motion.cpp:48-58

// The search-pattern tables of motion.cpp:63-65 as packed nibbles (value + 8): a lookup is two VALU ops on a literal instead of
// a dependent constant-memory load sitting on the serial chain (per lane in the SIMT row-team kernel).
__device__ __forceinline__ int hex2xC(int i) { return (int)((0x679A9767u >> (4 * i)) & 15) - 8; }      // {-1,-2,-1,1,2,1,-1,-2}
__device__ __forceinline__ int hex2yC(int i) { return (int)((0x8668AA86u >> (4 * i)) & 15) - 8; }      // {-2,0,2,2,0,-2,-2,0}
__device__ __forceinline__ int mod6m1C(int i) { return (int)((0x05432105u >> (4 * i)) & 15); }          // {5,0,1,2,3,4,5,0}
__device__ __forceinline__ int sq1xC(int i) { return (int)((0x997797888ull >> (4 * i)) & 15) - 8; }     // {0,0,0,-1,1,-1,-1,1,1}
__device__ __forceinline__ int sq1yC(int i) { return (int)((0x979788978ull >> (4 * i)) & 15) - 8; }     // {0,-1,1,0,0,-1,1,-1,1}

// sum over the team: a 16-lane row (DPP only) or the whole wave (row all-reduce + 4 readlanes; the wave is then one PU and
// its control flow is uniform)
template <int TEAM>
__device__ __forceinline__ int team_allsum(int v)
{
    if (TEAM == 8)
    {
        // eight lanes = half a DPP row: butterflies inside the quad, then the mirrored half (lane i <-> 7 - i sits in the other quad)
        v += dpp_all<0xB1>(v);          // quad_perm [1,0,3,2]
        v += dpp_all<0x4E>(v);          // quad_perm [2,3,0,1]
        v += dpp_all<0x141>(v);         // row_half_mirror
        return v;
    }
    v = row_allsum(v);
    if (TEAM == 64)
        v = __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
    return v;
}

template <typename P, int N, int TEAM, bool CHROMA = false>
struct RowTeam
{
    typedef typename Pk3<P>::T Q;
    static constexpr int IPT = N * N / 4 / TEAM;       // quads per lane: 1 (8x8), 4 (16x16 on 16 lanes, 32x32 on 64), 16 (64x64 on 64)
    static constexpr int TX = N / 4;                   // tiles per row
    // Candidate blocks are addressed as (uniform base) + (32-bit unsigned byte offset): the loads then take the scalar-base form
    // (global_load v, voffset, s[base]) and the per-load 64-bit pointer arithmetic disappears.  The base is the plane pointer moved back
    // by kBias elements so that offsets of candidates above / left of the picture origin stay non-negative; nothing is read at the base.
    static constexpr uint32_t kBias = 1u << 22;
    const char* base;                                  // (const char*)(planes - kBias), the same for every lane
    uint32_t org;                                      // byte offset of phase plane 0 at the PU origin from `base`
    const char* costBase;                              // (const char*)(mvcost centre - kCostBias)
    static constexpr uint32_t kCostBias = 1u << 16;
    int64_t planeElems;
    bool smallPlane;                                   // planeElems < 2^24: the phase offset fits a 24-bit multiply
    int stride;
    const uint16_t* cost;
    Mv3 qmvp;
    int s;                                             // lane within the team (0..TEAM-1)
    int qoff[IPT];                                     // element offset of this lane's quads inside the PU (row * stride + col)
    Q fq[IPT];
    int fu[IPT][4];
    s2v fp[IPT][2];                                    // the same source samples as packed pairs (0,1) and (2,3)
    bool pk16;                                         // depth <= 10: packed 16-bit SATD
    // bChromaSATD (4:2:0): the chroma block is (N/2)^2 per plane = N*N/16 quads; a lane holds one quad per pass.  8x8 PUs put Cb in
    // lanes 0-3 and Cr in lanes 4-7 of one pass, 16x16 / 32x32 PUs take one pass per plane with every lane busy.
    static constexpr int CPASS = N == 8 ? 1 : 2;
    const P* cref[CPASS];                               // reference chroma plane at this lane's quad (mv (0,0))
    int strideC, depth;
    int fuc[CPASS][4];
    bool cact;                                          // lane takes part in the chroma term

    __device__ __forceinline__ uint16_t cost_at(int i) const { return *reinterpret_cast<const uint16_t*>(costBase + (size_t)(((uint32_t)i + kCostBias) * 2u)); }
    __device__ __forceinline__ int mvcost(int qx, int qy) const { return (int)(uint16_t)(cost_at(qx - qmvp.x) + cost_at(qy - qmvp.y)); }
    template <typename T> __device__ __forceinline__ T ld_off(uint32_t byteOff) const { return ld_unaligned<T>(base + (size_t)byteOff); }

    __device__ __forceinline__ uint32_t cand(Mv3 q) const
    {
        // full-rate 24-bit multiplies instead of v_mul_lo_u32 / v_mad_u64_u32 (quarter rate) on the candidate chain: the stride is below
        // 2^23 (checked at dispatch), the phase index is 0..15 and the plane size is below 2^24 elements up to 4K (smallPlane)
        const int ph = (q.y & 3) * 4 + (q.x & 3);
        const uint32_t po = smallPlane ? (uint32_t)__umul24(ph, (int)planeElems) : (uint32_t)ph * (uint32_t)planeElems;
        return org + (po + (uint32_t)(__mul24(q.y >> 2, stride) + (q.x >> 2))) * (uint32_t)sizeof(P);
    }
    // subpelCompare(..., sad) (motion.cpp:1571) / sad() of the block at quarter-pel vector q, WITHOUT mv cost
    __device__ __forceinline__ int sad_q(Mv3 q) const
    {
        const uint32_t r = cand(q);
        unsigned acc = 0;
#pragma unroll
        for (int j = 0; j < IPT; j++)
            acc = Pk3<P>::sad(ld_off<Q>(r + (uint32_t)qoff[j]), fq[j], acc);
        return team_allsum<TEAM>((int)acc);
    }
    // subpelCompare(..., satd): 4x4 Hadamard tiles, rows of a tile in the 4 lanes of a DPP quad
    __device__ __forceinline__ int satd_q(Mv3 q) const
    {
        const uint32_t r = cand(q);
        const bool hi1 = s & 1, hi2 = s & 2;
        int acc = 0;
        if (sizeof(P) == 1 || pk16)                  // 8-bit: always (the 32-bit path below is not even compiled in)
        {
            const s2v one = { 1, 1 }, kh = { 1, -1 };
            const s2v k1 = hi1 ? -one : one, k2 = hi2 ? -one : one;
#pragma unroll
            for (int j = 0; j < IPT; j++)
            {
                s2v p01, p23;
                Pk16<P>::split(ld_off<Q>(r + (uint32_t)qoff[j]), p01, p23);
                const s2v d01 = fp[j][0] - p01, d23 = fp[j][1] - p23;
                const s2v A = d01 + d23, B = d01 - d23;                          // (d0+d2, d1+d3), (d0-d2, d1-d3)
                const s2v Ar = as_s2(__builtin_amdgcn_alignbit(as_u(A), as_u(A), 16)), Br = as_s2(__builtin_amdgcn_alignbit(as_u(B), as_u(B), 16));
                s2v m01 = Ar * kh + A, m23 = Br * kh + B;                        // (h0, -h1), (h2, -h3): signs do not matter below
                s2v pr = as_s2((uint32_t)__builtin_amdgcn_mov_dpp((int)as_u(m01), 0xB1, 0xF, 0xF, true));     // quad_perm [1,0,3,2]
                m01 = m01 * k1 + pr;
                pr = as_s2((uint32_t)__builtin_amdgcn_mov_dpp((int)as_u(m23), 0xB1, 0xF, 0xF, true));
                m23 = m23 * k1 + pr;
                pr = as_s2((uint32_t)__builtin_amdgcn_mov_dpp((int)as_u(m01), 0x4E, 0xF, 0xF, true));         // quad_perm [2,3,0,1]
                m01 = m01 * k2 + pr;
                pr = as_s2((uint32_t)__builtin_amdgcn_mov_dpp((int)as_u(m23), 0x4E, 0xF, 0xF, true));
                m23 = m23 * k2 + pr;
                acc = __builtin_amdgcn_sdot2(__builtin_elementwise_max(m01, -m01), one, acc, false);
                acc = __builtin_amdgcn_sdot2(__builtin_elementwise_max(m23, -m23), one, acc, false);
            }
            return team_allsum<TEAM>(acc) >> 1;
        }
#pragma unroll
        for (int j = 0; j < IPT; j++)
        {
            int p[4];
            Pk3<P>::unpack(ld_off<Q>(r + (uint32_t)qoff[j]), p);
            const int d0 = fu[j][0] - p[0], d1 = fu[j][1] - p[1], d2 = fu[j][2] - p[2], d3 = fu[j][3] - p[3];
            const int s01 = d0 + d1, e01 = d0 - d1, s23 = d2 + d3, e23 = d2 - d3;
            int m[4] = { s01 + s23, s01 - s23, e01 + e23, e01 - e23 };
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                const int pr = __builtin_amdgcn_mov_dpp(m[i], 0xB1, 0xF, 0xF, true);       // quad_perm [1,0,3,2]
                m[i] = hi1 ? pr - m[i] : m[i] + pr;
            }
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                const int pr = __builtin_amdgcn_mov_dpp(m[i], 0x4E, 0xF, 0xF, true);       // quad_perm [2,3,0,1]
                m[i] = hi2 ? pr - m[i] : m[i] + pr;
            }
            acc += iabs(m[0]) + iabs(m[1]) + iabs(m[2]) + iabs(m[3]);
        }
        // every tile's |H D H^T| sum is even (pixel.hip), so one >> 1 over the block equals the per-tile >> 1 of pixel.cpp:235
        return team_allsum<TEAM>(acc) >> 1;
    }
    // SATD of the Cb and Cr blocks predicted at quarter-pel luma vector q (= eighth-pel chroma), the chroma part of subpelCompare
    __device__ __forceinline__ int chroma_term(Mv3 q) const
    {
        const bool hi1 = s & 1, hi2 = s & 2;
        int acc = 0;
#pragma unroll
        for (int ps = 0; ps < CPASS; ps++)
        {
            int p[4];
            chroma_quad_hv(cref[ps], strideC, q.x, q.y, depth, p);
            const int d[4] = { fuc[ps][0] - p[0], fuc[ps][1] - p[1], fuc[ps][2] - p[2], fuc[ps][3] - p[3] };
            const int a = quad_row_hadamard_abs(d, hi1, hi2);
            acc += cact ? a : 0;
        }
        return team_allsum<TEAM>(acc) >> 1;                 // every 4x4 tile sum is even: one shift equals the per-tile >> 1 of satd_4x4
    }
    // MotionEstimate::subpelCompare: luma sad / satd + (bChromaSATD) the chroma SATD term
    __device__ __forceinline__ int sub_q(Mv3 q, int satd) const
    {
        const int v = satd ? satd_q(q) : sad_q(q);
        return CHROMA ? v + chroma_term(q) : v;
    }

    // mestar.h contract: K full-pel points measured together (all their loads in flight before the first reduction)
    template <int K>
    __device__ __forceinline__ void fullpel_costs(const int (&mx)[K], const int (&my)[K], int (&out)[K]) const
    {
        unsigned acc[K];
        int mvc[K];
#pragma unroll
        for (int k = 0; k < K; k++)
        {
            const uint32_t r = org + (uint32_t)(__mul24(my[k], stride) + mx[k]) * (uint32_t)sizeof(P);
            acc[k] = 0;
#pragma unroll
            for (int j = 0; j < IPT; j++)
                acc[k] = Pk3<P>::sad(ld_off<Q>(r + (uint32_t)qoff[j]), fq[j], acc[k]);
            mvc[k] = mvcost(mx[k] * 4, my[k] * 4);
        }
#pragma unroll
        for (int k = 0; k < K; k++)
            out[k] = team_allsum<TEAM>((int)acc[k]) + mvc[k];
    }
    __device__ __forceinline__ int fullpel_cost(int mx, int my, int shift) const
    {
        return sad_q(Mv3{ mx * 4, my * 4 }) + mvcost(mx << shift, my << shift);
    }
    __device__ __forceinline__ int cmp_q(Mv3 q, int satd) const { return satd ? satd_q(q) : sad_q(q); }
};

__device__ __forceinline__ Mv3 mv_clip3(Mv3 v, Mv3 lo, Mv3 hi)
{
    Mv3 r = { v.x > hi.x ? hi.x : v.x, v.y > hi.y ? hi.y : v.y };
    r.x = r.x < lo.x ? lo.x : r.x;
    r.y = r.y < lo.y ? lo.y : r.y;
    return r;
}
__device__ __forceinline__ bool mv_in3(Mv3 v, Mv3 lo, Mv3 hi) { return v.x >= lo.x && v.x <= hi.x && v.y >= lo.y && v.y <= hi.y; }
__device__ __forceinline__ int sext2c(int v) { return (v & 2) ? (v | ~3) : v; }

template <typename P, int N, int TEAM, bool CHROMA>
__global__ __launch_bounds__(256) void motion3_kernel(const P* __restrict__ fencPlane, int64_t strideF, int64_t strideR,
                                                      const int32_t* __restrict__ pu_xy, const int32_t* __restrict__ mvminA,
                                                      const int32_t* __restrict__ mvmaxA, const int32_t* __restrict__ qmvpA,
                                                      int numCand, const int32_t* __restrict__ mvcA, int merange, int method, int subme,
                                                      const uint16_t* __restrict__ mvcostTab, int n,
                                                      const P* __restrict__ planes, int64_t planeElems, DeriveRange dr,
                                                      int32_t* __restrict__ outMv, int32_t* __restrict__ outCost, ChromaPlanes cp, int depth)
{
    typedef RowTeam<P, N, TEAM, CHROMA> RT;
    typedef typename RT::Q Q;
    constexpr int TPB = 256 / TEAM;                         // PUs per workgroup
    RT c;
    c.s = threadIdx.x & (TEAM - 1);
    c.stride = (int)strideR;
    c.planeElems = planeElems;
    c.smallPlane = planeElems < (1 << 24);
    c.cost = mvcostTab;
    c.pk16 = depth <= 10;
    c.costBase = reinterpret_cast<const char*>(mvcostTab) - (size_t)RT::kCostBias * 2;
    c.base = reinterpret_cast<const char*>(planes) - (size_t)RT::kBias * sizeof(P);
    // XCD-aware block order (see motion2.hip): XCD x works on the x-th contiguous eighth of the raster-ordered PU list
    const int chunk = gridDim.x >> 3;
    const int lblock = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    const int pu = lblock * TPB + threadIdx.x / TEAM;
    if (pu >= n)
        return;                                             // whole rows leave together

    const int bx = pu_xy[2 * pu], by = pu_xy[2 * pu + 1];
    Mv3 mvmin, mvmax, qmvp;
    if (dr.enable)
    {
        qmvp = Mv3{ 0, 0 };
        if (dr.mvSrc && dr.srcIdx[pu] >= 0)
            qmvp = Mv3{ dr.mvSrc[2 * dr.srcIdx[pu]], dr.mvSrc[2 * dr.srcIdx[pu] + 1] };
        const SearchRange sr = search_range(dr.picW, dr.picH, dr.maxCUSize, merange, dr.refLagPixels, bx, by, qmvp.x, qmvp.y);
        mvmin = Mv3{ sr.minx, sr.miny };
        mvmax = Mv3{ sr.maxx, sr.maxy };
        if (c.s == 0)
        {
            dr.qmvpO[2 * pu] = qmvp.x; dr.qmvpO[2 * pu + 1] = qmvp.y;
            dr.mvminO[2 * pu] = mvmin.x; dr.mvminO[2 * pu + 1] = mvmin.y;
            dr.mvmaxO[2 * pu] = mvmax.x; dr.mvmaxO[2 * pu + 1] = mvmax.y;
        }
    }
    else
    {
        mvmin = Mv3{ mvminA[2 * pu], mvminA[2 * pu + 1] };
        mvmax = Mv3{ mvmaxA[2 * pu], mvmaxA[2 * pu + 1] };
        qmvp = Mv3{ qmvpA[2 * pu], qmvpA[2 * pu + 1] };
    }
    const Mv3 qmvmin = { mvmin.x * 4, mvmin.y * 4 }, qmvmax = { mvmax.x * 4, mvmax.y * 4 };
    c.qmvp = qmvp;
    c.org = (RT::kBias + (uint32_t)(by * (int)strideR + bx)) * (uint32_t)sizeof(P);
    {
        const P* f = fencPlane + (int64_t)by * strideF + bx;
#pragma unroll
        for (int j = 0; j < RT::IPT; j++)
        {
            const int t = j * (TEAM / 4) + (c.s >> 2), r = c.s & 3;     // tile-major: 4 consecutive lanes = the 4 rows of tile t
            const int row = (t / RT::TX) * 4 + r, col = (t % RT::TX) * 4;
            c.qoff[j] = (row * (int)strideR + col) * (int)sizeof(P);          // bytes
            c.fq[j] = ld_unaligned<Q>(f + (int64_t)row * strideF + col);
            Pk3<P>::unpack(c.fq[j], c.fu[j]);
            c.fp[j][0] = s2v{ (short)c.fu[j][0], (short)c.fu[j][1] };
            c.fp[j][1] = s2v{ (short)c.fu[j][2], (short)c.fu[j][3] };
        }
    }
    if (CHROMA)
    {
        c.strideC = (int)cp.strideRC;
        c.depth = depth;
        constexpr int TXC = N / 8;                              // 4x4 tiles per row of the chroma block
        const int t = c.s >> 2, r = c.s & 3;
        const int cbx = bx >> 1, cby = by >> 1;
#pragma unroll
        for (int ps = 0; ps < RT::CPASS; ps++)
        {
            // 8x8 PU: tile 0 = Cb, tile 1 = Cr, tiles 2-3 idle; larger PUs: pass 0 = Cb, pass 1 = Cr, tile t of the plane
            const int plane = N == 8 ? (t & 1) : ps;
            const int tt = N == 8 ? 0 : t;
            const int row = (tt / TXC) * 4 + r, col = (tt % TXC) * 4;
            const P* fc = (const P*)(plane ? cp.fencCr : cp.fencCb) + (int64_t)(cby + row) * cp.strideFC + cbx + col;
            load4(fc, c.fuc[ps]);
            c.cref[ps] = (const P*)(plane ? cp.refCr : cp.refCb) + (int64_t)(cby + row) * cp.strideRC + cbx + col;
        }
        c.cact = N == 8 ? (t < 2) : true;
    }

#define YOK(yy) (((yy) >= mvmin.y) & ((yy) <= mvmax.y))
#define LT1(v) do { const int v_ = (v); if (v_ < bcost) bcost = v_; } while (0)
#define FULLPEL(mx, my) (c.sad_q(Mv3{ (mx) * 4, (my) * 4 }) + c.mvcost((mx) * 4, (my) * 4))
    // ---- predictor, zero and candidates (motion.cpp:761-812)
    const Mv3 pmv = mv_clip3(qmvp, qmvmin, qmvmax);
    Mv3 bestpre = pmv;
    int bprecost = c.sub_q(pmv, 0);
    Mv3 bmv = { (pmv.x + 2) >> 2, (pmv.y + 2) >> 2 };
    int bcost = bprecost;
    if ((pmv.x & 3) | (pmv.y & 3))
        bcost = FULLPEL(bmv.x, bmv.y);
    if (pmv.x | pmv.y)
    {
        const int cst = FULLPEL(0, 0);
        if (cst < bcost)
        {
            bcost = cst;
            bmv.x = 0;
            bmv.y = max(min(0, mvmax.y), mvmin.y);
        }
    }
    for (int i = 0; i < numCand; i++)
    {
        const Mv3 raw = { mvcA[((int64_t)pu * numCand + i) * 2], mvcA[((int64_t)pu * numCand + i) * 2 + 1] };
        const Mv3 m = mv_clip3(raw, qmvmin, qmvmax);
        if ((m.x | m.y) && !(m.x == pmv.x && m.y == pmv.y) && !(m.x == bestpre.x && m.y == bestpre.y))
        {
            const int cst = c.sub_q(m, 0) + c.mvcost(m.x, m.y);
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
            const int c0 = FULLPEL(bmv.x, bmv.y - 1), c1 = FULLPEL(bmv.x, bmv.y + 1), c2 = FULLPEL(bmv.x - 1, bmv.y), c3 = FULLPEL(bmv.x + 1, bmv.y);
            if (YOK(bmv.y - 1)) LT1((c0 << 4) + 1);
            if (YOK(bmv.y + 1)) LT1((c1 << 4) + 3);
            LT1((c2 << 4) + 4);
            LT1((c3 << 4) + 12);
            if (!(bcost & 15))
                break;
            bmv.x -= sext2c((bcost >> 2) & 3);
            bmv.y -= sext2c(bcost & 3);
            bcost &= ~15;
        }
        while (--i && mv_in3(bmv, mvmin, mvmax));
        bcost >>= 4;
    }
    else if (meth == 1)
    {
        // X265_HEX_SEARCH, motion.cpp:855-944
        {
            const int c0 = FULLPEL(bmv.x - 2, bmv.y), c1 = FULLPEL(bmv.x - 1, bmv.y + 2), c2 = FULLPEL(bmv.x + 1, bmv.y + 2);
            const int c3 = FULLPEL(bmv.x + 2, bmv.y), c4 = FULLPEL(bmv.x + 1, bmv.y - 2), c5 = FULLPEL(bmv.x - 1, bmv.y - 2);
            bcost <<= 3;
            if (YOK(bmv.y)) LT1((c0 << 3) + 2);
            if (YOK(bmv.y + 2))
            {
                LT1((c1 << 3) + 3);
                LT1((c2 << 3) + 4);
            }
            if (YOK(bmv.y)) LT1((c3 << 3) + 5);
            if (YOK(bmv.y - 2))
            {
                LT1((c4 << 3) + 6);
                LT1((c5 << 3) + 7);
            }
        }
        if (bcost & 7)
        {
            int dir = (bcost & 7) - 2;
            if (YOK(bmv.y + hex2yC(dir + 1)))
            {
                bmv.x += hex2xC(dir + 1);
                bmv.y += hex2yC(dir + 1);
                for (int i = (hexRange >> 1) - 1; i > 0 && mv_in3(bmv, mvmin, mvmax); i--)
                {
                    const Mv3 a = { bmv.x + hex2xC(dir + 0), bmv.y + hex2yC(dir + 0) };
                    const Mv3 b = { bmv.x + hex2xC(dir + 1), bmv.y + hex2yC(dir + 1) };
                    const Mv3 d = { bmv.x + hex2xC(dir + 2), bmv.y + hex2yC(dir + 2) };
                    const int c0 = FULLPEL(a.x, a.y), c1 = FULLPEL(b.x, b.y), c2 = FULLPEL(d.x, d.y);
                    bcost &= ~7;
                    if (YOK(a.y)) LT1((c0 << 3) + 1);
                    if (YOK(b.y)) LT1((c1 << 3) + 2);
                    if (YOK(d.y)) LT1((c2 << 3) + 3);
                    if (!(bcost & 7))
                        break;
                    dir += (bcost & 7) - 2;
                    dir = mod6m1C(dir + 1);
                    bmv.x += hex2xC(dir + 1);
                    bmv.y += hex2yC(dir + 1);
                }
            }
        }
        bcost >>= 3;
        // square refine, motion.cpp:918-942
        int dir = 0;
        {
            const int c0 = FULLPEL(bmv.x, bmv.y - 1), c1 = FULLPEL(bmv.x, bmv.y + 1), c2 = FULLPEL(bmv.x - 1, bmv.y), c3 = FULLPEL(bmv.x + 1, bmv.y);
            const int c4 = FULLPEL(bmv.x - 1, bmv.y - 1), c5 = FULLPEL(bmv.x - 1, bmv.y + 1), c6 = FULLPEL(bmv.x + 1, bmv.y - 1), c7 = FULLPEL(bmv.x + 1, bmv.y + 1);
            if (YOK(bmv.y - 1) && c0 < bcost) { bcost = c0; dir = 1; }
            if (YOK(bmv.y + 1) && c1 < bcost) { bcost = c1; dir = 2; }
            if (c2 < bcost) { bcost = c2; dir = 3; }
            if (c3 < bcost) { bcost = c3; dir = 4; }
            if (YOK(bmv.y - 1) && c4 < bcost) { bcost = c4; dir = 5; }
            if (YOK(bmv.y + 1) && c5 < bcost) { bcost = c5; dir = 6; }
            if (YOK(bmv.y - 1) && c6 < bcost) { bcost = c6; dir = 7; }
            if (YOK(bmv.y + 1) && c7 < bcost) { bcost = c7; dir = 8; }
        }
        bmv.x += sq1xC(dir);
        bmv.y += sq1yC(dir);
    }
    else if (meth == 3)
        star_search(c, mvmin.x, mvmin.y, mvmax.x, mvmax.y, merange, bmv.x, bmv.y, bcost);      // X265_STAR_SEARCH (mestar.h)
    else if (meth == 5)
    {
        // X265_FULL_SEARCH, motion.cpp:1397-1441: raster order, strict '<' keeps the first minimum
        for (int ty = mvmin.y; ty <= mvmax.y; ty++)
            for (int tx = mvmin.x; tx <= mvmax.x; tx++)
            {
                const int cst = FULLPEL(tx, ty);
                if (cst < bcost)
                {
                    bcost = cst;
                    bmv.x = tx;
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
        bcost = c.mvcost(bmv.x, bmv.y);                // motion.cpp:1466-1471
    else
    {
        // motion.cpp:1504-1561
        const int hpelIters = kWorkloadC[subme][0], hpelDirs = kWorkloadC[subme][1];
        const int qpelIters = kWorkloadC[subme][2], qpelDirs = kWorkloadC[subme][3], hpelSatd = kWorkloadC[subme][4];
        int hpelcomp = 0;
        if (hpelSatd)
        {
            bcost = c.sub_q(bmv, 1) + c.mvcost(bmv.x, bmv.y);
            hpelcomp = 1;
        }
        for (int iter = 0; iter < hpelIters; iter++)
        {
            int bdir = 0;
            for (int i = 1; i <= hpelDirs; i++)
            {
                const Mv3 q = { bmv.x + sq1xC(i) * 2, bmv.y + sq1yC(i) * 2 };
                if ((q.y < qmvmin.y) | (q.y > qmvmax.y))
                    continue;
                const int cst = c.sub_q(q, hpelcomp) + c.mvcost(q.x, q.y);
                if (cst < bcost) { bcost = cst; bdir = i; }
            }
            if (bdir)
            {
                bmv.x += sq1xC(bdir) * 2;
                bmv.y += sq1yC(bdir) * 2;
            }
            else
                break;
        }
        if (!hpelSatd)
            bcost = c.sub_q(bmv, 1) + c.mvcost(bmv.x, bmv.y);
        for (int iter = 0; iter < qpelIters; iter++)
        {
            int bdir = 0;
            for (int i = 1; i <= qpelDirs; i++)
            {
                const Mv3 q = { bmv.x + sq1xC(i), bmv.y + sq1yC(i) };
                if ((q.y < qmvmin.y) | (q.y > qmvmax.y))
                    continue;
                const int cst = c.sub_q(q, 1) + c.mvcost(q.x, q.y);
                if (cst < bcost) { bcost = cst; bdir = i; }
            }
            if (bdir)
            {
                bmv.x += sq1xC(bdir);
                bmv.y += sq1yC(bdir);
            }
            else
                break;
        }
    }
#undef YOK
#undef LT1
#undef FULLPEL
    if (c.s == 0)
    {
        outMv[2 * pu] = bmv.x;
        outMv[2 * pu + 1] = bmv.y;
        outCost[pu] = bcost;
    }
    if (dr.predOut)
    {
        // Predict::predInterLumaPixel (predict.cpp:245-266) at the winner: copy / hpp / vpp / hvpp of the block == the block of phase plane
        // (y & 3) * 4 + (x & 3) at the full-pel part of the vector; each lane moves the quads it has been comparing all along
        const uint32_t r = c.cand(bmv);
        P* pd = (P*)dr.predOut + (int64_t)by * dr.predStride + bx;
#pragma unroll
        for (int j = 0; j < RT::IPT; j++)
        {
            const int t = j * (TEAM / 4) + (c.s >> 2), rr = c.s & 3;
            const int row = (t / RT::TX) * 4 + rr, col = (t % RT::TX) * 4;
            st_unaligned<Q>(pd + (int64_t)row * dr.predStride + col, c.template ld_off<Q>(r + (uint32_t)c.qoff[j]));
        }
    }
}

// returns 1 when handled (square PUs with planes), 0 otherwise
int motion3_dispatch(int depth, int size, const void* fencPlane, int64_t strideF, int64_t strideR, const int32_t* pu_xy,
                     const int32_t* mvmin, const int32_t* mvmax, const int32_t* qmvp, int numCand, const int32_t* mvc, int merange,
                     int method, int subme, const uint16_t* mvcost, int n, const void* planes, int64_t planeElems, const DeriveRange* drp,
                     int32_t* outMv, int32_t* outCost, hipStream_t st, int* rc, const ChromaPlanes* cpp)
{
    // 64x64 stays on the 4-wave team kernel of motion2.hip: one wave per 64x64 PU measured slower (58 vs 38 us per level)
    if (!planes || (size != 8 && size != 16 && size != 32) || strideR >= (1 << 23))
        return 0;
    DeriveRange dr{};
    if (drp) dr = *drp;
    // 8x8: eight PUs per wave (8-lane teams, a lane = two quads: all per-candidate control and address work is shared by twice as many PUs as
    // with 16-lane teams); 16x16: four PUs per wave; 32x32: one wave per PU
    static const bool team16 = getenv("X265HIP_ME8_TEAM16") != nullptr;
    const int tpb = size == 8 ? (team16 ? 16 : 32) : (size == 16 ? 16 : 4);     // (16x16 on 8-lane teams measured slower: 35 vs 28 us)
    const int blocks = (((n + tpb - 1) / tpb) + 7) & ~7;
    dim3 grid(blocks), block(256);
    ChromaPlanes cpn{};
    const bool chroma = cpp && cpp->enable;
    if (chroma) cpn = *cpp;
#define M3C(P, N, TEAM, CH) hipLaunchKernelGGL((motion3_kernel<P, N, TEAM, CH>), grid, block, 0, st, (const P*)fencPlane, strideF, strideR, pu_xy, mvmin, mvmax, qmvp, \
                                    numCand, mvc, merange, method, subme, mvcost, n, (const P*)planes, planeElems, dr, outMv, outCost, cpn, depth)
#define M3(P, N, TEAM) do { if (chroma) M3C(P, N, TEAM, true); else M3C(P, N, TEAM, false); } while (0)
    if (depth == 8)
    {
        if (size == 8) { if (team16) M3(uint8_t, 8, 16); else M3(uint8_t, 8, 8); }
        else if (size == 16) M3(uint8_t, 16, 16);
        else M3(uint8_t, 32, 64);
    }
    else
    {
        if (size == 8) { if (team16) M3(uint16_t, 8, 16); else M3(uint16_t, 8, 8); }
        else if (size == 16) M3(uint16_t, 16, 16);
        else M3(uint16_t, 32, 64);
    }
#undef M3C
#undef M3
    hipError_t e = hipGetLastError();
    *rc = e == hipSuccess ? X265HIP_OK : check_hip(e, "motion3_kernel");
    if (dr.predOut && dr.predDone && e == hipSuccess)
        *dr.predDone = 1;
    return 1;
}

} // namespace xh
