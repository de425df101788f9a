// sadsurf.hip — SAD surfaces behind x265hip_sadsurf_* (include/x265hip.h): the integer-pel candidates of the encoder's motion search as table loads.
//
// MotionEstimate::motionEstimate (reference source/encoder/motion.cpp:739-1569) measures an integer-pel candidate as
// sad(fenc, FENC_STRIDE, fref + mx + my * stride, stride) (:246-330) — a function of the source picture, the finished reference picture, the
// block's position and the vector, and of nothing the encoder decides.  One surface = one (source picture, reference picture) pair: for every
// aligned 16 / 32 / 64 block a 16 x 16 window of vectors, built here as the reference picture's rows arrive (the refpic worker calls
// sadsurf_rows_arrived after each band) and mirrored into page-locked host memory, where x265_amd/host/x265_hip_sadplanes.cpp reads them.
//
// The kernel is the search-window kernel of the design: one workgroup per 64 x 64 region (CTU), 16 waves.
//   stage    the source CTU (4 KB) and the reference window (64 + 2S)^2 around it (16 KB at S = 32) go to LDS once;
//   measure  wave b owns 16x16 block b of the CTU: its 256 source pixels sit in registers, and every vector of [-S, S)^2 is measured four at a
//            time with v_qsad_pk_u16_u8 (four SADs of 4 source pixels against a sliding 8-byte window of the reference row: 16 absolute
//            differences per lane and instruction, no alignment work), 64 instructions per four candidates; the 16 x (2S)^2 results
//            (u16, 128 KB at S = 32) stay in LDS — the exhaustive search's unique footprint per CTU is 4 KB + 16 KB in, nothing out;
//   decide   top down 64 -> 32 -> 16: a block's SAD at a vector is the sum of its 16x16 parts; cost = SAD + vector cost around the parent's
//            best vector; wave / workgroup minimum (DPP + LDS);
//   emit     the 16 x 16 window around each block's best vector is gathered out of the LDS surface (no second SAD pass) and written with the
//            block's origin into the CTU row's chunk of the table buffer.
// Values are pinned by tests/test_sadsurf.py (entries == sad<N, N> of the reference, pixel.cpp:40-55; origins == the same rule restated on the
// CPU by the test tier).  16-bit pictures (Main10 / Main12 builds) have a kernel of their own below: v_sad_u16, u32 surfaces, range <= 16.
#include "common.h"
#include "internal.h"
#include "refpic.h"
#include "tiles.h"
#include <cstring>
#include <type_traits>
#include <map>
#include <utility>

struct x265hip_srcpic
{
    int depth = 8, w = 0, h = 0, device = 0;
    int place = -1;                     // x265hip_srcpic_create_at; -(device + 1) for x265hip_srcpic_create
    int64_t pitch = 0;                  // bytes between rows of dLuma
    char* dLuma = nullptr;
    char* hStage = nullptr;             // page-locked staging copy (the encoder's buffer is ordinary memory)
    // (no stream: x265hip_srcpic_upload leases one from the pool, runtime.hip stream_lease)
    std::atomic<int> refs{ 1 };         // the creator's + one per SAD surface attached: a surface reads dLuma and files its buffers under `device` until it is freed,
                                        // so x265hip_srcpic_destroy only drops the creator's reference (a resized source buffer used to free under them)
};
static void srcpic_unref(x265hip_srcpic* sp);

namespace xh {

constexpr int kWin = X265HIP_SADSURF_WIN;

struct SurfLayout
{
    int blocksX[4], blocksY[4], per[4], entryBytes[4];
    int64_t originOff[4], tableOff[4];   // byte offsets inside a CTU row's chunk
    int64_t subpelOff[4];                 // sub-pel SATD tables (levels 1..3 when bit 4 of `levels` is set; 0 = not built)
    int64_t pitch;                        // bytes per CTU row
    int ctuCols, ctuRows;
};

// one surface's share of a launch: rows [row0, row0 + rows) of its table
struct SurfJob
{
    const uint8_t* ref; int64_t refStride;       // the job's reference pixel (0, 0): the surfaces of one launch may follow different reference pictures
    const uint8_t* src; int64_t srcPitch;
    char* out;                                    // device table buffer (chunk of CTU row 0 first)
    int S, lambda20, row0, rows;
    int level0;                                   // the 8x8 windows are wanted
};
constexpr int kMaxJobs = 16;

// one launch: rows of up to kMaxJobs surfaces that share the geometry and the table layout (same encoder, same levels) — of one reference picture or several
struct SurfArgs
{
    int picW, picH, marginX, marginY;
    int bufRows;                                  // rows of the padded picture in device memory (marginY above picture row 0)
    int64_t pitch;
    int64_t originOff[4], tableOff[4];
    int blocksX[4];
    int blocksY0;                                 // 8x8 blocks inside the picture, vertically
    int nJobs;
    int xcd;                                      // 1: XCD-aware CTU order (surf_ctu below)
    int64_t subpelOff3;                           // >= 0: the CTU's 49 sub-pel SATD entries of the 64x64 level are zeroed (subpel_satd_kernel_lds adds four quadrants into them)
    SurfJob job[kMaxJobs];
};

// Which CTU of the launch a workgroup takes.  Workgroups are dealt to the eight XCDs round-robin in dispatch order (x fastest), and the windows of neighbouring
// CTUs overlap by half either way: in the plain order a row's neighbours sit in eight different L2s and each fetches the shared half again.  XCD x instead walks a
// contiguous run of CTUs: of the `total` workgroups it receives n_x = total / 8 + (x < total % 8) — the linear ids x, x + 8, ... — and its i-th takes CTU
// start_x + i, start_x = x * (total / 8) + min(x, total % 8): a bijection for any grid, every workgroup the same work.
__device__ __forceinline__ void surf_ctu(int xcd, int& cx, int& rowIn)
{
    cx = blockIdx.x; rowIn = blockIdx.y;
    if (!xcd) return;
    const int total = gridDim.x * gridDim.y, L = blockIdx.y * gridDim.x + blockIdx.x;
    const int x = L & 7, i = L >> 3, per = total >> 3, rem = total & 7;
    const int c = x * per + (x < rem ? x : rem) + i;
    cx = c % (int)gridDim.x; rowIn = c / (int)gridDim.x;
}

} // namespace xh

struct x265hip_sadsurf
{
    x265hip_srcpic* src = nullptr;
    x265hip_refpic* ref = nullptr;       // null once the reference picture has gone (under g_ssLock)
    int workerPlace = -1;                // the worker that drives this surface (its reference picture's place)
    xh::Replica* rep = nullptr;          // the reference picture's replica at the source's place; null: the mirror itself (same place).  Worker only
    int S = 32, lambda20 = 0, levels = 14;
    xh::SurfLayout lay;
    char* dBuf = nullptr;                // ctuRows * pitch
    char* hBuf = nullptr;                // page-locked mirror, same layout
    size_t bytes = 0;
    int rowsBuilt = 0;                   // worker only
    std::atomic<int> ctuRowsReady{ 0 };
    x265hip_sadsurf_view view;
    std::atomic<bool> released{ false };
};

namespace xh {

static std::mutex g_ssLock;              // the surfaces lists of the mirrors and x265hip_sadsurf::ref
static std::mutex g_poolLock;
struct PoolEntry { char* d; char* h; };
static std::multimap<std::pair<size_t, int>, PoolEntry> g_pool;      // key: (bytes, device of d)      // table buffers of finished surfaces, by size (pinned allocations are expensive: ~ms)
static std::atomic<uint64_t> g_statAttached{ 0 }, g_statRows{ 0 }, g_statLaunches{ 0 };
static std::atomic<uint64_t> g_statReplicas{ 0 }, g_statPeerBands{ 0 }, g_statPeerBytes{ 0 };

static void layout_for(int w, int h, int depth, int levels, SurfLayout& L)
{
    memset(&L, 0, sizeof(L));
    L.ctuCols = (w + 63) / 64; L.ctuRows = (h + 63) / 64;
    int64_t off = 0;
    for (int l = 0; l < 4; l++)
    {
        if (!(levels >> l & 1))
            continue;                             // level not built: no room in the chunks, blocksX = 0
        const int N = 8 << l;
        L.blocksX[l] = w / N; L.blocksY[l] = h / N; L.per[l] = 64 / N;
        L.entryBytes[l] = (uint64_t)N * N * ((1u << depth) - 1) < 65536 ? 2 : 4;
        L.originOff[l] = off; off += (int64_t)L.per[l] * L.blocksX[l] * 4;
        off = (off + 15) & ~(int64_t)15;
        L.tableOff[l] = off; off += (int64_t)L.per[l] * L.blocksX[l] * kWin * kWin * L.entryBytes[l];
        off = (off + 15) & ~(int64_t)15;
        if (l && (levels & 16))
        {
            L.subpelOff[l] = off; off += (int64_t)L.per[l] * L.blocksX[l] * X265HIP_SADSURF_SUBPEL * 4;
            off = (off + 15) & ~(int64_t)15;
        }
    }
    L.pitch = (off + 255) & ~(int64_t)255;
}

// ---- the kernel ---------------------------------------------------------------------------------------------------------------------------

__device__ __forceinline__ int ss_log(int d) { return 31 - __clz(4 * d + 1); }                  // floor(log2(4 d + 1)): bits(4 d) = 2 ss_log(d) + 1
__device__ __forceinline__ uint64_t u64_min(uint64_t a, uint64_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
    {
        const uint32_t lo = __shfl_xor((uint32_t)v, off), hi = __shfl_xor((uint32_t)(v >> 32), off);
        v = u64_min(v, ((uint64_t)hi << 32) | lo);
    }
    return v;
}

// The scan of one block's candidates by a team of threads, four neighbouring vectors per step.  A thread owns the column group q (vectors
// vx = 4 q - S .. + 3) and the rows first, first + step, ...: its four horizontal terms are computed once.  `sums(c, s)` = the block's SADs at
// the candidates c .. c + 3.  cost = SAD + sVc[log(|vx - px|) + log(|vy - py|)] (the vector-cost table of this CTU's lambda), illegal vectors cost
// UINT_MAX; the thread's candidates are visited in ascending index, so `<` keeps the smallest index among equals.  Returns cost << 12 | index.
template <typename F>
__device__ __forceinline__ uint64_t ss_scan4(F sums, const uint32_t* sVc, int S, int q, int first, int step, int xlo, int xhi, int ylo, int yhi, int px, int py)
{
    const int D = 2 * S;
    uint32_t best = 0xFFFFFFFFu, bestC = 0;
    if (4 * q < D)
    {
        int lx[4];
        bool okx[4];
#pragma unroll
        for (int e = 0; e < 4; e++)
        {
            const int vx = 4 * q + e - S;
            lx[e] = ss_log(abs(vx - px));
            okx[e] = vx >= xlo && vx <= xhi;
        }
        for (int dyI = first; dyI < D; dyI += step)
        {
            const int vy = dyI - S;
            if (vy < ylo || vy > yhi)
                continue;
            const int ly = ss_log(abs(vy - py));
            const int c = dyI * D + 4 * q;
            uint32_t s[4];
            sums(c, s);
#pragma unroll
            for (int e = 0; e < 4; e++)
            {
                const uint32_t cost = okx[e] ? s[e] + sVc[lx[e] + ly] : 0xFFFFFFFFu;
                if (cost < best) { best = cost; bestC = (uint32_t)(c + e); }
            }
        }
    }
    return best == 0xFFFFFFFFu ? ~(uint64_t)0 : ((uint64_t)best << 12) | bestC;
}

__device__ __forceinline__ void ss_origin(uint64_t key, int S, int xlo, int xhi, int ylo, int yhi, int& bx, int& by, int& ox, int& oy)
{
    const int D = 2 * S;
    int c = (int)(key & 4095);
    if (key == ~(uint64_t)0) c = S * D + S;               // no legal candidate (cannot happen with margins >= S): vector (0, 0)
    by = c / D - S; bx = c - (c / D) * D - S;
    ox = bx - kWin / 2; oy = by - kWin / 2;
    if (ox > S - kWin) ox = S - kWin;
    if (ox < -S) ox = -S;
    if (oy > S - kWin) oy = S - kWin;
    if (oy < -S) oy = -S;
    if (ox > xhi - (kWin - 1)) ox = xhi - (kWin - 1);
    if (ox < xlo) ox = xlo;
    if (oy > yhi - (kWin - 1)) oy = yhi - (kWin - 1);
    if (oy < ylo) oy = ylo;
}

// four u16 SADs of one 16x16 block at the candidates c .. c + 3 (c a multiple of 4: one 8-byte LDS read), added to s[]
__device__ __forceinline__ void ss_add4(const uint16_t* surf, int c, uint32_t* s)
{
    const uint2 v = *(const uint2*)(surf + c);
    s[0] += v.x & 0xFFFFu; s[1] += v.x >> 16; s[2] += v.y & 0xFFFFu; s[3] += v.y >> 16;
}

// the same for the u32 surfaces of 16-bit pictures (one 16-byte LDS read)
__device__ __forceinline__ void ss_add4(const uint32_t* surf, int c, uint32_t* s)
{
    const uint4 v = *(const uint4*)(surf + c);
    s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
}

// what the phases of a workgroup share besides the surface
struct SsShared
{
    uint64_t red[16];
    int best[4][16][2];                                          // [level][block][vx, vy]
    int org[4][16][2];
    uint32_t vc[32];                                             // vector cost by log(|dx|) + log(|dy|)
};

// decide (top down 64 -> 32 -> 16) and emit the windows of levels 1..3 out of the LDS surface `sSurf` ([16][D][D], ST = u16 for 8-bit pictures, u32 for
// 16-bit ones); called by all 1024 threads of the workgroup after the surface is complete.  Leaves the level-1 origins in sh.org[1] (the 8x8 pass of
// the 8-bit kernel takes them from there).
template <typename ST>
__device__ __forceinline__ void ss_decide_emit(const SurfArgs& a, const SurfJob& jb, const ST* sSurf, SsShared& sh, int S, int x0, int y0, int cx, int cy)
{
    const int D = 2 * S, DD = D * D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint64_t* sRed = sh.red;
    int (*sBest)[16][2] = sh.best;
    int (*sOrg)[16][2] = sh.org;
    const uint32_t* sVc = sh.vc;
    // ---- decide, top down ----
    // level 3: the whole CTU, all 1024 threads: column group tid & 15, rows tid >> 4, + 64, ...
    const bool have3 = x0 + 64 <= a.picW && y0 + 64 <= a.picH;
    if (have3)
    {
        const int xlo = -a.marginX - x0, xhi = a.picW + a.marginX - 64 - x0, ylo = -a.marginY - y0, yhi = a.picH + a.marginY - 64 - y0;
        uint64_t k = ss_scan4([&](int c, uint32_t* s) { s[0] = s[1] = s[2] = s[3] = 0;
#pragma unroll
                                                       for (int b = 0; b < 16; b++) ss_add4(sSurf + b * DD, c, s); },
                              sVc, S, tid & 15, tid >> 4, 64, xlo, xhi, ylo, yhi, 0, 0);
        k = wave_min_u64(k);
        if (lane == 0) sRed[wave] = k;
        __syncthreads();
        if (tid == 0)
        {
            uint64_t m = sRed[0];
            for (int i = 1; i < 16; i++) m = u64_min(m, sRed[i]);
            int bx, by, ox, oy;
            ss_origin(m, S, xlo, xhi, ylo, yhi, bx, by, ox, oy);
            sBest[3][0][0] = bx; sBest[3][0][1] = by; sOrg[3][0][0] = ox; sOrg[3][0][1] = oy;
        }
    }
    __syncthreads();
    // level 2: 32x32 block g <-> threads [256 g, 256 g + 256)
    {
        const int g = tid >> 8, t = tid & 255, gx = g & 1, gy = g >> 1;
        const int x = x0 + gx * 32, y = y0 + gy * 32;
        const bool have = x + 32 <= a.picW && y + 32 <= a.picH;
        const int xlo = -a.marginX - x, xhi = a.picW + a.marginX - 32 - x, ylo = -a.marginY - y, yhi = a.picH + a.marginY - 32 - y;
        uint64_t k = ~(uint64_t)0;
        if (have)
        {
            const int px = have3 ? sBest[3][0][0] : 0, py = have3 ? sBest[3][0][1] : 0;
            const ST* s0 = sSurf + ((gy * 2) * 4 + gx * 2) * DD;
            k = ss_scan4([&](int c, uint32_t* s) { s[0] = s[1] = s[2] = s[3] = 0;
                                                   ss_add4(s0, c, s); ss_add4(s0 + DD, c, s); ss_add4(s0 + 4 * DD, c, s); ss_add4(s0 + 5 * DD, c, s); },
                         sVc, S, t & 15, t >> 4, 16, xlo, xhi, ylo, yhi, px, py);
            k = wave_min_u64(k);
        }
        if (lane == 0) sRed[wave] = k;
        __syncthreads();
        if (have && t == 0)
        {
            uint64_t m = u64_min(u64_min(sRed[4 * g], sRed[4 * g + 1]), u64_min(sRed[4 * g + 2], sRed[4 * g + 3]));
            int bx, by, ox, oy;
            ss_origin(m, S, xlo, xhi, ylo, yhi, bx, by, ox, oy);
            sBest[2][g][0] = bx; sBest[2][g][1] = by; sOrg[2][g][0] = ox; sOrg[2][g][1] = oy;
        }
    }
    __syncthreads();
    // level 1: 16x16 block b <-> wave b
    {
        const int b = wave, bx16 = b & 3, by16 = b >> 2;
        const int x = x0 + bx16 * 16, y = y0 + by16 * 16;
        const bool have = x + 16 <= a.picW && y + 16 <= a.picH;
        if (have)
        {
            const int g = (by16 >> 1) * 2 + (bx16 >> 1);
            const bool haveParent = x0 + (bx16 >> 1) * 32 + 32 <= a.picW && y0 + (by16 >> 1) * 32 + 32 <= a.picH;
            const int px = haveParent ? sBest[2][g][0] : 0, py = haveParent ? sBest[2][g][1] : 0;
            const int xlo = -a.marginX - x, xhi = a.picW + a.marginX - 16 - x, ylo = -a.marginY - y, yhi = a.picH + a.marginY - 16 - y;
            const ST* s0 = sSurf + b * DD;
            uint64_t k = ss_scan4([&](int c, uint32_t* s) { s[0] = s[1] = s[2] = s[3] = 0; ss_add4(s0, c, s); },
                                  sVc, S, lane & 15, lane >> 4, 4, xlo, xhi, ylo, yhi, px, py);
            k = wave_min_u64(k);
            if (lane == 0)
            {
                int vx, vy, ox, oy;
                ss_origin(k, S, xlo, xhi, ylo, yhi, vx, vy, ox, oy);
                sOrg[1][b][0] = ox; sOrg[1][b][1] = oy;
            }
        }
    }
    __syncthreads();

    // ---- emit: origins and the 16 x 16 windows, gathered from the LDS surface ----
    char* chunk = jb.out + (int64_t)cy * a.pitch;
    // level 1: wave b, 4 entries per lane
    {
        const int b = wave, bx16 = b & 3, by16 = b >> 2;
        if (x0 + bx16 * 16 + 16 <= a.picW && y0 + by16 * 16 + 16 <= a.picH)
        {
            const int ox = sOrg[1][b][0], oy = sOrg[1][b][1];
            const int64_t idx = (int64_t)by16 * a.blocksX[1] + cx * 4 + bx16;
            if (lane == 0)
                *(uint32_t*)(chunk + a.originOff[1] + idx * 4) = (uint32_t)(uint16_t)(int16_t)ox | ((uint32_t)(uint16_t)(int16_t)oy << 16);
            const int j = lane >> 2, i4 = (lane & 3) * 4;
            const ST* p = sSurf + ((size_t)b * D + (oy + S + j)) * D + (ox + S + i4);
            const ST v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
            if (sizeof(ST) == 2)
            {
                uint16_t* tab = (uint16_t*)(chunk + a.tableOff[1]) + idx * (kWin * kWin);
                *(uint2*)(tab + j * kWin + i4) = make_uint2((uint32_t)v0 | ((uint32_t)v1 << 16), (uint32_t)v2 | ((uint32_t)v3 << 16));
            }
            else
            {
                uint32_t* tab = (uint32_t*)(chunk + a.tableOff[1]) + idx * (kWin * kWin);
                *(uint4*)(tab + j * kWin + i4) = make_uint4((uint32_t)v0, (uint32_t)v1, (uint32_t)v2, (uint32_t)v3);
            }
        }
    }
    // level 2: group g, one entry per thread
    {
        const int g = tid >> 8, t = tid & 255, gx = g & 1, gy = g >> 1;
        if (x0 + gx * 32 + 32 <= a.picW && y0 + gy * 32 + 32 <= a.picH)
        {
            const int ox = sOrg[2][g][0], oy = sOrg[2][g][1];
            const int64_t idx = (int64_t)gy * a.blocksX[2] + cx * 2 + gx;
            if (t == 0)
                *(uint32_t*)(chunk + a.originOff[2] + idx * 4) = (uint32_t)(uint16_t)(int16_t)ox | ((uint32_t)(uint16_t)(int16_t)oy << 16);
            const int j = t >> 4, i = t & 15, b0 = (gy * 2) * 4 + gx * 2;
            const size_t c = (size_t)(oy + S + j) * D + (ox + S + i);
            ((uint32_t*)(chunk + a.tableOff[2]))[idx * (kWin * kWin) + t] =
                (uint32_t)sSurf[b0 * DD + c] + sSurf[(b0 + 1) * DD + c] + sSurf[(b0 + 4) * DD + c] + sSurf[(b0 + 5) * DD + c];
        }
    }
    // level 3: threads 0..255
    if (have3 && tid < 256)
    {
        const int ox = sOrg[3][0][0], oy = sOrg[3][0][1];
        const int64_t idx = cx;
        if (tid == 0)
            *(uint32_t*)(chunk + a.originOff[3] + idx * 4) = (uint32_t)(uint16_t)(int16_t)ox | ((uint32_t)(uint16_t)(int16_t)oy << 16);
        const int j = tid >> 4, i = tid & 15;
        const size_t c = (size_t)(oy + S + j) * D + (ox + S + i);
        uint32_t t = 0;
#pragma unroll
        for (int b = 0; b < 16; b++) t += sSurf[b * DD + c];
        ((uint32_t*)(chunk + a.tableOff[3]))[idx * (kWin * kWin) + tid] = t;
    }

}

__global__ __launch_bounds__(1024) void sadsurf_ctu_kernel(SurfArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ SsShared sh;

    int jn = 0, rowIn, cx;
    surf_ctu(a.xcd, cx, rowIn);
    while (jn + 1 < a.nJobs && rowIn >= a.job[jn].rows) { rowIn -= a.job[jn].rows; jn++; }
    const SurfJob& jb = a.job[jn];
    const int S = jb.S, D = 2 * S, DD = D * D;
    const int RW = 64 + D + 8;                                   // reference window row pitch (bytes): room for the last 24-byte read, multiple of 8
    uint8_t* sSrc = smem;                                        // [64][64]
    uint8_t* sRef = smem + 4096;                                 // [64 + D][RW]
    uint16_t* sSurf = (uint16_t*)(smem + 4096 + (size_t)(64 + D) * RW);      // [16][D][D]

    const int cy = jb.row0 + rowIn;
    const int x0 = cx * 64, y0 = cy * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.subpelOff3 >= 0 && cx < a.blocksX[3] && tid < X265HIP_SADSURF_SUBPEL)          // (blocksX: the blocks that lie inside the picture)
        ((uint32_t*)(jb.out + (int64_t)cy * a.pitch + a.subpelOff3))[(int64_t)cx * X265HIP_SADSURF_SUBPEL + tid] = 0;

    // ---- stage ----
    for (int i = tid; i < 64 * 16; i += 1024)
    {
        const int r = i >> 4, c4 = (i & 15) * 4;
        uint32_t v = 0;
        if (y0 + r < a.picH && x0 + c4 + 3 < a.picW)
            v = ld_global_unaligned<uint32_t>(jb.src + (int64_t)(y0 + r) * jb.srcPitch + x0 + c4);
        else if (y0 + r < a.picH)
            for (int k = 0; k < 4; k++)
                if (x0 + c4 + k < a.picW) v |= (uint32_t)jb.src[(int64_t)(y0 + r) * jb.srcPitch + x0 + c4 + k] << (8 * k);
        *(uint32_t*)(sSrc + r * 64 + c4) = v;
    }
    {
        const int dw = RW / 4;
        // the window may reach beyond the padded picture (small margins: CTU 16 pads 48 x 32; the last CTU row and column of any picture): those
        // positions belong to illegal vectors only — whatever is staged for them is never looked at — but the loads must stay inside the buffer
        const int yMax = a.bufRows - a.marginY - 1, xMax = a.picW + a.marginX - 4;
        for (int i = tid; i < (64 + D) * dw; i += 1024)
        {
            const int r = i / dw, c4 = (i - r * dw) * 4;
            const int y = min(max(y0 - S + r, -a.marginY), yMax), x = min(max(x0 - S + c4, -a.marginX), xMax);
            *(uint32_t*)(sRef + r * RW + c4) = ld_global_unaligned<uint32_t>(jb.ref + (int64_t)y * jb.refStride + x);
        }
    }
    if (tid < 32)
        sh.vc[tid] = (uint32_t)((jb.lambda20 * (2 * tid + 2) + 10) / 20);        // bits(4 |dx|) + bits(4 |dy|) = 2 (log + log) + 2
    __syncthreads();

    // ---- measure: wave b <-> 16x16 block b.  A lane owns an 8 x 8 patch of vectors: 16 accumulators of four u16 SADs each.  Reference row j of
    // the patch (24 bytes) is read once and meets source row j - dy for each of the patch's eight dy: 64 v_qsad_pk_u16_u8 per 24 bytes of LDS ----
    {
        const int b = wave, bx = b & 3, by = b >> 2;
        const bool inside = x0 + bx * 16 + 16 <= a.picW && y0 + by * 16 + 16 <= a.picH;
        if (inside)
        {
            uint32_t s[16][4];
#pragma unroll
            for (int r = 0; r < 16; r++)
            {
                // the block's pixels are the same for every lane of the wave: scalar registers, the qsad's second operand
                const uint4 v = *(const uint4*)(sSrc + (by * 16 + r) * 64 + bx * 16);
                s[r][0] = __builtin_amdgcn_readfirstlane(v.x); s[r][1] = __builtin_amdgcn_readfirstlane(v.y);
                s[r][2] = __builtin_amdgcn_readfirstlane(v.z); s[r][3] = __builtin_amdgcn_readfirstlane(v.w);
            }
            const int G = D / 8;
            for (int item = lane; item < G * G; item += 64)
            {
                const int dyg = item / G, xs = item - dyg * G;
                const uint8_t* base = sRef + (by * 16 + 8 * dyg) * RW + bx * 16 + 8 * xs;
                uint64_t acc[8][2];
#pragma unroll
                for (int i = 0; i < 8; i++) acc[i][0] = acc[i][1] = 0;
#pragma unroll
                for (int j = 0; j < 23; j++)
                {
                    const uint2* p = (const uint2*)(base + j * RW);
                    const uint2 p0 = p[0], p1 = p[1], p2 = p[2];
                    const uint32_t d[6] = { p0.x, p0.y, p1.x, p1.y, p2.x, p2.y };
#pragma unroll
                    for (int dyo = 0; dyo < 8; dyo++)
                    {
                        const int r = j - dyo;
                        if (r < 0 || r > 15)
                            continue;
#pragma unroll
                        for (int k = 0; k < 4; k++)
                        {
                            acc[dyo][0] = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)d[k + 1] << 32) | d[k], s[r][k], acc[dyo][0]);
                            acc[dyo][1] = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)d[k + 2] << 32) | d[k + 1], s[r][k], acc[dyo][1]);
                        }
                    }
                }
                // an accumulator's u16 lanes hold at most 16 * 16 * 255 = 65280: no carry between them
                uint16_t* o = sSurf + ((size_t)b * D + 8 * dyg) * D + 8 * xs;
#pragma unroll
                for (int dyo = 0; dyo < 8; dyo++)
                    *(uint4*)(o + dyo * D) = make_uint4((uint32_t)acc[dyo][0], (uint32_t)(acc[dyo][0] >> 32), (uint32_t)acc[dyo][1], (uint32_t)(acc[dyo][1] >> 32));
            }
        }
    }
    __syncthreads();

    ss_decide_emit<uint16_t>(a, jb, sSurf, sh, S, x0, y0, cx, cy);
    char* chunk = jb.out + (int64_t)cy * a.pitch;
    const auto& sOrg = sh.org;
    // level 0: the four 8x8 blocks of wave b's 16x16 block take their parent's window; their SADs are measured now, 4 vectors per lane (one row of
    // the window = 4 lanes): per source row two v_qsad_pk_u16_u8 on a 12-byte stretch of the reference row, brought to byte alignment with
    // v_alignbyte (the window's origin is arbitrary).  An 8x8 block whose parent is not inside the picture has no window: origin (-32768, -32768)
    {
        const int b = wave, bx16 = b & 3, by16 = b >> 2;
        const bool parent = x0 + bx16 * 16 + 16 <= a.picW && y0 + by16 * 16 + 16 <= a.picH;
        const int ox = parent ? sOrg[1][b][0] : 0, oy = parent ? sOrg[1][b][1] : 0;
#pragma unroll 1
        for (int q = 0; q < (jb.level0 ? 4 : 0); q++)
        {
            const int qx = q & 1, qy = q >> 1;
            const int x = x0 + bx16 * 16 + qx * 8, y = y0 + by16 * 16 + qy * 8;
            if (x + 8 > a.picW || y + 8 > a.picH)
                continue;
            const int64_t idx = (int64_t)(by16 * 2 + qy) * a.blocksX[0] + cx * 8 + bx16 * 2 + qx;
            if (!parent)
            {
                if (lane == 0)
                    *(uint32_t*)(chunk + a.originOff[0] + idx * 4) = 0x80008000u;
                *(uint2*)((uint16_t*)(chunk + a.tableOff[0]) + idx * (kWin * kWin) + lane * 4) = make_uint2(0, 0);
                continue;
            }
            if (lane == 0)
                *(uint32_t*)(chunk + a.originOff[0] + idx * 4) = (uint32_t)(uint16_t)(int16_t)ox | ((uint32_t)(uint16_t)(int16_t)oy << 16);
            const int j = lane >> 2, i4 = (lane & 3) * 4;
            const int col = bx16 * 16 + qx * 8 + ox + S + i4, sh = col & 3;
            const uint8_t* rp0 = sRef + (by16 * 16 + qy * 8 + oy + S + j) * RW + (col - sh);
            const uint8_t* sp0 = sSrc + (by16 * 16 + qy * 8) * 64 + bx16 * 16 + qx * 8;
            uint64_t acc = 0;
#pragma unroll
            for (int r = 0; r < 8; r++)
            {
                const uint2 sv = *(const uint2*)(sp0 + r * 64);
                const uint32_t s0 = __builtin_amdgcn_readfirstlane(sv.x), s1 = __builtin_amdgcn_readfirstlane(sv.y);
                const uint32_t* w = (const uint32_t*)(rp0 + r * RW);
                const uint32_t d0 = w[0], d1 = w[1], d2 = w[2], d3 = w[3];
                const uint32_t w0 = __builtin_amdgcn_alignbyte(d1, d0, sh), w1 = __builtin_amdgcn_alignbyte(d2, d1, sh), w2 = __builtin_amdgcn_alignbyte(d3, d2, sh);
                acc = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)w1 << 32) | w0, s0, acc);
                acc = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)w2 << 32) | w1, s1, acc);
            }
            uint16_t* tab = (uint16_t*)(chunk + a.tableOff[0]) + idx * (kWin * kWin);
            *(uint2*)(tab + j * kWin + i4) = make_uint2((uint32_t)acc, (uint32_t)(acc >> 32));
        }
    }
}

// ---- the same kernel for 16-bit pictures (Main10 / Main12 builds) --------------------------------------------------------------------------
// v_qsad_pk_u16_u8 is a byte instruction; 16-bit samples are measured with v_sad_u16 (two absolute differences per lane and instruction), and a 16x16
// SAD no longer fits 16 bits, so the surface holds u32 — 16 x (2 S)^2 x 4 bytes: S <= 16 keeps it in LDS (64 KB).  The reference window is staged twice,
// once as it is and once shifted by one sample, so that odd vectors read aligned dwords too.  A lane owns a 4 x 4 patch of vectors; the block's source
// rows go through scalar registers eight at a time.  Decide and emit are the shared template.
__global__ __launch_bounds__(1024) void sadsurf_ctu16_kernel(SurfArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ SsShared sh;

    int jn = 0, rowIn, cx;
    surf_ctu(a.xcd, cx, rowIn);
    while (jn + 1 < a.nJobs && rowIn >= a.job[jn].rows) { rowIn -= a.job[jn].rows; jn++; }
    const SurfJob& jb = a.job[jn];
    const int S = jb.S, D = 2 * S;
    const int RWD = (64 + D) / 2 + 4;                            // window row pitch in dwords (two samples each): room for the last 9-dword read, even
    uint32_t* sSrc = (uint32_t*)smem;                            // [64][32] dwords
    uint32_t* sRefA = sSrc + 64 * 32;                            // [64 + D][RWD]: dword k = samples (2k, 2k + 1) of the window row
    uint32_t* sRefB = sRefA + (64 + D) * RWD;                    // the same shifted by one sample: dword k = samples (2k + 1, 2k + 2)
    uint32_t* sSurf = sRefB + (64 + D) * RWD;                    // [16][D][D]

    const int cy = jb.row0 + rowIn;
    const int x0 = cx * 64, y0 = cy * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.subpelOff3 >= 0 && cx < a.blocksX[3] && tid < X265HIP_SADSURF_SUBPEL)          // (as in the 8-bit kernel: 10-bit pictures' tables are summed over quadrant jobs too)
        ((uint32_t*)(jb.out + (int64_t)cy * a.pitch + a.subpelOff3))[(int64_t)cx * X265HIP_SADSURF_SUBPEL + tid] = 0;
    const uint16_t* src = (const uint16_t*)jb.src;
    const uint16_t* ref = (const uint16_t*)jb.ref;
    const int64_t srcPitch = jb.srcPitch / 2;                    // samples

    // ---- stage ----
    for (int i = tid; i < 64 * 32; i += 1024)
    {
        const int r = i >> 5, c2 = (i & 31) * 2;
        uint32_t v = 0;
        if (y0 + r < a.picH)
        {
            if (x0 + c2 < a.picW) v = src[(int64_t)(y0 + r) * srcPitch + x0 + c2];
            if (x0 + c2 + 1 < a.picW) v |= (uint32_t)src[(int64_t)(y0 + r) * srcPitch + x0 + c2 + 1] << 16;
        }
        sSrc[r * 32 + c2 / 2] = v;
    }
    {
        // positions outside the padded picture's buffer belong to illegal vectors only: the loads are clamped into the buffer (see the 8-bit kernel)
        // (sample pairs start at even positions and the padded width is even: a pair is inside or outside as a whole; the third sample is clamped alone)
        const int yMax = a.bufRows - a.marginY - 1, xEnd = a.picW + a.marginX;
        for (int i = tid; i < (64 + D) * RWD; i += 1024)
        {
            const int r = i / RWD, k = i - r * RWD;
            const int y = min(max(y0 - S + r, -a.marginY), yMax), x = min(max(x0 - S + 2 * k, -a.marginX), xEnd - 2);
            const uint16_t* p = ref + (int64_t)y * jb.refStride;
            const uint32_t s0 = p[x], s1 = p[x + 1], s2 = p[min(x + 2, xEnd - 1)];
            sRefA[r * RWD + k] = s0 | (s1 << 16);
            sRefB[r * RWD + k] = s1 | (s2 << 16);
        }
    }
    if (tid < 32)
        sh.vc[tid] = (uint32_t)((jb.lambda20 * (2 * tid + 2) + 10) / 20);
    __syncthreads();

    // ---- measure: wave b <-> 16x16 block b; a lane owns 4 x 4 vectors (16 u32 accumulators); source rows eight at a time in scalar registers ----
    {
        const int b = wave, bx = b & 3, by = b >> 2;
        const bool inside = x0 + bx * 16 + 16 <= a.picW && y0 + by * 16 + 16 <= a.picH;
        if (inside)
        {
            const int G = D / 4;
            for (int item = lane; item < G * G; item += 64)
            {
                const int py4 = item / G, px4 = item - py4 * G;
                uint32_t acc[4][4];
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int c = 0; c < 4; c++) acc[i][c] = 0;
#pragma unroll 1
                for (int h = 0; h < 2; h++)
                {
                    uint32_t s[8][8];
#pragma unroll
                    for (int r = 0; r < 8; r++)
                    {
                        const uint4 v0 = *(const uint4*)(sSrc + (by * 16 + 8 * h + r) * 32 + bx * 8), v1 = *(const uint4*)(sSrc + (by * 16 + 8 * h + r) * 32 + bx * 8 + 4);
                        s[r][0] = __builtin_amdgcn_readfirstlane(v0.x); s[r][1] = __builtin_amdgcn_readfirstlane(v0.y);
                        s[r][2] = __builtin_amdgcn_readfirstlane(v0.z); s[r][3] = __builtin_amdgcn_readfirstlane(v0.w);
                        s[r][4] = __builtin_amdgcn_readfirstlane(v1.x); s[r][5] = __builtin_amdgcn_readfirstlane(v1.y);
                        s[r][6] = __builtin_amdgcn_readfirstlane(v1.z); s[r][7] = __builtin_amdgcn_readfirstlane(v1.w);
                    }
                    const int base = (by * 16 + 8 * h + 4 * py4) * RWD + bx * 8 + 2 * px4;
#pragma unroll
                    for (int j = 0; j < 11; j++)
                    {
                        const uint32_t* pa = sRefA + base + j * RWD;
                        const uint32_t* pb = sRefB + base + j * RWD;
                        uint32_t da[9], db[9];
#pragma unroll
                        for (int k = 0; k < 4; k++)
                        {
                            const uint2 va = *(const uint2*)(pa + 2 * k), vb = *(const uint2*)(pb + 2 * k);
                            da[2 * k] = va.x; da[2 * k + 1] = va.y; db[2 * k] = vb.x; db[2 * k + 1] = vb.y;
                        }
                        da[8] = pa[8]; db[8] = pb[8];
#pragma unroll
                        for (int dyo = 0; dyo < 4; dyo++)
                        {
                            const int r = j - dyo;
                            if (r < 0 || r > 7)
                                continue;
#pragma unroll
                            for (int k = 0; k < 8; k++)
                            {
                                acc[dyo][0] = __builtin_amdgcn_sad_u16(da[k], s[r][k], acc[dyo][0]);
                                acc[dyo][1] = __builtin_amdgcn_sad_u16(db[k], s[r][k], acc[dyo][1]);
                                acc[dyo][2] = __builtin_amdgcn_sad_u16(da[k + 1], s[r][k], acc[dyo][2]);
                                acc[dyo][3] = __builtin_amdgcn_sad_u16(db[k + 1], s[r][k], acc[dyo][3]);
                            }
                        }
                    }
                }
                uint32_t* o = sSurf + ((size_t)b * D + 4 * py4) * D + 4 * px4;
#pragma unroll
                for (int dyo = 0; dyo < 4; dyo++)
                    *(uint4*)(o + dyo * D) = make_uint4(acc[dyo][0], acc[dyo][1], acc[dyo][2], acc[dyo][3]);
            }
        }
    }
    __syncthreads();

    ss_decide_emit<uint32_t>(a, jb, sSurf, sh, S, x0, y0, cx, cy);
}


// ---- sub-pel SATD tables (round 4) ---------------------------------------------------------------------------------------------------------------
// What MotionEstimate::subpelCompare (reference encoder/motion.cpp:1571-1600) computes with the satd comparison — satd(source block, reference block
// at quarter-pel vector q), the reference block being the picture itself (q integer) or luma_hpp / vpp / hvpp of it, i.e. the mirror's phase plane
// 4 fy + fx — for the 7 x 7 vectors q = 4 c + (dx, dy) around the window's centre c = origin + WIN / 2 of every block of levels 1..3.  A job = one
// (block, vector); a wave runs 64 / T jobs at once, T = lanes per job (one 4x4 tile each per step), as pixcmp_kernel<SATD> does (pixel.hip); the SATD of
// a 16x16 / 32x32 / 64x64 block is the sum over its 4x4 tiles of (sum |H d H^T|) >> 1 (pixel.cpp:210-297: satd8<w, h> over satd_8x4).
struct SubpelArgs
{
    const void* pic; const void* planes; int64_t stride, planeElems;        // reference: pixel (0, 0) of the picture / of plane 0; planes 1..15 follow
    const void* src; int64_t srcPitch;                                      // source luma, bytes per row
    char* out;                                                              // the surface's device chunks
    int64_t pitch, originOff[4], subpelOff[4];
    int blocksX[4], blocksY[4], per[4];
    int row0, rows;
    int jobs[4];                                                            // (block, vector) jobs of levels 1..3 in this launch: prefix sums
    int xcd;                                                                // 1: XCD-aware job order (the grid is a multiple of 128)
};
template <typename P>
__global__ __launch_bounds__(256) void subpel_satd_kernel(SubpelArgs a)
{
    const int lane = threadIdx.x & 63;
    // XCD-aware order: workgroups are dealt to the eight XCDs round-robin, each XCD has its own L2, and consecutive jobs — the 49 vectors of a block, then the next
    // block of the row, whose phase-plane lines are the same 128-byte lines — want ONE L2.  The job order is cut into chunks of 16 workgroups, chunk c goes to
    // XCD c % 8: workgroup b (XCD b % 8, its i-th there, i = b / 8) takes place ((i / 16) * 8 + b % 8) * 16 + i % 16.  Chunks rather than one contiguous eighth
    // per XCD: the levels have few jobs each (64x64 blocks: 368 workgroups' worth per picture row set) and an eighth-per-XCD split leaves XCDs without work —
    // measured 95.9 us per launch against 77.2 us in the plain order, with the HBM fetch already down 5.3 x.  The grid is a multiple of 128.
    // X265HIP_SADSURF_XCD=0 (a.xcd == 0): the plain order
    const int xi = (int)(blockIdx.x >> 3), xq = (int)(blockIdx.x & 7);
    const int vblock = a.xcd ? (((xi >> 4) * 8 + xq) << 4) + (xi & 15) : (int)blockIdx.x;
    const int gwave = vblock * (blockDim.x >> 6) + (threadIdx.x >> 6), wavesTotal = gridDim.x * (blockDim.x >> 6);
    const P* pic = (const P*)a.pic;
    const P* planes = (const P*)a.planes;
    for (int l = 1; l < 4; l++)
    {
        const int N = 8 << l, tiles = (N >> 2) * (N >> 2), T = tiles >= 64 ? 64 : tiles, bpw = 64 / T, iters = tiles / T;
        const int n = a.jobs[l] - a.jobs[l - 1], sub = lane & (T - 1);
        const int rowBlocks = a.per[l] * a.blocksX[l];
        for (int job0 = gwave * bpw; job0 < n; job0 += wavesTotal * bpw)
        {
            const int job = job0 + lane / T;
            int jc = job < n ? job : n - 1;
            int v = jc % X265HIP_SADSURF_SUBPEL, b = jc / X265HIP_SADSURF_SUBPEL;           // b: block index inside the launch's rows of this level
            int cr = a.row0 + b / rowBlocks, k = b % rowBlocks;                             // chunk (row of 64 lines), index inside the chunk
            int by = cr * a.per[l] + k / a.blocksX[l];
            // block rows below the picture (a last CTU row that is not 64 lines tall) exist in the chunk's layout but not in the pictures: such lanes
            // (and the lanes past the last job) measure the picture's first block instead (its origin was written by this launch or an earlier one) and
            // store nothing
            const bool ok = job < n && by < a.blocksY[l];
            if (!ok) { v = 0; cr = 0; k = 0; by = 0; }
            const int bx = k % a.blocksX[l];
            const bool inPic = ok;
            char* chunk = a.out + (int64_t)cr * a.pitch;
            const int16_t* org = (const int16_t*)(chunk + a.originOff[l]) + 2 * k;
            const int cx = org[0] + kWin / 2, cy = org[1] + kWin / 2;
            const int qx = 4 * cx + v % 7 - 3, qy = 4 * cy + v / 7 - 3;
            const int phase = 4 * (qy & 3) + (qx & 3);
            const P* ref = (phase ? planes + (int64_t)phase * a.planeElems : pic) + (int64_t)(by * N + (qy >> 2)) * a.stride + bx * N + (qx >> 2);
            const P* srcB = (const P*)((const char*)a.src + (int64_t)(by * N) * a.srcPitch) + bx * N;
            const int64_t srcStride = a.srcPitch / (int64_t)sizeof(P);
            int acc = 0;
            for (int it = 0; it < iters; it++)
            {
                const int t = sub + it * T, ty = (t / (N >> 2)) * 4, tx = (t % (N >> 2)) * 4;
                int m[16];
                tile_diff(srcB + ty * srcStride + tx, srcStride, ref + ty * a.stride + tx, a.stride, m);
                hadamard4x4(m);
                acc += abs_sum16(m) >> 1;
            }
            acc = group_sum(acc, T);
            if (ok && inPic && sub == 0)
                ((uint32_t*)(chunk + a.subpelOff[l]))[(int64_t)k * X265HIP_SADSURF_SUBPEL + v] = (uint32_t)acc;
        }
    }
}

// ---- the same tables for 8-bit pictures, second form (round 6): the planes through LDS, the lanes are the vectors, packed 16-bit Hadamard --------------------
// subpel_satd_kernel above measures a (block, vector) per group of lanes, a lane per 4x4 tile: eight scattered 4-byte global loads and ~160 VALU instructions per
// tile and vector — the texture addresser and the VALU are both near their limits (ISA count; the planes' lines come out of L2).  Here a workgroup takes a
// block: the block's footprint in the sixteen phase planes — (Q + 1) rows of Q + 1 pixels around the window's centre, whole dwords — is staged in LDS ONCE for
// the 49 vectors, lane v measures vector v and walks the block's tiles: a tile's four rows are four ds_read2_b32 (the two dwords a row of four unaligned pixels
// lies in) and two v_perm_b32 each with the lane's own byte selectors, which split them into 16-bit pairs at the same time.  The Hadamard runs on packed pairs
// (|coefficients| <= 16 * 255; common.h), and it is LINEAR: the source tile's transform M s is made once per tile (64 lanes, one tile each, into LDS) and
// a lane accumulates sum |M s - M r| with v_sad_u16 — two coefficients and the accumulation per instruction.  v_sad_u16 compares unsigned halves: both
// transforms carry the same bias 0x4000 in their first sample, which reaches every coefficient as +-0x4000 and cancels in the difference.  M is the Hadamard up
// to the sign of every other coefficient (the rotate-and-multiply butterfly below), the same for both sides.  ~57 VALU instructions per tile and vector
// instead of ~160, no scattered global loads.  A tile's sum of magnitudes is even (it is congruent to the sum of the coefficients = 16 x the first sample), so
// shifting the block's total once equals pixel.cpp:210-297's shift per 8x4.
// Jobs of a launch, all of one size (one staging, 16 tiles per wave): the four 32x32 quadrants of every 64x64 block (their halved sums meet in the entry by
// atomicAdd; the window kernel zeroes those entries), the 32x32 blocks (the four waves take 16 tiles each), the 16x16 blocks four to a workgroup (a block per
// wave).  X265HIP_SUBPEL_LDS=0: the first form.
namespace sp8 {
// geometry of a staged block, in dwords: B bytes per pixel, Q x Q block -> Q + 1 rows of Q + 1 pixels starting at any pixel: whole dwords, Q B / 4 + 1 per row
template <int B, int Q> struct Lay { static constexpr int pitch = Q * B / 4 + 1, plane = (Q + 1) * pitch, block = 16 * plane; };
template <int B> constexpr int ref_dw() { return (4 * Lay<B, 16>::block > Lay<B, 32>::block ? 4 * Lay<B, 16>::block : Lay<B, 32>::block) + 4; }
struct Geo { int ok, x0, y0, sx, sy, k, cr, pad; };      // footprint's first pixel (x0, y0) in the reference, the block's (quadrant's) first pixel (sx, sy) in the source
}

// rows as ((c0, c1), (c2, c3)) pairs in, M * tile out (see above)
__device__ __forceinline__ void pk_hadamard4x4(s2v a[4], s2v b[4])
{
    const s2v kh = { 1, -1 };
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
        a[y] = Ar * kh + A; b[y] = Br * kh + B;
    }
}

// stage ONE block's footprint (Q x Q block, B bytes per pixel) with a group of T threads (a wave or the workgroup), g = the thread's number in the group; (x0, y0) is
// uniform over the group, so a load is a scalar base + a 32-bit offset.  Every thread issues ALL its loads before the first LDS store (a load -> wait -> store loop
// has one 4-byte load in flight per thread: twenty round trips to L2 per staging).  The picture (plane 0) has an allocation of its own; planes 1..15 follow one another.
template <int B, int Q, int T>
__device__ __forceinline__ void sp8_stage_ref(const SubpelArgs& a, int x0, int y0, int g, uint32_t* dst)
{
    constexpr int pitch = sp8::Lay<B, Q>::pitch, planeDw = sp8::Lay<B, Q>::plane, n15 = 15 * planeDw, it0 = (planeDw + T - 1) / T, it15 = (n15 + T - 1) / T;
    const int64_t first = ((int64_t)y0 * a.stride + x0) * B;
    // buffer loads: the 128-bit descriptor of a group-uniform base in scalar registers + ONE 32-bit offset register per load (a flat address is two, and the
    // compiler builds all of a batch's addresses before its first load: 157 registers with global_load, three waves per SIMD)
    const __amdgpu_buffer_rsrc_t d0 = __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)a.pic + (uintptr_t)first) & ~(uintptr_t)3), 0, (int)0xfffffff0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t d1 = __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)a.planes + (uintptr_t)(a.planeElems * B) + (uintptr_t)first) & ~(uintptr_t)3), 0, (int)0xfffffff0u, 0x00020000);
    const uint32_t stride = (uint32_t)a.stride * B, planeBytes = (uint32_t)a.planeElems * B;
    uint32_t v0[it0], v1[it15];
#pragma unroll
    for (int k = 0; k < it0; k++)
    {
        const int i0 = g + T * k, i = i0 < planeDw ? i0 : planeDw - 1, r = i / pitch, j = i - r * pitch;
        v0[k] = __builtin_amdgcn_raw_buffer_load_b32(d0, (int)((uint32_t)r * stride + 4u * (uint32_t)j), 0, 0);
    }
#pragma unroll
    for (int k = 0; k < it15; k++)
    {
        const int i0 = g + T * k, i = i0 < n15 ? i0 : n15 - 1, pp = i / planeDw, r1 = i - pp * planeDw, r = r1 / pitch, j = r1 - r * pitch;
        v1[k] = __builtin_amdgcn_raw_buffer_load_b32(d1, (int)((uint32_t)pp * planeBytes + (uint32_t)r * stride + 4u * (uint32_t)j), 0, 0);
    }
#pragma unroll
    for (int k = 0; k < it0; k++)
        if (g + T * k < planeDw) dst[g + T * k] = v0[k];
#pragma unroll
    for (int k = 0; k < it15; k++)
        if (g + T * k < n15) dst[planeDw + g + T * k] = v1[k];
}
// the transform of one source tile per thread: tile t of the Q x Q block whose first pixel is (sx, sy)
template <typename P, int Q>
__device__ __forceinline__ void sp8_stage_src(const SubpelArgs& a, int sx, int sy, int t, uint32_t* o)
{
    typedef typename std::conditional<sizeof(P) == 1, uint32_t, uint2>::type Row;
    constexpr int TX = Q / 4;
    const int ty = t / TX, tx = t - ty * TX;
    const P* s = (const P*)((const char*)a.src + (int64_t)(sy + 4 * ty) * a.srcPitch) + sx + 4 * tx;
    const int64_t srcStride = a.srcPitch / (int64_t)sizeof(P);
    s2v sa[4], sb[4];
#pragma unroll
    for (int y = 0; y < 4; y++)
        Pk16<P>::split(ld_global_unaligned<Row>(s + y * srcStride), sa[y], sb[y]);
    sa[0] = as_s2(as_u(sa[0]) | 0x4000u);
    pk_hadamard4x4(sa, sb);
    *(uint4*)o = make_uint4(as_u(sa[0]), as_u(sa[1]), as_u(sa[2]), as_u(sa[3]));
    *(uint4*)(o + 4) = make_uint4(as_u(sb[0]), as_u(sb[1]), as_u(sb[2]), as_u(sb[3]));
}

// sum over tiles [t0, t1) of a staged block of sum |M s - M r| for this lane's vector: `ref` = the lane's first dword in the staged footprint.  8 bit: a row of
// four pixels lies in two dwords, (selLo, selHi) pick its bytes into 16-bit pairs.  16 bit: in three, the pairs are dwords (d0, d1) or — an odd first pixel —
// the middles of (d0, d1) and (d1, d2): ONE selector (selLo) for both v_perm_b32
template <int B, int Q>
__device__ __forceinline__ uint32_t sp8_tiles(const uint32_t* ref, uint32_t selLo, uint32_t selHi, const uint32_t* hs, int t0, int t1)
{
    constexpr int pitch = sp8::Lay<B, Q>::pitch, TX = Q / 4;
    uint32_t acc = 0;
#pragma unroll 2
    for (int t = t0; t < t1; t++)
    {
        const int ty = t / TX, tx = t - ty * TX;
        const uint32_t* p = ref + 4 * ty * pitch + tx * B;
        s2v ra[4], rb[4];
#pragma unroll
        for (int y = 0; y < 4; y++)
        {
            const uint32_t d0 = p[y * pitch], d1 = p[y * pitch + 1];
            if (B == 1)
            {
                ra[y] = as_s2(__builtin_amdgcn_perm(d1, d0, selLo));
                rb[y] = as_s2(__builtin_amdgcn_perm(d1, d0, selHi));
            }
            else
            {
                const uint32_t d2 = p[y * pitch + 2];
                ra[y] = as_s2(__builtin_amdgcn_perm(d1, d0, selLo));
                rb[y] = as_s2(__builtin_amdgcn_perm(d2, d1, selLo));
            }
        }
        ra[0] = as_s2(as_u(ra[0]) | 0x4000u);
        pk_hadamard4x4(ra, rb);
        const uint4 h0 = *(const uint4*)(hs + 8 * t), h1 = *(const uint4*)(hs + 8 * t + 4);
        acc = __builtin_amdgcn_sad_u16(as_u(ra[0]), h0.x, acc); acc = __builtin_amdgcn_sad_u16(as_u(ra[1]), h0.y, acc);
        acc = __builtin_amdgcn_sad_u16(as_u(ra[2]), h0.z, acc); acc = __builtin_amdgcn_sad_u16(as_u(ra[3]), h0.w, acc);
        acc = __builtin_amdgcn_sad_u16(as_u(rb[0]), h1.x, acc); acc = __builtin_amdgcn_sad_u16(as_u(rb[1]), h1.y, acc);
        acc = __builtin_amdgcn_sad_u16(as_u(rb[2]), h1.z, acc); acc = __builtin_amdgcn_sad_u16(as_u(rb[3]), h1.w, acc);
    }
    return acc;
}

// P = uint8_t: 8-bit pictures; uint16_t: 10-bit pictures (|coefficients| <= 16 * 1023 = 16 368: inside 16 bits with the bias; 12-bit pictures keep the first form)
template <typename P>
__global__ __launch_bounds__(256) void subpel_satd_kernel_lds(SubpelArgs a)
{
    constexpr int B = (int)sizeof(P);
    typedef sp8::Lay<B, 16> L16;
    typedef sp8::Lay<B, 32> L32;
    __shared__ uint32_t sRef[sp8::ref_dw<B>()];
    __shared__ __attribute__((aligned(16))) uint32_t sHs[64 * 8];
    __shared__ uint32_t sPart[4 * 64];
    __shared__ sp8::Geo sGeo[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n1 = (a.jobs[1] - a.jobs[0]) / X265HIP_SADSURF_SUBPEL, n2 = (a.jobs[2] - a.jobs[1]) / X265HIP_SADSURF_SUBPEL, n3 = (a.jobs[3] - a.jobs[2]) / X265HIP_SADSURF_SUBPEL;
    const int total = 4 * n3 + n2 + ((n1 + 3) >> 2);
    // the lane's vector: quarter-pel offset (dx, dy) from the centre -> phase plane 4 fy + fx at the whole-pel offset (dx >> 2, dy >> 2) = -1 or 0
    const int v = lane < X265HIP_SADSURF_SUBPEL ? lane : 0;
    const int dx = v % 7 - 3, dy = v / 7 - 3, phase = 4 * (dy & 3) + (dx & 3);
    const int xi = (int)(blockIdx.x >> 3), xq = (int)(blockIdx.x & 7);
    const int first = a.xcd ? (((xi >> 4) * 8 + xq) << 4) + (xi & 15) : (int)blockIdx.x;          // (subpel_satd_kernel: chunks of 16 workgroups per XCD)
    for (int job = first; job < total; job += gridDim.x)
    {
        // a 64x64 block is four jobs, one per 32x32 quadrant: every job is ONE staging + 16 tiles per wave (a workgroup that walked the four quadrants itself was
        // the launch's long pole: a launch is ~180 CTUs, two rounds of workgroups).  A quadrant's sum is even like a tile's, so the halves add up to the block's
        // entry: atomicAdd into entries the window kernel of the same rows has zeroed (SurfArgs::subpelOff3)
        const int l = job < 4 * n3 ? 3 : job < 4 * n3 + n2 ? 2 : 1;
        const int b0 = l == 3 ? job >> 2 : l == 2 ? job - 4 * n3 : 4 * (job - 4 * n3 - n2), quad = l == 3 ? job & 3 : 0;
        const int nb = l == 1 ? (n1 - b0 < 4 ? n1 - b0 : 4) : 1, N = 8 << l;
        const int rowBlocks = a.per[l] * a.blocksX[l];
        __syncthreads();                                                             // the previous job's LDS is no longer read
        if (tid < 4)
        {
            sp8::Geo g = { 0, 0, 0, 0, 0, 0, 0, 0 };
            if (tid < nb)
            {
                const int b = b0 + tid;
                g.cr = a.row0 + b / rowBlocks; g.k = b % rowBlocks;
                const int by = g.cr * a.per[l] + g.k / a.blocksX[l], bx = g.k % a.blocksX[l];
                g.ok = by < a.blocksY[l];
                if (g.ok)
                {
                    const int16_t* org = (const int16_t*)(a.out + (int64_t)g.cr * a.pitch + a.originOff[l]) + 2 * g.k;
                    g.sx = bx * N + 32 * (quad & 1); g.sy = by * N + 32 * (quad >> 1);
                    g.x0 = g.sx + org[0] + kWin / 2 - 1; g.y0 = g.sy + org[1] + kWin / 2 - 1;
                }
            }
            sGeo[tid] = g;
        }
        __syncthreads();
        {
            // (blocks that do not exist carry the geometry of the picture's first block: valid addresses, nothing of them is kept)
            const sp8::Geo& gs = sGeo[l == 1 ? wave : 0];
            const int x0 = __builtin_amdgcn_readfirstlane(gs.x0), y0 = __builtin_amdgcn_readfirstlane(gs.y0);
            if (l == 1) sp8_stage_ref<B, 16, 64>(a, x0, y0, lane, sRef + wave * L16::block);
            else sp8_stage_ref<B, 32, 256>(a, x0, y0, tid, sRef);
            if (tid < 64)
            {
                const sp8::Geo& gt = sGeo[l == 1 ? tid >> 4 : 0];
                if (gt.ok)
                {
                    if (l == 1) sp8_stage_src<P, 16>(a, gt.sx, gt.sy, tid & 15, sHs + 8 * tid);
                    else sp8_stage_src<P, 32>(a, gt.sx, gt.sy, tid, sHs + 8 * tid);
                }
            }
        }
        __syncthreads();
        const sp8::Geo& g = sGeo[l == 1 ? wave : 0];
        uint32_t acc = 0;
        if (g.ok)
        {
            // alignment of the footprint's first pixel in its dword: the planes share one (planeElems and the stride are multiples of 4 bytes), the picture has its own
            // allocation.  col = the lane's first pixel in the staged row, counted in pixels from the row's first dword
            const uintptr_t base = phase ? (uintptr_t)a.planes : (uintptr_t)a.pic;
            const int col = (int)(((base + (uintptr_t)(((int64_t)g.y0 * a.stride + g.x0) * B)) & 3) / B) + 1 + (dx >> 2);
            constexpr int perDw = 4 / B;
            const int sh = col & (perDw - 1), dw = col / perDw;
            const uint32_t selLo = B == 1 ? ((uint32_t)sh | 0x0c000c00u | ((uint32_t)(sh + 1) << 16)) : (sh ? 0x05040302u : 0x03020100u);
            const uint32_t selHi = (uint32_t)(sh + 2) | 0x0c000c00u | ((uint32_t)(sh + 3) << 16);
            if (l == 1)
                acc = sp8_tiles<B, 16>(sRef + wave * L16::block + phase * L16::plane + (1 + (dy >> 2)) * L16::pitch + dw, selLo, selHi, sHs + wave * 16 * 8, 0, 16);
            else
                acc = sp8_tiles<B, 32>(sRef + phase * L32::plane + (1 + (dy >> 2)) * L32::pitch + dw, selLo, selHi, sHs, 16 * wave, 16 * wave + 16);
        }
        if (l == 1)
        {
            if (g.ok && lane < X265HIP_SADSURF_SUBPEL)
                ((uint32_t*)(a.out + (int64_t)g.cr * a.pitch + a.subpelOff[l]))[(int64_t)g.k * X265HIP_SADSURF_SUBPEL + lane] = acc >> 1;
        }
        else
        {
            sPart[tid] = acc;
            __syncthreads();
            if (g.ok && tid < X265HIP_SADSURF_SUBPEL)
            {
                uint32_t* o = (uint32_t*)(a.out + (int64_t)g.cr * a.pitch + a.subpelOff[l]) + (int64_t)g.k * X265HIP_SADSURF_SUBPEL + tid;
                const uint32_t sum = (sPart[tid] + sPart[64 + tid] + sPart[128 + tid] + sPart[192 + tid]) >> 1;
                if (l == 3) atomicAdd(o, sum);
                else *o = sum;
            }
        }
    }
}

static size_t surf16_lds_bytes(int S)
{
    const int D = 2 * S, RWD = (64 + D) / 2 + 4;
    return (size_t)64 * 32 * 4 + (size_t)2 * (64 + D) * RWD * 4 + (size_t)16 * D * D * 4;
}

static size_t surf_lds_bytes(int S)
{
    const int D = 2 * S, RW = 64 + D + 8;
    return 4096 + (size_t)(64 + D) * RW + (size_t)16 * D * D * 2;
}

// rows of `ss` that the reference's uploaded rows allow now: [ss->rowsBuilt, returned value)
static int rows_possible(x265hip_sadsurf* ss)
{
    x265hip_refpic* rp = ss->ref;
    if (!rp || rp->failed.load() || ss->released)
        return ss->rowsBuilt;
    const bool complete = rp->uploaded >= rp->marginY + rp->picH + rp->marginY;
    const int finalPic = rp->uploaded - rp->marginY;           // picture rows [.., finalPic) are on the device
    int r1 = ss->rowsBuilt;
    while (r1 < ss->lay.ctuRows && (complete || 64 * (r1 + 1) + ss->S <= finalPic))
        r1++;
    return r1;
}

// the replica of rp at `place`, created on first use (worker thread)
static Replica* replica_at(x265hip_refpic* rp, int place)
{
    for (Replica* r : rp->replicas)
        if (r->place == place) return r;
    const int dev = place_device(place);
    if (dev < 0 || hipSetDevice(dev) != hipSuccess)
        return nullptr;
    Replica* r = new Replica;
    r->place = place; r->device = dev;
    const size_t bytes = (size_t)rp->planeElems * rp->B;
    // (the planes: 16 x the picture, plane 0 unused, as in the mirror itself; without them the replica serves integer-pel surfaces only)
    if (hipStreamCreateWithFlags(&r->st, hipStreamNonBlocking) != hipSuccess || hipMalloc((void**)&r->dPic, bytes) != hipSuccess)
    {
        if (r->st) (void)hipStreamDestroy(r->st);
        delete r;
        (void)hipSetDevice(rp->device);
        return nullptr;
    }
    if (dev != rp->device)
    {
        // direct access between the two GPUs where the fabric offers it (xGMI inside a node); hipMemcpyPeerAsync works either way
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, dev, rp->device) == hipSuccess && can && hipDeviceEnablePeerAccess(rp->device, 0) != hipSuccess)
            (void)hipGetLastError();                       // already enabled
    }
    if (hipMalloc((void**)&r->dPlanes, 16 * bytes) != hipSuccess) { (void)hipGetLastError(); r->dPlanes = nullptr; }
    (void)hipSetDevice(rp->device);
    rp->replicas.push_back(r);
    g_statReplicas++;
    return r;
}

// Build what the references' rows allow of every surface attached to the pictures in `rps` (worker thread; the bands' uploads have been synchronised).
// Surfaces that share geometry, table layout and the device they are built on go into ONE launch, whichever reference picture they follow (a job
// carries its own reference pointer): the three surfaces a P frame attaches within microseconds of each other — one per reference picture — are
// 1530 CTUs together, 7 rounds of the 224 CUs the job server leaves free at 0.98 fill, where three launches of 510 CTUs are 3 rounds each at 0.76.
// Surfaces whose source lives at another place are built from the replica there, after the rows it lacks have been pushed device to device.
//
// 148 KB of LDS = one workgroup per CU: a launch takes ceil(CTUs / free CUs) rounds of the same ~54 us whatever the last round holds.  While a
// reference picture is still arriving, its newest rows — the ones no search is waiting for yet — are left out when they would only open a round that
// stays less than 3/4 full; they go with the next band, or with the picture's last band, which takes everything.  X265HIP_SADSURF_ROUNDS=0: every launch
// takes all it can.
struct SurfItem { x265hip_sadsurf* ss; x265hip_refpic* rp; Replica* rep; int dev; int take; bool arriving; };

static void fail_refs(const std::vector<SurfItem>& items, size_t from, size_t to)
{
    for (size_t k = from; k < to; k++) items[k].rp->failed = 1;
}

static void progress_multi(const std::vector<x265hip_refpic*>& rps)
{
    static const bool rounds = !(getenv("X265HIP_SADSURF_ROUNDS") && !atoi(getenv("X265HIP_SADSURF_ROUNDS")));
    std::vector<SurfItem> items;
    for (x265hip_refpic* rp : rps)
    {
        if (rp->failed.load()) continue;
        std::vector<x265hip_sadsurf*> list;
        {
            std::lock_guard<std::mutex> g(g_ssLock);
            list = rp->surfaces;
        }
        const bool arriving = rp->uploaded < rp->marginY + rp->picH + rp->marginY;
        for (x265hip_sadsurf* ss : list)
        {
            if (ss->ref != rp) continue;
            const int take = rows_possible(ss) - ss->rowsBuilt;
            if (take > 0)
                items.push_back(SurfItem{ ss, rp, ss->rep, ss->rep ? ss->rep->device : rp->device, take, arriving });
        }
    }
    if (items.empty())
        return;
    // same device, geometry and table layout next to each other (stable: the order of `rps` and of the surfaces inside a group is kept)
    auto same_group = [](const SurfItem& x, const SurfItem& y) {
        return x.dev == y.dev && x.ss->levels == y.ss->levels && x.rp->depth == y.rp->depth && x.rp->picW == y.rp->picW && x.rp->picH == y.rp->picH &&
               x.rp->marginX == y.rp->marginX && x.rp->marginY == y.rp->marginY && x.rp->bufRows == y.rp->bufRows && x.ss->lay.pitch == y.ss->lay.pitch; };
    {
        std::vector<SurfItem> sorted;
        std::vector<bool> used(items.size(), false);
        for (size_t a0 = 0; a0 < items.size(); a0++)
        {
            if (used[a0]) continue;
            for (size_t b0 = a0; b0 < items.size(); b0++)
                if (!used[b0] && same_group(items[a0], items[b0])) { used[b0] = true; sorted.push_back(items[b0]); }
        }
        items.swap(sorted);
    }
    size_t g0 = 0;
    while (g0 < items.size())
    {
        size_t g1 = g0 + 1;
        while (g1 < items.size() && same_group(items[g0], items[g1])) g1++;
        // whole rounds: rows of pictures that are still arriving may stay behind
        if (rounds)
        {
            const int cols = items[g0].ss->lay.ctuCols, cus = free_compute_units(items[g0].dev);
            int rows = 0, spare = 0;
            for (size_t k = g0; k < g1; k++) { rows += items[k].take; if (items[k].arriving) spare += items[k].take; }
            if (cols > 0 && cols <= cus && rows * cols > cus)
            {
                const int rowsPerRound = cus / cols;
                int excess = rows % rowsPerRound;
                if (excess * 4 < rowsPerRound * 3 && excess <= spare)
                    for (size_t k = g1; k-- > g0 && excess > 0;)
                        if (items[k].arriving)
                        {
                            const int d = items[k].take < excess ? items[k].take : excess;
                            items[k].take -= d; excess -= d;
                        }
            }
        }
        size_t i = g0;
        while (i < g1)
        {
            SurfArgs a;
            memset(&a, 0, sizeof(a));
            static const bool xcdOrderCtu = !(getenv("X265HIP_SADSURF_XCD") && !atoi(getenv("X265HIP_SADSURF_XCD")));
            a.xcd = xcdOrderCtu ? 1 : 0;
            size_t in[kMaxJobs];
            int upto[kMaxJobs], rows = 0, maxS = 0;
            for (; i < g1 && a.nJobs < kMaxJobs; i++)
            {
                const SurfItem& it = items[i];
                if (it.take <= 0)
                    continue;
                x265hip_sadsurf* ss = it.ss;
                const char* dPic = it.rep ? it.rep->dPic : it.rp->dPic;
                SurfJob& j = a.job[a.nJobs];
                j.ref = (const uint8_t*)dPic + ((size_t)it.rp->marginY * it.rp->stride + it.rp->marginX) * it.rp->B; j.refStride = it.rp->stride;   // stride in samples
                j.src = (const uint8_t*)ss->src->dLuma; j.srcPitch = ss->src->pitch;
                j.out = ss->dBuf; j.S = ss->S; j.lambda20 = ss->lambda20; j.row0 = ss->rowsBuilt; j.rows = it.take;
                j.level0 = (ss->levels & 1) && it.rp->depth == 8;
                in[a.nJobs] = i; upto[a.nJobs] = ss->rowsBuilt + it.take;
                rows += j.rows;
                if (ss->S > maxS) maxS = ss->S;
                a.nJobs++;
            }
            if (!a.nJobs)
                break;
            const SurfItem& first = items[in[0]];
            x265hip_refpic* rp0 = first.rp;
            const int dev = first.dev;
            hipStream_t st = first.rep ? first.rep->st : rp0->st;
            if (hipSetDevice(dev) != hipSuccess) { fail_refs(items, g0, g1); return; }
            // the reconstructed rows a replica lacks, straight from the owner's device memory (on the replica's stream, waited for: the launch below may
            // run on another reference's stream)
            for (int k = 0; k < a.nJobs; k++)
            {
                const SurfItem& it = items[in[k]];
                if (!it.rep || it.rep->copied >= it.rp->uploaded)
                    continue;
                const size_t off = (size_t)it.rep->copied * it.rp->stride * it.rp->B, bytes = (size_t)(it.rp->uploaded - it.rep->copied) * it.rp->stride * it.rp->B;
                if (hipMemcpyPeerAsync(it.rep->dPic + off, it.rep->device, it.rp->dPic + off, it.rp->device, bytes, it.rep->st) != hipSuccess ||
                    hipStreamSynchronize(it.rep->st) != hipSuccess)
                {
                    set_error(X265HIP_EHIP, "sadsurf: device-to-device push of reference rows failed");
                    fail_refs(items, g0, g1);
                    (void)hipSetDevice(rp0->device);
                    return;
                }
                g_statPeerBands++;
                g_statPeerBytes += bytes;
                it.rep->copied = it.rp->uploaded;
                // the replica's own sub-pel planes of the rows whose support (y - 3 .. y + 4) has arrived — the same filter launch as the mirror's band
                // (refpic.hip process()), on the replica's device and stream; waited for below with the stream the surfaces are built on
                const int phaseEnd = it.rep->copied - 4;
                if (it.rep->dPlanes && phaseEnd > it.rep->phaseDone)
                {
                    x265hip_refpic* rp = it.rp;
                    const char* org = it.rep->dPic + ((size_t)rp->marginY * rp->stride + rp->marginX) * rp->B;
                    char* porg = it.rep->dPlanes + ((size_t)rp->marginY * rp->stride + rp->marginX) * rp->B;
                    DevSpan spanP(X265HIP_CLK_PLANES, it.rep->st);
                    if (build_subpel_rows(rp->depth, org, rp->stride, -rp->marginX + 4, rp->picW + rp->marginX - 4, it.rep->phaseDone - rp->marginY, phaseEnd - rp->marginY, porg,
                                          rp->planeElems, it.rep->st) || hipStreamSynchronize(it.rep->st) != hipSuccess)
                    {
                        set_error(X265HIP_EHIP, "sadsurf: sub-pel planes of a replica failed");
                        fail_refs(items, g0, g1);
                        (void)hipSetDevice(rp0->device);
                        return;
                    }
                    spanP.end();
                    spanP.bytes = (uint64_t)(phaseEnd - it.rep->phaseDone) * (uint64_t)(rp->picW + 2 * rp->marginX - 8) * 16 * rp->B;
                    spanP.commit();
                    it.rep->phaseDone = phaseEnd;
                }
            }
            const SurfLayout& lay = first.ss->lay;            // same picture size and levels: same layout
            a.picW = rp0->picW; a.picH = rp0->picH; a.marginX = rp0->marginX; a.marginY = rp0->marginY; a.bufRows = rp0->bufRows;
            a.pitch = lay.pitch;
            for (int l = 0; l < 4; l++) { a.originOff[l] = lay.originOff[l]; a.tableOff[l] = lay.tableOff[l]; a.blocksX[l] = lay.blocksX[l]; }
            a.blocksY0 = lay.blocksY[0];
            // (8- and 10-bit pictures with sub-pel tables: the window kernel zeroes the 64x64 level's entries, which subpel_satd_kernel_lds then adds into)
            a.subpelOff3 = (first.ss->levels & 16) && (rp0->depth == 8 || rp0->depth == 10) ? lay.subpelOff[3] : -1;
            // the dynamic LDS limit is a per-device attribute of the kernel
            static std::atomic<uint64_t> attrSet{ 0 };
            if (!(attrSet.load() >> dev & 1))
            {
                if (hipFuncSetAttribute((const void*)sadsurf_ctu_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)surf_lds_bytes(32)) != hipSuccess ||
                    hipFuncSetAttribute((const void*)sadsurf_ctu16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)surf16_lds_bytes(16)) != hipSuccess)
                {
                    set_error(X265HIP_EHIP, "sadsurf: cannot raise the dynamic LDS limit");
                    fail_refs(items, g0, g1);
                    (void)hipSetDevice(rp0->device);
                    return;
                }
                attrSet |= (uint64_t)1 << dev;
            }
            // the launch between two events of its own stream: the kernel's device time (x265hip_device_time, x265hip_sadsurf_stats)
            DevSpan span(X265HIP_CLK_SADSURF, st);
            if (rp0->depth == 8)
                hipLaunchKernelGGL(sadsurf_ctu_kernel, dim3(lay.ctuCols, rows), dim3(1024), surf_lds_bytes(maxS), st, a);
            else
                hipLaunchKernelGGL(sadsurf_ctu16_kernel, dim3(lay.ctuCols, rows), dim3(1024), surf16_lds_bytes(maxS), st, a);
            bool bad = hipGetLastError() != hipSuccess;
            span.end();
            DevSpan span2(X265HIP_CLK_SUBPEL, st);
            // the sub-pel SATD tables of the rows just built (same stream: the origins are there)
            for (int k = 0; k < a.nJobs && !bad; k++)
            {
                const SurfItem& it = items[in[k]];
                if (!(it.ss->levels & 16) || (it.rep && !it.rep->dPlanes))
                    continue;
                x265hip_refpic* rp = it.rp;
                SubpelArgs sa;
                memset(&sa, 0, sizeof(sa));
                // (a surface at another place than its reference's mirror: the replica's rows and the replica's own planes)
                sa.pic = (it.rep ? it.rep->dPic : rp->dPic) + ((size_t)rp->marginY * rp->stride + rp->marginX) * rp->B;
                sa.planes = (it.rep ? it.rep->dPlanes : rp->dPlanes) + ((size_t)rp->marginY * rp->stride + rp->marginX) * rp->B;
                sa.stride = rp->stride; sa.planeElems = rp->planeElems;
                sa.src = a.job[k].src; sa.srcPitch = a.job[k].srcPitch;
                sa.out = a.job[k].out; sa.pitch = lay.pitch;
                sa.row0 = a.job[k].row0; sa.rows = a.job[k].rows;
                for (int l = 1; l < 4; l++)
                {
                    sa.originOff[l] = lay.originOff[l]; sa.subpelOff[l] = lay.subpelOff[l];
                    sa.blocksX[l] = lay.blocksX[l]; sa.blocksY[l] = lay.blocksY[l]; sa.per[l] = lay.per[l];
                    sa.jobs[l] = sa.jobs[l - 1] + sa.rows * lay.per[l] * lay.blocksX[l] * X265HIP_SADSURF_SUBPEL;
                }
                const int waves = (sa.jobs[1] + 3) / 4 + (sa.jobs[2] - sa.jobs[1]) + (sa.jobs[3] - sa.jobs[2]);
                static const bool xcdOrder = !(getenv("X265HIP_SADSURF_XCD") && !atoi(getenv("X265HIP_SADSURF_XCD")));
                sa.xcd = xcdOrder ? 1 : 0;
                const int grid = ((waves / 4 + 1 < 4096 ? waves / 4 + 1 : 4096) + 127) & ~127;
                static const bool ldsForm = !(getenv("X265HIP_SUBPEL_LDS") && !atoi(getenv("X265HIP_SUBPEL_LDS")));
                if ((rp->depth == 8 || rp->depth == 10) && ldsForm && rp->stride * rp->B % 4 == 0 && rp->planeElems * rp->B % 4 == 0)
                {
                    // a workgroup per 32x32 quadrant of a 64x64 block, per 32x32 block, per four 16x16 blocks (subpel_satd_kernel_lds); a multiple of 128 for the XCD order
                    const int blocks1 = sa.jobs[1] / X265HIP_SADSURF_SUBPEL, blocks2 = (sa.jobs[2] - sa.jobs[1]) / X265HIP_SADSURF_SUBPEL, blocks3 = (sa.jobs[3] - sa.jobs[2]) / X265HIP_SADSURF_SUBPEL;
                    const int groups = 4 * blocks3 + blocks2 + (blocks1 + 3) / 4;
                    // (the entries of the 64x64 blocks are sums over four workgroups: the window kernel in front of this launch has zeroed them, a.subpelOff3)
                    if (rp->depth == 8)
                        hipLaunchKernelGGL(subpel_satd_kernel_lds<uint8_t>, dim3(((groups < 16384 ? groups : 16384) + 127) & ~127), dim3(256), 0, st, sa);
                    else
                        hipLaunchKernelGGL(subpel_satd_kernel_lds<uint16_t>, dim3(((groups < 16384 ? groups : 16384) + 127) & ~127), dim3(256), 0, st, sa);
                }
                else if (rp->depth == 8)
                    hipLaunchKernelGGL(subpel_satd_kernel<uint8_t>, dim3(grid), dim3(256), 0, st, sa);
                else
                    hipLaunchKernelGGL(subpel_satd_kernel<uint16_t>, dim3(grid), dim3(256), 0, st, sa);
                bad = bad || hipGetLastError() != hipSuccess;
                // satd W x H = 2 W H B per call
                span2.bytes += (uint64_t)sa.rows * lay.ctuCols * 3 * 64 * 64 * X265HIP_SADSURF_SUBPEL * 2 * rp->B;
            }
            span2.end();
            for (int k = 0; k < a.nJobs; k++)
            {
                // SURVEY.md §8d, "batched exhaustive search of one block over an R x R window counts the unique footprint", applied to what a workgroup
                // stages: ONE 64x64 source block and ONE (64 + R - 1)^2 window per CTU (R = 2 S; the 16x16 / 32x32 / 64x64 searches of the CTU all read
                // that one staging, the 32 / 64 SADs are sums of the 16x16 ones), plus the bytes the CTU really emits: a 16 x 16 window of entries and an
                // origin per block of the levels built (the surfaces themselves never leave LDS)
                const int64_t R = 2 * a.job[k].S;
                span.bytes += (uint64_t)a.job[k].rows * lay.ctuCols * (64 * 64 + (64 + R - 1) * (64 + R - 1)) * rp0->B;
                for (int l = a.job[k].level0 ? 0 : 1; l < 4; l++)
                {
                    int blockRows = 0;
                    for (int r = a.job[k].row0; r < a.job[k].row0 + a.job[k].rows; r++)
                        for (int jj = 0; jj < lay.per[l]; jj++)
                            blockRows += r * lay.per[l] + jj < lay.blocksY[l];
                    span.bytes += (uint64_t)blockRows * lay.blocksX[l] * (kWin * kWin * lay.entryBytes[l] + 4);
                }
            }
            for (int k = 0; k < a.nJobs && !bad; k++)
            {
                x265hip_sadsurf* ss = items[in[k]].ss;
                const size_t off = (size_t)a.job[k].row0 * lay.pitch, bytes = (size_t)a.job[k].rows * lay.pitch;
                bad = hipMemcpyAsync(ss->hBuf + off, ss->dBuf + off, bytes, hipMemcpyDeviceToHost, st) != hipSuccess;
            }
            if (bad || hipStreamSynchronize(st) != hipSuccess)
            {
                set_error(X265HIP_EHIP, "sadsurf: launch, copy or synchronisation failed");
                fail_refs(items, g0, g1);
                (void)hipSetDevice(rp0->device);
                return;
            }
            span.commit();
            if (span2.bytes) span2.commit();
            g_statRows += rows;
            g_statLaunches++;
            for (int k = 0; k < a.nJobs; k++)
            {
                x265hip_sadsurf* ss = items[in[k]].ss;
                ss->rowsBuilt = upto[k];
                ss->ctuRowsReady.store(upto[k], std::memory_order_release);
            }
            (void)hipSetDevice(rp0->device);
        }
        g0 = g1;
    }
}

// One workgroup per CTU: a band of one CTU row of one or two surfaces fills a fraction of the chip for the same ~54 us as a full one.  A picture that is
// still arriving therefore lets its rows wait for company — until the pending CTUs reach X265HIP_SADSURF_BATCH (default 224), the picture is complete, or
// X265HIP_SADSURF_DEFER bands (default 8) have gone by: the searches of a frame start at least the reference-lag rows behind the band, so nobody is waiting
// for the newest rows yet; a search that does arrive early measures its candidates on the host, same values (round 4's bench clip: 6 k of 12 million
// lookups at 2 bands, 20-25 k at 8 and at 16).  True: leave this picture's rows for later.
static bool rows_wait(x265hip_refpic* rp)
{
    static const int batch = getenv("X265HIP_SADSURF_BATCH") ? atoi(getenv("X265HIP_SADSURF_BATCH")) : 224;
    static const int maxDefer = getenv("X265HIP_SADSURF_DEFER") ? atoi(getenv("X265HIP_SADSURF_DEFER")) : 8;
    if (batch <= 0)
        return false;
    std::vector<x265hip_sadsurf*> list;
    {
        std::lock_guard<std::mutex> g(g_ssLock);
        list = rp->surfaces;
    }
    const bool complete = rp->uploaded >= rp->marginY + rp->picH + rp->marginY;
    int pending = 0;
    for (x265hip_sadsurf* ss : list)
        if (ss->ref == rp)
            pending += (rows_possible(ss) - ss->rowsBuilt) * ss->lay.ctuCols;
    if (!complete && pending < batch && rp->ssDeferred < maxDefer)
    {
        if (pending) rp->ssDeferred++;
        return true;
    }
    rp->ssDeferred = 0;
    return false;
}

void sadsurf_rows_arrived(x265hip_refpic* rp)
{
    if (!rows_wait(rp))
        progress_multi(std::vector<x265hip_refpic*>{ rp });
}

static void free_surface(x265hip_sadsurf* ss)
{
    x265hip_srcpic* src = ss->src;
    {
        std::lock_guard<std::mutex> g(g_poolLock);
        g_pool.insert({ { ss->bytes, src->device }, PoolEntry{ ss->dBuf, ss->hBuf } });
    }
    delete ss;
    ::srcpic_unref(src);
}

// attach jobs that reached the worker together (refpic.hip RefWorker::run): each surface is set up, then the reference pictures concerned make ONE
// pass of progress_multi — the surfaces of one source picture against its two or three reference pictures share a launch
void sadsurf_attach_batch(const std::vector<RefJob>& jobs)
{
    std::vector<x265hip_refpic*> rps;
    for (const RefJob& j : jobs)
    {
        x265hip_sadsurf* ss = j.ss;
        // the worker has seen every band queued before this job, so `uploaded` is what the surface can start from
        if (!ss->ref || j.epoch != ss->ref->epoch.load() || hipSetDevice(ss->ref->device) != hipSuccess)
            continue;
        if (ss->src->place != ss->ref->place && !(ss->rep = replica_at(ss->ref, ss->src->place)))
        {
            set_error(X265HIP_ENOMEM, "sadsurf: no replica of the reference picture at place %d", ss->src->place);
            ss->ref->failed = 1;
            continue;
        }
        bool seen = false;
        for (x265hip_refpic* rp : rps) seen = seen || rp == ss->ref;
        // together with whatever rows the reference's other surfaces are waiting with (a reference picture that is complete — the usual case —
        // gives the new surface all its rows at once)
        if (!seen && !rows_wait(ss->ref)) rps.push_back(ss->ref);
    }
    if (!rps.empty())
        progress_multi(rps);
}

void sadsurf_job(const RefJob& j)
{
    x265hip_sadsurf* ss = j.ss;
    if (j.kind == 1)
    {
        sadsurf_attach_batch(std::vector<RefJob>{ j });
        return;
    }
    // release
    {
        std::lock_guard<std::mutex> g(g_ssLock);
        if (ss->ref)
        {
            auto& v = ss->ref->surfaces;
            for (size_t i = 0; i < v.size(); i++)
                if (v[i] == ss) { v[i] = v.back(); v.pop_back(); break; }
            ss->ref = nullptr;
        }
    }
    free_surface(ss);
}

void sadsurf_detach_all(x265hip_refpic* rp)
{
    std::lock_guard<std::mutex> g(g_ssLock);
    for (x265hip_sadsurf* ss : rp->surfaces)
        ss->ref = nullptr;
    rp->surfaces.clear();
}

} // namespace xh

using namespace xh;

extern "C" {

static x265hip_srcpic* srcpic_create(int place, int depth, int width, int height);

x265hip_srcpic* x265hip_srcpic_create(int depth, int width, int height)
{
    if (ensure_device()) return nullptr;
    int dev = 0;
    (void)hipGetDevice(&dev);
    return srcpic_create(place_of_device(dev), depth, width, height);
}

x265hip_srcpic* x265hip_srcpic_create_at(int place, int depth, int width, int height)
{
    const int dev = place >= 0 ? place_device(place) : -1;
    if (dev < 0)
    {
        set_error(X265HIP_EINVAL, "x265hip_srcpic_create_at: no place %d (x265hip_places)", place);
        return nullptr;
    }
    int cur = 0;
    const bool had = hipGetDevice(&cur) == hipSuccess;
    if (hipSetDevice(dev) != hipSuccess) { set_error(X265HIP_EHIP, "x265hip_srcpic_create_at: hipSetDevice(%d)", dev); return nullptr; }
    x265hip_srcpic* sp = srcpic_create(place, depth, width, height);
    if (had) (void)hipSetDevice(cur);
    return sp;
}

static x265hip_srcpic* srcpic_create(int place, int depth, int width, int height)
{
    if (!valid_depth(depth) || width < 16 || height < 16 || width > 16384 || height > 16384)
    {
        set_error(X265HIP_EINVAL, "x265hip_srcpic_create: depth %d %dx%d", depth, width, height);
        return nullptr;
    }
    x265hip_srcpic* sp = new x265hip_srcpic;
    sp->depth = depth; sp->w = width; sp->h = height;
    sp->pitch = ((int64_t)width * (depth == 8 ? 1 : 2) + 255) & ~(int64_t)255;
    (void)hipGetDevice(&sp->device);
    sp->place = place;
    const size_t bytes = (size_t)sp->pitch * height;
    // no stream of its own (3.9 ms each, ~40 source pictures alive in a 1080p encoder): the upload leases one
    if (hipMalloc((void**)&sp->dLuma, bytes + 256) != hipSuccess ||
        pinned_alloc((void**)&sp->hStage, bytes) != hipSuccess)
    {
        set_error(X265HIP_ENOMEM, "x265hip_srcpic_create: %zu bytes", bytes);
        x265hip_srcpic_destroy(sp);
        return nullptr;
    }
    return sp;
}

int x265hip_srcpic_upload(x265hip_srcpic* sp, const void* hostLuma, int64_t stride)
{
    if (!sp || !hostLuma || stride < sp->w) return set_error(X265HIP_EINVAL, "x265hip_srcpic_upload: bad arguments");
    int cur = 0;
    const bool had = hipGetDevice(&cur) == hipSuccess;
    int e = check_hip(hipSetDevice(sp->device), "hipSetDevice");
    if (e) return e;
    const size_t B = sp->depth == 8 ? 1 : 2;                 // `stride` counts samples, like every stride of the ABI
    for (int y = 0; y < sp->h; y++)
        memcpy(sp->hStage + (size_t)y * sp->pitch, (const char*)hostLuma + (size_t)y * stride * B, sp->w * B);
    hipStream_t st = stream_lease(sp->device);
    if (!st) e = set_error(X265HIP_EHIP, "x265hip_srcpic_upload: no stream");
    else
    {
        if (!(e = check_hip(hipMemcpyAsync(sp->dLuma, sp->hStage, (size_t)sp->pitch * sp->h, hipMemcpyHostToDevice, st), "srcpic h2d")))
            e = check_hip(hipStreamSynchronize(st), "srcpic sync");
        stream_return(sp->device, st);
    }
    if (had && cur != sp->device) (void)hipSetDevice(cur);
    return e;
}

void x265hip_srcpic_destroy(x265hip_srcpic* sp)
{
    if (sp) srcpic_unref(sp);
}

} // extern "C"
static void srcpic_unref(x265hip_srcpic* sp)
{
    if (sp->refs.fetch_sub(1) != 1) return;
    int cur = 0;
    const bool had = hipGetDevice(&cur) == hipSuccess;
    (void)hipSetDevice(sp->device);
    if (sp->dLuma) (void)device_free(sp->dLuma);
    if (sp->hStage) (void)pinned_free(sp->hStage);
    if (had && cur != sp->device) (void)hipSetDevice(cur);
    delete sp;
}
extern "C" {

x265hip_sadsurf* x265hip_sadsurf_attach(x265hip_srcpic* src, x265hip_refpic* ref, int searchRange, int lambda20)
{
    return x265hip_sadsurf_attach_levels(src, ref, searchRange, lambda20, 14);
}

x265hip_sadsurf* x265hip_sadsurf_attach_levels(x265hip_srcpic* src, x265hip_refpic* ref, int searchRange, int lambda20, int levels)
{
    if (ensure_device()) return nullptr;
    if (!src || !ref || src->depth != ref->depth || src->w != ref->picW || src->h != ref->picH || searchRange < 8 || searchRange > (src->depth == 8 ? 32 : 16) || (searchRange & 3) ||
        lambda20 < 0 || lambda20 > (1 << 20) || ref->marginX < searchRange + 8 || ref->marginY < searchRange || (levels & ~31) || (levels & 14) != 14 || ((levels & 1) && src->depth != 8))
    {
        set_error(X265HIP_EINVAL, "x265hip_sadsurf_attach: pictures do not match, or range %d (8..32 for 8-bit pictures, 8..16 for 16-bit ones) / lambda %d / margins out of bounds", searchRange, lambda20);
        return nullptr;
    }
    x265hip_sadsurf* ss = new x265hip_sadsurf;
    ss->src = src; ss->ref = ref; ss->S = searchRange; ss->lambda20 = lambda20; ss->levels = levels;
    layout_for(src->w, src->h, src->depth, levels, ss->lay);
    ss->bytes = (size_t)ss->lay.pitch * ss->lay.ctuRows;
    {
        std::lock_guard<std::mutex> g(g_poolLock);
        auto it = g_pool.find({ ss->bytes, src->device });
        if (it != g_pool.end()) { ss->dBuf = it->second.d; ss->hBuf = it->second.h; g_pool.erase(it); }
    }
    if (!ss->dBuf)
    {
        // the table is written by the kernel that runs where the SOURCE picture lives
        int cur = 0;
        const bool had = hipGetDevice(&cur) == hipSuccess;
        (void)hipSetDevice(src->device);
        const bool ok = hipMalloc((void**)&ss->dBuf, ss->bytes) == hipSuccess && pinned_alloc((void**)&ss->hBuf, ss->bytes) == hipSuccess;
        if (had && cur != src->device) (void)hipSetDevice(cur);
        if (!ok)
        {
            set_error(X265HIP_ENOMEM, "x265hip_sadsurf_attach: %zu bytes", ss->bytes);
            if (ss->dBuf) (void)device_free(ss->dBuf);
            delete ss;
            return nullptr;
        }
    }
    src->refs.fetch_add(1);
    memset(&ss->view, 0, sizeof(ss->view));
    for (int l = 0; l < 4; l++)
    {
        x265hip_sadsurf_level& v = ss->view.level[l];
        if (!(levels >> l & 1))
            continue;                             // origin == NULL: level not built
        v.blocksX = ss->lay.blocksX[l]; v.blocksY = ss->lay.blocksY[l]; v.entryBytes = ss->lay.entryBytes[l]; v.blocksPerCtuRow = ss->lay.per[l];
        v.origin = (const int16_t*)(ss->hBuf + ss->lay.originOff[l]);
        v.table = ss->hBuf + ss->lay.tableOff[l];
        v.subpel = ss->lay.subpelOff[l] ? (const uint32_t*)(ss->hBuf + ss->lay.subpelOff[l]) : nullptr;
    }
    ss->view.ctuRowPitch = ss->lay.pitch;
    ss->view.ctuRowsReady = reinterpret_cast<const int*>(&ss->ctuRowsReady);
    {
        std::lock_guard<std::mutex> g(g_ssLock);
        ref->surfaces.push_back(ss);
    }
    g_statAttached++;
    ss->workerPlace = ref->place;                 // the release must queue behind the attach: same worker
    RefWorker::worker(ref->place).push(RefJob{ ref, 0, ref->epoch.load(), 1, ss });
    return ss;
}

const x265hip_sadsurf_view* x265hip_sadsurf_get_view(x265hip_sadsurf* ss) { return ss ? &ss->view : nullptr; }

void x265hip_sadsurf_release(x265hip_sadsurf* ss)
{
    if (!ss) return;
    ss->released = true;
    RefWorker::worker(ss->workerPlace).push(RefJob{ nullptr, 0, 0, 2, ss });
}

int x265hip_peer_stats(uint64_t* replicas, uint64_t* bands, uint64_t* bytes)
{
    if (replicas) *replicas = g_statReplicas.load();
    if (bands) *bands = g_statPeerBands.load();
    if (bytes) *bytes = g_statPeerBytes.load();
    return X265HIP_OK;
}

int x265hip_sadsurf_stats(uint64_t* attached, uint64_t* ctuRows, uint64_t* launches, uint64_t* kernelNs)
{
    if (kernelNs) (void)x265hip_device_time(X265HIP_CLK_SADSURF, nullptr, kernelNs, nullptr);
    if (launches) *launches = g_statLaunches.load();
    if (attached) *attached = g_statAttached.load();
    if (ctuRows) *ctuRows = g_statRows.load();
    return X265HIP_OK;
}

} // extern "C"
