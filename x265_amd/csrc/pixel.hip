// pixel.hip — batched pixel comparisons (SAD / SAD_x3 / SAD_x4 / SATD / SA8D / SSE / psyCost) and block arithmetic.
//
// Reference semantics (bit-exact): source/common/pixel.cpp — sad :40, sad_x3 :74, sad_x4 :96, sse :167,
// satd_4x4 :210, satd_8x4 :239, _sa8d_8x8 :299, sa8d_16x16 :341, sa8d8/sa8d16 :352/:367, psyCost_pp :726,
// ssd_s :379, sub_ps :815, add_ps :829, addAvg :842, pixelavg_pp :545, blockcopy_* :759-812; ipfilter.cpp:40 (p2s).
//
// Mapping (all comparison kernels): the unit of lane work is one 4x4 TILE of a block.  A W x H block has
// (W/4)(H/4) tiles; T = min(64, pow2(tiles)) lanes cooperate on one block, so one wave64 carries 64/T blocks at a
// time (16 8x8 blocks, 4 16x16 blocks, one 32x32 block; a 64x64 block takes 4 passes).  A lane loads its 4 rows
// with one unaligned dword (u8) or dwordx2 (u16) each, does the 4x4 Hadamard / abs-diff in registers, and the
// per-block value is a DPP/shuffle sum over the T lanes.  For the 8x8 Hadamard (SA8D, psyCost) the four 4x4
// quadrants of an 8x8 sit in one DPP quad: H8 = H2 (x) H4, so the 8x8 transform is the 4x4 transform of each
// quadrant followed by a 2x2 butterfly ACROSS the quad (v_*_dpp quad_perm), no LDS involved.
#include "common.h"
#include "tiles.h"
#include "internal.h"

namespace xh {

enum { OP_SAD = X265HIP_CMP_SAD, OP_SATD = X265HIP_CMP_SATD, OP_SA8D = X265HIP_CMP_SA8D, OP_SA8D8 = X265HIP_CMP_SA8D8,
       OP_PSY = X265HIP_CMP_PSY, OP_SSE = 16 };

// tile index -> pixel coordinates inside the block
template <int OP>
__device__ __forceinline__ void tile_xy(int t, int w, int& x, int& y)
{
    if (OP == OP_SA8D)
    {
        // 16 consecutive tiles = one 16x16 (4 quads = four 8x8); needed because sa8d_16x16 rounds once per 16x16
        int n16x = w >= 16 ? (w >> 4) : 1;
        int b16 = t >> 4, b8 = (t >> 2) & 3, q = t & 3;
        x = (b16 % n16x) * 16 + (b8 & 1) * 8 + (q & 1) * 4;
        y = (b16 / n16x) * 16 + (b8 >> 1) * 8 + (q >> 1) * 4;
    }
    else if (OP == OP_SA8D8 || OP == OP_PSY)
    {
        int n8x = w >= 8 ? (w >> 3) : 1;
        int b8 = t >> 2, q = t & 3;
        x = (b8 % n8x) * 8 + (q & 1) * 4;
        y = (b8 / n8x) * 8 + (q >> 1) * 4;
    }
    else
    {
        int tilesX = w >> 2;
        y = (t / tilesX) * 4;
        x = (t % tilesX) * 4;
    }
}

// ---- the comparison kernel ------------------------------------------------------------------------------------
// divA: jobs per A offset (1 normally, K for sad_x3/x4 where K candidates share one fenc block)
template <typename P, int OP>
__global__ __launch_bounds__(256) void pixcmp_kernel(const P* __restrict__ A, int64_t sA, const P* __restrict__ B, int64_t sB,
                                                     const int32_t* __restrict__ offA, const int32_t* __restrict__ offB,
                                                     int n, int w, int h, int divA, int32_t* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int tiles = (w >> 2) * (h >> 2);
    const int T = tiles >= 64 ? 64 : pow2_ceil(tiles);
    const int bpw = 64 / T;
    const int iters = (tiles + T - 1) / T;
    const int sub = lane & (T - 1);
    const int wavesTotal = gridDim.x * (blockDim.x >> 6);
    const int gwave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const bool small4 = (w == 4 && h == 4);   // cu[BLOCK_4x4].sa8d == satd_4x4 (pixel.cpp:1163); psy 4x4 uses satd too

    for (long long job0 = (long long)gwave * bpw; job0 < n; job0 += (long long)wavesTotal * bpw)
    {
        const long long job = job0 + lane / T;
        const bool jobOk = job < n;
        const long long jc = jobOk ? job : n - 1;
        const P* a = A + offA[jc / divA];
        const P* b = B + offB[jc];
        int acc = 0;
        for (int it = 0; it < iters; it++)
        {
            const int t = sub + it * T;
            const bool live = jobOk && t < tiles;
            int x = 0, y = 0;
            if (live)
                tile_xy<OP>(t, w, x, y);
            const P* ta = a + y * sA + x;
            const P* tb = b + y * sB + x;
            if (OP == OP_SAD)
            {
                acc += live ? tile_sad(ta, sA, tb, sB) : 0;
            }
            else if (OP == OP_SATD)
            {
                int m[16];
                tile_diff(ta, sA, tb, sB, m);
                hadamard4x4(m);
                // every 4x4 raw sum is even (all 16 coefficients share the parity of the tile's element sum), so
                // satd_8x4's (left + right) >> 1 (pixel.cpp:260) equals the sum of the per-tile >> 1 (pixel.cpp:235)
                acc += live ? (abs_sum16(m) >> 1) : 0;
            }
            else if (OP == OP_SA8D || OP == OP_SA8D8)
            {
                int m[16];
                tile_diff(ta, sA, tb, sB, m);
                if (!live)
                {
#pragma unroll
                    for (int i = 0; i < 16; i++) m[i] = 0;
                }
                hadamard4x4(m);
                if (small4)
                    acc += abs_sum16(m) >> 1;
                else
                {
                    int raw8 = quad_sa8d_raw(m, lane);
                    if (OP == OP_SA8D && w >= 16)
                    {
                        int s16 = group_sum((lane & 3) == 0 ? raw8 : 0, 16);
                        acc += (lane & 15) == 0 ? ((s16 + 2) >> 2) : 0;
                    }
                    else
                        acc += (lane & 3) == 0 ? ((raw8 + 2) >> 2) : 0;
                }
            }
            else if (OP == OP_PSY)
            {
                int ms[16], mr[16];
                tile_load(ta, sA, ms);
                tile_load(tb, sB, mr);
                if (!live)
                {
#pragma unroll
                    for (int i = 0; i < 16; i++) { ms[i] = 0; mr[i] = 0; }
                }
                int sumS = 0, sumR = 0;
#pragma unroll
                for (int i = 0; i < 16; i++) { sumS += ms[i]; sumR += mr[i]; }
                hadamard4x4(ms);
                hadamard4x4(mr);
                if (small4)
                {
                    int es = (abs_sum16(ms) >> 1) - (sumS >> 2);
                    int er = (abs_sum16(mr) >> 1) - (sumR >> 2);
                    acc += iabs(es - er);
                }
                else
                {
                    int rawS = quad_sa8d_raw(ms, lane), rawR = quad_sa8d_raw(mr, lane);
                    int sadS = quad_sum(sumS), sadR = quad_sum(sumR);
                    int es = ((rawS + 2) >> 2) - (sadS >> 2);
                    int er = ((rawR + 2) >> 2) - (sadR >> 2);
                    acc += (lane & 3) == 0 ? iabs(es - er) : 0;
                }
            }
        }
        acc = group_sum(acc, T);
        if (jobOk && sub == 0)
            out[job] = acc;
    }
}

// sum of squared differences; PA/PB are pixel or int16.  hasB == false: ssd_s (sum of squares of A)
template <typename PA, typename PB>
__global__ __launch_bounds__(256) void sse_kernel(const PA* __restrict__ A, int64_t sA, const PB* __restrict__ B, int64_t sB,
                                                  const int32_t* __restrict__ offA, const int32_t* __restrict__ offB,
                                                  int n, int w, int h, bool hasB, uint64_t* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int tiles = (w >> 2) * (h >> 2);
    const int T = tiles >= 64 ? 64 : pow2_ceil(tiles);
    const int bpw = 64 / T;
    const int iters = (tiles + T - 1) / T;
    const int sub = lane & (T - 1);
    const int tilesX = w >> 2;
    const int wavesTotal = gridDim.x * (blockDim.x >> 6);
    const int gwave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    for (long long job0 = (long long)gwave * bpw; job0 < n; job0 += (long long)wavesTotal * bpw)
    {
        const long long job = job0 + lane / T;
        const bool jobOk = job < n;
        const long long jc = jobOk ? job : n - 1;
        const PA* a = A + offA[jc];
        const PB* b = hasB ? B + offB[jc] : nullptr;
        unsigned long long acc = 0;
        for (int it = 0; it < iters; it++)
        {
            const int t = sub + it * T;
            const bool live = jobOk && t < tiles;
            const int y = live ? (t / tilesX) * 4 : 0, x = live ? (t % tilesX) * 4 : 0;
            // u8: a tile sums to <= 16 * 255^2, u32 is plenty; 16-bit types can reach 65535^2 per element: widen per element
            typename std::conditional<sizeof(PA) == 1, unsigned, unsigned long long>::type s = 0;
#pragma unroll
            for (int r = 0; r < 4; r++)
            {
                int va[4], vb[4] = { 0, 0, 0, 0 };
                load4(a + (y + r) * sA + x, va);
                if (hasB)
                    load4(b + (y + r) * sB + x, vb);
#pragma unroll
                for (int c = 0; c < 4; c++)
                {
                    const unsigned dlt = (unsigned)(va[c] - vb[c]);
                    s += dlt * dlt;               // (-d)^2 == d^2 mod 2^32 and d^2 < 2^32
                }
            }
            acc += live ? s : 0;
        }
        acc = group_sum64(acc, T);
        if (jobOk && sub == 0)
            out[job] = acc;
    }
}

// ---- block arithmetic (elementwise; V elements per lane) ---------------------------------------------------------
enum { EW_SUB_PS, EW_ADD_PS, EW_ADDAVG, EW_PIXELAVG, EW_COPY_PP, EW_COPY_SP, EW_COPY_PS, EW_COPY_SS, EW_P2S };

template <typename T>
__device__ __forceinline__ int ld_elem(const T* p) { return (int)*p; }

// D = f(S0, S1): element types are template parameters; one lane produces V adjacent elements of one row
template <int EW, typename TD, typename T0, typename T1, int V>
__global__ __launch_bounds__(256) void ew_kernel(TD* __restrict__ D, int64_t sD, const T0* __restrict__ S0, int64_t s0,
                                                 const T1* __restrict__ S1, int64_t s1,
                                                 const int32_t* __restrict__ offD, const int32_t* __restrict__ off0,
                                                 const int32_t* __restrict__ off1, int n, int w, int h, int depth)
{
    const int vecX = w / V;
    const int per = vecX * h;
    const long long total = (long long)n * per;
    const int maxVal = (1 << depth) - 1;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x)
    {
        const int job = (int)(idx / per);
        const int p = (int)(idx - (long long)job * per);
        const int y = p / vecX, x = (p - y * vecX) * V;
        TD* d = D + offD[job] + y * sD + x;
        const T0* a = S0 + off0[job] + y * s0 + x;
        const T1* b = (EW == EW_SUB_PS || EW == EW_ADD_PS || EW == EW_ADDAVG || EW == EW_PIXELAVG) ? S1 + off1[job] + y * s1 + x : nullptr;
#pragma unroll
        for (int i = 0; i < V; i++)
        {
            int va = ld_elem(a + i), r;
            if (EW == EW_SUB_PS) r = va - ld_elem(b + i);
            else if (EW == EW_ADD_PS) r = clip3i(0, maxVal, va + ld_elem(b + i));
            else if (EW == EW_ADDAVG)
            {
                const int shiftNum = 14 + 1 - depth;                      // IF_INTERNAL_PREC + 1 - X265_DEPTH (pixel.cpp:845)
                const int offset = (1 << (shiftNum - 1)) + 2 * 8192;      // + 2 * IF_INTERNAL_OFFS
                r = clip3i(0, maxVal, (va + ld_elem(b + i) + offset) >> shiftNum);
            }
            else if (EW == EW_PIXELAVG) r = (va + ld_elem(b + i) + 1) >> 1;
            else if (EW == EW_P2S) r = (int)(int16_t)((int16_t)(va << (14 - depth)) - (int16_t)8192);
            else r = va;                                                   // copies (sp truncates to pixel, ps/ss widen/keep)
            d[i] = (TD)r;
        }
    }
}

template <typename P>
static int launch_pixcmp(int op, int w, int h, const void* A, int64_t sA, const void* B, int64_t sB, const int32_t* offA,
                         const int32_t* offB, int n, int divA, int32_t* out, hipStream_t st)
{
    const int tiles = (w >> 2) * (h >> 2);
    const int T = tiles >= 64 ? 64 : pow2_ceil(tiles);
    const long long waves = ((long long)n + (64 / T) - 1) / (64 / T);
    dim3 grid(grid_for((waves + 3) / 4)), block(256);
    const P* a = (const P*)A;
    const P* b = (const P*)B;
    switch (op)
    {
    case OP_SAD:   hipLaunchKernelGGL((pixcmp_kernel<P, OP_SAD>), grid, block, 0, st, a, sA, b, sB, offA, offB, n, w, h, divA, out); break;
    case OP_SATD:  hipLaunchKernelGGL((pixcmp_kernel<P, OP_SATD>), grid, block, 0, st, a, sA, b, sB, offA, offB, n, w, h, divA, out); break;
    case OP_SA8D:  hipLaunchKernelGGL((pixcmp_kernel<P, OP_SA8D>), grid, block, 0, st, a, sA, b, sB, offA, offB, n, w, h, divA, out); break;
    case OP_SA8D8: hipLaunchKernelGGL((pixcmp_kernel<P, OP_SA8D8>), grid, block, 0, st, a, sA, b, sB, offA, offB, n, w, h, divA, out); break;
    case OP_PSY:   hipLaunchKernelGGL((pixcmp_kernel<P, OP_PSY>), grid, block, 0, st, a, sA, b, sB, offA, offB, n, w, h, divA, out); break;
    default: return set_error(X265HIP_EINVAL, "pixcmp: unknown op %d", op);
    }
    XH_LAUNCH_CHECK("pixcmp_kernel");
    return X265HIP_OK;
}

template <typename PA, typename PB>
static int launch_sse(int w, int h, const void* A, int64_t sA, const void* B, int64_t sB, const int32_t* offA,
                      const int32_t* offB, int n, uint64_t* out, hipStream_t st)
{
    const int tiles = (w >> 2) * (h >> 2);
    const int T = tiles >= 64 ? 64 : pow2_ceil(tiles);
    const long long waves = ((long long)n + (64 / T) - 1) / (64 / T);
    dim3 grid(grid_for((waves + 3) / 4)), block(256);
    hipLaunchKernelGGL((sse_kernel<PA, PB>), grid, block, 0, st, (const PA*)A, sA, (const PB*)B, sB, offA, offB, n, w, h,
                       B != nullptr, out);
    XH_LAUNCH_CHECK("sse_kernel");
    return X265HIP_OK;
}

template <int EW, typename TD, typename T0, typename T1>
static int launch_ew(int w, int h, int depth, void* D, int64_t sD, const void* S0, int64_t s0, const void* S1, int64_t s1,
                     const int32_t* offD, const int32_t* off0, const int32_t* off1, int n, hipStream_t st)
{
    const int V = (w & 3) ? 2 : 4;
    const long long total = (long long)n * (w / V) * h;
    dim3 grid(grid_for((total + 255) / 256)), block(256);
    if (V == 4)
        hipLaunchKernelGGL((ew_kernel<EW, TD, T0, T1, 4>), grid, block, 0, st, (TD*)D, sD, (const T0*)S0, s0, (const T1*)S1, s1,
                           offD, off0, off1, n, w, h, depth);
    else
        hipLaunchKernelGGL((ew_kernel<EW, TD, T0, T1, 2>), grid, block, 0, st, (TD*)D, sD, (const T0*)S0, s0, (const T1*)S1, s1,
                           offD, off0, off1, n, w, h, depth);
    XH_LAUNCH_CHECK("ew_kernel");
    return X265HIP_OK;
}

static bool cmp_shape_ok(int op, int w, int h)
{
    if (!valid_block(w, h) || (w & 3) || (h & 3))
        return false;
    if (op == OP_SA8D)
        return w == h && (w == 4 || w == 8 || w == 16 || w == 32 || w == 64);
    if (op == OP_PSY)
        return w == h && (w == 4 || !(w & 7));
    if (op == OP_SA8D8)
        return !(w & 7) && !(h & 7);
    return true;
}

} // namespace xh

using namespace xh;

#define XH_ARGS_CHECK(cond, ...) do { if (!(cond)) return set_error(X265HIP_EINVAL, __VA_ARGS__); } while (0)

extern "C" {

int x265hip_pixcmp_batch(int op, int depth, int w, int h, const void* planeA, int64_t strideA, const void* planeB,
                         int64_t strideB, const int32_t* offA, const int32_t* offB, int n, int32_t* out, void* stream)
{
    XH_CHECK_DEV();
    XH_ARGS_CHECK(valid_depth(depth), "pixcmp: depth %d", depth);
    XH_ARGS_CHECK(op >= 0 && op <= OP_PSY && cmp_shape_ok(op, w, h), "pixcmp: op %d does not take %dx%d", op, w, h);
    XH_ARGS_CHECK(n >= 0, "pixcmp: n %d", n);
    if (!n) return X265HIP_OK;
    return depth == 8 ? launch_pixcmp<uint8_t>(op, w, h, planeA, strideA, planeB, strideB, offA, offB, n, 1, out, as_stream(stream))
                      : launch_pixcmp<uint16_t>(op, w, h, planeA, strideA, planeB, strideB, offA, offB, n, 1, out, as_stream(stream));
}

int x265hip_sad_xn_batch(int K, int depth, int w, int h, const void* fenc, int64_t strideF, const void* ref, int64_t strideR,
                         const int32_t* offF, const int32_t* offRef, int n, int32_t* out, void* stream)
{
    XH_CHECK_DEV();
    XH_ARGS_CHECK(valid_depth(depth), "sad_xn: depth %d", depth);
    XH_ARGS_CHECK(K == 3 || K == 4, "sad_xn: K %d", K);
    XH_ARGS_CHECK(cmp_shape_ok(OP_SAD, w, h) && n >= 0, "sad_xn: %dx%d n %d", w, h, n);
    if (!n) return X265HIP_OK;
    return depth == 8 ? launch_pixcmp<uint8_t>(OP_SAD, w, h, fenc, strideF, ref, strideR, offF, offRef, n * K, K, out, as_stream(stream))
                      : launch_pixcmp<uint16_t>(OP_SAD, w, h, fenc, strideF, ref, strideR, offF, offRef, n * K, K, out, as_stream(stream));
}

int x265hip_sse_pp_batch(int depth, int w, int h, const void* planeA, int64_t strideA, const void* planeB, int64_t strideB,
                         const int32_t* offA, const int32_t* offB, int n, uint64_t* out, void* stream)
{
    XH_CHECK_DEV();
    XH_ARGS_CHECK(valid_depth(depth) && cmp_shape_ok(OP_SAD, w, h) && n >= 0, "sse_pp: depth %d %dx%d n %d", depth, w, h, n);
    if (!n) return X265HIP_OK;
    return depth == 8 ? launch_sse<uint8_t, uint8_t>(w, h, planeA, strideA, planeB, strideB, offA, offB, n, out, as_stream(stream))
                      : launch_sse<uint16_t, uint16_t>(w, h, planeA, strideA, planeB, strideB, offA, offB, n, out, as_stream(stream));
}

int x265hip_sse_ss_batch(int w, int h, const int16_t* planeA, int64_t strideA, const int16_t* planeB, int64_t strideB,
                         const int32_t* offA, const int32_t* offB, int n, uint64_t* out, void* stream)
{
    XH_CHECK_DEV();
    XH_ARGS_CHECK(cmp_shape_ok(OP_SAD, w, h) && n >= 0, "sse_ss: %dx%d n %d", w, h, n);
    if (!n) return X265HIP_OK;
    return launch_sse<int16_t, int16_t>(w, h, planeA, strideA, planeB, strideB, offA, offB ? offB : offA, n, out, as_stream(stream));
}

int x265hip_sub_ps_batch(int depth, int w, int h, int16_t* resi, int64_t strideD, const void* planeA, int64_t strideA,
                         const void* planeB, int64_t strideB, const int32_t* offD, const int32_t* offA, const int32_t* offB,
                         int n, void* stream)
{
    XH_CHECK_DEV();
    XH_ARGS_CHECK(valid_depth(depth) && valid_block(w, h) && n >= 0, "sub_ps: depth %d %dx%d n %d", depth, w, h, n);
    if (!n) return X265HIP_OK;
    return depth == 8 ? launch_ew<EW_SUB_PS, int16_t, uint8_t, uint8_t>(w, h, depth, resi, strideD, planeA, strideA, planeB, strideB, offD, offA, offB, n, as_stream(stream))
                      : launch_ew<EW_SUB_PS, int16_t, uint16_t, uint16_t>(w, h, depth, resi, strideD, planeA, strideA, planeB, strideB, offD, offA, offB, n, as_stream(stream));
}

int x265hip_add_ps_batch(int depth, int w, int h, void* recon, int64_t strideD, const void* pred, int64_t strideA,
                         const int16_t* resi, int64_t strideR, const int32_t* offD, const int32_t* offA, const int32_t* offR,
                         int n, void* stream)
{
    XH_CHECK_DEV();
    XH_ARGS_CHECK(valid_depth(depth) && valid_block(w, h) && n >= 0, "add_ps: depth %d %dx%d n %d", depth, w, h, n);
    if (!n) return X265HIP_OK;
    return depth == 8 ? launch_ew<EW_ADD_PS, uint8_t, uint8_t, int16_t>(w, h, depth, recon, strideD, pred, strideA, resi, strideR, offD, offA, offR, n, as_stream(stream))
                      : launch_ew<EW_ADD_PS, uint16_t, uint16_t, int16_t>(w, h, depth, recon, strideD, pred, strideA, resi, strideR, offD, offA, offR, n, as_stream(stream));
}

int x265hip_addavg_batch(int depth, int w, int h, const int16_t* src0, int64_t stride0, const int16_t* src1, int64_t stride1,
                         void* dst, int64_t strideD, const int32_t* off0, const int32_t* off1, const int32_t* offD,
                         int n, void* stream)
{
    XH_CHECK_DEV();
    XH_ARGS_CHECK(valid_depth(depth) && valid_block(w, h) && n >= 0, "addAvg: depth %d %dx%d n %d", depth, w, h, n);
    if (!n) return X265HIP_OK;
    return depth == 8 ? launch_ew<EW_ADDAVG, uint8_t, int16_t, int16_t>(w, h, depth, dst, strideD, src0, stride0, src1, stride1, offD, off0, off1, n, as_stream(stream))
                      : launch_ew<EW_ADDAVG, uint16_t, int16_t, int16_t>(w, h, depth, dst, strideD, src0, stride0, src1, stride1, offD, off0, off1, n, as_stream(stream));
}

int x265hip_pixelavg_pp_batch(int depth, int w, int h, void* dst, int64_t strideD, const void* src0, int64_t stride0,
                              const void* src1, int64_t stride1, const int32_t* offD, const int32_t* off0, const int32_t* off1,
                              int n, void* stream)
{
    XH_CHECK_DEV();
    XH_ARGS_CHECK(valid_depth(depth) && valid_block(w, h) && n >= 0, "pixelavg: depth %d %dx%d n %d", depth, w, h, n);
    if (!n) return X265HIP_OK;
    return depth == 8 ? launch_ew<EW_PIXELAVG, uint8_t, uint8_t, uint8_t>(w, h, depth, dst, strideD, src0, stride0, src1, stride1, offD, off0, off1, n, as_stream(stream))
                      : launch_ew<EW_PIXELAVG, uint16_t, uint16_t, uint16_t>(w, h, depth, dst, strideD, src0, stride0, src1, stride1, offD, off0, off1, n, as_stream(stream));
}

int x265hip_copy_batch(int kind, int depth, int w, int h, void* dst, int64_t strideD, const void* src, int64_t strideS,
                       const int32_t* offD, const int32_t* offS, int n, void* stream)
{
    XH_CHECK_DEV();
    XH_ARGS_CHECK(valid_depth(depth) && valid_block(w, h) && n >= 0 && kind >= 0 && kind <= 3, "copy: kind %d depth %d %dx%d n %d", kind, depth, w, h, n);
    if (!n) return X265HIP_OK;
    hipStream_t st = as_stream(stream);
    const bool b8 = depth == 8;
    switch (kind)
    {
    case 0: return b8 ? launch_ew<EW_COPY_PP, uint8_t, uint8_t, uint8_t>(w, h, depth, dst, strideD, src, strideS, nullptr, 0, offD, offS, nullptr, n, st)
                      : launch_ew<EW_COPY_PP, uint16_t, uint16_t, uint16_t>(w, h, depth, dst, strideD, src, strideS, nullptr, 0, offD, offS, nullptr, n, st);
    case 1: return b8 ? launch_ew<EW_COPY_SP, uint8_t, int16_t, uint8_t>(w, h, depth, dst, strideD, src, strideS, nullptr, 0, offD, offS, nullptr, n, st)
                      : launch_ew<EW_COPY_SP, uint16_t, int16_t, uint16_t>(w, h, depth, dst, strideD, src, strideS, nullptr, 0, offD, offS, nullptr, n, st);
    case 2: return b8 ? launch_ew<EW_COPY_PS, int16_t, uint8_t, uint8_t>(w, h, depth, dst, strideD, src, strideS, nullptr, 0, offD, offS, nullptr, n, st)
                      : launch_ew<EW_COPY_PS, int16_t, uint16_t, uint16_t>(w, h, depth, dst, strideD, src, strideS, nullptr, 0, offD, offS, nullptr, n, st);
    default: return launch_ew<EW_COPY_SS, int16_t, int16_t, int16_t>(w, h, depth, dst, strideD, src, strideS, nullptr, 0, offD, offS, nullptr, n, st);
    }
}

int x265hip_p2s_batch(int depth, int w, int h, const void* src, int64_t strideS, int16_t* dst, int64_t strideD,
                      const int32_t* offS, const int32_t* offD, int n, void* stream)
{
    XH_CHECK_DEV();
    XH_ARGS_CHECK(valid_depth(depth) && valid_block(w, h) && n >= 0, "p2s: depth %d %dx%d n %d", depth, w, h, n);
    if (!n) return X265HIP_OK;
    return depth == 8 ? launch_ew<EW_P2S, int16_t, uint8_t, uint8_t>(w, h, depth, dst, strideD, src, strideS, nullptr, 0, offD, offS, nullptr, n, as_stream(stream))
                      : launch_ew<EW_P2S, int16_t, uint16_t, uint16_t>(w, h, depth, dst, strideD, src, strideS, nullptr, 0, offD, offS, nullptr, n, as_stream(stream));
}

} // extern "C"

// ---- picture border extension (reference: pixel.cpp:1027-1041 extendPicBorder / PicYuv margins picyuv.cpp:87-115) ----
namespace xh {
template <typename P>
__global__ __launch_bounds__(256) void extend_border_kernel(P* __restrict__ pic, int64_t stride, int picW, int picH, int marginX, int marginY)
{
    // only the margin pixels are visited: two full-width bands above / below, then the left / right strips of the picture rows
    const int fullW = picW + 2 * marginX;
    const long long bands = 2LL * marginY * fullW, total = bands + 2LL * marginX * picH;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    {
        int x, y;
        if (i < bands)
        {
            const int row = (int)(i / fullW);
            y = row < marginY ? row - marginY : picH + (row - marginY);
            x = (int)(i % fullW) - marginX;
        }
        else
        {
            const long long k = i - bands;
            const int xx = (int)(k % (2 * marginX));
            y = (int)(k / (2 * marginX));
            x = xx < marginX ? xx - marginX : picW + (xx - marginX);
        }
        const int sy = y < 0 ? 0 : (y >= picH ? picH - 1 : y), sx = x < 0 ? 0 : (x >= picW ? picW - 1 : x);
        pic[(int64_t)y * stride + x] = pic[(int64_t)sy * stride + sx];
    }
}
} // namespace xh

extern "C" int x265hip_extend_border(int depth, void* picOrigin, int64_t stride, int picW, int picH, int marginX, int marginY, void* stream)
{
    XH_CHECK_DEV();
    if (!valid_depth(depth) || picW < 1 || picH < 1 || marginX < 0 || marginY < 0)
        return set_error(X265HIP_EINVAL, "extend_border: depth %d pic %dx%d margins %d,%d", depth, picW, picH, marginX, marginY);
    const long long total = 2LL * marginY * (picW + 2 * marginX) + 2LL * marginX * picH;
    if (total <= 0) return X265HIP_OK;
    dim3 grid(grid_for((total + 255) / 256)), block(256);
    if (depth == 8)
        hipLaunchKernelGGL((extend_border_kernel<uint8_t>), grid, block, 0, as_stream(stream), (uint8_t*)picOrigin, stride, picW, picH, marginX, marginY);
    else
        hipLaunchKernelGGL((extend_border_kernel<uint16_t>), grid, block, 0, as_stream(stream), (uint16_t*)picOrigin, stride, picW, picH, marginX, marginY);
    XH_LAUNCH_CHECK("extend_border_kernel");
    return X265HIP_OK;
}

// ---- sa8d of several CU sizes in one launch (frame pass step 4) -------------------------------------------------------------
namespace xh {
struct Sa8dLevels { Sa8dLevel l[4]; int firstBlock[5]; };

template <typename P>
__global__ __launch_bounds__(256) void sa8d_levels_kernel(const P* __restrict__ A, int64_t sA, const P* __restrict__ B, int64_t sB, Sa8dLevels lv)
{
    int li = 0;
#pragma unroll
    for (int i = 1; i < 4; i++)
        if ((int)blockIdx.x >= lv.firstBlock[i]) li = i;
    const Sa8dLevel L = lv.l[li];
    const int w = L.size;
    const int lane = threadIdx.x & 63;
    const int tiles = (w >> 2) * (w >> 2);
    const int T = tiles >= 64 ? 64 : tiles;                    // 4 (8x8), 16, 64, 64
    const int bpw = 64 / T, iters = tiles / T, sub = lane & (T - 1);
    const int wave = ((int)blockIdx.x - lv.firstBlock[li]) * 4 + (threadIdx.x >> 6);
    const long long job = (long long)wave * bpw + lane / T;
    const bool jobOk = job < L.n;
    const long long jc = jobOk ? job : L.n - 1;
    const P* a = A + L.offA[jc];
    const P* b = B + L.offB[jc];
    int acc = 0;
    for (int it = 0; it < iters; it++)
    {
        const int t = sub + it * T;
        int x, y;
        const int n16x = w >= 16 ? (w >> 4) : 1;
        const int b16 = t >> 4, b8 = (t >> 2) & 3, q = t & 3;
        x = (b16 % n16x) * 16 + (b8 & 1) * 8 + (q & 1) * 4;
        y = (b16 / n16x) * 16 + (b8 >> 1) * 8 + (q >> 1) * 4;
        int m[16];
        tile_diff(a + y * sA + x, sA, b + y * sB + x, sB, m);
        if (!jobOk)
        {
#pragma unroll
            for (int i = 0; i < 16; i++) m[i] = 0;
        }
        hadamard4x4(m);
        const int raw8 = quad_sa8d_raw(m, lane);
        if (w >= 16)
        {
            const int s16 = group_sum((lane & 3) == 0 ? raw8 : 0, 16);
            acc += (lane & 15) == 0 ? ((s16 + 2) >> 2) : 0;
        }
        else
            acc += (lane & 3) == 0 ? ((raw8 + 2) >> 2) : 0;
    }
    acc = group_sum(acc, T);
    if (jobOk && sub == 0)
        L.out[job] = acc;
}

int sa8d_levels(int depth, const void* planeA, int64_t strideA, const void* planeB, int64_t strideB, const Sa8dLevel* levels, int nLevels,
                hipStream_t st)
{
    Sa8dLevels lv{};
    int blocks = 0;
    for (int i = 0; i < 4; i++)
    {
        lv.firstBlock[i] = blocks;
        if (i < nLevels && levels[i].n > 0)
        {
            if (levels[i].size != 8 && levels[i].size != 16 && levels[i].size != 32 && levels[i].size != 64)
                return set_error(X265HIP_EINVAL, "sa8d_levels: size %d", levels[i].size);
            lv.l[i] = levels[i];
            const int tiles = (levels[i].size / 4) * (levels[i].size / 4), T = tiles >= 64 ? 64 : tiles;
            const long long waves = ((long long)levels[i].n + 64 / T - 1) / (64 / T);
            blocks += (int)((waves + 3) / 4);
        }
        else
            lv.l[i] = Sa8dLevel{ nullptr, nullptr, nullptr, 0, 8 };
    }
    lv.firstBlock[4] = blocks;
    // levels with n == 0 own no blocks; give them an unreachable first block so the level lookup skips them
    for (int i = 3; i >= 0; i--)
        if (lv.l[i].n == 0) lv.firstBlock[i] = lv.firstBlock[i + 1];
    if (!blocks) return X265HIP_OK;
    if (depth == 8)
        hipLaunchKernelGGL((sa8d_levels_kernel<uint8_t>), dim3(blocks), dim3(256), 0, st, (const uint8_t*)planeA, strideA, (const uint8_t*)planeB, strideB, lv);
    else
        hipLaunchKernelGGL((sa8d_levels_kernel<uint16_t>), dim3(blocks), dim3(256), 0, st, (const uint16_t*)planeA, strideA, (const uint16_t*)planeB, strideB, lv);
    XH_LAUNCH_CHECK("sa8d_levels_kernel");
    return X265HIP_OK;
}
} // namespace xh

// ---- sa8d pyramid: the four CU sizes' costs from ONE Hadamard pass over the picture (frame pass step 4) ------------------------------
// cu[].sa8d of a 16x16 block is ((sum of its four raw 8x8 Hadamard sums) + 2) >> 2 and the 32x32 / 64x64 costs are sums of 16x16 costs
// (pixel.cpp:336-376), so every level derives from the raw 8x8 sums: a lane takes a 4x4 tile, a DPP quad an 8x8, sixteen lanes a 16x16 block.  CU lists are the raster-ordered complete
// CUs of each size inside the picture (what framepass.hip builds), so indices are computed, not looked up.
namespace xh {
template <typename P>
__global__ __launch_bounds__(256) void sa8d_pyramid_kernel(const P* __restrict__ A, int64_t sA, const P* __restrict__ B, int64_t sB, int W, int H,
                                                           int32_t* __restrict__ o64, int32_t* __restrict__ o32, int32_t* __restrict__ o16, int32_t* __restrict__ o8)
{
    // one wave = one 64x64 region of the 64-aligned grid, four passes of one 32x32 quadrant each (16 lanes = a 16x16 block): the 32 and 64
    // totals are wave-local sums, so nothing needs zeroing or atomics
    const int lane = threadIdx.x & 63;
    const int rw = (W + 63) >> 6, rh = (H + 63) >> 6;
    const int region = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (region >= rw * rh)
        return;
    const int x64 = (region % rw) * 64, y64 = (region / rw) * 64;
    const int sub = lane & 15, b8 = sub >> 2, q = sub & 3, blk = lane >> 4;
    int raw8[4];
#pragma unroll
    for (int it = 0; it < 4; it++)
    {
        const int x16 = x64 + (it & 1) * 32 + (blk & 1) * 16, y16 = y64 + (it >> 1) * 32 + (blk >> 1) * 16;
        const int x = x16 + (b8 & 1) * 8 + (q & 1) * 4, y = y16 + (b8 >> 1) * 8 + (q >> 1) * 4;
        int m[16];
        // tiles past the picture edge read the margins (or, far outside, a clamped position); their sums are dropped below
        const int cx = x < W ? x : W - 4, cy = y < H ? y : H - 4;
        tile_diff(A + (int64_t)cy * sA + cx, sA, B + (int64_t)cy * sB + cx, sB, m);
        hadamard4x4(m);
        raw8[it] = quad_sa8d_raw(m, lane);
    }
    int s64 = 0;
#pragma unroll
    for (int it = 0; it < 4; it++)
    {
        const int x16 = x64 + (it & 1) * 32 + (blk & 1) * 16, y16 = y64 + (it >> 1) * 32 + (blk >> 1) * 16;
        const int x8 = x16 + (b8 & 1) * 8, y8 = y16 + (b8 >> 1) * 8;
        if (q == 0 && x8 + 8 <= W && y8 + 8 <= H)
            o8[(y8 >> 3) * (W >> 3) + (x8 >> 3)] = (raw8[it] + 2) >> 2;
        const int s16 = group_sum(q == 0 ? raw8[it] : 0, 16);
        const int c16 = (s16 + 2) >> 2;
        if (sub == 0 && x16 + 16 <= W && y16 + 16 <= H)
            o16[(y16 >> 4) * (W >> 4) + (x16 >> 4)] = c16;
        const int s32 = __builtin_amdgcn_readlane(c16, 0) + __builtin_amdgcn_readlane(c16, 16) + __builtin_amdgcn_readlane(c16, 32) + __builtin_amdgcn_readlane(c16, 48);
        const int x32 = x64 + (it & 1) * 32, y32 = y64 + (it >> 1) * 32;
        if (lane == 0 && x32 + 32 <= W && y32 + 32 <= H)
            o32[(y32 >> 5) * (W >> 5) + (x32 >> 5)] = s32;
        s64 += s32;
    }
    if (lane == 0 && x64 + 64 <= W && y64 + 64 <= H)
        o64[(y64 >> 6) * (W >> 6) + (x64 >> 6)] = s64;
}

int sa8d_pyramid(int depth, const void* planeA, int64_t strideA, const void* planeB, int64_t strideB, int W, int H, int32_t* const out[4], hipStream_t st)
{
    const int regions = ((W + 63) >> 6) * ((H + 63) >> 6);
    dim3 grid((unsigned)((regions + 3) / 4)), block(256);
    if (depth == 8)
        hipLaunchKernelGGL((sa8d_pyramid_kernel<uint8_t>), grid, block, 0, st, (const uint8_t*)planeA, strideA, (const uint8_t*)planeB, strideB, W, H, out[0], out[1], out[2], out[3]);
    else
        hipLaunchKernelGGL((sa8d_pyramid_kernel<uint16_t>), grid, block, 0, st, (const uint16_t*)planeA, strideA, (const uint16_t*)planeB, strideB, W, H, out[0], out[1], out[2], out[3]);
    XH_LAUNCH_CHECK("sa8d_pyramid_kernel");
    return X265HIP_OK;
}
} // namespace xh

// ---- border extension of up to four planes (Y, Cb, Cr; the four lowres hpel planes) in one launch (frame pass step 5) -----------------------------------
namespace xh {
struct BorderPlanes { void* pic[4]; int64_t stride[4]; int w[4], h[4], mx[4], my[4]; long long first[5]; };

template <typename P>
__global__ __launch_bounds__(256) void extend_border3_kernel(BorderPlanes bp)
{
    const long long total = bp.first[4];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    {
        const int pl = i >= bp.first[3] ? 3 : (i >= bp.first[2] ? 2 : (i >= bp.first[1] ? 1 : 0));
        const long long k0 = i - bp.first[pl];
        const int W = bp.w[pl], H = bp.h[pl], MX = bp.mx[pl], MY = bp.my[pl], fullW = W + 2 * MX;
        const long long bands = 2LL * MY * fullW;
        int x, y;
        if (k0 < bands)
        {
            const int row = (int)(k0 / fullW);
            y = row < MY ? row - MY : H + (row - MY);
            x = (int)(k0 % fullW) - MX;
        }
        else
        {
            const long long k = k0 - bands;
            const int xx = (int)(k % (2 * MX));
            y = (int)(k / (2 * MX));
            x = xx < MX ? xx - MX : W + (xx - MX);
        }
        const int sy = y < 0 ? 0 : (y >= H ? H - 1 : y), sx = x < 0 ? 0 : (x >= W ? W - 1 : x);
        P* pic = (P*)bp.pic[pl];
        pic[(int64_t)y * bp.stride[pl] + x] = pic[(int64_t)sy * bp.stride[pl] + sx];
    }
}

int extend_border_planes(int depth, int nPlanes, void* const* pics, const int64_t* strides, const int* w, const int* h, const int* mx, const int* my,
                         hipStream_t st)
{
    BorderPlanes bp{};
    long long total = 0;
    if (nPlanes < 1 || nPlanes > 4)
        return set_error(X265HIP_EINVAL, "extend_border_planes: %d planes", nPlanes);
    for (int i = 0; i < 4; i++)
    {
        bp.first[i] = total;
        if (i < nPlanes)
        {
            bp.pic[i] = pics[i]; bp.stride[i] = strides[i]; bp.w[i] = w[i]; bp.h[i] = h[i]; bp.mx[i] = mx[i]; bp.my[i] = my[i];
            total += 2LL * my[i] * (w[i] + 2 * mx[i]) + 2LL * mx[i] * h[i];
        }
        else
        {
            bp.w[i] = bp.h[i] = 1;
        }
    }
    bp.first[4] = total;
    for (int i = nPlanes; i < 4; i++) bp.first[i] = total;
    dim3 grid(grid_for((total + 255) / 256)), block(256);
    if (depth == 8) hipLaunchKernelGGL((extend_border3_kernel<uint8_t>), grid, block, 0, st, bp);
    else hipLaunchKernelGGL((extend_border3_kernel<uint16_t>), grid, block, 0, st, bp);
    XH_LAUNCH_CHECK("extend_border3_kernel");
    return X265HIP_OK;
}
} // namespace xh
