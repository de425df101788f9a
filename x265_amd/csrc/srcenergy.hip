// srcenergy.hip — AC-energy planes of a SOURCE picture (x265hip_source_energy, include/x265hip.h).
//
// psyCost_pp (reference source/common/pixel.cpp:726-757), the psycho-visual term of every RD cost at psy-rd > 0, compares the AC energy of
// the source block with that of a candidate reconstruction, 8x8 by 8x8: energy = sa8d_8x8(block, 0) - (sum(block) >> 2) (for 4x4 blocks
// satd_4x4(block, 0) - (sum >> 2)).  The reference recomputes the SOURCE half at every call — dozens of times per block during mode decision —
// although it is a function of the source picture and the position only.  So, like the fractional planes of a reference picture (refpic.hip),
// it is computed once per picture: one launch per plane gives the energy of every aligned 8x8 and 4x4 block; the table slot
// (x265_amd/host/x265_hip_srcplanes.cpp) then looks the source half up and only computes the reconstruction half.
#include "common.h"
#include <cstring>

namespace xh {

// 8-point Hadamard butterflies on eight values (unnormalised)
__device__ __forceinline__ void had8(int (&v)[8])
{
#pragma unroll
    for (int s = 1; s < 8; s <<= 1)
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (!(i & s))
            {
                const int a = v[i], b = v[i + s];
                v[i] = a + b;
                v[i + s] = a - b;
            }
}

// one thread = one aligned 8x8 block: E8 = ((sum |H8 X H8^T| + 2) >> 2) - (sum X >> 2)   (pixel.cpp:299-339, :741-742),
// and its four 4x4 blocks: E4 = (sum |H4 X H4^T| >> 1) - (sum X >> 2)                      (pixel.cpp:210-237, :752)
template <typename P>
__global__ __launch_bounds__(128) void source_energy_kernel(const P* __restrict__ plane, int64_t stride, int bw, int bh, int32_t* __restrict__ e8, int32_t* __restrict__ e4)
{
    const int idx = blockIdx.x * 128 + threadIdx.x;
    if (idx >= bw * bh)
        return;
    const int by = idx / bw, bx = idx - by * bw;
    const P* p = plane + (int64_t)by * 8 * stride + bx * 8;
    int m[8][8];
#pragma unroll
    for (int y = 0; y < 8; y++)
    {
        int lo[4], hi[4];
        load4(p + (int64_t)y * stride, lo);
        load4(p + (int64_t)y * stride + 4, hi);
#pragma unroll
        for (int x = 0; x < 4; x++) { m[y][x] = lo[x]; m[y][x + 4] = hi[x]; }
    }
    // ---- the four 4x4 blocks
#pragma unroll
    for (int q = 0; q < 4; q++)
    {
        const int oy = (q >> 1) * 4, ox = (q & 1) * 4;
        int t[4][4], sum = 0, had = 0;
#pragma unroll
        for (int y = 0; y < 4; y++)
        {
            const int a = m[oy + y][ox], b = m[oy + y][ox + 1], c = m[oy + y][ox + 2], d = m[oy + y][ox + 3];
            sum += a + b + c + d;
            const int s01 = a + b, d01 = a - b, s23 = c + d, d23 = c - d;
            t[y][0] = s01 + s23; t[y][1] = s01 - s23; t[y][2] = d01 + d23; t[y][3] = d01 - d23;
        }
#pragma unroll
        for (int x = 0; x < 4; x++)
        {
            const int s01 = t[0][x] + t[1][x], d01 = t[0][x] - t[1][x], s23 = t[2][x] + t[3][x], d23 = t[2][x] - t[3][x];
            had += iabs(s01 + s23) + iabs(s01 - s23) + iabs(d01 + d23) + iabs(d01 - d23);
        }
        e4[(int64_t)(by * 2 + (q >> 1)) * (bw * 2) + bx * 2 + (q & 1)] = (had >> 1) - (sum >> 2);
    }
    // ---- the 8x8 block
    int sum = 0;
#pragma unroll
    for (int y = 0; y < 8; y++)
    {
#pragma unroll
        for (int x = 0; x < 8; x++) sum += m[y][x];
        had8(m[y]);
    }
    int had = 0;
#pragma unroll
    for (int x = 0; x < 8; x++)
    {
        int col[8];
#pragma unroll
        for (int y = 0; y < 8; y++) col[y] = m[y][x];
        had8(col);
#pragma unroll
        for (int y = 0; y < 8; y++) had += iabs(col[y]);
    }
    e8[idx] = ((had + 2) >> 2) - (sum >> 2);
}

} // namespace xh

using namespace xh;

namespace xh {
// per calling thread: a stream, a device buffer and page-locked staging, kept between calls (a picture's worth of planes arrives per frame)
struct EnergyScratch
{
    hipStream_t st = nullptr;
    char* d = nullptr; char* h = nullptr;
    size_t cap = 0;
    int device = -1;
};
static thread_local EnergyScratch t_es;
}

extern "C" int x265hip_source_energy(int depth, const void* hostPlane, int64_t stride, int width, int height, int32_t* hostE8, int32_t* hostE4)
{
    XH_CHECK_DEV();
    if (!valid_depth(depth) || !hostPlane || !hostE8 || !hostE4 || width < 8 || height < 8 || stride < width)
        return set_error(X265HIP_EINVAL, "source_energy: depth %d %dx%d stride %lld", depth, width, height, (long long)stride);
    // complete 8x8 blocks only (what psyCost_pp can be asked about: CUs and TUs are aligned to their size)
    const int bw = width >> 3, bh = height >> 3, B = depth == 8 ? 1 : 2;
    const size_t planeBytes = (size_t)stride * (bh * 8) * B, e8Bytes = (size_t)bw * bh * 4, e4Bytes = e8Bytes * 4, need = planeBytes + e8Bytes + e4Bytes;
    EnergyScratch& es = t_es;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (es.device != dev || es.cap < need)
    {
        if (es.d) (void)device_free(es.d);
        if (es.h) (void)hipHostFree(es.h);
        if (es.st && es.device != dev) { (void)hipStreamDestroy(es.st); es.st = nullptr; }
        es.d = es.h = nullptr; es.cap = 0; es.device = dev;
        if (!es.st && hipStreamCreateWithFlags(&es.st, hipStreamNonBlocking) != hipSuccess)
            return set_error(X265HIP_EHIP, "source_energy: stream");
        if (hipMalloc((void**)&es.d, need) != hipSuccess || hipHostMalloc((void**)&es.h, need, hipHostMallocDefault) != hipSuccess)
            return set_error(X265HIP_ENOMEM, "source_energy: %zu bytes", need);
        es.cap = need;
    }
    memcpy(es.h, hostPlane, planeBytes);
    if (hipMemcpyAsync(es.d, es.h, planeBytes, hipMemcpyHostToDevice, es.st) != hipSuccess)
        return set_error(X265HIP_EHIP, "source_energy: upload");
    int32_t* dE8 = (int32_t*)(es.d + planeBytes);
    int32_t* dE4 = dE8 + (size_t)bw * bh;
    const dim3 grid((bw * bh + 127) / 128), block(128);
    DevSpan span(X265HIP_CLK_ENERGY, es.st);
    if (depth == 8)
        hipLaunchKernelGGL((source_energy_kernel<uint8_t>), grid, block, 0, es.st, (const uint8_t*)es.d, stride, bw, bh, dE8, dE4);
    else
        hipLaunchKernelGGL((source_energy_kernel<uint16_t>), grid, block, 0, es.st, (const uint16_t*)es.d, stride, bw, bh, dE8, dE4);
    XH_LAUNCH_CHECK("source_energy_kernel");
    span.end();
    if (hipMemcpyAsync(es.h + planeBytes, dE8, e8Bytes + e4Bytes, hipMemcpyDeviceToHost, es.st) != hipSuccess || hipStreamSynchronize(es.st) != hipSuccess)
        return set_error(X265HIP_EHIP, "source_energy: download");
    span.commit();
    memcpy(hostE8, es.h + planeBytes, e8Bytes);
    memcpy(hostE4, es.h + planeBytes + e8Bytes, e4Bytes);
    return X265HIP_OK;
}
