// smallops.hip — the remaining small table primitives around the path: pixel_var (cu[].var, pixel.cpp:704), explicit weighted prediction
// (weight_pp / weight_sp, pixel.cpp:493-543), the 64x64 intra-scan downscales (scale1D_128to64 / scale2D_64to32, pixel.cpp:559-602) and
// transpose<N> (pixel.cpp:485).  All elementwise or one small reduction; HBM-bound.
#include "common.h"

namespace xh {

// cu[].var: sum | (sum of squares) << 32, both 32-bit with the reference's wrap-around (uint32_t accumulators)
template <typename P>
__global__ __launch_bounds__(256) void var_kernel(const P* __restrict__ plane, int64_t stride, const int32_t* __restrict__ off, int size, int n,
                                                  unsigned long long* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int job = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (job >= n) return;
    const P* p = plane + off[job];
    const int qx = size >> 2, quads = qx * size;
    uint32_t sum = 0, sqr = 0;
    for (int q = lane; q < quads; q += 64)
    {
        int v[4];
        load4(p + (int64_t)(q / qx) * stride + (q % qx) * 4, v);
#pragma unroll
        for (int i = 0; i < 4; i++) { sum += (uint32_t)v[i]; sqr += (uint32_t)v[i] * (uint32_t)v[i]; }
    }
    for (int o = 32; o; o >>= 1)
    {
        sum += (uint32_t)__shfl_xor((int)sum, o);
        sqr += (uint32_t)__shfl_xor((int)sqr, o);
    }
    if (lane == 0)
        out[job] = (unsigned long long)sum | ((unsigned long long)sqr << 32);
}

// weight_pp (SRC = pixel, lifted by 14 - depth) / weight_sp (SRC = int16 14-bit intermediate, + IF_INTERNAL_OFFS)
template <typename S, typename P>
__global__ __launch_bounds__(256) void weight_kernel(const S* __restrict__ src, int64_t ss, P* __restrict__ dst, int64_t ds, int width, int height,
                                                     int w0, int round, int shift, int offset, int depth)
{
    const long long total = (long long)width * height;
    const int maxv = (1 << depth) - 1;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    {
        const int y = (int)(i / width), x = (int)(i % width);
        const int s = (int)src[(int64_t)y * ss + x];
        const int lifted = std::is_same<S, int16_t>::value ? s + 8192 : (int)(int16_t)(s << (14 - depth));
        const int v = ((w0 * lifted + round) >> shift) + offset;
        dst[(int64_t)y * ds + x] = (P)(v < 0 ? 0 : (v > maxv ? maxv : v));
    }
}

template <typename P>
__global__ __launch_bounds__(256) void scale1d_kernel(const P* __restrict__ src, P* __restrict__ dst, long long total)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    {
        const long long job = i >> 7;
        const int e = (int)(i & 127);                                  // 0..63 top half, 64..127 left half
        const P* s = src + job * 256 + (e >> 6) * 128 + 2 * (e & 63);
        dst[i] = (P)(((int)s[0] + (int)s[1] + 1) >> 1);
    }
}

template <typename P>
__global__ __launch_bounds__(256) void scale2d_kernel(const P* __restrict__ plane, int64_t stride, const int32_t* __restrict__ off, P* __restrict__ dst, long long total)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    {
        const int job = (int)(i >> 10), e = (int)(i & 1023), y = e >> 5, x = e & 31;
        const P* p = plane + off[job] + (int64_t)(2 * y) * stride + 2 * x;
        dst[i] = (P)(((int)p[0] + (int)p[1] + (int)p[stride] + (int)p[stride + 1] + 2) >> 2);
    }
}

template <typename P>
__global__ __launch_bounds__(256) void transpose_kernel(const P* __restrict__ plane, int64_t stride, const int32_t* __restrict__ off, P* __restrict__ dst,
                                                        int size, long long total)
{
    const int sq = size * size;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    {
        const int job = (int)(i / sq), e = (int)(i % sq), k = e / size, l = e % size;
        dst[i] = plane[off[job] + (int64_t)l * stride + k];
    }
}

static bool square_size(int s) { return s == 4 || s == 8 || s == 16 || s == 32 || s == 64; }

} // namespace xh

using namespace xh;

extern "C" {

int x265hip_var_batch(int depth, int size, const void* plane, int64_t stride, const int32_t* off, int n, uint64_t* out, void* stream)
{
    XH_CHECK_DEV();
    if (!valid_depth(depth) || !square_size(size) || n < 0)
        return set_error(X265HIP_EINVAL, "var_batch: depth %d size %d n %d", depth, size, n);
    if (!n) return X265HIP_OK;
    dim3 grid((n + 3) / 4), block(256);
    if (depth == 8) hipLaunchKernelGGL((var_kernel<uint8_t>), grid, block, 0, as_stream(stream), (const uint8_t*)plane, stride, off, size, n, (unsigned long long*)out);
    else hipLaunchKernelGGL((var_kernel<uint16_t>), grid, block, 0, as_stream(stream), (const uint16_t*)plane, stride, off, size, n, (unsigned long long*)out);
    XH_LAUNCH_CHECK("var_kernel");
    return X265HIP_OK;
}

int x265hip_weight_pp(int depth, const void* src, void* dst, int64_t stride, int width, int height, int w0, int round, int shift, int offset, void* stream)
{
    XH_CHECK_DEV();
    if (!valid_depth(depth) || width < 1 || height < 1 || shift < 0 || shift > 30)
        return set_error(X265HIP_EINVAL, "weight_pp: depth %d %dx%d shift %d", depth, width, height, shift);
    dim3 grid(grid_for(((long long)width * height + 255) / 256)), block(256);
    if (depth == 8) hipLaunchKernelGGL((weight_kernel<uint8_t, uint8_t>), grid, block, 0, as_stream(stream), (const uint8_t*)src, stride, (uint8_t*)dst, stride, width, height, w0, round, shift, offset, depth);
    else hipLaunchKernelGGL((weight_kernel<uint16_t, uint16_t>), grid, block, 0, as_stream(stream), (const uint16_t*)src, stride, (uint16_t*)dst, stride, width, height, w0, round, shift, offset, depth);
    XH_LAUNCH_CHECK("weight_kernel(pp)");
    return X265HIP_OK;
}

int x265hip_weight_sp(int depth, const int16_t* src, void* dst, int64_t srcStride, int64_t dstStride, int width, int height, int w0, int round, int shift,
                      int offset, void* stream)
{
    XH_CHECK_DEV();
    if (!valid_depth(depth) || width < 1 || height < 1 || shift < 0 || shift > 30)
        return set_error(X265HIP_EINVAL, "weight_sp: depth %d %dx%d shift %d", depth, width, height, shift);
    dim3 grid(grid_for(((long long)width * height + 255) / 256)), block(256);
    if (depth == 8) hipLaunchKernelGGL((weight_kernel<int16_t, uint8_t>), grid, block, 0, as_stream(stream), src, srcStride, (uint8_t*)dst, dstStride, width, height, w0, round, shift, offset, depth);
    else hipLaunchKernelGGL((weight_kernel<int16_t, uint16_t>), grid, block, 0, as_stream(stream), src, srcStride, (uint16_t*)dst, dstStride, width, height, w0, round, shift, offset, depth);
    XH_LAUNCH_CHECK("weight_kernel(sp)");
    return X265HIP_OK;
}

int x265hip_scale1d_128to64_batch(int depth, const void* src, void* dst, int n, void* stream)
{
    XH_CHECK_DEV();
    if (!valid_depth(depth) || n < 0) return set_error(X265HIP_EINVAL, "scale1d_128to64: depth %d n %d", depth, n);
    if (!n) return X265HIP_OK;
    const long long total = (long long)n * 128;
    dim3 grid(grid_for((total + 255) / 256)), block(256);
    if (depth == 8) hipLaunchKernelGGL((scale1d_kernel<uint8_t>), grid, block, 0, as_stream(stream), (const uint8_t*)src, (uint8_t*)dst, total);
    else hipLaunchKernelGGL((scale1d_kernel<uint16_t>), grid, block, 0, as_stream(stream), (const uint16_t*)src, (uint16_t*)dst, total);
    XH_LAUNCH_CHECK("scale1d_kernel");
    return X265HIP_OK;
}

int x265hip_scale2d_64to32_batch(int depth, const void* plane, int64_t stride, const int32_t* off, void* dst, int n, void* stream)
{
    XH_CHECK_DEV();
    if (!valid_depth(depth) || n < 0) return set_error(X265HIP_EINVAL, "scale2d_64to32: depth %d n %d", depth, n);
    if (!n) return X265HIP_OK;
    const long long total = (long long)n * 1024;
    dim3 grid(grid_for((total + 255) / 256)), block(256);
    if (depth == 8) hipLaunchKernelGGL((scale2d_kernel<uint8_t>), grid, block, 0, as_stream(stream), (const uint8_t*)plane, stride, off, (uint8_t*)dst, total);
    else hipLaunchKernelGGL((scale2d_kernel<uint16_t>), grid, block, 0, as_stream(stream), (const uint16_t*)plane, stride, off, (uint16_t*)dst, total);
    XH_LAUNCH_CHECK("scale2d_kernel");
    return X265HIP_OK;
}

int x265hip_transpose_batch(int depth, int size, const void* plane, int64_t stride, const int32_t* off, void* dst, int n, void* stream)
{
    XH_CHECK_DEV();
    if (!valid_depth(depth) || !square_size(size) || n < 0) return set_error(X265HIP_EINVAL, "transpose: depth %d size %d n %d", depth, size, n);
    if (!n) return X265HIP_OK;
    const long long total = (long long)n * size * size;
    dim3 grid(grid_for((total + 255) / 256)), block(256);
    if (depth == 8) hipLaunchKernelGGL((transpose_kernel<uint8_t>), grid, block, 0, as_stream(stream), (const uint8_t*)plane, stride, off, (uint8_t*)dst, size, total);
    else hipLaunchKernelGGL((transpose_kernel<uint16_t>), grid, block, 0, as_stream(stream), (const uint16_t*)plane, stride, off, (uint16_t*)dst, size, total);
    XH_LAUNCH_CHECK("transpose_kernel");
    return X265HIP_OK;
}

} // extern "C"
