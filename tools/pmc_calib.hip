// pmc_calib.hip — known-byte-count micro-kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 in OUR access patterns
// (MI355X_MICROARCH.md §HBM: only 16 B / lane streaming reads are calibrated there; "calibrate on a known byte count in your own access
// pattern before trusting an absolute").  Each kernel touches every byte of the buffer exactly once; buffers are sized well past the
// 256 MiB Infinity Cache by the driver script (tools/pmc_calibrate.py), so bytes moved == bytes of the buffer.
//   calib_read4u   4-byte loads at a 1-byte misalignment, consecutive lanes 4 bytes apart: what the motion kernels issue for candidates
//   calib_read8u   8-byte loads, 2-byte misalignment (16-bit pixels)
//   calib_read16   16-byte aligned loads (the guide's calibrated pattern; control)
//   calib_write4 / calib_write16   stores of the same shapes
#include <hip/hip_runtime.h>
#include <cstdint>

template <typename T> __device__ __forceinline__ T ldu(const void* p) { T v; __builtin_memcpy(&v, p, sizeof(T)); return v; }

__global__ void calib_read4u(const uint8_t* p, size_t n, uint32_t* sink)
{
    uint32_t acc = 0;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i + 5 <= n; i += (size_t)gridDim.x * blockDim.x * 4)
        acc += ldu<uint32_t>(p + i + 1);
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void calib_read8u(const uint8_t* p, size_t n, uint32_t* sink)
{
    uint32_t acc = 0;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i + 10 <= n; i += (size_t)gridDim.x * blockDim.x * 8)
    {
        const uint2 v = ldu<uint2>(p + i + 2);
        acc += v.x ^ v.y;
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void calib_read16(const uint4* p, size_t n16, uint32_t* sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    {
        const uint4 v = p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void calib_write4(uint32_t* p, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        p[i] = (uint32_t)i;
}
__global__ void calib_write16(uint4* p, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_uint4((uint32_t)i, 1, 2, 3);
}

extern "C" int pmc_calib_run(size_t bytes, int reps)
{
    uint8_t* buf = nullptr;
    uint32_t* sink = nullptr;
    if (hipMalloc((void**)&buf, bytes + 64) != hipSuccess || hipMalloc((void**)&sink, 4) != hipSuccess) return 1;
    if (hipMemset(buf, 1, bytes + 64) != hipSuccess) return 2;
    const dim3 grid(256 * 8), block(256);
    for (int r = 0; r < reps; r++)
    {
        hipLaunchKernelGGL(calib_read4u, grid, block, 0, 0, buf, bytes, sink);
        hipLaunchKernelGGL(calib_read8u, grid, block, 0, 0, buf, bytes, sink);
        hipLaunchKernelGGL(calib_read16, grid, block, 0, 0, (const uint4*)buf, bytes / 16, sink);
        hipLaunchKernelGGL(calib_write4, grid, block, 0, 0, (uint32_t*)buf, bytes / 4);
        hipLaunchKernelGGL(calib_write16, grid, block, 0, 0, (uint4*)buf, bytes / 16);
    }
    const int rc = hipDeviceSynchronize() == hipSuccess ? 0 : 3;
    (void)hipFree(buf);
    (void)hipFree(sink);
    return rc;
}
