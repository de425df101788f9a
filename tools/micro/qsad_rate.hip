// Issue-rate probe for the SAD instructions of gfx950 (profiling aid, not part of the product):
// v_sad_u8 (4 abs-diffs per lane), v_qsad_pk_u16_u8 (16), v_mqsad_u32_u8 (16, masked) — 8 independent accumulators per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int MODE> __global__ __launch_bounds__(256) void probe(const uint32_t* in, uint32_t* out, int iters)
{
    uint32_t a = in[threadIdx.x], b = in[threadIdx.x + 256];
    uint64_t w = ((uint64_t)b << 32) | a;
    uint32_t s[8] = { 0 };
    uint64_t q[8] = { 0 };
    for (int i = 0; i < iters; i++)
    {
#pragma unroll
        for (int k = 0; k < 8; k++)
        {
            if (MODE == 0) s[k] = __builtin_amdgcn_sad_u8(a + k, b, s[k]);
            else if (MODE == 1) q[k] = __builtin_amdgcn_qsad_pk_u16_u8(w + k, b, q[k]);
            else q[k] = __builtin_amdgcn_mqsad_pk_u16_u8(w + k, b, q[k]);
        }
        a += 0x01010101u; w += 0x0101010101010101ull;
    }
    uint64_t r = 0;
    for (int k = 0; k < 8; k++) r += s[k] + q[k];
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)r ^ (uint32_t)(r >> 32);
}

int main()
{
    uint32_t *in, *out;
    hipMalloc(&in, 4096); hipMalloc(&out, 4 * 256 * 4096);
    hipMemset(in, 0x5a, 4096);
    const int iters = 4096, blocks = 2048;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; mode++)
    {
        for (int rep = 0; rep < 2; rep++)
        {
            hipEventRecord(e0);
            if (mode == 0) probe<0><<<blocks, 256>>>(in, out, iters);
            else if (mode == 1) probe<1><<<blocks, 256>>>(in, out, iters);
            else probe<2><<<blocks, 256>>>(in, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep)
            {
                const double waveInstr = (double)blocks * 4 * iters * 8;
                printf("%s: %.3f ms, %.1f G wave-instr/s, %.2f T abs-diff/s\n", mode == 0 ? "v_sad_u8" : mode == 1 ? "v_qsad_pk_u16_u8" : "v_mqsad_pk_u16_u8", ms,
                       waveInstr / ms * 1e-6, waveInstr * 64 * (mode ? 16 : 4) / ms * 1e-9);
            }
        }
    }
    return 0;
}
