// common.h — shared device/host helpers for the gfx950 kernels of libx265hip.
// Wave size is 64 on CDNA4; every cross-lane idiom below is written for that and for nothing else.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/x265hip.h"

namespace xh {

constexpr int kWave = 64;

// ---- error plumbing (host) ---------------------------------------------------------------------------------
int set_error(int code, const char* fmt, ...);
int check_hip(hipError_t e, const char* what);
int ensure_device();                       // lazy hipSetDevice + capability check; X265HIP_ENODEV when absent
// hipFree synchronises EVERY stream of the device (measured: 1.45 s beside a resident kernel, tools/micro/mailbox_diag), so the library never calls it
// directly: device_free() first asks the resident CU-job servers (cuserve.hip) to leave, frees, and lets the next submitter start them again.
hipError_t device_free(void* p);
// runtime.hip: a non-blocking stream of `device` for the duration of one operation.  hipStreamCreateWithFlags takes 3.9 ms on the MI355X box
// (tools/micro/create_cost): short-lived users share a pool instead of owning one each; streams go back, they are never destroyed.
// runtime.hip: workgroups that stay on the chip (the CU-job server's, one per mailbox slot, each alone on a CU as far as a kernel that needs most of a
// CU's LDS is concerned): resident_workgroups(device, +n / -n) keeps the count per device, free_compute_units(device) = CUs of the device minus that count
void resident_workgroups(int device, int delta);
int free_compute_units(int device);
hipStream_t stream_lease(int device);
void stream_return(int device, hipStream_t st);
hipError_t pinned_alloc(void** out, size_t bytes);      // runtime.hip: page-locked host memory for the big buffers (huge pages + hipHostRegister)
hipError_t pinned_free(void* p);
void servers_pause();                      // cuserve.hip
void servers_resume();
inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
bool valid_depth(int depth);
bool valid_block(int w, int h);            // multiples of 2 up to 64 (luma PU shapes and their chroma halves)

#define XH_CHECK_DEV()      do { int e_ = xh::ensure_device(); if (e_) return e_; } while (0)
#define XH_LAUNCH_CHECK(nm) do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return xh::check_hip(e_, nm); } while (0)

// ---- device-time ledger (x265hip_device_time): a span = the launches between begin() and end() on one stream; commit() after that stream has been
// synchronised adds the elapsed time to the clock.  Events are per thread and per clock; a span that could not get its events measures nothing.
struct DevSpan
{
    int clk; hipStream_t st; hipEvent_t e0 = nullptr, e1 = nullptr;
    uint64_t bytes = 0;                              // SURVEY.md §8d algorithmic bytes of the span's work, where the module states them
    DevSpan(int clock, hipStream_t stream);          // records the first event
    void end();                                      // records the second event
    void commit();                                   // after the stream's synchronisation
};

// grid sizing for batch kernels: enough 256-thread workgroups to fill 256 CUs several times over, grid-stride beyond
inline int grid_for(long long workgroups_needed, int cap = 256 * 16)
{
    if (workgroups_needed < 1) workgroups_needed = 1;
    return (int)(workgroups_needed < cap ? workgroups_needed : cap);
}

// ---- unaligned loads ---------------------------------------------------------------------------------------
// x265 block pointers carry no alignment promise (primitives.h:133; TestBench slides by odd offsets).  gfx950
// global/LDS accesses are unaligned-capable, and memcpy lowers to one global_load_dword / dwordx2 / dwordx4.
template <typename T, typename P>
__device__ __forceinline__ T ld_unaligned(const P* p)
{
    T v;
    __builtin_memcpy(&v, p, sizeof(T));
    return v;
}
// the same through a pointer the compiler cannot prove to be global memory (one read out of a descriptor in memory): the address-space cast
// turns flat_load (aperture check, both wait counters) into global_load
#define XH_GLOBAL_AS __attribute__((address_space(1)))
template <typename T, typename P>
__device__ __forceinline__ T ld_global_unaligned(const P* p)
{
    T v;
    __builtin_memcpy(&v, (const XH_GLOBAL_AS char*)p, sizeof(T));
    return v;
}
template <typename T, typename P>
__device__ __forceinline__ void st_unaligned(P* p, T v)
{
    __builtin_memcpy(p, &v, sizeof(T));
}

// ---- packed 16-bit SATD arithmetic.  Every value of a 4x4 Hadamard of pixel differences is at most 16 * (2^depth - 1): 4080 at 8 bit,
// 16368 at 10 bit — inside int16 — so two values share a register (v_pk_add / v_pk_sub / v_pk_mad / v_pk_max, one DPP move for two),
// which halves the instruction count of the sub-pel comparisons.  12-bit pixels (65520) keep the 32-bit path.
typedef short s2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s2v as_s2(uint32_t v) { return __builtin_bit_cast(s2v, v); }
__device__ __forceinline__ uint32_t as_u(s2v v) { return __builtin_bit_cast(uint32_t, v); }
template <typename P> struct Pk16;
template <> struct Pk16<uint8_t>
{
    // bytes 0,1 -> (lo, hi) halves of one register, bytes 2,3 -> the other (v_perm_b32; selector 0x0c = constant zero byte)
    static __device__ __forceinline__ void split(uint32_t a, s2v& p01, s2v& p23)
    {
        p01 = as_s2(__builtin_amdgcn_perm(0u, a, 0x0c010c00u));
        p23 = as_s2(__builtin_amdgcn_perm(0u, a, 0x0c030c02u));
    }
};
template <> struct Pk16<uint16_t>
{
    static __device__ __forceinline__ void split(uint2 a, s2v& p01, s2v& p23) { p01 = as_s2(a.x); p23 = as_s2(a.y); }
};

// 4 horizontally adjacent pixels as ints
__device__ __forceinline__ void load4(const uint8_t* p, int v[4])
{
    uint32_t x = ld_unaligned<uint32_t>(p);
    v[0] = x & 255; v[1] = (x >> 8) & 255; v[2] = (x >> 16) & 255; v[3] = x >> 24;
}
__device__ __forceinline__ void load4(const uint16_t* p, int v[4])
{
    uint2 x = ld_unaligned<uint2>(p);
    v[0] = x.x & 0xffff; v[1] = x.x >> 16; v[2] = x.y & 0xffff; v[3] = x.y >> 16;
}
__device__ __forceinline__ void load4(const int16_t* p, int v[4])
{
    uint2 x = ld_unaligned<uint2>(p);
    v[0] = (int)(int16_t)(x.x & 0xffff); v[1] = (int)x.x >> 16; v[2] = (int)(int16_t)(x.y & 0xffff); v[3] = (int)x.y >> 16;
}
__device__ __forceinline__ void store4(uint8_t* p, const int v[4])
{
    st_unaligned<uint32_t>(p, (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24));
}
__device__ __forceinline__ void store4(uint16_t* p, const int v[4])
{
    uint2 x;
    x.x = (uint32_t)v[0] | ((uint32_t)v[1] << 16);
    x.y = (uint32_t)v[2] | ((uint32_t)v[3] << 16);
    st_unaligned<uint2>(p, x);
}
__device__ __forceinline__ void store4(int16_t* p, const int v[4])
{
    uint2 x;
    x.x = ((uint32_t)v[0] & 0xffff) | ((uint32_t)v[1] << 16);
    x.y = ((uint32_t)v[2] & 0xffff) | ((uint32_t)v[3] << 16);
    st_unaligned<uint2>(p, x);
}
// 2 adjacent elements (chroma 2xN / 6xN shapes)
template <typename P>
__device__ __forceinline__ void load2(const P* p, int v[2]) { v[0] = p[0]; v[1] = p[1]; }

// ---- wave-level reductions ---------------------------------------------------------------------------------
// DPP quad permutes: cross-lane inside aligned groups of 4 lanes without touching LDS.
__device__ __forceinline__ int quad_xor1(int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true); }  // [1,0,3,2]
__device__ __forceinline__ int quad_xor2(int v) { return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true); }  // [2,3,0,1]
__device__ __forceinline__ int quad_sum(int v) { v += quad_xor1(v); v += quad_xor2(v); return v; }

// sum over aligned groups of `g` lanes (g = power of two, 1..64, wave-uniform); every lane of a group gets the sum
__device__ __forceinline__ int group_sum(int v, int g)
{
    if (g > 1) v += quad_xor1(v);
    if (g > 2) v += quad_xor2(v);
    for (int o = 4; o < g; o <<= 1)
        v += __shfl_xor(v, o, kWave);
    return v;
}
__device__ __forceinline__ unsigned long long group_sum64(unsigned long long v, int g)
{
    for (int o = 1; o < g; o <<= 1)
        v += __shfl_xor(v, o, kWave);
    return v;
}
__device__ __forceinline__ int wave_sum(int v) { return group_sum(v, kWave); }

__device__ __forceinline__ int clip3i(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }

// smallest power of two >= v (v in 1..64)
__host__ __device__ inline int pow2_ceil(int v)
{
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// ---- filter and transform tables (HEVC normative constants; reference copies: common/constants.cpp:250-344) ----
// Luma 8-tap / chroma 4-tap DCT-IF coefficients.
__device__ __constant__ const int8_t kLumaFilter[4][8] = {
    { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 }, { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
__device__ __constant__ const int8_t kChromaFilter[8][4] = {
    { 0, 64, 0, 0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 }, { -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } };

// ---- DPP all-reduce over a 16-lane row: the sum lands in all 16 lanes (row_ror:8,4,2,1), no LDS, no readlane -------------------
template <int CTRL>
__device__ __forceinline__ int dpp_all(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ int row_allsum(int v)
{
    v += dpp_all<0x128>(v);
    v += dpp_all<0x124>(v);
    v += dpp_all<0x122>(v);
    v += dpp_all<0x121>(v);
    return v;
}

// filter tap i of phase idx; NT = 8 (luma) or 4 (chroma)
template <int NT>
__device__ __forceinline__ int filter_tap(int idx, int i)
{
    if (NT == 8) return kLumaFilter[idx][i];
    return kChromaFilter[idx][i];
}

} // namespace xh
