// internal.h — cross-file helpers of libx265hip that are NOT part of the C ABI (used by the frame pass only).
#pragma once
#include "common.h"
#include "searchrange.h"

namespace xh {

// sa8d(src, pred) of up to 4 CU sizes in ONE launch (the frame pass's mode costs; same arithmetic as pixcmp_kernel<SA8D>)
struct Sa8dLevel { const int32_t* offA; const int32_t* offB; int32_t* out; int n; int size; };
int sa8d_levels(int depth, const void* planeA, int64_t strideA, const void* planeB, int64_t strideB, const Sa8dLevel* levels, int nLevels,
                hipStream_t st);

// the same four costs (out[l] for CU size 64 >> l, raster order of the complete CUs inside a W x H picture) from one Hadamard pass
int sa8d_pyramid(int depth, const void* planeA, int64_t strideA, const void* planeB, int64_t strideB, int W, int H, int32_t* const out[4], hipStream_t st);

// Predict::predInterLumaPixel for 8x8 PUs reading the pre-filtered quarter-pel planes (a phase-selected block copy)
int pred_from_planes(int depth, int size, const void* planes, int64_t planeElems, int64_t strideR, void* dst, int64_t strideD,
                     const int32_t* pu_xy, const int32_t* qmv, int n, hipStream_t st);

// up to four residual chains (x265hip_residual_chain_batch) of different TU sizes / planes / quantisers in ONE launch (frame.hip)
struct ChainJob
{
    int size;
    const void* fenc; int64_t sF; const void* pred; int64_t sP; void* recon; int64_t sR;
    const int32_t* offF; const int32_t* offP; const int32_t* offR; const int32_t* quantCoeff;
    int qBits, add, dqScale, dqShift;
    int16_t* level; uint32_t* numSig; uint64_t* dist; int n;
};
int residual_chain_multi(int depth, const ChainJob* jobs, int count, hipStream_t st);

// border extension of up to four planes in one launch
int extend_border_planes(int depth, int nPlanes, void* const* pics, const int64_t* strides, const int* w, const int* h, const int* mx, const int* my,
                         hipStream_t st);

} // namespace xh
