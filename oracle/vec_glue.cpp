// oracle/vec_glue.cpp — TEST / BASELINE INFRASTRUCTURE ONLY (bench.py's second cpu_baseline leg, kind "reference+vec").
//
// The reference's own SIMD for the transform rows that needs no assembler: source/common/vec/{dct-sse3,dct-ssse3,dct-sse41}.cpp
// (idct8/16/32, dct16/32, dequant_scaling as compiler intrinsics), installed by setupInstrinsicPrimitives (vec-primitives.cpp:59-80)
// when primitives.cpp is compiled with -DENABLE_ASSEMBLY=1 (primitives.cpp:260-265).  That configuration also expects the nasm half of the
// build: setupAssemblyPrimitives (asm-primitives.cpp, needs the .asm objects) and the four cpu-a.asm helpers.  nasm is not in the image, so
// this TU supplies exactly what the reference's own no-assembly branch supplies for them (primitives.cpp:288-303: NOPs) and an
// assembly table that installs nothing.  The encoder is then run with --asm SSE4.1 (cpu_detect() answers 0 through the NOP cpuid; the
// names are taken as given, param.cpp:1445-1456).  None of this is linked into the product.
#include "common.h"
#include "primitives.h"

namespace X265_NS {
void setupAssemblyPrimitives(EncoderPrimitives&, int) {}
}

extern "C" {
int PFX(cpu_cpuid_test)(void) { return 0; }
void PFX(cpu_emms)(void) {}
void PFX(cpu_cpuid)(uint32_t, uint32_t* eax, uint32_t*, uint32_t*, uint32_t*) { *eax = 0; }
void PFX(cpu_xgetbv)(uint32_t, uint32_t*, uint32_t*) {}
void PFX(cpu_neon_test)(void) {}
int PFX(cpu_fast_neon_mrc_test)(void) { return 0; }
}
