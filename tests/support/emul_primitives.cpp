// emul_primitives.cpp — TEST INFRASTRUCTURE ONLY: the setupAssemblyPrimitives of the CPU-tier emulation build (oracle/_ref/x265_emul_*bit).
// The product's x265_hip_primitives.cpp also carries the per-call shims of every slot, which bind the whole device library; the emulation
// build only exercises the two seams that do not need a GPU to be checked for plumbing — the lookahead session and the reference-picture
// mirrors — against tests/support/libx265hip_emul.so, so its table setup is just the lookup slots.
#include <cstring>
#include <mutex>

#include "common.h"
#include "primitives.h"

namespace X265_NS {
void x265hip_install_lookup_slots(EncoderPrimitives& p);        // x265_amd/host/x265_hip_refplanes.cpp
void x265hip_install_psy_slots(EncoderPrimitives& p);           // x265_amd/host/x265_hip_srcplanes.cpp
void x265hip_install_cuserve_slots(EncoderPrimitives& p);       // x265_amd/host/x265_hip_cuserve.cpp
// the reference's C table for the bindings' "what the slot did before" (same function as in the product's x265_hip_primitives.cpp)
const EncoderPrimitives& x265hip_c_table()
{
    static EncoderPrimitives c;
    static bool ready = false;
    static std::mutex once;
    std::lock_guard<std::mutex> g(once);
    if (!ready)
    {
        memset(&c, 0, sizeof(c));
        setupCPrimitives(c);
        setupAliasPrimitives(c);
        ready = true;
    }
    return c;
}
void setupInstrinsicPrimitives(EncoderPrimitives&, int) {}
void setupAssemblyPrimitives(EncoderPrimitives& p, int) { setupAliasPrimitives(p); x265hip_install_lookup_slots(p); x265hip_install_psy_slots(p); x265hip_install_cuserve_slots(p); }
}
extern "C" {
int PFX(cpu_cpuid_test)(void) { return 0; }
void PFX(cpu_emms)(void) {}
void PFX(cpu_cpuid)(uint32_t, uint32_t* eax, uint32_t* ebx, uint32_t* ecx, uint32_t* edx) { *eax = *ebx = *ecx = *edx = 0; }
void PFX(cpu_xgetbv)(uint32_t, uint32_t* eax, uint32_t* edx) { *eax = *edx = 0; }
void PFX(cpu_neon_test)(void) {}
int PFX(cpu_fast_neon_mrc_test)(void) { return 0; }
}
