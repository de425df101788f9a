// How fast does the CPU read page-locked memory the GPU has just written?  (profiling aid, not part of the product)
// Random 2-byte reads / 512-byte block reads from hipHostMalloc'd memory (default, non-coherent, write-combined flags) vs malloc, after a D2H copy.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t bytes = 64 << 20;
    char* d; hipMalloc(&d, bytes); hipMemset(d, 0x5a, bytes);
    struct { const char* name; unsigned flags; int kind; } kinds[] = {
        { "hipHostMallocDefault", hipHostMallocDefault, 1 }, { "hipHostMallocNonCoherent", hipHostMallocNonCoherent, 1 },
        { "hipHostMallocCoherent", hipHostMallocCoherent, 1 }, { "hipHostMallocWriteCombined", hipHostMallocWriteCombined, 1 },
        { "malloc + hipHostRegister", 0, 2 }, { "malloc (memcpy from pinned)", 0, 0 } };
    std::vector<uint32_t> idx(1 << 20);
    uint32_t s = 12345;
    for (auto& i : idx) { s = s * 1664525u + 1013904223u; i = (s >> 8) % (bytes / 512); }
    for (auto& k : kinds)
    {
        char* h = nullptr; char* stage = nullptr;
        if (k.kind == 1) { if (hipHostMalloc((void**)&h, bytes, k.flags) != hipSuccess) { printf("%s: alloc failed\n", k.name); continue; } }
        else { h = (char*)aligned_alloc(4096, bytes); memset(h, 1, bytes); if (k.kind == 2) hipHostRegister(h, bytes, hipHostRegisterDefault); }
        if (k.kind == 0) { hipHostMalloc((void**)&stage, bytes, hipHostMallocDefault); hipMemcpy(stage, d, bytes, hipMemcpyDeviceToHost); memcpy(h, stage, bytes); }
        else hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost);
        hipDeviceSynchronize();
        // 1. dependent-free random u16 reads, one per 512-byte block
        double t0 = now(); uint64_t acc = 0;
        for (uint32_t i : idx) acc += *(const uint16_t*)(h + (size_t)i * 512 + ((i * 7) & 510));
        double t1 = now();
        // 2. sequential read of everything
        uint64_t acc2 = 0; for (size_t i = 0; i < bytes; i += 8) acc2 += *(const uint64_t*)(h + i);
        double t2 = now();
        // 3. the same random reads again (now possibly cached)
        for (uint32_t i : idx) acc += *(const uint16_t*)(h + (size_t)i * 512 + ((i * 7) & 510));
        double t3 = now();
        printf("%-30s random u16 read %.1f ns each (cold), sequential %.2f GB/s, random again %.1f ns  [%llu]\n", k.name, (t1 - t0) / idx.size() * 1e9, bytes / (t2 - t1) * 1e-9,
               (t3 - t2) / idx.size() * 1e9, (unsigned long long)(acc + acc2));
        if (k.kind == 1) hipHostFree(h); else { if (k.kind == 2) hipHostUnregister(h); free(h); }
        if (stage) hipHostFree(stage);
    }
    return 0;
}
