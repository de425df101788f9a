// lasession.hip — the device-resident lookahead session behind x265hip_la_* (include/x265hip.h): what an x265 build binds at
// the seam where the reference itself batches lookahead work, CostEstimateGroup::add / finishBatch / estimateFrameCost
// (reference source/encoder/slicetype.cpp:3027-3214).
//
// x265's Lookahead keeps, per queued frame, a Lowres (common/lowres.h:152): four half-resolution planes, per-8x8-block intra
// costs and AQ factors, and per (list, distance) the vectors and costs of every motion search it has done.  The session mirrors
// exactly that in HBM — a frame SLOT holds the four padded planes, intraCost, invQscale and the (list, distance) vector store —
// so that one estimate is a handful of slot indices and a whole batch of estimates (x265 queues up to 512, slicetype.cpp:3037)
// is three launches:
//     lookahead_p_kernel      every list search of the batch (P estimates complete, B estimates one pair per list)   [lookahead.hip]
//     lookahead_pcost_kernel  P estimates whose search was done earlier
//     lookahead_bidir_kernel  every B estimate
// Results come back in ONE device-to-host copy of a batch-contiguous output block; the searched vectors are also scattered into
// the slot's store on the device, where later estimates of the same frame find them without any upload.
#include "common.h"
#include "internal.h"
#include <cstring>
#include <mutex>
#include <vector>

namespace xh {

struct LaSlot
{
    char*    buffers = nullptr;      // 4 * planeElems pixels (Lowres::buffer[0..3])
    int32_t* intraCost = nullptr;    // [ncu]
    int32_t* invQscale = nullptr;    // [ncu], meaningful when hasInvQ
    int32_t* store = nullptr;        // [2][maxDist][3 * ncu]: mvs (2 ncu) then mvCosts (ncu)
    bool     live = false, hasInvQ = false;
    std::vector<uint8_t> valid;      // [2 * maxDist]
    // searches done ahead of their request (x265hip_la_search): [2][maxDist][4 variants = bidir * 2 + (numSlices > 1)][3 * ncu], allocated on first use
    int32_t* ahead = nullptr;
    struct AheadInfo { uint8_t valid = 0; int32_t rows = 0, slices = 0; };
    std::vector<AheadInfo> aheadInfo; // [2 * maxDist * 4]
};

struct ScatterJob { const int32_t* src; int32_t* dst; };

} // namespace xh

struct x265hip_la
{
    x265hip_la_config cfg;
    int device = 0, B = 1, ncu = 0;
    hipStream_t st = nullptr;
    std::mutex lock;
    std::vector<xh::LaSlot> slots;
    std::vector<char*> wbufs;        // weighted plane sets (4 * planeElems each), handed out per batch
    int wbufsUsed = 0;
    uint16_t* mvcost = nullptr;      // centred MVD cost row of the lookahead QP
    uint32_t epoch = 0;
    // batch scratch (grown on demand)
    int capEst = 0;
    char* dOut = nullptr; char* hOut = nullptr; size_t outBytes = 0;       // per-estimate outputs, device + pinned mirror
    char* hStage = nullptr; size_t hStageBytes = 0;                        // page-locked staging block of x265hip_la_set_frame
    uint64_t* dSync = nullptr;                                               // [2 capEst][ncu]
    char* dDesc = nullptr; char* hDesc = nullptr; size_t descBytes = 0;     // pairs / pcost pairs / bframes / scatter jobs
    uint64_t statBatches = 0, statEstimates = 0, statSearches = 0;
    uint64_t statAheadLaunched = 0, statAheadUsed = 0, statSearchLaunches = 0, statSearchPairs = 0;
};

namespace xh {

int place_device(int place);                     // runtime.hip
static constexpr int kMvcostHalf = 2 * 32768;    // x265's own row: 2 * BC_MAX_MV quarter-pels each side (bitcost.h:45)

static int la_enter(x265hip_la* la)
{
    if (!la) return set_error(X265HIP_EINVAL, "x265hip_la: null session");
    return check_hip(hipSetDevice(la->device), "hipSetDevice");
}

// per-estimate output block: [mvs0 2ncu][mvc0 ncu][mvs1 2ncu][mvc1 ncu] int32, [rowSatds H] int32, [lowresCosts ncu] u16 (padded to 8 bytes)
static size_t est_block_bytes(const x265hip_la* la)
{
    size_t b = (size_t)6 * la->ncu * 4 + (size_t)la->cfg.heightInCU * 4 + (size_t)la->ncu * 2;
    return (b + 63) & ~(size_t)63;
}

static int la_grow(x265hip_la* la, int n)
{
    if (n <= la->capEst) return X265HIP_OK;
    int cap = la->capEst ? la->capEst : 16;
    while (cap < n) cap *= 2;
    if (la->dOut) (void)device_free(la->dOut);
    if (la->hOut) (void)hipHostFree(la->hOut);
    if (la->dSync) (void)device_free(la->dSync);
    if (la->dDesc) (void)device_free(la->dDesc);
    if (la->hDesc) (void)hipHostFree(la->hDesc);
    la->dOut = la->hOut = la->dDesc = la->hDesc = nullptr; la->dSync = nullptr; la->capEst = 0;
    const size_t blk = est_block_bytes(la);
    la->outBytes = blk * cap + (size_t)cap * 3 * 4 * sizeof(int64_t);        // + est words of the three kernels
    la->descBytes = (size_t)cap * (2 * sizeof(x265hip_lookahead_pair) + sizeof(x265hip_lookahead_pair) + sizeof(x265hip_lookahead_bframe) + 4 * sizeof(ScatterJob));
    int e;
    if ((e = check_hip(hipMalloc((void**)&la->dOut, la->outBytes), "la out"))) return e;
    if ((e = check_hip(hipHostMalloc((void**)&la->hOut, la->outBytes, hipHostMallocDefault), "la out pinned"))) return e;
    if ((e = check_hip(hipMalloc((void**)&la->dSync, (size_t)2 * cap * la->ncu * sizeof(uint64_t)), "la sync"))) return e;
    if ((e = check_hip(hipMemsetAsync(la->dSync, 0, (size_t)2 * cap * la->ncu * sizeof(uint64_t), la->st), "la sync zero"))) return e;
    if ((e = check_hip(hipMalloc((void**)&la->dDesc, la->descBytes), "la desc"))) return e;
    if ((e = check_hip(hipHostMalloc((void**)&la->hDesc, la->descBytes, hipHostMallocDefault), "la desc pinned"))) return e;
    la->capEst = cap;
    return X265HIP_OK;
}

__global__ __launch_bounds__(256) void la_scatter_kernel(const ScatterJob* __restrict__ jobs, int words)
{
    const ScatterJob j = jobs[blockIdx.y];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < words; i += gridDim.x * 256)
        j.dst[i] = j.src[i];
}

static inline int32_t* store_of(const x265hip_la* la, const LaSlot& s, int list, int dist)
{
    return s.store + ((size_t)list * la->cfg.maxDist + dist) * 3 * la->ncu;
}

static inline int ahead_index(const x265hip_la* la, int list, int dist, int bidir, int slices)
{
    return ((list * la->cfg.maxDist + dist) * 2 + (bidir ? 1 : 0)) * 2 + (slices > 1 ? 1 : 0);
}
static inline int32_t* ahead_of(const x265hip_la* la, const LaSlot& s, int idx) { return s.ahead + (size_t)idx * 3 * la->ncu; }

} // namespace xh

using namespace xh;

extern "C" {

x265hip_la* x265hip_la_create(const x265hip_la_config* cfg)
{
    if (ensure_device()) return nullptr;
    if (!cfg || !valid_depth(cfg->depth) || cfg->width < 8 || cfg->lines < 8 || cfg->stride < cfg->width || cfg->planeElems < cfg->stride * cfg->lines ||
        cfg->planeElems >= (1 << 24) || cfg->stride >= (1 << 23) || cfg->padOffset < 0 || cfg->padOffset >= cfg->planeElems ||
        cfg->widthInCU < 1 || cfg->heightInCU < 1 || cfg->maxDist < 2 || cfg->maxDist > 18 || cfg->numSlots < 2 || cfg->numSlots > 4096)
    {
        set_error(X265HIP_EINVAL, "x265hip_la_create: bad configuration");
        return nullptr;
    }
    x265hip_la* la = new x265hip_la;
    la->cfg = *cfg;
    la->B = cfg->depth == 8 ? 1 : 2;
    la->ncu = cfg->widthInCU * cfg->heightInCU;
    (void)hipGetDevice(&la->device);
    la->slots.resize(cfg->numSlots);
    for (auto& s : la->slots) { s.valid.assign(2 * cfg->maxDist, 0); s.aheadInfo.assign(2 * cfg->maxDist * 4, xh::LaSlot::AheadInfo()); }
    bool ok = hipStreamCreateWithFlags(&la->st, hipStreamNonBlocking) == hipSuccess;
    // BitCost row of the lookahead QP (X265_LOOKAHEAD_QP = 12 + 6 * (depth - 8), common.h:232)
    std::vector<uint16_t> tab(2 * kMvcostHalf + 1);
    ok = ok && !x265hip_mvcost_table(12 + 6 * (cfg->depth - 8), cfg->depth, tab.data(), kMvcostHalf);
    ok = ok && hipMalloc((void**)&la->mvcost, tab.size() * 2) == hipSuccess &&
         hipMemcpy(la->mvcost, tab.data(), tab.size() * 2, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok)
    {
        set_error(X265HIP_EHIP, "x265hip_la_create: stream / cost table");
        x265hip_la_destroy(la);
        return nullptr;
    }
    return la;
}

x265hip_la* x265hip_la_create_at(int place, const x265hip_la_config* cfg)
{
    const int dev = place >= 0 ? xh::place_device(place) : -1;
    if (dev < 0) { set_error(X265HIP_EINVAL, "x265hip_la_create_at: no place %d (x265hip_places)", place); return nullptr; }
    int cur = 0;
    const bool had = hipGetDevice(&cur) == hipSuccess;
    if (hipSetDevice(dev) != hipSuccess) { set_error(X265HIP_EHIP, "x265hip_la_create_at: hipSetDevice(%d)", dev); return nullptr; }
    x265hip_la* la = x265hip_la_create(cfg);
    if (had) (void)hipSetDevice(cur);
    return la;
}

void x265hip_la_destroy(x265hip_la* la)
{
    if (!la) return;
    (void)hipSetDevice(la->device);
    if (la->st) (void)hipStreamSynchronize(la->st);
    for (auto& s : la->slots)
    {
        if (s.buffers) (void)device_free(s.buffers);
        if (s.intraCost) (void)device_free(s.intraCost);
        if (s.invQscale) (void)device_free(s.invQscale);
        if (s.store) (void)device_free(s.store);
        if (s.ahead) (void)device_free(s.ahead);
    }
    for (char* w : la->wbufs) (void)device_free(w);
    if (la->mvcost) (void)device_free(la->mvcost);
    if (la->dOut) (void)device_free(la->dOut);
    if (la->hOut) (void)hipHostFree(la->hOut);
    if (la->dSync) (void)device_free(la->dSync);
    if (la->dDesc) (void)device_free(la->dDesc);
    if (la->hDesc) (void)hipHostFree(la->hDesc);
    if (la->hStage) (void)hipHostFree(la->hStage);
    if (la->st) (void)hipStreamDestroy(la->st);
    delete la;
}

int x265hip_la_set_frame(x265hip_la* la, int slot, const void* buffers, const int32_t* intraCost, const int32_t* invQscale)
{
    int e = la_enter(la);
    if (e) return e;
    if (slot < 0 || slot >= (int)la->slots.size() || !buffers || !intraCost)
        return set_error(X265HIP_EINVAL, "x265hip_la_set_frame: slot %d", slot);
    std::lock_guard<std::mutex> g(la->lock);
    LaSlot& s = la->slots[slot];
    const size_t planeBytes = (size_t)4 * la->cfg.planeElems * la->B, cuBytes = (size_t)la->ncu * 4;
    if (!s.buffers)
    {
        if ((e = check_hip(hipMalloc((void**)&s.buffers, planeBytes), "la slot planes"))) return e;
        if ((e = check_hip(hipMalloc((void**)&s.intraCost, cuBytes), "la slot intraCost"))) return e;
        if ((e = check_hip(hipMalloc((void**)&s.invQscale, cuBytes), "la slot invQscale"))) return e;
        if ((e = check_hip(hipMalloc((void**)&s.store, (size_t)2 * la->cfg.maxDist * 3 * cuBytes), "la slot vectors"))) return e;
    }
    // the caller's arrays are pageable: a direct hipMemcpyAsync page-locks and unlocks them per call (an ioctl and an munmap each: 2.7 % + 0.9 % of the
    // bound encoder's CPU time in round 3's profile); they go through the session's own page-locked staging block instead
    const size_t stageBytes = planeBytes + 2 * cuBytes;
    if (la->hStageBytes < stageBytes)
    {
        if (la->hStage) (void)hipHostFree(la->hStage);
        la->hStage = nullptr; la->hStageBytes = 0;
        if ((e = check_hip(hipHostMalloc((void**)&la->hStage, stageBytes, hipHostMallocDefault), "la staging"))) return e;
        la->hStageBytes = stageBytes;
    }
    memcpy(la->hStage, buffers, planeBytes);
    memcpy(la->hStage + planeBytes, intraCost, cuBytes);
    if (invQscale) memcpy(la->hStage + planeBytes + cuBytes, invQscale, cuBytes);
    if ((e = check_hip(hipMemcpyAsync(s.buffers, la->hStage, planeBytes, hipMemcpyHostToDevice, la->st), "la planes h2d"))) return e;
    if ((e = check_hip(hipMemcpyAsync(s.intraCost, la->hStage + planeBytes, cuBytes, hipMemcpyHostToDevice, la->st), "la intraCost h2d"))) return e;
    s.hasInvQ = invQscale != nullptr;
    if (invQscale && (e = check_hip(hipMemcpyAsync(s.invQscale, la->hStage + planeBytes + cuBytes, cuBytes, hipMemcpyHostToDevice, la->st), "la invQscale h2d"))) return e;
    // the staging block is reused by the next call
    if ((e = check_hip(hipStreamSynchronize(la->st), "la set_frame sync"))) return e;
    std::fill(s.valid.begin(), s.valid.end(), 0);            // Lowres::init resets every search (lowres.cpp:289-295)
    std::fill(s.aheadInfo.begin(), s.aheadInfo.end(), LaSlot::AheadInfo());
    s.live = true;
    return X265HIP_OK;
}

int x265hip_la_put_vectors(x265hip_la* la, int slot, int list, int dist, const int32_t* mvs, const int32_t* mvCosts)
{
    int e = la_enter(la);
    if (e) return e;
    if (slot < 0 || slot >= (int)la->slots.size() || list < 0 || list > 1 || dist < 0 || dist >= la->cfg.maxDist || !mvs || !mvCosts || !la->slots[slot].live)
        return set_error(X265HIP_EINVAL, "x265hip_la_put_vectors: slot %d list %d dist %d", slot, list, dist);
    std::lock_guard<std::mutex> g(la->lock);
    LaSlot& s = la->slots[slot];
    int32_t* d = store_of(la, s, list, dist);
    // small (96 KB at 1080p): the runtime copies them through its own staging buffers, no page-locking
    if ((e = check_hip(hipMemcpyAsync(d, mvs, (size_t)la->ncu * 8, hipMemcpyHostToDevice, la->st), "la vectors h2d"))) return e;
    if ((e = check_hip(hipMemcpyAsync(d + 2 * la->ncu, mvCosts, (size_t)la->ncu * 4, hipMemcpyHostToDevice, la->st), "la vector costs h2d"))) return e;
    if ((e = check_hip(hipStreamSynchronize(la->st), "la put_vectors sync"))) return e;
    s.valid[list * la->cfg.maxDist + dist] = 1;
    return X265HIP_OK;
}

int x265hip_la_has_vectors(x265hip_la* la, int slot, int list, int dist)
{
    if (!la || slot < 0 || slot >= (int)la->slots.size() || list < 0 || list > 1 || dist < 0 || dist >= la->cfg.maxDist) return 0;
    std::lock_guard<std::mutex> g(la->lock);
    return la->slots[slot].live && la->slots[slot].valid[list * la->cfg.maxDist + dist];
}

int x265hip_la_weights_analyse(x265hip_la* la, int slotB, int slotRef, uint64_t fencSsd, uint64_t fencSum, uint64_t refSsd, uint64_t refSum,
                               x265hip_weight_param* chosen, int* isWeighted, int* weightedId)
{
    int e = la_enter(la);
    if (e) return e;
    if (slotB < 0 || slotB >= (int)la->slots.size() || slotRef < 0 || slotRef >= (int)la->slots.size() || !la->slots[slotB].live || !la->slots[slotRef].live ||
        !chosen || !isWeighted || !weightedId)
        return set_error(X265HIP_EINVAL, "x265hip_la_weights_analyse: slots %d %d", slotB, slotRef);
    std::lock_guard<std::mutex> g(la->lock);
    if (la->wbufsUsed == (int)la->wbufs.size())
    {
        char* w = nullptr;
        if ((e = check_hip(hipMalloc((void**)&w, (size_t)4 * la->cfg.planeElems * la->B), "la weighted planes"))) return e;
        la->wbufs.push_back(w);
    }
    char* wbuf = la->wbufs[la->wbufsUsed];
    const x265hip_la_config& c = la->cfg;
    const LaSlot& fb = la->slots[slotB];
    const LaSlot& fr = la->slots[slotRef];
    const int paddedLines = (int)(c.planeElems / c.stride);
    e = x265hip_lookahead_weights_analyse(c.depth, fb.buffers + c.padOffset * la->B, fr.buffers, c.planeElems, c.stride, c.padOffset, paddedLines, c.width,
                                          c.lines, fb.intraCost, fencSsd, fencSum, refSsd, refSum, wbuf, chosen, isWeighted, la->st);
    if (e) return e;
    *weightedId = -1;
    if (*isWeighted)
        *weightedId = la->wbufsUsed++;        // stays reserved until the next estimate batch has used it
    return X265HIP_OK;
}

int x265hip_la_estimate_batch(x265hip_la* la, x265hip_la_estimate* est, int n, int numRowsPerSlice, int numSlices)
{
    return x265hip_la_estimate_batch_ahead(la, est, n, numRowsPerSlice, numSlices, nullptr, 0);
}

int x265hip_la_has_ahead(x265hip_la* la, int slot, int list, int dist, int bidir, int numRowsPerSlice, int numSlices)
{
    if (!la || slot < 0 || slot >= (int)la->slots.size() || list < 0 || list > 1 || dist < 1 || dist >= la->cfg.maxDist) return 0;
    std::lock_guard<std::mutex> g(la->lock);
    const LaSlot& s = la->slots[slot];
    const LaSlot::AheadInfo& a = s.aheadInfo[ahead_index(la, list, dist, bidir, numSlices)];
    return s.live && a.valid && a.rows == numRowsPerSlice && a.slices == numSlices;
}

int x265hip_la_estimate_batch_ahead(x265hip_la* la, x265hip_la_estimate* est, int n, int numRowsPerSlice, int numSlices,
                                    const x265hip_la_search* ahead, int nAhead)
{
    int e = la_enter(la);
    if (e) return e;
    if (n < 0 || (n && !est) || nAhead < 0 || (nAhead && !ahead) || numSlices < 1 || numRowsPerSlice < 1 ||
        (long long)numRowsPerSlice * (numSlices - 1) >= la->cfg.heightInCU)
        return set_error(X265HIP_EINVAL, "x265hip_la_estimate_batch: n %d ahead %d slices %d x %d", n, nAhead, numSlices, numRowsPerSlice);
    std::lock_guard<std::mutex> g(la->lock);
    // the weighted plane sets of this batch go back to the pool when the call ends; a big batch on a fade must not pin its peak for the session's life
    struct Release
    {
        x265hip_la* l;
        ~Release()
        {
            l->wbufsUsed = 0;
            while (l->wbufs.size() > 8) { (void)device_free(l->wbufs.back()); l->wbufs.pop_back(); }
        }
    } release{ la };
    if (!n && !nAhead) return X265HIP_OK;
    if ((e = la_grow(la, n + nAhead))) return e;
    const x265hip_la_config& c = la->cfg;
    const int ncu = la->ncu, H = c.heightInCU, W = c.widthInCU, B = la->B;
    const size_t blk = est_block_bytes(la);
    // descriptor block layout (host mirror hDesc, same offsets on the device)
    x265hip_lookahead_pair* hPairs = (x265hip_lookahead_pair*)la->hDesc;                      // [2 cap] searches
    x265hip_lookahead_pair* hPcost = hPairs + 2 * la->capEst;                                 // [cap]
    x265hip_lookahead_bframe* hBf = (x265hip_lookahead_bframe*)(hPcost + la->capEst);         // [cap]
    ScatterJob* hSc = (ScatterJob*)(hBf + la->capEst);                                        // [4 cap]
    auto dev_of = [&](const void* h) { return la->dDesc + ((const char*)h - la->hDesc); };
    // est words live in the tail of dOut (3 * cap * 4 int64 reserved): [nPairs][4] search pairs, then [nPcost][4], then [nBf][2]; the worst
    // case (all B, both lists searched, + the searches ahead) is 8 + 2 words per estimate and 4 per search ahead
    int64_t* dEstSearch = (int64_t*)(la->dOut + blk * la->capEst);
    int nPairs = 0, nPcost = 0, nBf = 0, nSc = 0;
    std::vector<int> pairOfEst(n, -1), pcostOfEst(n, -1), bfOfEst(n, -1);
    // vectors this call produces: they become visible to later calls (valid[] / aheadInfo[]) only after the final synchronise succeeded, and
    // a later estimate of THIS call that reuses them reads the producer's block directly (the slot store is filled by the scatter at the end)
    struct Produced { int slot, list, dist; const int32_t* where; };
    std::vector<Produced> produced;
    struct AheadDone { int slot, idx, rows, slices; };
    std::vector<AheadDone> aheadDone;
    auto slice_geom = [&](int rows, int slices) { return (rows == numRowsPerSlice && slices == numSlices) ? 0 : (rows | slices << 16); };
    for (int i = 0; i < n; i++)
    {
        x265hip_la_estimate& q = est[i];
        const bool bidir = q.p1 != q.b;
        if (q.b < 0 || q.b >= (int)la->slots.size() || q.p0 < 0 || q.p0 >= (int)la->slots.size() || q.p1 < 0 || q.p1 >= (int)la->slots.size() ||
            !la->slots[q.b].live || !la->slots[q.p0].live || !la->slots[q.p1].live || q.dist0 < 1 || q.dist0 >= c.maxDist ||
            (bidir && (q.dist1 < 1 || q.dist1 >= c.maxDist)) || !q.lowresCosts || !q.rowSatds ||
            (q.search0 && (!q.mvs0 || !q.mvCosts0)) || (bidir && q.search1 && (!q.mvs1 || !q.mvCosts1)) ||
            q.weightedId >= la->wbufsUsed || (q.weightedId >= 0 && !q.search0))
            return set_error(X265HIP_EINVAL, "x265hip_la_estimate_batch: estimate %d (b %d p0 %d p1 %d)", i, q.b, q.p0, q.p1);
        LaSlot& fb = la->slots[q.b];
        char* outBlk = la->dOut + blk * i;
        int32_t* oMvs[2] = { (int32_t*)outBlk, (int32_t*)outBlk + 3 * ncu };
        int32_t* oRows = (int32_t*)outBlk + 6 * ncu;
        uint16_t* oLc = (uint16_t*)(oRows + H);
        const char* fencOrg = fb.buffers + c.padOffset * B;
        const int32_t* invQ = fb.hasInvQ ? fb.invQscale : nullptr;
        const int32_t* useMvs[2] = { nullptr, nullptr };
        bool searchedHere0 = false;
        for (int l = 0; l < (bidir ? 2 : 1); l++)
        {
            const int dist = l ? q.dist1 : q.dist0, search = l ? q.search1 : q.search0;
            int32_t* st = store_of(la, fb, l, dist);
            if (!search)
            {
                const int32_t* where = fb.valid[l * c.maxDist + dist] ? st : nullptr;
                for (const Produced& pr : produced)
                    if (pr.slot == q.b && pr.list == l && pr.dist == dist) where = pr.where;
                if (!where)
                    return set_error(X265HIP_EINVAL, "x265hip_la_estimate_batch: estimate %d reuses list %d distance %d which the session has not seen "
                                     "(x265hip_la_put_vectors)", i, l, dist);
                useMvs[l] = where;
                continue;
            }
            const LaSlot::AheadInfo& ai = fb.aheadInfo[ahead_index(la, l, dist, bidir, numSlices)];
            if (fb.ahead && ai.valid && ai.rows == numRowsPerSlice && ai.slices == numSlices)
            {
                // searched ahead of this request, same variant: the estimate gets the vectors as if it had searched
                const int32_t* src = ahead_of(la, fb, ahead_index(la, l, dist, bidir, numSlices));
                useMvs[l] = src;
                hSc[nSc].src = src; hSc[nSc].dst = st; nSc++;
                hSc[nSc].src = src; hSc[nSc].dst = oMvs[l]; nSc++;
                produced.push_back(Produced{ q.b, l, dist, src });
                la->statAheadUsed++;
                continue;
            }
            const LaSlot& fr = la->slots[l ? q.p1 : q.p0];
            const char* refBuf = (!l && q.weightedId >= 0) ? la->wbufs[q.weightedId] : fr.buffers;
            x265hip_lookahead_pair& p = hPairs[nPairs];
            memset(&p, 0, sizeof(p));
            p.fenc = fencOrg;
            p.ref = refBuf + c.padOffset * B;
            p.intraCost = fb.intraCost;
            p.mvs = oMvs[l];
            p.mvCosts = oMvs[l] + 2 * ncu;
            p.lowresCosts = oLc;
            p.rowSatds = oRows;
            p.sync = la->dSync + (size_t)nPairs * ncu;
            p.invQscale = invQ;
            p.bidirList = bidir ? 1 : 0;
            if (!bidir) { pairOfEst[i] = nPairs; searchedHere0 = true; }
            nPairs++;
            useMvs[l] = oMvs[l];
            hSc[nSc].src = oMvs[l];
            hSc[nSc].dst = st;
            nSc++;
            produced.push_back(Produced{ q.b, l, dist, oMvs[l] });
            la->statSearches++;
        }
        if (bidir)
        {
            x265hip_lookahead_bframe& f = hBf[nBf];
            f.fenc = fencOrg;
            f.ref0 = la->slots[q.p0].buffers + c.padOffset * B;       // NOTE: the weighted reference is not used for bidir (slicetype.cpp:3322)
            f.ref1 = la->slots[q.p1].buffers + c.padOffset * B;
            f.mvs0 = useMvs[0]; f.mvCosts0 = useMvs[0] + 2 * ncu;
            f.mvs1 = useMvs[1]; f.mvCosts1 = useMvs[1] + 2 * ncu;
            f.lowresCosts = oLc;
            f.rowSatds = oRows;
            f.invQscale = invQ;
            bfOfEst[i] = nBf++;
        }
        else if (!searchedHere0)
        {
            x265hip_lookahead_pair& p = hPcost[nPcost];
            memset(&p, 0, sizeof(p));
            p.intraCost = fb.intraCost;
            p.mvCosts = (int32_t*)useMvs[0] + 2 * ncu;
            p.lowresCosts = oLc;
            p.rowSatds = oRows;
            p.invQscale = invQ;
            pcostOfEst[i] = nPcost++;
        }
    }
    // the searches ahead: same kernel, same launch; vectors go straight into the slot's ahead store, the rest of a pair's outputs into a scratch block
    for (int j = 0; j < nAhead; j++)
    {
        const x265hip_la_search& a = ahead[j];
        if (a.b < 0 || a.b >= (int)la->slots.size() || a.ref < 0 || a.ref >= (int)la->slots.size() || !la->slots[a.b].live || !la->slots[a.ref].live ||
            a.list < 0 || a.list > 1 || a.dist < 1 || a.dist >= c.maxDist || a.numSlices < 1 || a.numRowsPerSlice < 1 || a.numRowsPerSlice > 0xffff ||
            a.numSlices > 0x7fff || (long long)a.numRowsPerSlice * (a.numSlices - 1) >= H || a.weightedId >= la->wbufsUsed || (a.weightedId >= 0 && a.list))
            return set_error(X265HIP_EINVAL, "x265hip_la_estimate_batch: search ahead %d (b %d ref %d list %d dist %d)", j, a.b, a.ref, a.list, a.dist);
        LaSlot& fb = la->slots[a.b];
        const int idx = ahead_index(la, a.list, a.dist, a.bidir, a.numSlices);
        if (fb.valid[a.list * c.maxDist + a.dist] || (fb.aheadInfo[idx].valid && fb.aheadInfo[idx].rows == a.numRowsPerSlice && fb.aheadInfo[idx].slices == a.numSlices))
            continue;                          // already there
        bool dup = false;
        for (const AheadDone& d : aheadDone) dup = dup || (d.slot == a.b && d.idx == idx);
        if (dup) continue;
        if (!fb.ahead && (e = check_hip(hipMalloc((void**)&fb.ahead, (size_t)2 * c.maxDist * 4 * 3 * ncu * 4), "la slot searches ahead"))) return e;
        char* scratchBlk = la->dOut + blk * (n + j);
        int32_t* oRows = (int32_t*)scratchBlk + 6 * ncu;
        int32_t* dst = ahead_of(la, fb, idx);
        x265hip_lookahead_pair& p = hPairs[nPairs];
        memset(&p, 0, sizeof(p));
        p.fenc = fb.buffers + c.padOffset * B;
        p.ref = ((!a.list && a.weightedId >= 0) ? la->wbufs[a.weightedId] : la->slots[a.ref].buffers) + c.padOffset * B;
        p.intraCost = fb.intraCost;
        p.mvs = dst;
        p.mvCosts = dst + 2 * ncu;
        p.lowresCosts = (uint16_t*)(oRows + H);
        p.rowSatds = oRows;
        p.sync = la->dSync + (size_t)nPairs * ncu;
        p.invQscale = fb.hasInvQ ? fb.invQscale : nullptr;
        p.bidirList = a.bidir ? 1 : 0;
        p.sliceGeom = slice_geom(a.numRowsPerSlice, a.numSlices);
        nPairs++;
        aheadDone.push_back(AheadDone{ a.b, idx, a.numRowsPerSlice, a.numSlices });
        la->statAheadLaunched++;
    }
    int64_t* dEstPcost = dEstSearch + (size_t)nPairs * 4;
    int64_t* dEstBf = dEstPcost + (size_t)nPcost * 4;
    if ((e = check_hip(hipMemcpyAsync(la->dDesc, la->hDesc, la->descBytes, hipMemcpyHostToDevice, la->st), "la desc h2d"))) return e;
    if (++la->epoch == 0)
    {
        // 2^32 launches later: start over on a clean scratch
        if ((e = check_hip(hipMemsetAsync(la->dSync, 0, (size_t)2 * la->capEst * ncu * sizeof(uint64_t), la->st), "la sync zero"))) return e;
        la->epoch = 1;
    }
    DevSpan spanSearch(X265HIP_CLK_LA_SEARCH, la->st);
    if (nPairs)
    {
        if ((e = x265hip_lookahead_cost_p_batch(c.depth, (const x265hip_lookahead_pair*)dev_of(hPairs), nPairs, c.stride, c.planeElems, W, H, numRowsPerSlice,
                                                numSlices, la->mvcost + kMvcostHalf, la->epoch, dEstSearch, la->st)))
            return e;
        la->statSearchLaunches++;
        la->statSearchPairs += nPairs;
    }
    spanSearch.end();
    DevSpan spanOther(X265HIP_CLK_LA_OTHER, la->st);
    if (nPcost && (e = x265hip_lookahead_pcost_batch((const x265hip_lookahead_pair*)dev_of(hPcost), nPcost, W, H, dEstPcost, la->st)))
        return e;
    if (nBf && (e = x265hip_lookahead_bidir_batch(c.depth, (const x265hip_lookahead_bframe*)dev_of(hBf), nBf, c.stride, c.planeElems, W, H, dEstBf, la->st)))
        return e;
    if (nSc)
    {
        hipLaunchKernelGGL(la_scatter_kernel, dim3(8, nSc), dim3(256), 0, la->st, (const ScatterJob*)dev_of(hSc), 3 * ncu);
        XH_LAUNCH_CHECK("la_scatter_kernel");
    }
    spanOther.end();
    const size_t used = blk * n, estWords = (size_t)(nPairs + nPcost) * 4 + (size_t)nBf * 2;
    if (used && (e = check_hip(hipMemcpyAsync(la->hOut, la->dOut, used, hipMemcpyDeviceToHost, la->st), "la out d2h"))) return e;
    if ((e = check_hip(hipMemcpyAsync(la->hOut + blk * la->capEst, dEstSearch, estWords * 8, hipMemcpyDeviceToHost, la->st), "la est d2h"))) return e;
    if ((e = check_hip(hipStreamSynchronize(la->st), "la batch sync"))) return e;
    if (nPairs) spanSearch.commit();
    if (nPcost || nBf || nSc) spanOther.commit();
    const int64_t* hEst = (const int64_t*)(la->hOut + blk * la->capEst);
    const int64_t* hEstPcost = hEst + (size_t)nPairs * 4;
    const int64_t* hEstBf = hEstPcost + (size_t)nPcost * 4;
    for (int i = 0; i < nPairs; i++)
        if (hEst[4 * i + 3])
            return set_error(X265HIP_EHIP, "x265hip_la_estimate_batch: row handshake of search %d timed out", i);
    // everything arrived: the produced vectors are now the session's
    for (const Produced& pr : produced)
        la->slots[pr.slot].valid[pr.list * c.maxDist + pr.dist] = 1;
    for (const AheadDone& d : aheadDone)
    {
        LaSlot::AheadInfo& ai = la->slots[d.slot].aheadInfo[d.idx];
        ai.valid = 1; ai.rows = d.rows; ai.slices = d.slices;
    }
    for (int i = 0; i < n; i++)
    {
        x265hip_la_estimate& q = est[i];
        const bool bidir = q.p1 != q.b;
        const char* outBlk = la->hOut + blk * i;
        const int32_t* oMvs[2] = { (const int32_t*)outBlk, (const int32_t*)outBlk + 3 * ncu };
        const int32_t* oRows = (const int32_t*)outBlk + 6 * ncu;
        if (q.search0) { memcpy(q.mvs0, oMvs[0], (size_t)ncu * 8); memcpy(q.mvCosts0, oMvs[0] + 2 * ncu, (size_t)ncu * 4); }
        if (bidir && q.search1) { memcpy(q.mvs1, oMvs[1], (size_t)ncu * 8); memcpy(q.mvCosts1, oMvs[1] + 2 * ncu, (size_t)ncu * 4); }
        memcpy(q.rowSatds, oRows, (size_t)H * 4);
        memcpy(q.lowresCosts, oRows + H, (size_t)ncu * 2);
        if (bidir)
        {
            q.costEst = hEstBf[2 * bfOfEst[i]];
            q.costEstAq = hEstBf[2 * bfOfEst[i] + 1];
            q.intraMbs = 0;
        }
        else
        {
            const int64_t* w = pairOfEst[i] >= 0 ? hEst + 4 * pairOfEst[i] : hEstPcost + 4 * pcostOfEst[i];
            q.costEst = w[0];
            q.costEstAq = w[1];
            q.intraMbs = (int32_t)w[2];
        }
    }
    if (n) la->statBatches++;
    la->statEstimates += n;
    return X265HIP_OK;
}

int x265hip_la_stats(x265hip_la* la, uint64_t* batches, uint64_t* estimates, uint64_t* searches)
{
    if (!la) return set_error(X265HIP_EINVAL, "x265hip_la_stats: null session");
    std::lock_guard<std::mutex> g(la->lock);
    if (batches) *batches = la->statBatches;
    if (estimates) *estimates = la->statEstimates;
    if (searches) *searches = la->statSearches;
    return X265HIP_OK;
}

int x265hip_la_stats_ahead(x265hip_la* la, uint64_t* launchedAhead, uint64_t* usedAhead, uint64_t* searchLaunches, uint64_t* searchPairs)
{
    if (!la) return set_error(X265HIP_EINVAL, "x265hip_la_stats_ahead: null session");
    std::lock_guard<std::mutex> g(la->lock);
    if (launchedAhead) *launchedAhead = la->statAheadLaunched;
    if (usedAhead) *usedAhead = la->statAheadUsed;
    if (searchLaunches) *searchLaunches = la->statSearchLaunches;
    if (searchPairs) *searchPairs = la->statSearchPairs;
    return X265HIP_OK;
}

} // extern "C"
