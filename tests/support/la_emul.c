/* la_emul.c — TEST INFRASTRUCTURE ONLY.  The lookahead-session part of the C ABI (include/x265hip.h, x265hip_la_*) implemented on the CPU
 * with the oracle's restatement (oracle/x265_oracle*.c), built into tests/support/libx265hip_emul.so.  It exists so that the x265-side binding
 * (x265_amd/host/x265_hip_lookahead.cpp: slot management, batching, what is written back into x265's Lowres arrays) can be proven
 * byte-identical against the unmodified reference encoder in this GPU-less container: oracle/_ref/x265_emul_8bit = reference objects + the
 * binding + THIS library, compared with oracle/_ref/x265_8bit by tests/test_lookahead_binding.py.  It is never linked into, loaded by or
 * shipped with the product (x265_amd/libx265hip.so); on a GPU the same binding runs on the HIP kernels (tests/test_x265_dropin.py). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include "../../include/x265hip.h"
#include "../../oracle/x265_oracle.h"

/* oracle/x265_oracle_pix.inc (LookaheadTLD::weightsAnalyse); exported by the oracle library, not listed in its header */
int orc_weights_analyse_8(const uint8_t* fenc, const uint8_t* const refBuf[4], uint8_t* const outBuf[4], intptr_t stride, intptr_t padOffset, int paddedLines,
                          int width, int lines, const int32_t* intraCost, uint64_t fencSsd, uint64_t fencSum, uint64_t refSsd, uint64_t refSum,
                          int32_t chosen[3], int depth);
int orc_weights_analyse_16(const uint16_t* fenc, const uint16_t* const refBuf[4], uint16_t* const outBuf[4], intptr_t stride, intptr_t padOffset, int paddedLines,
                           int width, int lines, const int32_t* intraCost, uint64_t fencSsd, uint64_t fencSum, uint64_t refSsd, uint64_t refSum,
                           int32_t chosen[3], int depth);

typedef struct slot
{
    void*    buffers;
    int32_t* intraCost;
    int32_t* invQscale;
    int32_t* store;        /* [2][maxDist][3 ncu] */
    uint8_t  valid[2 * 18];
    int      live, hasInvQ;
    int32_t* ahead;        /* [2][maxDist][4][3 ncu]: searches done ahead of their request (x265hip_la_search) */
    uint8_t  aheadValid[2 * 18 * 4];
    int32_t  aheadRows[2 * 18 * 4], aheadSlices[2 * 18 * 4];
} slot;

struct x265hip_la
{
    x265hip_la_config c;
    int B, ncu;
    slot* slots;
    void* wbuf[1024];        /* as many weighted plane sets as one batch can ask for (x265 queues up to 512 estimates) */
    int wbufsUsed;
    uint16_t* mvcost;
    uint64_t batches, estimates, searches, aheadLaunched, aheadUsed, searchLaunches, searchPairs;
};

static char g_err[256] = "";
/* X265HIP_EMUL_FAIL=<entry point>[:<n>]: the named entry point fails from its n-th call on (n = 1 when omitted), the way a device that runs out of
 * memory or is lost fails — the bindings must then carry on with the reference's own host code and still produce the reference's bytes
 * (tests/test_fallback.py; SURVEY.md §8b "Errors").  Entry points: la_create, la_set_frame, la_weights, la_put_vectors, la_estimate, refpic_create,
 * refpic_reset, rows_final, source_energy, srcpic_create, srcpic_upload, sadsurf_attach, cuserve_open, cuserve_submit, cuserve_submit_sao, cuserve_job, saostats_job (from its n-th job
 * on the service accepts jobs and never does them: the ready words stay as they are and x265hip_cuserve_poke reports the failure — a device that
 * dies while a CU is in flight). */
static int fail_now(const char* name)
{
    static const char* spec = NULL;
    static int asked = 0, from = 1, calls = 0;
    static char want[32];
    if (!asked)
    {
        asked = 1;
        spec = getenv("X265HIP_EMUL_FAIL");
        if (spec)
        {
            const char* c = strchr(spec, ':');
            size_t n = c ? (size_t)(c - spec) : strlen(spec);
            if (n > 31) n = 31;
            memcpy(want, spec, n); want[n] = 0;
            if (c) from = atoi(c + 1);
        }
    }
    if (!spec || strcmp(want, name))
        return 0;
    if (__atomic_add_fetch(&calls, 1, __ATOMIC_RELAXED) < from)
        return 0;
    snprintf(g_err, sizeof(g_err), "emulated failure of %s (X265HIP_EMUL_FAIL)", name);
    return 1;
}
int x265hip_device_count(void) { return 1; }
int x265hip_init(int device) { (void)device; return 0; }
const char* x265hip_last_error(void) { return g_err; }

static int g_nPlaces;
x265hip_la* x265hip_la_create(const x265hip_la_config* cfg);
x265hip_la* x265hip_la_create_at(int place, const x265hip_la_config* cfg)
{
    if (place < 0 || place >= g_nPlaces) { snprintf(g_err, sizeof(g_err), "emul: no place %d", place); return NULL; }
    return x265hip_la_create(cfg);
}
x265hip_la* x265hip_la_create(const x265hip_la_config* cfg)
{
    if (fail_now("la_create")) return NULL;
    x265hip_la* la = (x265hip_la*)calloc(1, sizeof(*la));
    la->c = *cfg;
    la->B = cfg->depth == 8 ? 1 : 2;
    la->ncu = cfg->widthInCU * cfg->heightInCU;
    la->slots = (slot*)calloc(cfg->numSlots, sizeof(slot));
    la->mvcost = (uint16_t*)malloc((4 * 32768 + 1) * sizeof(uint16_t));
    orc_mvcost_table(12 + 6 * (cfg->depth - 8), cfg->depth, la->mvcost);
    return la;
}

void x265hip_la_destroy(x265hip_la* la)
{
    if (!la) return;
    for (int i = 0; i < la->c.numSlots; i++)
    {
        free(la->slots[i].buffers); free(la->slots[i].intraCost); free(la->slots[i].invQscale); free(la->slots[i].store);
        free(la->slots[i].ahead);
    }
    for (int i = 0; i < 1024; i++) free(la->wbuf[i]);
    free(la->slots); free(la->mvcost); free(la);
}

int x265hip_la_set_frame(x265hip_la* la, int slotIdx, const void* buffers, const int32_t* intraCost, const int32_t* invQscale)
{
    if (fail_now("la_set_frame")) return X265HIP_EHIP;
    slot* s = &la->slots[slotIdx];
    const size_t pb = (size_t)4 * la->c.planeElems * la->B, cb = (size_t)la->ncu * 4;
    if (!s->buffers)
    {
        s->buffers = malloc(pb); s->intraCost = (int32_t*)malloc(cb); s->invQscale = (int32_t*)malloc(cb);
        s->store = (int32_t*)malloc((size_t)2 * la->c.maxDist * 3 * cb);
    }
    memcpy(s->buffers, buffers, pb);
    memcpy(s->intraCost, intraCost, cb);
    s->hasInvQ = invQscale != NULL;
    if (invQscale) memcpy(s->invQscale, invQscale, cb);
    memset(s->valid, 0, sizeof(s->valid));
    memset(s->aheadValid, 0, sizeof(s->aheadValid));
    s->live = 1;
    return 0;
}

static int32_t* store_of(x265hip_la* la, slot* s, int list, int dist) { return s->store + ((size_t)list * la->c.maxDist + dist) * 3 * la->ncu; }

int x265hip_la_put_vectors(x265hip_la* la, int slotIdx, int list, int dist, const int32_t* mvs, const int32_t* mvCosts)
{
    if (fail_now("la_put_vectors")) return X265HIP_EHIP;
    slot* s = &la->slots[slotIdx];
    int32_t* d = store_of(la, s, list, dist);
    memcpy(d, mvs, (size_t)la->ncu * 8);
    memcpy(d + 2 * la->ncu, mvCosts, (size_t)la->ncu * 4);
    s->valid[list * la->c.maxDist + dist] = 1;
    return 0;
}

int x265hip_la_has_vectors(x265hip_la* la, int slotIdx, int list, int dist) { return la->slots[slotIdx].live && la->slots[slotIdx].valid[list * la->c.maxDist + dist]; }

#define PLANES(T, buf, out) do { for (int k_ = 0; k_ < 4; k_++) (out)[k_] = (const T*)(buf) + (size_t)k_ * la->c.planeElems + la->c.padOffset; } while (0)

int x265hip_la_weights_analyse(x265hip_la* la, int slotB, int slotRef, uint64_t fencSsd, uint64_t fencSum, uint64_t refSsd, uint64_t refSum,
                               x265hip_weight_param* chosen, int* isWeighted, int* weightedId)
{
    if (fail_now("la_weights")) return X265HIP_EHIP;
    const x265hip_la_config* c = &la->c;
    slot* fb = &la->slots[slotB];
    slot* fr = &la->slots[slotRef];
    if (la->wbufsUsed >= 1024) { snprintf(g_err, sizeof(g_err), "emul: too many weighted references in one batch"); return X265HIP_EINVAL; }
    if (!la->wbuf[la->wbufsUsed]) la->wbuf[la->wbufsUsed] = malloc((size_t)4 * c->planeElems * la->B);
    void* w = la->wbuf[la->wbufsUsed];
    const int paddedLines = (int)(c->planeElems / c->stride);
    int32_t ch[3];
    int r;
    if (c->depth == 8)
    {
        const uint8_t* rb[4]; uint8_t* ob[4];
        for (int k = 0; k < 4; k++) { rb[k] = (const uint8_t*)fr->buffers + (size_t)k * c->planeElems; ob[k] = (uint8_t*)w + (size_t)k * c->planeElems; }
        r = orc_weights_analyse_8((const uint8_t*)fb->buffers + c->padOffset, rb, ob, c->stride, c->padOffset, paddedLines, c->width, c->lines, fb->intraCost,
                                  fencSsd, fencSum, refSsd, refSum, ch, c->depth);
    }
    else
    {
        const uint16_t* rb[4]; uint16_t* ob[4];
        for (int k = 0; k < 4; k++) { rb[k] = (const uint16_t*)fr->buffers + (size_t)k * c->planeElems; ob[k] = (uint16_t*)w + (size_t)k * c->planeElems; }
        r = orc_weights_analyse_16((const uint16_t*)fb->buffers + c->padOffset, rb, ob, c->stride, c->padOffset, paddedLines, c->width, c->lines, fb->intraCost,
                                   fencSsd, fencSum, refSsd, refSum, ch, c->depth);
    }
    *isWeighted = r;
    chosen->inputWeight = ch[0]; chosen->log2WeightDenom = ch[1]; chosen->inputOffset = ch[2]; chosen->wtPresent = r;
    *weightedId = r ? la->wbufsUsed++ : -1;
    return 0;
}

static int ahead_index(const x265hip_la* la, int list, int dist, int bidir, int slices) { return ((list * la->c.maxDist + dist) * 2 + (bidir ? 1 : 0)) * 2 + (slices > 1 ? 1 : 0); }
static int32_t* ahead_of(x265hip_la* la, slot* s, int idx)
{
    if (!s->ahead) s->ahead = (int32_t*)calloc((size_t)2 * la->c.maxDist * 4 * 3 * la->ncu, 4);
    return s->ahead + (size_t)idx * 3 * la->ncu;
}

int x265hip_la_has_ahead(x265hip_la* la, int slotIdx, int list, int dist, int bidir, int numRowsPerSlice, int numSlices)
{
    const slot* s = &la->slots[slotIdx];
    const int idx = ahead_index(la, list, dist, bidir, numSlices);
    return s->live && s->aheadValid[idx] && s->aheadRows[idx] == numRowsPerSlice && s->aheadSlices[idx] == numSlices;
}

/* one list search in the variant asked for, into dst ([3 ncu]); the oracle's P / B passes with only that list searched */
static void search_ahead(x265hip_la* la, const x265hip_la_search* a, int32_t* dst)
{
    const x265hip_la_config* c = &la->c;
    const int ncu = la->ncu, W = c->widthInCU, H = c->heightInCU;
    slot* fb = &la->slots[a->b];
    const void* refbuf = (!a->list && a->weightedId >= 0) ? la->wbuf[a->weightedId] : la->slots[a->ref].buffers;
    uint16_t* lc = (uint16_t*)malloc((size_t)ncu * 2); int32_t* rs = (int32_t*)malloc((size_t)H * 4);
    int32_t* other = (int32_t*)calloc((size_t)ncu * 3, 4);
    int32_t intraMbs = 0; int64_t aq = 0;
#define AHEAD(T, SFX) do { \
        const T* r[4]; PLANES(T, refbuf, r); \
        const T* fenc = (const T*)fb->buffers + c->padOffset; \
        if (!a->bidir) \
            orc_lookahead_cost_p_aq_##SFX(fenc, r, c->stride, W, H, a->numRowsPerSlice, a->numSlices, c->depth, fb->intraCost, la->mvcost + 2 * 32768, \
                                          dst, dst + 2 * ncu, lc, rs, &intraMbs, NULL, 1, &aq); \
        else \
        { \
            const int32_t ds[2] = { !a->list, a->list }; \
            orc_lookahead_cost_b_aq_##SFX(fenc, r, r, c->stride, W, H, a->numRowsPerSlice, a->numSlices, c->depth, la->mvcost + 2 * 32768, ds, \
                                          a->list ? other : dst, (a->list ? other : dst) + 2 * ncu, a->list ? dst : other, (a->list ? dst : other) + 2 * ncu, \
                                          lc, rs, NULL, NULL); \
        } } while (0)
    if (c->depth == 8) AHEAD(uint8_t, 8); else AHEAD(uint16_t, 16);
#undef AHEAD
    free(lc); free(rs); free(other);
}

int x265hip_la_estimate_batch(x265hip_la* la, x265hip_la_estimate* est, int n, int numRowsPerSlice, int numSlices)
{
    return x265hip_la_estimate_batch_ahead(la, est, n, numRowsPerSlice, numSlices, NULL, 0);
}

int x265hip_la_estimate_batch_ahead(x265hip_la* la, x265hip_la_estimate* est, int n, int numRowsPerSlice, int numSlices,
                                    const x265hip_la_search* ahead, int nAhead)
{
    if (fail_now("la_estimate")) return X265HIP_EHIP;
    const x265hip_la_config* c = &la->c;
    const int ncu = la->ncu, W = c->widthInCU, H = c->heightInCU;
    int anySearch = 0;
    for (int i = 0; i < n; i++)
    {
        x265hip_la_estimate* q = &est[i];
        slot* fb = &la->slots[q->b];
        const int bidir = q->p1 != q->b;
        const int32_t* invQ = fb->hasInvQ ? fb->invQscale : NULL;
        int32_t* st0 = store_of(la, fb, 0, q->dist0);
        int32_t* st1 = bidir ? store_of(la, fb, 1, q->dist1) : NULL;
        if ((!q->search0 && !fb->valid[q->dist0]) || (bidir && !q->search1 && !fb->valid[c->maxDist + q->dist1]))
        {
            snprintf(g_err, sizeof(g_err), "emul: estimate %d reuses vectors the session has not seen", i);
            return X265HIP_EINVAL;
        }
        /* searched ahead of this request in the same variant: the vectors are the search's result (the device copies them the same way) */
        int served0 = 0, served1 = 0;
        for (int l = 0; l < (bidir ? 2 : 1); l++)
        {
            const int dist = l ? q->dist1 : q->dist0;
            if (!(l ? q->search1 : q->search0) || !x265hip_la_has_ahead(la, q->b, l, dist, bidir, numRowsPerSlice, numSlices)) continue;
            memcpy(l ? st1 : st0, ahead_of(la, fb, ahead_index(la, l, dist, bidir, numSlices)), (size_t)ncu * 12);
            if (l) served1 = 1; else served0 = 1;
            la->aheadUsed++;
        }
        const void* ref0buf = q->weightedId >= 0 && !served0 ? la->wbuf[q->weightedId] : la->slots[q->p0].buffers;
        int64_t aq = 0;
        int32_t intraMbs = 0;
        int64_t cost;
        const int32_t ds[2] = { q->search0 && !served0, q->search1 && !served1 };
#define RUN(T, SFX) do { \
            const T* r0[4]; const T* r0u[4]; const T* r1[4]; \
            PLANES(T, ref0buf, r0); PLANES(T, la->slots[q->p0].buffers, r0u); PLANES(T, la->slots[q->p1].buffers, r1); \
            const T* fenc = (const T*)fb->buffers + c->padOffset; \
            if (!bidir) \
                cost = orc_lookahead_cost_p_aq_##SFX(fenc, r0, c->stride, W, H, numRowsPerSlice, numSlices, c->depth, fb->intraCost, la->mvcost + 2 * 32768, \
                                                     st0, st0 + 2 * ncu, q->lowresCosts, q->rowSatds, &intraMbs, invQ, ds[0], &aq); \
            else \
            { \
                /* list 0 searches the weighted planes when there are any; the bi-predictive candidates use the unweighted ones (slicetype.cpp:3322): \
                 * the oracle's B pass takes one plane set per list, so a weighted list 0 is searched first as its own pass */ \
                int32_t dsl[2] = { ds[0], ds[1] }; \
                if (q->weightedId >= 0 && ds[0]) \
                { \
                    uint16_t* lc = (uint16_t*)malloc((size_t)ncu * 2); int32_t* rs = (int32_t*)malloc((size_t)H * 4); \
                    int32_t* m1 = (int32_t*)malloc((size_t)ncu * 12); \
                    const int32_t only0[2] = { 1, 0 }; \
                    memset(m1, 0, (size_t)ncu * 12); \
                    orc_lookahead_cost_b_aq_##SFX(fenc, r0, r1, c->stride, W, H, numRowsPerSlice, numSlices, c->depth, la->mvcost + 2 * 32768, only0, \
                                                  st0, st0 + 2 * ncu, m1, m1 + 2 * ncu, lc, rs, NULL, NULL); \
                    free(lc); free(rs); free(m1); \
                    dsl[0] = 0; \
                } \
                cost = orc_lookahead_cost_b_aq_##SFX(fenc, r0u, r1, c->stride, W, H, numRowsPerSlice, numSlices, c->depth, la->mvcost + 2 * 32768, dsl, \
                                                     st0, st0 + 2 * ncu, st1, st1 + 2 * ncu, q->lowresCosts, q->rowSatds, invQ, &aq); \
            } \
        } while (0)
        if (c->depth == 8) RUN(uint8_t, 8); else RUN(uint16_t, 16);
#undef RUN
        if (q->search0) { memcpy(q->mvs0, st0, (size_t)ncu * 8); memcpy(q->mvCosts0, st0 + 2 * ncu, (size_t)ncu * 4); fb->valid[q->dist0] = 1; la->searches += !served0; }
        if (bidir && q->search1) { memcpy(q->mvs1, st1, (size_t)ncu * 8); memcpy(q->mvCosts1, st1 + 2 * ncu, (size_t)ncu * 4); fb->valid[c->maxDist + q->dist1] = 1; la->searches += !served1; }
        if (ds[0] || ds[1]) { anySearch = 1; la->searchPairs += ds[0] + ds[1]; }
        q->costEst = cost;
        q->costEstAq = aq;
        q->intraMbs = bidir ? 0 : intraMbs;
    }
    for (int j = 0; j < nAhead; j++)
    {
        const x265hip_la_search* a = &ahead[j];
        slot* fb = &la->slots[a->b];
        const int idx = ahead_index(la, a->list, a->dist, a->bidir, a->numSlices);
        if (fb->valid[a->list * c->maxDist + a->dist] || (fb->aheadValid[idx] && fb->aheadRows[idx] == a->numRowsPerSlice && fb->aheadSlices[idx] == a->numSlices))
            continue;
        search_ahead(la, a, ahead_of(la, fb, idx));
        fb->aheadValid[idx] = 1; fb->aheadRows[idx] = a->numRowsPerSlice; fb->aheadSlices[idx] = a->numSlices;
        la->aheadLaunched++; la->searchPairs++; anySearch = 1;
    }
    la->wbufsUsed = 0;
    la->batches += n > 0;
    la->estimates += n;
    la->searchLaunches += anySearch;
    return 0;
}

int x265hip_la_stats(x265hip_la* la, uint64_t* batches, uint64_t* estimates, uint64_t* searches)
{
    if (batches) *batches = la->batches;
    if (estimates) *estimates = la->estimates;
    if (searches) *searches = la->searches;
    return 0;
}

int x265hip_la_stats_ahead(x265hip_la* la, uint64_t* launchedAhead, uint64_t* usedAhead, uint64_t* searchLaunches, uint64_t* searchPairs)
{
    if (launchedAhead) *launchedAhead = la->aheadLaunched;
    if (usedAhead) *usedAhead = la->aheadUsed;
    if (searchLaunches) *searchLaunches = la->searchLaunches;
    if (searchPairs) *searchPairs = la->searchPairs;
    return 0;
}

/* ---------------------------------------------------------------- reference-picture mirrors, emulated ------------------------------- *
 * Synchronous: rows_final() filters the newly final band with the oracle's luma filters (orc_interp_*_pp, the restatement of
 * ipfilter.cpp:79-369) before it returns.  Same contract as the device implementation otherwise. */
struct x265hip_refpic
{
    int depth, B, picW, picH, marginX, marginY, bufRows;
    int64_t stride, planeElems;
    const char* hostBase;
    char* planes;        /* 15 planes */
    int phaseDone;       /* buffer rows */
    int rowsReady;
    int rowsFinal;       /* picture rows the caller declared final (the whole padded picture once >= picH) */
    struct x265hip_sadsurf* surfaces;      /* attached SAD surfaces (singly linked) */
    int place;                             /* x265hip_refpic_create_at; -1 for x265hip_refpic_create */
    int repPlace[16], repCopied[16], nRep; /* emulated replicas at other places: how many buffer rows each has been "pushed" */
};
/* places (x265hip_places): the emulation has no devices; it keeps the bookkeeping of the exchange — which replicas exist, how many bands and bytes
 * the device implementation would have pushed from GPU to GPU — so that the binding's placement logic can be tested on the CPU tier */
static uint64_t g_peerReplicas, g_peerBands, g_peerBytes;
int x265hip_places(int n, const int* devices)
{
    if (n < 1 || n > 64 || !devices || n < g_nPlaces) { snprintf(g_err, sizeof(g_err), "emul: x265hip_places: %d", n); return -1; }
    g_nPlaces = n;
    return 0;
}
int x265hip_peer_stats(uint64_t* replicas, uint64_t* bands, uint64_t* bytes)
{
    if (replicas) *replicas = g_peerReplicas;
    if (bands) *bands = g_peerBands;
    if (bytes) *bytes = g_peerBytes;
    return 0;
}
static void sadsurf_progress(struct x265hip_sadsurf* ss);
static void sadsurf_detach_all(x265hip_refpic* rp);

x265hip_refpic* x265hip_refpic_create(int depth, int picW, int picH, int64_t stride, int marginX, int marginY, int bufRows, const void* hostBase)
{
    if (fail_now("refpic_create")) return NULL;
    x265hip_refpic* rp = (x265hip_refpic*)calloc(1, sizeof(*rp));
    rp->depth = depth; rp->B = depth == 8 ? 1 : 2; rp->picW = picW; rp->picH = picH; rp->marginX = marginX; rp->marginY = marginY; rp->bufRows = bufRows;
    rp->stride = stride; rp->planeElems = stride * bufRows; rp->hostBase = (const char*)hostBase;
    rp->planes = (char*)calloc((size_t)15 * rp->planeElems, rp->B);
    rp->phaseDone = 4;
    rp->rowsReady = -(1 << 30);
    rp->place = -1;
    return rp;
}
x265hip_refpic* x265hip_refpic_create_at(int place, int depth, int picW, int picH, int64_t stride, int marginX, int marginY, int bufRows, const void* hostBase)
{
    if (place < 0 || place >= g_nPlaces) { snprintf(g_err, sizeof(g_err), "emul: no place %d", place); return NULL; }
    x265hip_refpic* rp = x265hip_refpic_create(depth, picW, picH, stride, marginX, marginY, bufRows, hostBase);
    if (rp) rp->place = place;
    return rp;
}
void x265hip_refpic_destroy(x265hip_refpic* rp) { if (rp) { sadsurf_detach_all(rp); free(rp->planes); free(rp); } }
int x265hip_refpic_reset(x265hip_refpic* rp)
{
    if (fail_now("refpic_reset")) return X265HIP_EHIP;
    sadsurf_detach_all(rp);
    for (int i = 0; i < rp->nRep; i++) rp->repCopied[i] = 0;
    rp->phaseDone = 4; rp->rowsFinal = 0;
    __atomic_store_n(&rp->rowsReady, -(1 << 30), __ATOMIC_RELEASE);
    return 0;
}
const void* x265hip_refpic_plane(x265hip_refpic* rp, int phase) { return rp->planes + (size_t)(phase - 1) * rp->planeElems * rp->B; }
int x265hip_refpic_rows_ready(x265hip_refpic* rp) { return __atomic_load_n(&rp->rowsReady, __ATOMIC_ACQUIRE); }
const int* x265hip_refpic_rows_ready_ptr(x265hip_refpic* rp) { return &rp->rowsReady; }
int x265hip_refpic_wait(x265hip_refpic* rp) { (void)rp; return 0; }

static void sadsurf_progress_all(x265hip_refpic* rp);

int x265hip_refpic_rows_final(x265hip_refpic* rp, int rowsFinal)
{
    if (fail_now("rows_final")) return X265HIP_EHIP;
    if (rowsFinal > rp->rowsFinal) rp->rowsFinal = rowsFinal;
    sadsurf_progress_all(rp);
    const int complete = rowsFinal >= rp->picH;
    const int finalRows = complete ? rp->marginY + rp->picH + rp->marginY : rp->marginY + rowsFinal;
    const int phaseEnd = finalRows - 4;
    if (phaseEnd <= rp->phaseDone) return 0;
    const int w = rp->picW + 2 * rp->marginX - 8, h = phaseEnd - rp->phaseDone;
    const size_t off = (size_t)rp->phaseDone * rp->stride + 4;            /* first computed element of the band (buffer coordinates) */
    for (int yf = 0; yf < 4; yf++)
        for (int xf = 0; xf < 4; xf++)
        {
            if (!(xf | yf)) continue;
            char* dst = rp->planes + ((size_t)(yf * 4 + xf - 1) * rp->planeElems + off) * rp->B;
            const char* src = rp->hostBase + off * rp->B;
#define FILT(T, SFX) do { \
                /* the oracle's filters work on PU-sized blocks (their intermediate buffer holds 64 x 71 samples): tile the band */ \
                for (int ty = 0; ty < h; ty += 64) \
                    for (int tx = 0; tx < w; tx += 64) \
                    { \
                        const int bw = w - tx < 64 ? w - tx : 64, bh = h - ty < 64 ? h - ty : 64; \
                        const T* s_ = (const T*)src + (size_t)ty * rp->stride + tx; \
                        T* d_ = (T*)dst + (size_t)ty * rp->stride + tx; \
                        if (!yf) orc_interp_horiz_pp_##SFX(8, s_, rp->stride, d_, rp->stride, bw, bh, xf, rp->depth); \
                        else if (!xf) orc_interp_vert_pp_##SFX(8, s_, rp->stride, d_, rp->stride, bw, bh, yf, rp->depth); \
                        else orc_interp_hv_pp_##SFX(8, s_, rp->stride, d_, rp->stride, bw, bh, xf, yf, rp->depth); \
                    } } while (0)
            if (rp->depth == 8) FILT(uint8_t, 8); else FILT(uint16_t, 16);
#undef FILT
        }
    rp->phaseDone = phaseEnd;
    __atomic_store_n(&rp->rowsReady, phaseEnd - rp->marginY, __ATOMIC_RELEASE);
    return 0;
}


/* ---------------------------------------------------------------- source-picture energy planes, emulated ------------------------------ *
 * the source half of psyCost_pp (pixel.cpp:726-757) with the oracle's sa8d / satd restatements against a zero block */
int x265hip_device_time(int clock, uint64_t* spans, uint64_t* nanoseconds, uint64_t* algorithmicBytes)
{
    (void)clock;
    if (algorithmicBytes) *algorithmicBytes = 0;
    if (spans) *spans = 0;
    if (nanoseconds) *nanoseconds = 0;
    return 0;                           /* no device: nothing to time */
}

int x265hip_source_energy(int depth, const void* hostPlane, int64_t stride, int width, int height, int32_t* hostE8, int32_t* hostE4)
{
    if (fail_now("source_energy")) return X265HIP_EHIP;
    const int bw = width >> 3, bh = height >> 3;
    static const uint16_t zero[64];
    for (int by = 0; by < bh; by++)
        for (int bx = 0; bx < bw; bx++)
        {
#define ENERGY(T, SFX) do { \
            const T* p = (const T*)hostPlane + (int64_t)by * 8 * stride + bx * 8; \
            int sum = 0; \
            for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) sum += p[y * stride + x]; \
            hostE8[by * bw + bx] = orc_sa8d_##SFX(p, stride, (const T*)zero, 0, 8) - (sum >> 2); \
            for (int q = 0; q < 4; q++) \
            { \
                const T* p4 = p + (q >> 1) * 4 * stride + (q & 1) * 4; \
                int s4 = 0; \
                for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) s4 += p4[y * stride + x]; \
                hostE4[(by * 2 + (q >> 1)) * (bw * 2) + bx * 2 + (q & 1)] = orc_satd_##SFX(p4, stride, (const T*)zero, 0, 4, 4) - (s4 >> 2); \
            } } while (0)
            if (depth == 8) ENERGY(uint8_t, 8); else ENERGY(uint16_t, 16);
#undef ENERGY
        }
    return 0;
}


/* ---------------------------------------------------------------- SAD surfaces, emulated ------------------------------------------------- *
 * Synchronous: attach() and the reference picture's rows_final() build every CTU row whose reference rows are final with the oracle
 * (oracle/x265_oracle_sadsurf.inc) before they return.  Same contract as the device implementation otherwise. */
void orc_sadsurf_rows_8(const uint8_t* src, intptr_t srcStride, const uint8_t* ref, intptr_t refStride, int picW, int picH, int marginX, int marginY,
                        int S, int lambda20, int row0, int row1, int16_t* const origin[4], uint32_t* const table[4]);
void orc_sadsurf_rows_16(const uint16_t* src, intptr_t srcStride, const uint16_t* ref, intptr_t refStride, int picW, int picH, int marginX, int marginY,
                         int S, int lambda20, int row0, int row1, int16_t* const origin[4], uint32_t* const table[4]);

void orc_sadsurf_subpel_rows_8(const uint8_t* src, intptr_t srcStride, const uint8_t* ref, intptr_t refStride, int picW, int picH, int depth, int row0, int row1,
                               int16_t* const origin[4], uint32_t* const subpel[4]);
void orc_sadsurf_subpel_rows_16(const uint16_t* src, intptr_t srcStride, const uint16_t* ref, intptr_t refStride, int picW, int picH, int depth, int row0, int row1,
                                int16_t* const origin[4], uint32_t* const subpel[4]);

struct x265hip_srcpic { int depth, B, w, h; char* luma; int place; int refs; };     /* refs: the creator + the surfaces attached (as in the library) */
static void srcpic_unref(x265hip_srcpic* sp) { if (__atomic_sub_fetch(&sp->refs, 1, __ATOMIC_ACQ_REL) == 0) { free(sp->luma); free(sp); } }

x265hip_srcpic* x265hip_srcpic_create(int depth, int width, int height)
{
    if (fail_now("srcpic_create")) return NULL;
    x265hip_srcpic* sp = (x265hip_srcpic*)calloc(1, sizeof(*sp));
    sp->depth = depth; sp->B = depth == 8 ? 1 : 2; sp->w = width; sp->h = height;
    sp->luma = (char*)malloc((size_t)width * height * sp->B);
    sp->place = -1;
    sp->refs = 1;
    return sp;
}
x265hip_srcpic* x265hip_srcpic_create_at(int place, int depth, int width, int height)
{
    if (place < 0 || place >= g_nPlaces) { snprintf(g_err, sizeof(g_err), "emul: no place %d", place); return NULL; }
    x265hip_srcpic* sp = x265hip_srcpic_create(depth, width, height);
    if (sp) sp->place = place;
    return sp;
}
int x265hip_srcpic_upload(x265hip_srcpic* sp, const void* hostLuma, int64_t stride)
{
    if (fail_now("srcpic_upload")) return X265HIP_EHIP;
    for (int y = 0; y < sp->h; y++)
        memcpy(sp->luma + (size_t)y * sp->w * sp->B, (const char*)hostLuma + (size_t)y * stride * sp->B, (size_t)sp->w * sp->B);
    return 0;
}
void x265hip_srcpic_destroy(x265hip_srcpic* sp) { if (sp) srcpic_unref(sp); }

struct x265hip_sadsurf
{
    x265hip_srcpic* src;
    x265hip_refpic* ref;                 /* NULL once the reference picture has gone (reset / destroy) */
    struct x265hip_sadsurf* next;
    int S, lambda20, ctuRows, ctuRowsReady, levels;
    x265hip_sadsurf_view view;
    int64_t originOff[X265HIP_SADSURF_LEVELS], tableOff[X265HIP_SADSURF_LEVELS];
    char* buf;                                   /* ctuRows chunks of view.ctuRowPitch bytes, the layout of x265hip_sadsurf_level */
    int16_t* origin[X265HIP_SADSURF_LEVELS];     /* the oracle's whole-picture arrays */
    uint32_t* wide[X265HIP_SADSURF_LEVELS];
    uint32_t* subpel[X265HIP_SADSURF_LEVELS];    /* [blocks][49] when the sub-pel tables are built */
    int64_t subpelOff[X265HIP_SADSURF_LEVELS];
};
static uint64_t g_ssAttached, g_ssRows;

static void sadsurf_progress(x265hip_sadsurf* ss)
{
    x265hip_refpic* rp = ss->ref;
    if (!rp) return;
    const int complete = rp->rowsFinal >= rp->picH;
    if (ss->src->place != rp->place && ss->ctuRowsReady < ss->ctuRows && (complete || 64 * (ss->ctuRowsReady + 1) + ss->S <= rp->rowsFinal))
    {
        /* the device implementation builds this surface from a replica of the picture at the source's place: rows pushed GPU to GPU */
        int k = 0;
        while (k < rp->nRep && rp->repPlace[k] != ss->src->place) k++;
        if (k == rp->nRep && k < 16) { rp->repPlace[k] = ss->src->place; rp->repCopied[k] = 0; rp->nRep++; g_peerReplicas++; }
        const int uploaded = complete ? rp->picH + 2 * rp->marginY : rp->marginY + rp->rowsFinal;
        if (k < 16 && rp->repCopied[k] < uploaded)
        {
            g_peerBands++;
            g_peerBytes += (uint64_t)(uploaded - rp->repCopied[k]) * rp->stride * rp->B;
            rp->repCopied[k] = uploaded;
        }
    }
    while (ss->ctuRowsReady < ss->ctuRows)
    {
        const int r = ss->ctuRowsReady;
        if (!complete && 64 * (r + 1) + ss->S > rp->rowsFinal)      /* the lowest reference line a candidate of this row touches */
            break;
        const char* refOrg = rp->hostBase + ((size_t)rp->marginY * rp->stride + rp->marginX) * rp->B;
        if (rp->depth == 8)
            orc_sadsurf_rows_8((const uint8_t*)ss->src->luma, ss->src->w, (const uint8_t*)refOrg, rp->stride, rp->picW, rp->picH, rp->marginX, rp->marginY,
                               ss->S, ss->lambda20, r, r + 1, ss->origin, ss->wide);
        else
            orc_sadsurf_rows_16((const uint16_t*)ss->src->luma, ss->src->w, (const uint16_t*)refOrg, rp->stride, rp->picW, rp->picH, rp->marginX, rp->marginY,
                                ss->S, ss->lambda20, r, r + 1, ss->origin, ss->wide);
        if (ss->levels & 16)
        {
            if (rp->depth == 8)
                orc_sadsurf_subpel_rows_8((const uint8_t*)ss->src->luma, ss->src->w, (const uint8_t*)refOrg, rp->stride, rp->picW, rp->picH, rp->depth, r, r + 1,
                                          ss->origin, ss->subpel);
            else
                orc_sadsurf_subpel_rows_16((const uint16_t*)ss->src->luma, ss->src->w, (const uint16_t*)refOrg, rp->stride, rp->picW, rp->picH, rp->depth, r, r + 1,
                                           ss->origin, ss->subpel);
        }
        char* chunk = ss->buf + (size_t)r * ss->view.ctuRowPitch;
        for (int l = 0; l < X265HIP_SADSURF_LEVELS; l++)
        {
            if (!(ss->levels >> l & 1)) continue;
            const x265hip_sadsurf_level* v = &ss->view.level[l];
            for (int j = 0; j < v->blocksPerCtuRow; j++)
            {
                const int by = r * v->blocksPerCtuRow + j;
                if (by >= v->blocksY) break;
                for (int bx = 0; bx < v->blocksX; bx++)
                {
                    const size_t b = (size_t)by * v->blocksX + bx, k = (size_t)j * v->blocksX + bx;
                    memcpy(chunk + ss->originOff[l] + k * 4, ss->origin[l] + 2 * b, 4);
                    for (int e = 0; e < 256; e++)
                        if (v->entryBytes == 2) ((uint16_t*)(chunk + ss->tableOff[l]))[k * 256 + e] = (uint16_t)ss->wide[l][b * 256 + e];
                        else ((uint32_t*)(chunk + ss->tableOff[l]))[k * 256 + e] = ss->wide[l][b * 256 + e];
                    if (ss->subpel[l])
                        memcpy(chunk + ss->subpelOff[l] + k * X265HIP_SADSURF_SUBPEL * 4, ss->subpel[l] + b * X265HIP_SADSURF_SUBPEL, X265HIP_SADSURF_SUBPEL * 4);
                }
            }
        }
        g_ssRows++;
        __atomic_store_n(&ss->ctuRowsReady, r + 1, __ATOMIC_RELEASE);
    }
}

/* rows_final() runs on the encoder's frame-filter threads, attach() / release() on its analysis threads: one lock around the lists (the device
 * implementation serialises the same operations on its worker thread) */
static pthread_mutex_t g_ssLock = PTHREAD_MUTEX_INITIALIZER;
static void sadsurf_progress_all(x265hip_refpic* rp)
{
    pthread_mutex_lock(&g_ssLock);
    for (x265hip_sadsurf* s = rp->surfaces; s; s = s->next) sadsurf_progress(s);
    pthread_mutex_unlock(&g_ssLock);
}
static void sadsurf_detach_all(x265hip_refpic* rp)
{
    pthread_mutex_lock(&g_ssLock);
    for (x265hip_sadsurf* s = rp->surfaces; s; )
    {
        x265hip_sadsurf* n = s->next;
        s->ref = NULL; s->next = NULL;
        s = n;
    }
    rp->surfaces = NULL;
    pthread_mutex_unlock(&g_ssLock);
}

x265hip_sadsurf* x265hip_sadsurf_attach(x265hip_srcpic* src, x265hip_refpic* ref, int searchRange, int lambda20)
{
    return x265hip_sadsurf_attach_levels(src, ref, searchRange, lambda20, 14);
}
x265hip_sadsurf* x265hip_sadsurf_attach_levels(x265hip_srcpic* src, x265hip_refpic* ref, int searchRange, int lambda20, int levels)
{
    if (fail_now("sadsurf_attach")) return NULL;
    if (!src || !ref || src->depth != ref->depth || src->w != ref->picW || src->h != ref->picH || searchRange < 8 || searchRange > 32 || (searchRange & 3) ||
        lambda20 < 0 || lambda20 > (1 << 20) || (levels & ~31) || (levels & 14) != 14)
    {
        snprintf(g_err, sizeof(g_err), "emul: sadsurf_attach: mismatched pictures or range %d", searchRange);
        return NULL;
    }
    /* the sub-pel tables: as the library, only where the mirror lives; and — this is the scalar restatement, 49 filtered comparisons per block — only for
     * pictures the CPU tier's clips have (X265HIP_EMUL_SUBPEL_MAX_PIXELS, default 416 x 240); beyond that the level's `subpel` stays NULL */
    {
        const char* lim = getenv("X265HIP_EMUL_SUBPEL_MAX_PIXELS");
        const long maxPix = lim ? atol(lim) : 416L * 240L;
        if ((long)src->w * src->h > maxPix) levels &= ~16;          /* (a surface built from a replica has sub-pel tables too: the replica computes its own planes) */
    }
    x265hip_sadsurf* ss = (x265hip_sadsurf*)calloc(1, sizeof(*ss));
    ss->src = src; ss->ref = ref; ss->S = searchRange; ss->lambda20 = lambda20; ss->levels = levels;
    __atomic_add_fetch(&src->refs, 1, __ATOMIC_ACQ_REL);
    ss->ctuRows = (src->h + 63) / 64;
    int64_t off = 0;
    for (int l = 0; l < X265HIP_SADSURF_LEVELS; l++)
    {
        if (!(levels >> l & 1)) continue;         /* not built: origin == NULL in the view, NULL arrays for the oracle */
        const int log2n = 3 + l, N = 1 << log2n;
        x265hip_sadsurf_level* v = &ss->view.level[l];
        v->blocksX = src->w >> log2n; v->blocksY = src->h >> log2n; v->blocksPerCtuRow = 64 / N;
        v->entryBytes = (uint64_t)N * N * ((1u << src->depth) - 1) < 65536 ? 2 : 4;
        const size_t nb = (size_t)v->blocksX * v->blocksY;
        ss->origin[l] = (int16_t*)calloc(nb ? nb : 1, 4);
        ss->wide[l] = (uint32_t*)calloc(nb ? nb : 1, (size_t)256 * 4);
        ss->originOff[l] = off; off += (int64_t)v->blocksPerCtuRow * v->blocksX * 4;
        ss->tableOff[l] = off; off += (int64_t)v->blocksPerCtuRow * v->blocksX * 256 * v->entryBytes;
        if (l && (levels & 16))
        {
            ss->subpel[l] = (uint32_t*)calloc(nb ? nb : 1, (size_t)X265HIP_SADSURF_SUBPEL * 4);
            ss->subpelOff[l] = off; off += (int64_t)v->blocksPerCtuRow * v->blocksX * X265HIP_SADSURF_SUBPEL * 4;
        }
    }
    ss->view.ctuRowPitch = off;
    ss->buf = (char*)calloc((size_t)ss->ctuRows, (size_t)off);
    for (int l = 0; l < X265HIP_SADSURF_LEVELS; l++)
    {
        if (!(levels >> l & 1)) continue;
        ss->view.level[l].origin = (const int16_t*)(ss->buf + ss->originOff[l]);
        ss->view.level[l].table = ss->buf + ss->tableOff[l];
        ss->view.level[l].subpel = ss->subpel[l] ? (const uint32_t*)(ss->buf + ss->subpelOff[l]) : NULL;
    }
    ss->view.ctuRowsReady = &ss->ctuRowsReady;
    pthread_mutex_lock(&g_ssLock);
    ss->next = ref->surfaces;
    ref->surfaces = ss;
    g_ssAttached++;
    sadsurf_progress(ss);
    pthread_mutex_unlock(&g_ssLock);
    return ss;
}
const x265hip_sadsurf_view* x265hip_sadsurf_get_view(x265hip_sadsurf* ss) { return ss ? &ss->view : NULL; }
void x265hip_sadsurf_release(x265hip_sadsurf* ss)
{
    if (!ss) return;
    pthread_mutex_lock(&g_ssLock);
    if (ss->ref)
    {
        x265hip_sadsurf** pp = &ss->ref->surfaces;
        while (*pp && *pp != ss) pp = &(*pp)->next;
        if (*pp) *pp = ss->next;
    }
    pthread_mutex_unlock(&g_ssLock);
    for (int l = 0; l < X265HIP_SADSURF_LEVELS; l++) { free(ss->origin[l]); free(ss->wide[l]); free(ss->subpel[l]); }
    free(ss->buf);
    srcpic_unref(ss->src);
    free(ss);
}
int x265hip_sadsurf_stats(uint64_t* attached, uint64_t* ctuRows, uint64_t* launches, uint64_t* kernelNs)
{
    if (kernelNs) *kernelNs = 0;
    if (launches) *launches = 0;
    if (attached) *attached = g_ssAttached;
    if (ctuRows) *ctuRows = g_ssRows;
    return 0;
}

/* ---- CU residual quad-tree jobs (include/x265hip.h, x265hip_cuserve_*): the restatement (oracle/x265_oracle_rqt.c) behind the same slot / submit /
 * ready-word protocol; the job is done inside x265hip_cuserve_submit.  X265HIP_EMUL_FAIL=cuserve_open | cuserve_submit[:n] makes
 * the respective call fail (fail_now above). */
int orc_cujob_run_8(const x265hip_cujob* j, const uint8_t* pixels, x265hip_cujob_unit* units, int16_t* levels, int16_t* resi, uint32_t seq);
int orc_cujob_run_16(const x265hip_cujob* j, const uint16_t* pixels, x265hip_cujob_unit* units, int16_t* levels, int16_t* resi, uint32_t seq);
typedef struct cu_slot
{
    x265hip_cujob job;
    x265hip_cujob_unit units[X265HIP_CUJOB_MAX_UNITS];
    _Alignas(64) unsigned char pixels[X265HIP_CUJOB_PIXEL_BYTES];
    _Alignas(64) int16_t levels[X265HIP_CUJOB_MAX_ELEMS];
    _Alignas(64) int16_t resi[X265HIP_CUJOB_MAX_ELEMS];
    uint32_t seq;
} cu_slot;
struct x265hip_cuserve { int slots, mode; cu_slot* slot; uint64_t jobs; int lost; };
int x265hip_cuserve_open(int slots, int mode, x265hip_cuserve** out)
{
    if (fail_now("cuserve_open")) return X265HIP_ENOMEM;
    if (!out || slots < 1 || slots > 256) return X265HIP_EINVAL;
    x265hip_cuserve* cs = (x265hip_cuserve*)calloc(1, sizeof(*cs));
    cs->slots = slots; cs->mode = mode;
    cs->slot = (cu_slot*)aligned_alloc(64, sizeof(cu_slot) * slots);
    memset(cs->slot, 0, sizeof(cu_slot) * slots);
    *out = cs;
    return 0;
}
int x265hip_cuserve_open_at(int place, int slots, int mode, x265hip_cuserve** out)
{
    if (place < 0 || place >= g_nPlaces) { snprintf(g_err, sizeof(g_err), "emul: no place %d", place); return X265HIP_EINVAL; }
    return x265hip_cuserve_open(slots, mode, out);
}
int x265hip_cuserve_close(x265hip_cuserve* cs) { if (cs) { free(cs->slot); free(cs); } return 0; }
int x265hip_cuserve_slot(x265hip_cuserve* cs, int slot, x265hip_cujob** job, void** pixels, const x265hip_cujob_unit** units, const int16_t** levels,
                         const int16_t** resi)
{
    if (!cs || slot < 0 || slot >= cs->slots) return X265HIP_EINVAL;
    cu_slot* s = cs->slot + slot;
    if (job) *job = &s->job;
    if (pixels) *pixels = s->pixels;
    if (units) *units = s->units;
    if (levels) *levels = s->levels;
    if (resi) *resi = s->resi;
    return 0;
}
int x265hip_cuserve_submit(x265hip_cuserve* cs, int slot, uint32_t* seq)
{
    if (fail_now("cuserve_submit")) return X265HIP_EHIP;
    if (!cs || slot < 0 || slot >= cs->slots || !seq) return X265HIP_EINVAL;
    cu_slot* s = cs->slot + slot;
    int sHi, sLo;
    if (s->job.log2CUSize < 4 || s->job.log2CUSize > 6 || x265hipi_cujob_levels(&s->job, &sHi, &sLo) < 1) return X265HIP_EINVAL;
    *seq = ++s->seq;
    __atomic_fetch_add(&cs->jobs, 1, __ATOMIC_RELAXED);
    if (__atomic_load_n(&cs->lost, __ATOMIC_RELAXED) || fail_now("cuserve_job"))
    {
        __atomic_store_n(&cs->lost, 1, __ATOMIC_RELAXED);
        return 0;                       /* accepted, never done */
    }
    if (s->job.bitDepth == 8) orc_cujob_run_8(&s->job, s->pixels, s->units, s->levels, s->resi, *seq);
    else orc_cujob_run_16(&s->job, (const uint16_t*)s->pixels, s->units, s->levels, s->resi, *seq);
    return 0;
}
int orc_saojob_run_8(const x265hip_saojob* j, const uint8_t* pixels, x265hip_cujob_unit* units, int32_t* out, uint32_t seq);
int x265hip_cuserve_submit_sao(x265hip_cuserve* cs, int slot, const x265hip_saojob* job, uint32_t* seq)
{
    if (fail_now("cuserve_submit_sao")) return X265HIP_EHIP;
    if (!cs || slot < 0 || slot >= cs->slots || !job || !seq || job->bitDepth != 8 || job->planes < 1 || job->planes > 3 ||
        x265hipi_saojob_pixel_bytes(job) > X265HIP_CUJOB_PIXEL_BYTES)
        return X265HIP_EINVAL;
    cu_slot* s = cs->slot + slot;
    *seq = ++s->seq;
    __atomic_fetch_add(&cs->jobs, 1, __ATOMIC_RELAXED);
    if (__atomic_load_n(&cs->lost, __ATOMIC_RELAXED) || fail_now("saostats_job"))
    {
        __atomic_store_n(&cs->lost, 1, __ATOMIC_RELAXED);
        return 0;
    }
    orc_saojob_run_8(job, s->pixels, s->units, (int32_t*)s->levels, *seq);
    return 0;
}
int orc_intrajob_run(const x265hip_intrajob* j, const void* pixels, x265hip_cujob_unit* units, int32_t* out, uint32_t seq);
int x265hip_cuserve_submit_intra(x265hip_cuserve* cs, int slot, const x265hip_intrajob* job, uint32_t* seq)
{
    if (fail_now("cuserve_submit_intra")) return X265HIP_EHIP;
    if (!cs || slot < 0 || slot >= cs->slots || !job || !seq || (job->bitDepth != 8 && job->bitDepth != 10 && job->bitDepth != 12) ||
        job->mark != X265HIP_INTRAJOB_MARK || job->log2Size < 3 || job->log2Size > 5)
        return X265HIP_EINVAL;
    cu_slot* s = cs->slot + slot;
    *seq = ++s->seq;
    __atomic_fetch_add(&cs->jobs, 1, __ATOMIC_RELAXED);
    if (__atomic_load_n(&cs->lost, __ATOMIC_RELAXED) || fail_now("intrascan_job"))
    {
        __atomic_store_n(&cs->lost, 1, __ATOMIC_RELAXED);
        return 0;
    }
    orc_intrajob_run(job, s->pixels, s->units, (int32_t*)s->levels, *seq);
    return 0;
}
int x265hip_cuserve_poke(x265hip_cuserve* cs, int slot) { (void)slot; return cs && __atomic_load_n(&cs->lost, __ATOMIC_RELAXED) ? X265HIP_EHIP : 0; }
int x265hip_cuserve_stats(x265hip_cuserve* cs, uint64_t* jobs, uint64_t* serverStarts, uint64_t* deviceNs)
{
    if (jobs) *jobs = cs ? cs->jobs : 0;
    if (serverStarts) *serverStarts = 0;
    if (deviceNs) *deviceNs = 0;
    return 0;
}
