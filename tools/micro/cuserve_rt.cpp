// cuserve_rt.cpp — round trip of the PRODUCT's CU residual quad-tree jobs through the C ABI (include/x265hip.h, x265hip_cuserve_*): a host thread
// fills its slot, submits, and spins on the units' ready words, as x265_amd/host/x265_hip_cuserve.cpp does.  (measurement aid)
//   cuserve_rt <mode 0|1> [iters] [stamps 0|1]   1, 4 and 16 submitting threads; 32x32 and 64x64 CUs (4:2:0, 8 bit, one transform size);
//                                                stamps 1: the chain's stage stamps (job.reserved, x265hip_cujob_unit::reserved) of the first luma and the first Cb unit
#include <atomic>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "../../include/x265hip.h"

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 1, iters = argc > 2 ? atoi(argv[2]) : 3000, stamps = argc > 3 ? atoi(argv[3]) : 0;
    if (x265hip_device_count() < 1 || x265hip_init(0)) { fprintf(stderr, "no device: %s\n", x265hip_last_error()); return 2; }
    x265hip_cuserve* cs = NULL;
    if (x265hip_cuserve_open(16, mode, &cs)) { fprintf(stderr, "open: %s\n", x265hip_last_error()); return 2; }
    std::atomic<bool> failed(false);
    // CUSERVE_RT_ONLY=cu5|cu6|sao, CUSERVE_RT_THREADS=<n>: one job shape from one thread count — the counter passes (tools/exp/gpu.sh pmc) want launches of one kind
    const char* only = getenv("CUSERVE_RT_ONLY");
    const int onlyT = getenv("CUSERVE_RT_THREADS") ? atoi(getenv("CUSERVE_RT_THREADS")) : 0;
    for (int log2cu = 5; log2cu <= 6; log2cu++)
        for (int T : { 1, 4, 16 })
        {
            if ((only && (only[0] != 'c' || only[2] != '0' + log2cu)) || (onlyT && T != onlyT)) continue;
            std::vector<std::vector<double>> lat(T), first(T), dev(T);
            std::vector<std::vector<double>> st[2][7];
            for (auto& a : st) for (auto& b : a) b.resize(T);
            std::atomic<int> go(0);
            auto body = [&](int t)
            {
                x265hip_init(0);
                x265hip_cujob* job; void* pixels; const x265hip_cujob_unit* units; const int16_t* levels; const int16_t* resi;
                x265hip_cuserve_slot(cs, t, &job, &pixels, &units, &levels, &resi);
                const int N = 1 << log2cu, bytes = 2 * (N * N + N * N / 2);
                std::vector<unsigned char> src(bytes);
                uint32_t s = 1234 + t;
                for (auto& b : src) { s = s * 1664525u + 1013904223u; b = (unsigned char)(128 + ((s >> 24) & 15)); }
                while (!go.load()) {}
                for (int i = 0; i < iters + 100 && !failed; i++)
                {
                    src[i % bytes] ^= 3;
                    const double t0 = now_us();
                    memset(job, 0, sizeof(*job));
                    job->log2CUSize = log2cu; job->log2TrMax = 5; job->log2TrMin = 5; job->chroma = 1; job->bitDepth = 8; job->quantOffset = 85; job->signHide = 1; job->reserved = stamps;
                    for (int p = 0; p < 3; p++) { job->qpRem[p] = 2; job->qpPer[p] = 5; job->quantScale[p] = 20560; job->dequantScale[p] = 51; }
                    memcpy(pixels, src.data(), bytes);
                    uint32_t seq = 0;
                    if (x265hip_cuserve_submit(cs, t, &seq)) { fprintf(stderr, "submit: %s\n", x265hip_last_error()); failed = true; break; }
                    int sHi, sLo;
                    x265hipi_cujob_levels(job, &sHi, &sLo);
                    const int per = 1 << (log2cu - 5), nUnits = 3 * per * per;
                    double tFirst = 0;
                    for (int u = 0; u < nUnits && !failed; u++)
                    {
                        uint64_t spins = 0;
                        if (u == 0)
                        {
                            // the forward half of the first luma unit: what Quant::transformNxN waits for first
                            while (__atomic_load_n(&units[0].ready, __ATOMIC_ACQUIRE) != seq && now_us() - t0 < 2e6) __builtin_ia32_pause();
                            tFirst = now_us() - t0;
                        }
                        while (__atomic_load_n(&units[u].readyInv, __ATOMIC_ACQUIRE) != seq)
                        {
                            __builtin_ia32_pause();
                            if ((++spins & 1023) == 0)
                            {
                                if (x265hip_cuserve_poke(cs, t) < 0) { fprintf(stderr, "poke: %s\n", x265hip_last_error()); failed = true; break; }
                                if (now_us() - t0 > 2e6) { fprintf(stderr, "job %d of thread %d: unit %d not ready after 2 s\n", i, t, u); failed = true; break; }
                            }
                        }
                    }
                    volatile int16_t sink = levels[0] + resi[0]; (void)sink;
                    const double t1 = now_us();
                    if (i >= 100) { lat[t].push_back(t1 - t0); first[t].push_back(tFirst); dev[t].push_back(units[0].fwdTicks * 0.01); }
                    if (i >= 100 && stamps)
                        for (int w = 0; w < 2; w++)
                        {
                            const x265hip_cujob_unit& un = units[w ? per * per : 0];
                            const uint32_t v[7] = { un.reserved[0] & 0xffff, un.reserved[0] >> 16, un.reserved[1] & 0xffff, un.reserved[1] >> 16, un.fwdTicks, un.reserved[2] & 0xffff,
                                                    un.reserved[2] >> 16 };
                            for (int k = 0; k < 7; k++) st[w][k][t].push_back(v[k] * 0.01);
                        }
                }
            };
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++) th.emplace_back(body, t);
            const double w0 = now_us();
            go = 1;
            for (auto& x : th) x.join();
            const double wall = now_us() - w0;
            if (failed) { printf("FAILED (mode %d, CU %d, %d threads)\n", mode, 1 << log2cu, T); x265hip_cuserve_close(cs); return 1; }
            std::vector<double> all, f, dv;
            for (auto& v : lat) all.insert(all.end(), v.begin(), v.end());
            for (auto& v : first) f.insert(f.end(), v.begin(), v.end());
            for (auto& v : dev) dv.insert(dv.end(), v.begin(), v.end());
            std::sort(dv.begin(), dv.end());
            std::sort(all.begin(), all.end()); std::sort(f.begin(), f.end());
            double sum = 0; for (double x : all) sum += x;
            printf("%s, %dx%d CU (4:2:0, 8 bit, 32x32 transforms), %2d thread%s: whole job mean %6.1f us, median %6.1f, p99 %6.1f; first luma unit forward half median %6.1f us (%4.1f us of it on the device, doorbell seen -> ready word issued); %.0f jobs/s in total\n",
                   mode ? "one launch per job" : "resident server   ", 1 << log2cu, 1 << log2cu, T, T > 1 ? "s" : " ", sum / all.size(), all[all.size() / 2],
                   all[(size_t)(all.size() * 0.99)], f[f.size() / 2], dv[dv.size() / 2], (double)T * (iters + 100) / (wall * 1e-6));
            if (stamps)
                for (int w = 0; w < 2; w++)
                {
                    double med[7];
                    for (int k = 0; k < 7; k++)
                    {
                        std::vector<double> a;
                        for (auto& v : st[w][k]) a.insert(a.end(), v.begin(), v.end());
                        std::sort(a.begin(), a.end());
                        med[k] = a.empty() ? 0 : a[a.size() / 2];
                    }
                    printf("      %s unit, us since the doorbell was seen (medians): chain starts %.2f, forward transform done %.2f, quantised %.2f, sign hiding done %.2f, ready issued %.2f, "
                           "inverse transform done %.2f, readyInv issued %.2f\n", w ? "first Cb  " : "first luma", med[0], med[1], med[2], med[3], med[4], med[5], med[6]);
                }
            fflush(stdout);
        }
    // ---- SAO statistics jobs (x265hip_saojob): a whole 64x64 CTU, three planes, every class; what SAO::calcSaoStatsCTU's seam hands over per CTU
    for (int T : { 1, 4, 16 })
    {
        if ((only && only[0] != 's') || (onlyT && T != onlyT)) continue;
        std::vector<std::vector<double>> luma(T), whole(T), dev0(T), dev2(T);
        std::vector<double> sst[2][5];
        std::atomic<int> go(0);
        auto body = [&](int t)
        {
            x265hip_init(0);
            x265hip_cujob* job; void* pixels; const x265hip_cujob_unit* units; const int16_t* levels; const int16_t* resi;
            x265hip_cuserve_slot(cs, t, &job, &pixels, &units, &levels, &resi);
            x265hip_saojob sj;
            memset(&sj, 0, sizeof(sj));
            sj.bitDepth = 8; sj.planes = 3; sj.eo23 = 1; sj.reserved = stamps;
            for (int p = 0; p < 3; p++)
            {
                const int n = p ? 32 : 64, po = p ? 2 : 0;
                sj.plane[p].w = sj.plane[p].h = (uint16_t)n;
                for (int c = 0; c < 5; c++) { sj.plane[p].x0[c] = c == 1 || c >= 3; sj.plane[p].y0[c] = c >= 2; sj.plane[p].x1[c] = (uint8_t)(n - 5 + po); sj.plane[p].y1[c] = (uint8_t)(n - 4 + po); }
            }
            const int bytes = x265hipi_saojob_pixel_bytes(&sj);
            std::vector<unsigned char> src(bytes);
            uint32_t s = 99 + t;
            for (auto& b : src) { s = s * 1664525u + 1013904223u; b = (unsigned char)(120 + ((s >> 24) & 31)); }
            while (!go.load()) {}
            for (int i = 0; i < iters + 100 && !failed; i++)
            {
                src[i % bytes] ^= 5;
                const double t0 = now_us();
                memcpy(pixels, src.data(), bytes);
                uint32_t seq = 0;
                if (x265hip_cuserve_submit_sao(cs, t, &sj, &seq)) { fprintf(stderr, "submit_sao: %s\n", x265hip_last_error()); failed = true; break; }
                double tl = 0;
                for (int p = 0; p < 3 && !failed; p++)
                {
                    uint64_t spins = 0;
                    while (__atomic_load_n(&units[p].ready, __ATOMIC_ACQUIRE) != seq)
                        if ((++spins & 1023) == 0 && x265hip_cuserve_poke(cs, t) < 0) { fprintf(stderr, "poke: %s\n", x265hip_last_error()); failed = true; break; }
                    if (p == 0) tl = now_us() - t0;
                }
                const double tw = now_us() - t0;
                if (i >= 100) { luma[t].push_back(tl); whole[t].push_back(tw); dev0[t].push_back(units[0].fwdTicks * 0.01); dev2[t].push_back(units[2].fwdTicks * 0.01); }
                if (i >= 100 && stamps && t == 0)
                    for (int p = 0; p < 3; p += 2)
                        for (int k = 0; k < 5; k++)
                            sst[p / 2][k].push_back(((units[p].reserved[k >> 1] >> (16 * (k & 1))) & 0xffff) * 0.01);
            }
        };
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back(body, t);
        const double w0 = now_us();
        go = 1;
        for (auto& x : th) x.join();
        const double wall = now_us() - w0;
        auto med = [](std::vector<std::vector<double>>& v) { std::vector<double> a; for (auto& x : v) a.insert(a.end(), x.begin(), x.end()); std::sort(a.begin(), a.end()); return a.empty() ? 0.0 : a[a.size() / 2]; };
        printf("%s, SAO statistics of a 64x64 CTU (4:2:0, 8 bit, 5 classes x 3 planes), %2d thread%s: luma ready median %6.1f us (%4.1f us of it on the device), all planes %6.1f us (%4.1f on the device); "
               "%.0f jobs/s in total\n", mode ? "one launch per job" : "resident server   ", T, T > 1 ? "s" : " ", med(luma), med(dev0), med(whole), med(dev2), (double)T * (iters + 100) / (wall * 1e-6));
        if (stamps)
            for (int w = 0; w < 2; w++)
            {
                double m[5];
                for (int k = 0; k < 5; k++) { std::sort(sst[w][k].begin(), sst[w][k].end()); m[k] = sst[w][k].empty() ? 0 : sst[w][k][sst[w][k].size() / 2]; }
                printf("      %s plane, us since the doorbell was seen (medians): plane starts %.2f, histograms cleared %.2f, samples classified %.2f, edge classes totalled %.2f, published %.2f\n",
                       w ? "Cr  " : "luma", m[0], m[1], m[2], m[3], m[4]);
            }
        fflush(stdout);
    }
    // ---- inverse jobs (x265hip_cujob::coefMode == X265HIP_CUJOB_INVERSE): one 32x32 luma unit's levels in, its reconstructed residual + distortions out — what
    // Quant::invtransformNxN's seam would hand over behind Quant::rdoQuant.  Timed: packing (4 KB through the BAR), submit, the wait, the 2 KB copy back
    for (int T : { 1, 4, 16 })
    {
        if ((only && only[0] != 'i') || (onlyT && T != onlyT)) continue;
        std::vector<std::vector<double>> lat(T), dev(T);
        std::atomic<int> go(0);
        auto body = [&](int t)
        {
            x265hip_init(0);
            x265hip_cujob* job; void* pixels; const x265hip_cujob_unit* units; const int16_t* levels; const int16_t* resi;
            x265hip_cuserve_slot(cs, t, &job, &pixels, &units, &levels, &resi);
            std::vector<unsigned char> blob(2048 + 2048);
            uint32_t s = 4321 + t;
            for (int i = 0; i < 2048; i++) { s = s * 1664525u + 1013904223u; blob[i] = (unsigned char)(128 + ((s >> 24) & 15)); }
            int16_t* lv = (int16_t*)(blob.data() + 2048);
            for (int i = 0; i < 1024; i++) { s = s * 1664525u + 1013904223u; lv[i] = (int16_t)(((i & 31) + (i >> 5) < 12 && (s >> 28) < 6) ? (int)((s >> 20) & 7) - 3 : 0); }
            int16_t back[1024];
            while (!go.load()) {}
            for (int i = 0; i < iters + 100 && !failed; i++)
            {
                lv[0] = (int16_t)(1 + (i & 3));
                const double t0 = now_us();
                memset(job, 0, sizeof(*job));
                job->log2CUSize = 5; job->log2TrMax = 5; job->log2TrMin = 5; job->chroma = 0; job->bitDepth = 8; job->quantOffset = 85; job->coefMode = X265HIP_CUJOB_INVERSE;
                job->qpRem[0] = 2; job->qpPer[0] = 5; job->quantScale[0] = 20560; job->dequantScale[0] = 51;
                memcpy(pixels, blob.data(), blob.size());
                uint32_t seq = 0;
                if (x265hip_cuserve_submit(cs, t, &seq)) { fprintf(stderr, "submit (inverse): %s\n", x265hip_last_error()); failed = true; break; }
                uint64_t spins = 0;
                while (__atomic_load_n(&units[0].readyInv, __ATOMIC_ACQUIRE) != seq)
                {
                    __builtin_ia32_pause();
                    if ((++spins & 1023) == 0 && (x265hip_cuserve_poke(cs, t) < 0 || now_us() - t0 > 2e6)) { fprintf(stderr, "inverse job %d of thread %d did not come back\n", i, t); failed = true; break; }
                }
                memcpy(back, resi, sizeof(back));
                volatile int16_t sink = back[5]; (void)sink;
                const double t1 = now_us();
                while (__atomic_load_n(&units[0].ready, __ATOMIC_ACQUIRE) != seq && now_us() - t0 < 2e6) __builtin_ia32_pause();
                if (i >= 100) { lat[t].push_back(t1 - t0); dev[t].push_back(units[0].fwdTicks * 0.01); }
            }
        };
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back(body, t);
        go = 1;
        for (auto& x : th) x.join();
        if (failed) { printf("FAILED (mode %d, inverse jobs, %d threads)\n", mode, T); x265hip_cuserve_close(cs); return 1; }
        std::vector<double> all, dv;
        for (auto& v : lat) all.insert(all.end(), v.begin(), v.end());
        for (auto& v : dev) dv.insert(dv.end(), v.begin(), v.end());
        std::sort(all.begin(), all.end()); std::sort(dv.begin(), dv.end());
        printf("%s, inverse job of a 32x32 luma unit (8 bit), %2d thread%s: pack + submit + wait + copy back median %6.1f us, p99 %6.1f (%4.1f us of it on the device)\n",
               mode ? "one launch per job" : "resident server   ", T, T > 1 ? "s" : " ", all[all.size() / 2], all[(size_t)(all.size() * 0.99)], dv[dv.size() / 2]);
        fflush(stdout);
    }
    uint64_t jobs = 0, starts = 0, ns = 0;
    x265hip_cuserve_stats(cs, &jobs, &starts, &ns);
    printf("%llu jobs, %llu server starts, %.1f us of device time per job\n", (unsigned long long)jobs, (unsigned long long)starts, jobs ? ns * 1e-3 / jobs : 0.0);
    x265hip_cuserve_close(cs);
    return 0;
}
