#!/usr/bin/env python3
"""bench.py — the contract bench (BASELINE.json metric: encode fps, 1080p preset medium, --me hex).

`value` is REAL encode fps: the reference encoder's own objects with the binding's translation units (x265_amd/host/*.cpp) and libx265hip.so behind them —
integration/_build/x265_hip_8bit — encoding a synthetic 1920x1080 clip with `--preset medium --me hex` (INTEGRATION.md says what the GPU serves).  One "step" = CHUNK = 12
frames of the clip; K steps are encoded in one run of ONE encoder, bracketed by barrier + synchronize, wall clock of this process (the encoder's own "encoded N
frames in Xs" figure is reported beside it as `cli_fps`).  At N > 1 the job is the same and the encoder is still one: its device work is spread over the N GPUs
(X265HIP_DEVICES; rank 0 runs it, the other ranks wait at the fences) — strong scaling; the chunk form (N encoders, one per GPU) and BASELINE
configs[4]'s 8K shape are reported beside it, outside the timed region.  The same clip is then encoded by the unmodified reference encoder with the same
arguments (oracle/_ref/x265_8bit built with the reference's own Release flags: `cpu_baseline`, kind "reference"; `cpu_baseline_vec`: with its intrinsics
transforms), and the bitstreams must be byte-identical.  `roofline` is the kernel with the most device time in the timed encode, priced from the library's own
HIP events around every launch (x265hip_device_time); `rooflines` holds the other named kernels and the stand-alone probes; `gpu_duty_cycle` = device time /
wall clock.

Beside it (--frame-pass), `frame_pass` keeps the device-resident hot path of round 1 (quarter-pel planes, top-down motion search, prediction, residual
chains, sa8d, borders on HBM-resident pictures, F frame chains per GPU, recon exchanged over RCCL when N > 1) with its own roofline.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line (DESIGN.md §5 explains every field)."""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, DEPTH, QP, MERANGE, SUBME = 1920, 1080, 8, 28, 57, 2
VALU_SAD_CEILING_T = 95.2          # T absolute differences/s: v_qsad_pk_u16_u8 on the whole chip (tools/micro/qsad_rate)
HBM_PEAK_GBPS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MIN_TIMED_S = 0.5                 # every timed region lasts at least this long, whatever --steps says
CHUNK = 12                        # frames per step of the real encode
# frame-pass stages between the 12 HIP events of x265hip_framepass_stage_ms: quarter-pel planes, the four search levels (the 8x8 level also writes
# the luma prediction), chroma prediction, ALL residual chains (one launch: Y 32x32 + 8x8, Cb / Cr 16x16 + 4x4), sa8d, borders; "-" = unused slots
STAGES = ["planes", "me64", "me32", "me16", "me8", "pred_c", "chains", "-", "sa8d", "--", "border"]


def cpu_baseline(frames):
    import numpy as np
    """The oracle's C restatement of the SAME frame pass, single thread, on `frames` 1080p frames (checker code used
    here only as the reported CPU baseline)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from frame_oracle import oracle_frame_pass
    from x265_amd.synth import make_scene_yuv
    sc = make_scene_yuv(W, H, depth=DEPTH, seed=4321)
    crop = lambda a, hh, ww: np.ascontiguousarray(a[:hh, :ww])  # noqa: E731
    oracle_frame_pass(crop(sc["src"], 136, 200), crop(sc["ref"], 136, 200), depth=DEPTH, qp=QP,
                      src_c=(crop(sc["src_cb"], 68, 100), crop(sc["src_cr"], 68, 100)),
                      ref_c=(crop(sc["ref_cb"], 68, 100), crop(sc["ref_cr"], 68, 100)))   # load + warm
    t0 = time.perf_counter()
    ref, ref_c = sc["ref"], (sc["ref_cb"], sc["ref_cr"])
    for i in range(frames):
        r = oracle_frame_pass(sc["src"], ref, depth=DEPTH, qp=QP, merange=MERANGE, method=1, subme=SUBME,
                              src_c=(sc["src_cb"], sc["src_cr"]), ref_c=ref_c)
        ref = np.ascontiguousarray(r["recon"][96:96 + H, 96:96 + W])
        ref_c = tuple(np.ascontiguousarray(p[48:48 + H // 2, 48:48 + W // 2]) for p in r["recon_c"])
    dt = time.perf_counter() - t0
    return {"value": frames / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d frame passes of the same 1920x1080 workload (oracle/x265_oracle_frame.c, gcc -O2, 1 thread, %.1f s)" % (frames, dt)}


def _cpu_chain(frames, start_at=0.0):
    """One independent chain of `frames` oracle frame passes in a worker process, started at wall-clock time `start_at` (so that all workers
    run at the same time); returns (start, end) wall-clock times."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from frame_oracle import oracle_frame_pass
    from x265_amd.synth import make_scene_yuv
    sc = make_scene_yuv(W, H, depth=DEPTH, seed=4321 + os.getpid() % 7)
    crop = lambda a, hh, ww: np.ascontiguousarray(a[:hh, :ww])  # noqa: E731
    oracle_frame_pass(crop(sc["src"], 136, 200), crop(sc["ref"], 136, 200), depth=DEPTH, qp=QP,
                      src_c=(crop(sc["src_cb"], 68, 100), crop(sc["src_cr"], 68, 100)),
                      ref_c=(crop(sc["ref_cb"], 68, 100), crop(sc["ref_cr"], 68, 100)))
    while time.time() < start_at:
        time.sleep(0.01)
    t0 = time.time()
    ref, ref_c = sc["ref"], (sc["ref_cb"], sc["ref_cr"])
    for _ in range(frames):
        r = oracle_frame_pass(sc["src"], ref, depth=DEPTH, qp=QP, merange=MERANGE, method=1, subme=SUBME,
                              src_c=(sc["src_cb"], sc["src_cr"]), ref_c=ref_c)
        ref = np.ascontiguousarray(r["recon"][96:96 + H, 96:96 + W])
        ref_c = tuple(np.ascontiguousarray(p[48:48 + H // 2, 48:48 + W // 2]) for p in r["recon_c"])
    return t0, time.time()


def cpu_baseline_parallel(frames_per_chain=4, max_procs=64, timeout_s=120, start_delay_s=10.0):
    """The same port on many host cores: one independent frame chain per worker process (frame-level parallelism, like the GPU's chains);
    throughput = all frames / (last end - first start).  Context beside the single-core figure.  Plain subprocesses with a hard timeout:
    this leg must never be able to hang the bench."""
    procs = max(1, min(max_procs, (os.cpu_count() or 1)))
    start_at = time.time() + start_delay_s             # workers build their scene first (a few seconds), then all start together
    code = "import sys; sys.path.insert(0, %r); import bench; print('SPAN %%.6f %%.6f' %% bench._cpu_chain(%d, %.3f))" % (ROOT, frames_per_chain, start_at)
    env = dict(os.environ, OMP_NUM_THREADS="1", HIP_VISIBLE_DEVICES="")
    ps = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env) for _ in range(procs)]
    spans, deadline = [], time.time() + timeout_s
    for p in ps:
        try:
            out, _ = p.communicate(timeout=max(1.0, deadline - time.time()))
            for line in out.splitlines():
                if line.startswith("SPAN"):
                    spans.append(tuple(float(v) for v in line.split()[1:3]))
        except subprocess.TimeoutExpired:
            p.kill()
            p.communicate()
    if not spans:
        return {"error": "no worker finished within %d s" % timeout_s}
    wall = max(e for _, e in spans) - min(s for s, _ in spans)
    late = sum(1 for s0, _ in spans if s0 > start_at + 0.5)
    return {"value": len(spans) * frames_per_chain / wall, "unit": "frames/s", "cores": len(spans), "kind": "port",
            "sample": "%d processes x %d chained frame passes of the same workload started together, %.1f s%s"
                      % (len(spans), frames_per_chain, wall, (" (%d workers started late)" % late) if late else "")}


def _pmc_file():
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_bytes.txt")))
    return files[-1] if files else None


PMC_KERNELS = {"planes": "subpel_planes", "me64": "motion2_kernel<unsigned char, 64", "me32": "motion3_kernel<unsigned char, 32",
               "me16": "motion3_kernel<unsigned char, 16", "me8": "motion3_kernel<unsigned char, 8", "pred_c": "pred_chroma_kernel",
               "chains": "residual_chain_multi_kernel", "sa8d": "sa8d_pyramid_kernel", "border": "extend_border3_kernel", "-": "\0", "--": "\0"}


def pmc_traffic(stage):
    """HBM bytes per launch of the stage's kernel from the newest COMMITTED rocprofv3 PMC passes (tools/collect_profiles.sh: FETCH_SIZE and
    WRITE_SIZE in separate runs of this same workload; the file says which FETCH correction it applies, see tools/pmc_calibrate).
    (bytes, file, note) or (None, None, None)."""
    f = _pmc_file()
    if not f:
        return None, None, None
    note = "from the committed profile, not collected in this run"
    for line in open(f):
        if line.startswith("# fetch_correction"):
            note += "; " + line[1:].strip()
        if PMC_KERNELS[stage] in line:
            cols = line.split()
            try:
                return int((float(cols[-2]) + float(cols[-1])) * 1024), os.path.relpath(f, ROOT), note
            except ValueError:
                continue
    return None, None, None


def pmc_lookahead():
    """HBM bytes per launch of lookahead_p_kernel from the newest committed PMC passes of `bench.py --lookahead-probe-only`
    (profiles/r*_pmc_lookahead.txt; FETCH_SIZE and WRITE_SIZE in separate runs, corrected as the file says).  (bytes, file, note) or Nones."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_lookahead.txt")))
    if not files:
        return None, None, None
    note = "from the committed profile of this same probe, not collected in this run"
    for line in open(files[-1]):
        if line.startswith("# fetch_correction"):
            note += "; " + line[1:].strip()
        if "lookahead_p_kernel" in line and not line.startswith("#"):
            cols = line.split()
            try:
                return int((float(cols[-2]) + float(cols[-1])) * 1024), os.path.relpath(files[-1], ROOT), note
            except ValueError:
                continue
    return None, None, None


def pmc_frame_bytes():
    """Sum over the frame pass's kernels (each launched once per frame) of PMC FETCH + WRITE bytes per launch."""
    f = _pmc_file()
    if not f:
        return None, None
    total = 0
    for line in open(f):
        if line.startswith("#"):
            continue
        cols = line.split()
        try:
            total += int((float(cols[-2]) + float(cols[-1])) * 1024)
        except (ValueError, IndexError):
            continue
    return (total or None), os.path.relpath(f, ROOT)


def valu_per_frame():
    """VALU wave-instructions of one frame pass = sum over its kernels of SQ_INSTS_VALU per launch, from the committed rocprofv3 SQ pass
    (profiles/r*_pmc_sq.txt; every kernel of the pass is launched once per frame).  None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_sq.txt")))
    if not files:
        return None, None
    total = 0
    for line in open(files[-1]):
        if line.startswith("#") or "{" not in line:
            continue
        try:
            total += json.loads(line[line.index("{"):])["SQ_INSTS_VALU"]
        except (ValueError, KeyError):
            continue
    return (total or None), os.path.relpath(files[-1], ROOT)


def frame_pass_bench(args, rank, local_rank, world, steps, warmup):
    """The device-resident hot path (round 1's headline): F frame passes per GPU per step on HBM-resident pictures, recon exchanged over
    RCCL when N > 1.  The timed loop is repeated until it has run for at least MIN_TIMED_S.  Returns the `frame_pass` object (rank 0) or None."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from x265_amd import hipprim as hp
    from x265_amd.framepass import FramePass, MARGIN, algorithmic_bytes
    L = hp.lib()
    dev = torch.device("cuda", local_rank)
    stream = torch.cuda.current_stream().cuda_stream or None

    # ---- synthetic 4:2:0 input resident in HBM: a pool of source pictures per rank + the first reference.  A picture is ONE flat
    # u8 tensor [Y | Cb | Cr] (padded planes back to back) so the reference exchange is a single send/recv.
    from x265_amd.framepass import YuvStruct
    from x265_amd.synth import make_scene_yuv
    S, R = W + 2 * MARGIN, H + 2 * MARGIN
    MC = MARGIN // 2
    SC, RC = W // 2 + 2 * MC, H // 2 + 2 * MC
    YB, CB = S * R, SC * RC

    def flat(y, cb, cr):
        parts = [np.pad(y, MARGIN, mode="edge"), np.pad(cb, MC, mode="edge"), np.pad(cr, MC, mode="edge")]
        return torch.from_numpy(np.concatenate([np.ascontiguousarray(p).reshape(-1) for p in parts])).to(dev)

    def yuv(t):
        b = t.data_ptr()
        return YuvStruct(b + MARGIN * S + MARGIN, b + YB + MC * SC + MC, b + YB + CB + MC * SC + MC, S, SC)

    NPOOL = 4
    pool, ref0 = [], None
    for i in range(NPOOL):
        sc = make_scene_yuv(W, H, depth=DEPTH, seed=4321 + 17 * rank + i)
        pool.append(flat(sc["src"], sc["src_cb"], sc["src_cr"]))
        if i == 0:
            ref0 = flat(sc["ref"], sc["ref_cb"], sc["ref_cr"])
    from x265_amd.exchange import ReferenceRing
    # F frame chains per GPU (x265 runs several frame encoders per device the same way): chain j of rank g encodes frames
    # (step * N + g) * F + j; its reference is the reconstruction chain j-1 produced one step earlier (chain 0 takes the last
    # chain of rank g-1 through the RCCL ring).  The F passes of a step are independent and run on F streams, so the small
    # motion-search levels of one frame overlap the wide ones of another.
    F = max(1, args.frames_in_flight)
    ring = ReferenceRing(ref0, torch.empty_like(ref0), rank, world)
    preds = [torch.zeros_like(ref0) for _ in range(F)]
    recons = [[torch.empty_like(ref0), torch.empty_like(ref0)] for _ in range(F)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(F)] if F > 1 else [None]
    fps_ = [FramePass(W, H, depth=DEPTH, qp=QP, merange=MERANGE, method=hp.HEX_SEARCH, subme=SUBME) for _ in range(F)]
    fp = fps_[0]

    from x265_amd.exchange import FrameChains
    chains = FrameChains(ring, F, [ring.current] + [ref0] * (F - 1))

    def run_pass(h, src, ref, pred, rec, sh):
        a, b, c, d = yuv(src), yuv(ref), yuv(pred), yuv(rec)
        hp.check(L.x265hip_framepass_run_yuv(h, C.byref(a), C.byref(b), C.byref(c), C.byref(d), MARGIN, MARGIN, sh))

    # Dependencies between the F chains are per picture, tracked with events (no per-step join of all streams):
    #   done[j][k & 1]  chain j finished its pass of step k (its recon is complete)
    #   got[k & 1]      the reference for chain 0 of step k has arrived (the exchange of step k finished)
    done = [[torch.cuda.Event(), torch.cuda.Event()] for _ in range(F)]
    got = [torch.cuda.Event(), torch.cuda.Event()]

    def launch(j, k, ref):
        src, rec = pool[(k + j) % NPOOL], recons[j][k & 1]
        st = streams[j]
        if st is None:                                          # F == 1: everything in order on the current stream
            run_pass(fps_[j].h, src, ref, preds[j], rec, stream)
            return rec
        p = (k - 1) & 1
        if k == 0:
            st.wait_stream(torch.cuda.current_stream())          # input upload
        else:
            st.wait_event(got[k & 1] if j == 0 else done[j - 1][p])          # producer of this pass's reference
        if k > 1:
            # `rec` was the reference of the next chain (or went through the exchange) at step k-1: its reader must be done
            if j < F - 1:
                st.wait_event(done[j + 1][p])
            else:
                st.wait_event(got[p])
                st.wait_event(done[0][p])                        # single rank: chain 0 read it in place
        run_pass(fps_[j].h, src, ref, preds[j], rec, st.cuda_stream)
        done[j][k & 1].record(st)
        return rec

    def before_exchange(k):
        # the transfer reads the last chain's reconstruction of step k-1 and refills the inbox chain 0 read at step k-2
        if streams[0] is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(done[F - 1][(k - 1) & 1])
            if k >= 2:
                cur.wait_event(done[0][k & 1])

    def after_exchange(k):
        got[k & 1].record(torch.cuda.current_stream())

    def step():
        # x265_amd/exchange.py FrameChains: start the hand-over, launch chains 1..F-1, wait for the incoming reference, launch chain 0
        chains.step(launch, before_exchange, after_exchange)

    def profile_step():
        k = chains.k
        run_pass(fp.h, pool[k % NPOOL], chains.refs[0], preds[0], recons[0][k & 1], stream)
        chains.k = k + 1

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    fence()
    # the timed region lasts at least MIN_TIMED_S whatever --steps says: `steps` is raised (on every rank alike) from a first estimate
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    fence()
    est = (time.perf_counter() - t0) / 10
    if world > 1:
        t = torch.tensor([est], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        est = float(t.item())
    steps = max(steps, int(MIN_TIMED_S / max(est, 1e-6)) + 1)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    host_dt = time.perf_counter() - t0                     # host time to enqueue the steps (launch-bound check)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt * 1e3 / steps
    fps = world * F * steps / dt

    # ---- the dominant kernel's duration UNDER THE SAME LOAD as `fps` (F chains in flight): HIP events of chain 0's own stream
    hp.check(L.x265hip_framepass_set_profiling(fp.h, 1))
    acc_l = np.zeros(11)
    nload = 30
    ms9 = (C.c_float * 11)()
    for _ in range(nload):
        step()
        hp.check(L.x265hip_framepass_stage_ms(fp.h, ms9))
        acc_l += np.array(list(ms9))
    hp.check(L.x265hip_framepass_set_profiling(fp.h, 0))
    stage_ms_loaded = dict(zip(STAGES, (acc_l / nload).round(5).tolist()))
    fence()

    # ---- roofline of the dominant kernel: same workload, same stream, HIP events at the stage boundaries
    hp.check(L.x265hip_framepass_set_profiling(fp.h, 1))
    acc = np.zeros(11)
    nprof = max(10, min(steps, 50))
    for _ in range(nprof):
        profile_step()
        hp.check(L.x265hip_framepass_stage_ms(fp.h, ms9))
        acc += np.array(list(ms9))
    hp.check(L.x265hip_framepass_set_profiling(fp.h, 0))
    stage_ms = dict(zip(STAGES, (acc / nprof).round(5).tolist()))
    fence()

    if world > 1:
        dist.barrier()
    if rank != 0:
        return None
    ab = algorithmic_bytes(W, H, DEPTH, MERANGE)
    dom = max(STAGES, key=lambda s: stage_ms_loaded[s])
    # A motion-search level is a chain of the reference's sad / sad_x3 / sad_x4 / interpolate + satd slot calls.  ALGORITHMIC bytes per
    # launch = SURVEY.md §8d's per-call figures (sad 2WHB, sad_x3 4WHB, sad_x4 5WHB, an interpolated candidate adds the filter's
    # in + out bytes) summed over the calls the reference's search issues per PU on THIS workload, counted with the pinned CPU
    # oracle by tools/count_me_units.py (the search is bit-exact, so the GPU walks the same candidates), x the PUs of the launch.
    # These per-call bytes are mostly served by L2 / MALL (a candidate re-reads its neighbour's pixels); what HBM must deliver is the
    # unique footprint (source + reference window once + 44 B per PU), reported as `unique_footprint`, and what it did deliver is the
    # PMC figure `traffic` / `hbm_counter_frac`.
    n_by_stage = {"me64": 480, "me32": 1980, "me16": 8040, "me8": 32400}
    ME_BYTES_PER_PU = {"me64": 283422.6, "me32": 64257.7, "me16": 15408.4, "me8": 4067.6}   # tools/count_me_units.py, seed 4321
    ME_CALLS_PER_PU = {"me64": 17.58, "me32": 16.23, "me16": 15.15, "me8": 15.09}
    S_, R_ = W + 2 * MARGIN, H + 2 * MARGIN
    launch_ms = stage_ms_loaded[dom]                       # measured with F chains in flight, i.e. under the load `value` is measured under
    unique = None
    if dom in n_by_stage:
        n = n_by_stage[dom]
        dom_bytes = int(ME_BYTES_PER_PU[dom] * n)
        ub = W * H + (W + 2 * (MERANGE + 4)) * (H + 2 * (MERANGE + 4)) + n * (12 + 32)
        unique = {"bytes_per_launch": ub, "achieved": round(ub / (launch_ms * 1e-3) / 1e9, 2),
                  "frac": round(ub / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5)}
        names = {"me64": "motion2_kernel<u8,64,4,planes>", "me32": "motion3_kernel<u8,32,64>", "me16": "motion3_kernel<u8,16,16>",
                 "me8": "motion3_kernel<u8,8,16>"}
        kernel = "%s (%s: %d PUs x %.0f B = %.1f reference slot calls per PU, SURVEY 8d per-call bytes — cache-served, see unique_footprint / hbm_counter_frac)" % (
            names[dom], dom, n, ME_BYTES_PER_PU[dom], ME_CALLS_PER_PU[dom])
    elif dom == "planes":
        dom_bytes = 17 * S_ * R_                       # read the padded reference once, write 16 planes
        kernel = "subpel_planes_kernel<u8> (16 quarter-pel planes of the padded reference)"
    else:
        dom_bytes = {"pred_c": ab["pred"] // 2, "chains": ab["chain"] * 3 // 2, "sa8d": ab["sa8d"], "border": ab["border"], "-": 0, "--": 0}[dom]
        kernel = dom
    achieved = dom_bytes / (launch_ms * 1e-3) / 1e9
    traffic, traffic_src, traffic_note = pmc_traffic(dom)
    frame_bytes, _ = pmc_frame_bytes()
    # second roofline that actually binds at F frames in flight: VALU issue.  256 CUs x 4 SIMDs, one wave64 VALU instruction per
    # 4 cycles per SIMD, 2.4 GHz (MI355X_MICROARCH.md) = 614 G wave-instructions / s.
    vpf, vsrc = valu_per_frame()
    valu = None
    if vpf:
        peak = 256 * 4 / 4 * 2.4e9
        valu = {"wave_instr_per_frame": vpf, "source": vsrc, "peak_wave_instr_per_s": peak, "achieved_wave_instr_per_s": round(vpf * fps),
                "frac": round(vpf * fps / peak, 4)}
    out = {
        "value": round(fps, 2), "unit": "frame passes/s", "steps": steps, "ms_per_step": round(ms_per_step, 4), "timed_s": round(dt, 3),
        "workload": "1920x1080 8-bit 4:2:0, --me hex --merange 57 --subme 2, qp 28: frame pass = quarter-pel planes + top-down 2Nx2N "
                    "motion search (64/32/16/8) + luma/chroma prediction + dct/quant/dequant/idct/recon/sse chain (Y, Cb, Cr) + sa8d + borders; "
                    "F independent frame passes per GPU per step on F streams (x265 frame threads), each referencing the previous chain's recon; "
                    "the last chain's recon goes to the next rank (RCCL send/recv) when N > 1.  A builder-defined composition of reference "
                    "call sequences, not the encoder's own decisions (DESIGN.md §5)",
        "frames_per_step": world * F, "frames_in_flight_per_gpu": F, "pus_per_frame": 42900, "tus_per_frame": 8100,
        "roofline": {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "traffic_source": traffic_src, "traffic_note": traffic_note,
                     "hbm_counter_frac": round(traffic / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5) if traffic else None,
                     "algorithmic_bytes_per_launch": dom_bytes, "launch_ms": launch_ms, "launch_ms_solo": stage_ms[dom],
                     "launch_ms_note": "HIP events on the kernel's own stream with F chains in flight (solo single-stream figure beside it)",
                     "unique_footprint": unique},
        "whole_pass": {"pmc_bytes_per_frame": frame_bytes, "achieved": round(frame_bytes * fps / 1e9, 1) if frame_bytes else None,
                       "frac": round(frame_bytes * fps / 1e9 / HBM_PEAK_GBPS, 4) if frame_bytes else None,
                       "note": "sum over the pass's kernels of PMC FETCH + WRITE bytes per launch (committed profile) x frame passes/s of this run"},
        "valu_issue": valu,
        "stage_ms": stage_ms_loaded, "stage_ms_solo": stage_ms, "host_enqueue_ms_per_step": round(host_dt * 1e3 / steps, 4),
    }
    if world == 1 and args.cpu_frames > 0:
        out["cpu_port"] = cpu_baseline(args.cpu_frames)
        if not args.quick:
            out["cpu_port_parallel"] = cpu_baseline_parallel()
    return out


def lookahead_kernel_probe(L, hp, np_mod, pairs=8):
    """The GPU kernel that dominates the encode's device work, lookahead_p_kernel (CostEstimateGroup::estimateCUCost for a batch of (frame,
    reference) pairs), launched live on its own stream with HIP events around it: `pairs` pairs of the 960x544 lowres geometry of a 1080p
    source, one slice (x265's batch mode).  Returns (ms per launch, blocks per launch)."""
    np = np_mod
    from x265_amd.hipprim import DevBuf, LookaheadPair, check
    from x265_amd.synth import make_scene
    sc = make_scene(W, H, DEPTH, seed=4321)
    M = 96
    lw, lh = ((W // 2 + 7) // 8) * 8, ((H // 2 + 7) // 8) * 8
    ls = lw + 2 * 32
    ls += (32 - ls % 32) % 32
    mx = my = 32
    pe = (lh + 2 * my) * ls
    org = my * ls + mx
    wcu, hcu = lw // 8, lh // 8
    ncu = wcu * hcu
    st = C.c_void_p()
    check(L.x265hip_stream_create(C.byref(st)))
    planes = []
    for key in ("ref", "src"):
        src = np.ascontiguousarray(np.pad(sc[key], ((M, M), (M, M + 8)), mode="edge"))
        ds = DevBuf(src)
        pl = DevBuf.zeros((4, lh + 2 * my, ls), src.dtype)
        ptrs = (C.c_void_p * 4)(*[pl.at(k * pe + org) for k in range(4)])
        check(L.x265hip_lowres_init(DEPTH, ds.at(M * src.shape[1] + M), src.shape[1], ptrs, ls, lw, lh, mx, my, st))
        planes.append((pl, ds))
    icost, imode = DevBuf.zeros((ncu,), np.int32), DevBuf.zeros((ncu,), np.uint8)
    check(L.x265hip_lowres_intra_estimate(DEPTH, planes[1][0].at(org), ls, wcu, hcu, icost.ptr, imode.ptr, None, None, st))
    descs, keep = (LookaheadPair * pairs)(), []
    for i in range(pairs):
        o = [DevBuf.zeros((ncu, 2), np.int32), DevBuf.zeros((ncu,), np.int32), DevBuf.zeros((ncu,), np.uint16), DevBuf.zeros((hcu,), np.int32),
             DevBuf.zeros((ncu,), np.uint64)]
        d = descs[i]
        d.fenc, d.ref, d.intraCost = planes[1][0].at(org), planes[0][0].at(org), icost.ptr
        d.mvs, d.mvCosts, d.lowresCosts, d.rowSatds, d.sync = [b.ptr for b in o]
        keep.append(o)
    ddesc = DevBuf(np.frombuffer(bytes(descs), np.uint8))
    est = DevBuf.zeros((pairs, 4), np.int64)
    half = 2 * 32768
    tab = np.zeros(2 * half + 1, np.uint16)
    check(L.x265hip_mvcost_table(12, DEPTH, tab.ctypes.data, half))
    dtab = DevBuf(tab)
    ev0, ev1 = C.c_void_p(), C.c_void_p()
    check(L.x265hip_event_create(C.byref(ev0)))
    check(L.x265hip_event_create(C.byref(ev1)))
    epoch, total, reps = 0, 0.0, 0
    for it in range(3 + 20):
        epoch += 1
        check(L.x265hip_event_record(ev0, st))
        check(L.x265hip_lookahead_cost_p_batch(DEPTH, ddesc.ptr, pairs, ls, pe, wcu, hcu, hcu, 1, dtab.at(half), epoch, est.ptr, st))
        check(L.x265hip_event_record(ev1, st))
        ms = C.c_float()
        check(L.x265hip_event_elapsed_ms(ev0, ev1, C.byref(ms)))
        if it >= 3:
            total += ms.value
            reps += 1
    check(L.x265hip_stream_sync(st))
    assert not est.get()[:, 3].any()
    L.x265hip_event_destroy(ev0)
    L.x265hip_event_destroy(ev1)
    L.x265hip_stream_destroy(st)
    return total / reps, pairs * ncu


def parse_served(lines):
    """The encoder's X265HIP_VERBOSE lines (x265_amd/host/*.cpp) as numbers: the device-time ledger of x265hip_device_time per clock, the SAD
    surfaces' rows and launches, the lookahead's search launches and pairs."""
    import re
    out = {"clocks": {}}
    for l in lines:
        if "device time" in l:
            for m in re.finditer(r"(lookahead searches|other lookahead kernels|sub-pel plane bands|SAD surfaces|source energy planes|CU residual quad-tree jobs|sub-pel SATD tables) ([\d.]+) ms in (\d+) launch groups \((\d+) algorithmic bytes\)", l):
                out["clocks"][m.group(1)] = {"ms": float(m.group(2)), "launch_groups": int(m.group(3)), "algorithmic_bytes": int(m.group(4))}
            m = re.search(r"total ([\d.]+) ms", l)
            if m:
                out["device_ms"] = float(m.group(1))
        m = re.search(r"\((\d+) surfaces, (\d+) CTU rows in (\d+) launches", l)
        if m:
            out["surfaces"], out["ctu_rows"], out["surface_launches"] = int(m.group(1)), int(m.group(2)), int(m.group(3))
        m = re.search(r"(\d+) search launches of ([\d.]+) \(frame, reference\) pairs", l)
        if m:
            out["search_launches"], out["pairs_per_launch"] = int(m.group(1)), float(m.group(2))
        m = re.search(r"cuserve: (\d+) CU residual quad-trees \(CU >= (\d+)\) handed to the GPU as jobs \(([^,]+), ([\d.]+) ms of device time.*?: (\d+) forward transform\+quant units and "
                      r"(\d+) inverse units served, (\d+) \+ (\d+) calls of those CUs computed on the host; (\d+) waits of (\d+) cycles on average; (\d+) CUs not submitted", l)
        if m:
            out["cu"] = {"jobs": int(m.group(1)), "min_cu": int(m.group(2)), "handoff": m.group(3), "device_ms": float(m.group(4)), "forward_units": int(m.group(5)),
                         "inverse_units": int(m.group(6)), "host_computed": int(m.group(7)) + int(m.group(8)), "waits": int(m.group(9)), "wait_cycles": int(m.group(10)),
                         "not_submitted": int(m.group(11))}
        m = re.search(r"saostats: SAO statistics of (\d+) CTU planes .*? measured by the GPU in (\d+) jobs, (\d+) planes on the host; (\d+) waits of (\d+) cycles", l)
        if m:
            out["sao"] = {"planes": int(m.group(1)), "jobs": int(m.group(2)), "host_planes": int(m.group(3)), "waits": int(m.group(4)), "wait_cycles": int(m.group(5))}
        m = re.search(r"intrascan: the 35-mode sa8d scans of (\d+) blocks .*? measured by the GPU in (\d+) jobs, (\d+) scans on the host; (\d+) jobs left ahead when predInterSearch "
                      r"(?:returned|was entered), (\d+) of them adopted, (\d+) never asked for; (\d+) waits of (\d+) cycles", l)
        if m:
            out["intra"] = {"scans_served": int(m.group(1)), "jobs": int(m.group(2)), "host_scans": int(m.group(3)), "ahead": int(m.group(4)), "adopted": int(m.group(5)),
                            "never_asked_for": int(m.group(6)), "waits": int(m.group(7)), "wait_cycles": int(m.group(8))}
    return out


def encode_bench(args, rank, local_rank, world, fence, allmax):
    """Real encode fps of ONE encoder on ONE clip: the K * CHUNK-frame segment of make_clip(seed 4321), 1920x1080, --preset medium --me hex.
    N = 1: integration/_build/x265_hip_8bit on GPU 0.  N > 1: the SAME job, one encoder whose device work is spread over the N GPUs of the node (X265HIP_DEVICES=0..N-1:
    reference-picture mirrors and source pictures take the GPUs in turn, SAD surfaces are built where the source picture lives from replicas of the reference
    pictures fed band by band device to device — x265's frame threads <-> GPUs, the reconstructed-reference exchange of BASELINE's north_star inside libx265hip.so),
    run by rank 0 while the other ranks hold their GPUs and wait at the fences: strong scaling, total work fixed.  Timed: wall clock between two fences around the
    encoder run (after a W * CHUNK-frame warm-up run).  Afterwards, outside the timed region: the unmodified reference encoder on the same clip with the same
    arguments (`cpu_baseline`; byte-identity of the two bitstreams), the reference with its intrinsics transforms (`cpu_baseline_vec`), and at N > 1 the chunk
    form (N encoders, one per GPU, each a closed-GOP repetition of the segment, the host cores split between them) and BASELINE configs[4]'s shape (7680x4320)."""
    import hashlib
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import encode_fps as ef
    ref, hip = os.path.join(ef.REF, "x265_8bit"), os.path.join(ef.INTEG, "x265_hip_8bit")
    if not (os.path.exists(ref) and os.path.exists(hip)):
        raise SystemExit("bench.py: oracle/_ref/x265_8bit and integration/_build/x265_hip_8bit are missing — build them where /root/reference exists "
                         "(python -c 'import __graft_entry__ as g; g.build()'); they travel to the GPU box with the tree")
    from x265_amd.synth import make_clip
    frames, wframes = args.steps * CHUNK, max(args.warmup, 0) * CHUNK
    total = max(frames, wframes)
    clip = "/tmp/x265hip_bench_%d_r%d.yuv" % (os.getpid(), rank)
    if rank == 0 or world > 1:
        make_clip(clip, W, H, total, seed=4321)              # (every rank keeps a copy: the chunk form beside the timed leg encodes a repetition per rank)
    cores = os.cpu_count() or 1
    quota = host_cpu_quota()
    usable = int(min(cores, quota)) if quota else cores      # CPUs the container's cgroup grants (16 of the 256 shown on the MI355X box)
    base = ["--input", clip, "--input-res", "%dx%d" % (W, H), "--input-depth", "8", "--fps", "30", "--preset", "medium", "--me", "hex", "--hash", "1"]
    threads, threads_note, pools = [], "x265's defaults", cores
    if usable < cores:
        # x265 sizes its pool from the cores the kernel shows and does not see the cgroup's quota: 256 pool threads on 16 CPUs' worth of quota are throttled in
        # bursts.  Both encoders get the thread arguments at which the REFERENCE is fastest on this box (profiles/r06_v1_threads_sweep.txt, -O3 builds: the reference
        # 28.8 fps at --pools 16 -F 5, 31.1-31.7 at pools 20-24 with -F 5-6; the bound encoder 45.7-46.7 in every one of those cells): a pool of 1.5 x the quota
        # and six frame threads (ThreadPool::getFrameThreadsCount, threadpool.cpp:661-676, would pick 5 from the shown cores)
        pools, ft = int(round(1.5 * usable)), 6
        threads = ["--pools", str(pools), "--frame-threads", str(ft)]
        threads_note = ("--pools %d --frame-threads %d: the reference's best thread arguments on this box (cgroup quota %s of %d shown cores; profiles/r06_v1_threads_sweep.txt), "
                        "the same for both encoders" % (pools, ft, quota, cores))
    visible = os.environ.get("HIP_VISIBLE_DEVICES")
    all_devs = visible.split(",") if visible else [str(i) for i in range(world)]
    same_dev = os.environ.get("X265HIP_BENCH_SAME_DEVICE") == "1"          # debugging aid for 1-GPU boxes: every place is device 0
    env = dict(os.environ, X265HIP_VERBOSE="1", X265HIP="require", HIP_VISIBLE_DEVICES=all_devs[0] if (world == 1 or same_dev) else ",".join(all_devs[:world]))
    if world > 1:
        env["X265HIP_DEVICES"] = ",".join("0" if same_dev else str(i) for i in range(world))
    out_hip, out_ref = clip + ".gpu.hevc", clip + ".ref.hevc"
    res = {}
    try:
        if wframes and rank == 0:
            ef._run(hip, base + threads + ["--frames", str(wframes)], out_hip, env=env)
        fence()
        t0 = time.perf_counter()
        # (X265HIP_BENCH_TEST_FAIL=1 at N > 1: the one encoder is given an argument it rejects — exercises the fallback below on a box where nothing fails)
        force_fail = ["--preset", "nosuchpreset"] if (world > 1 and os.environ.get("X265HIP_BENCH_TEST_FAIL") == "1") else []
        r = ef._run(hip, base + threads + ["--frames", str(frames)] + force_fail, out_hip, env=env) if rank == 0 else {"rc": 0, "fps": None, "served": []}
        fence()
        dt = time.perf_counter() - t0
        failed = allmax(1.0 if r["rc"] else 0.0) != 0.0
        if failed and world == 1:
            raise SystemExit("bench.py: x265_hip_8bit failed: " + r["tail"])
        res = {"dt": dt, "frames": frames, "cli_fps": r["fps"], "served": r["served"], "pools": pools, "threads_note": threads_note}
        if failed:
            # N > 1 and the one encoder over all GPUs did not finish (its cross-device path has never run between two physical GPUs on the pool this was built on):
            # the line falls back to the chunk form below as its timed leg, and says so
            res["one_encoder_error"] = (r.get("tail") or "x265_hip_8bit over all GPUs failed on rank 0")[-400:]
        if not args.no_ref_encoder and rank == 0:
            t0 = time.perf_counter()
            r0 = ef._run(ref, base + threads + ["--frames", str(frames)], out_ref)
            dt_ref = time.perf_counter() - t0
            same = r0["rc"] == 0 and os.path.exists(out_hip) and not failed and hashlib.sha256(open(out_ref, "rb").read()).digest() == hashlib.sha256(open(out_hip, "rb").read()).digest()
            res["reference"] = {"cli_fps": r0["fps"], "wall_s": round(dt_ref, 2), "rc": r0["rc"], "byte_identical": same,
                                "bitstream_bytes": os.path.getsize(out_ref) if r0["rc"] == 0 else 0}
            # second baseline leg: the reference with its own intrinsics path (source/common/vec: SSE3 idct8/16/32, SSSE3 dct16/32, SSE4.1 dequant_scaling;
            # oracle/Makefile `vec`), --asm SSE4.1.  The nasm half (AVX2 / AVX-512 kernels) cannot be assembled in this image.
            vec = os.path.join(ef.REF, "x265_vec_8bit")
            if os.path.exists(vec):
                out_vec = clip + ".vec.hevc"
                try:
                    # (--no-info on both sides of the comparison: the options SEI carries the cpuid and would differ by those bytes alone)
                    t0 = time.perf_counter()
                    rv = ef._run(vec, base + threads + ["--frames", str(frames), "--asm", "SSE4.1", "--no-info"], out_vec)
                    dt_vec = time.perf_counter() - t0
                    rn = ef._run(ref, base + threads + ["--frames", str(min(frames, 40)), "--no-info"], out_ref + ".ni")
                    rm = ef._run(vec, base + threads + ["--frames", str(min(frames, 40)), "--asm", "SSE4.1", "--no-info"], out_vec + ".ni")
                    same_v = rn["rc"] == 0 and rm["rc"] == 0 and open(out_ref + ".ni", "rb").read() == open(out_vec + ".ni", "rb").read()
                    res["reference_vec"] = {"cli_fps": rv["fps"], "wall_s": round(dt_vec, 2), "rc": rv["rc"], "same_bitstream_without_info_sei": same_v}
                finally:
                    for q in (out_vec, out_vec + ".ni", out_ref + ".ni"):
                        if os.path.exists(q):
                            os.remove(q)
        if world > 1:
            fence()
            # ---- beside the timed leg, the chunk form: N encoders, one per GPU, each a closed-GOP repetition of the segment, the host's CPUs split between them
            per_rank = max(4, usable // world)
            envc = dict(os.environ, X265HIP_VERBOSE="1", X265HIP="require", HIP_VISIBLE_DEVICES=all_devs[0] if same_dev else all_devs[local_rank])
            out_chunk = clip + ".chunk.hevc"
            t0 = time.perf_counter()
            rc_ = ef._run(hip, base + ["--pools", str(per_rank), "--frames", str(frames)], out_chunk, env=envc)
            fence()
            dtc = allmax(time.perf_counter() - t0)
            bad = allmax(1.0 if rc_["rc"] else 0.0)
            # every chunk must be the reference's bitstream of the segment with the same arguments (rank 0 encodes that once)
            ref_chunk = clip + ".chunkref.hevc"
            if rank == 0:
                rr = ef._run(ref, base + ["--pools", str(per_rank), "--frames", str(frames)], ref_chunk)
                want = hashlib.sha256(open(ref_chunk, "rb").read()).hexdigest() if rr["rc"] == 0 else ""
            else:
                want = ""
            if world > 1:
                import torch.distributed as dist
                box = [want]
                dist.broadcast_object_list(box, src=0)
                want = box[0]
            mine = hashlib.sha256(open(out_chunk, "rb").read()).hexdigest() if (rc_["rc"] == 0 and os.path.exists(out_chunk)) else "-"
            differing = allmax(0.0 if (want and mine == want) else 1.0)
            for q in (out_chunk, ref_chunk):
                if os.path.exists(q):
                    os.remove(q)
            res["chunk_form"] = {"encoders": world, "fps": round(world * frames / dtc, 3) if not bad else None, "wall_s": round(dtc, 2), "pools_per_encoder": per_rank,
                                 "byte_identical_to_reference": differing == 0.0,
                                 "note": "%d encoders at the same time, one per GPU, each encoding a repetition of the segment as a closed-GOP chunk with --pools %d (the host's %d "
                                         "usable CPUs split); all frames of all encoders / wall clock" % (world, per_rank, usable)}
            # ---- ... and BASELINE configs[4]'s shape: 7680x4320 preset medium, one encoder over the N GPUs, with the reference beside it (a few frames: a reported
            # shape, not the metric)
            if rank == 0 and not failed:
                clip8 = "/tmp/x265hip_bench8k_%d.yuv" % os.getpid()
                out8, out8r = clip8 + ".gpu.hevc", clip8 + ".ref.hevc"
                try:
                    f8 = 8
                    make_clip(clip8, 7680, 4320, f8, seed=4321)
                    a8 = ["--input", clip8, "--input-res", "7680x4320", "--input-depth", "8", "--fps", "30", "--preset", "medium", "--me", "hex", "--hash", "1", "--frames", str(f8)] + threads
                    t0 = time.perf_counter(); r8 = ef._run(hip, a8, out8, env=env); d8 = time.perf_counter() - t0
                    t0 = time.perf_counter(); r8r = ef._run(ref, a8, out8r); d8r = time.perf_counter() - t0
                    same8 = r8["rc"] == 0 and r8r["rc"] == 0 and open(out8, "rb").read() == open(out8r, "rb").read()
                    res["configs4_8k"] = {"frames": f8, "places": world, "fps": round(f8 / d8, 3) if r8["rc"] == 0 else None, "cli_fps": r8["fps"],
                                          "reference_fps": round(f8 / d8r, 3) if r8r["rc"] == 0 else None, "reference_cli_fps": r8r["fps"], "byte_identical": same8,
                                          "exchange": [l for l in r8["served"] if "places:" in l]}
                except Exception as e:  # noqa: BLE001 — a secondary block must never cost the run its headline
                    res["configs4_8k"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
                finally:
                    for q in (clip8, out8, out8r):
                        if os.path.exists(q):
                            os.remove(q)
            fence()
    finally:
        for p in (clip, out_hip, out_ref):
            if os.path.exists(p):
                os.remove(p)
    return res


def sadsurf_probe(np_mod):
    """tools/sadsurf_bench.py's `frame` mode inside this process: the search-window kernel on whole 1080p pictures (510 CTUs per launch), timed by
    the library's own HIP events around each launch."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import sadsurf_bench as sb
    import x265_amd.hipprim as hp
    L = hp.lib()
    def planes_clock():
        v = [C.c_uint64() for _ in range(3)]
        L.x265hip_device_time(2, C.byref(v[0]), C.byref(v[1]), C.byref(v[2]))          # X265HIP_CLK_PLANES
        return [x.value for x in v]
    buf, stride, rows, srcs = sb.pictures(W, H, 2, 4321)
    sb.run_mode(hp, L, "frame", W, H, 1, 32, 1, buf, stride, rows, srcs[:1])           # warm-up: kernels loaded, buffers pooled
    p0 = planes_clock()
    r = sb.run_mode(hp, L, "frame", W, H, 2, 32, 4, buf, stride, rows, srcs)
    p1 = planes_clock()
    r["algorithmic_bytes_per_ctu"] = sb.unit_bytes(32)
    # in frame mode the whole reference picture becomes final at once: the sub-pel plane kernel runs over a whole padded picture per repetition
    r["planes"] = {"spans": p1[0] - p0[0], "ns": p1[1] - p0[1], "bytes": p1[2] - p0[2]}
    return r


def build_flags():
    """oracle/_ref/build_flags.txt (written by oracle/Makefile): the flags every reference object — the CPU baseline's and the ones the bound encoder links —
    and the binding's own TUs were compiled with"""
    try:
        d = dict(l.strip().split("=", 1) for l in open(os.path.join(ROOT, "oracle", "_ref", "build_flags.txt")) if "=" in l)
    except OSError:
        return None
    return {"reference_objects": d.get("REF_OPT"), "binding_tus": d.get("HOST_OPT"), "compiler": d.get("CXX"),
            "note": "the reference's own default build: CMake Release on GCC = -O3 -DNDEBUG (source/CMakeLists.txt:2-7) plus -ffast-math -mstackrealign -fno-exceptions "
                    "(:307-319); the same objects are the CPU baseline, the pinned oracle library and what the bound encoder links (rounds 1-5 used -O2)"}


def host_cpu_quota():
    """CPUs the container may use: the cgroup-v2 quota (cpu.max: "<quota> <period>" or "max ...") — on the MI355X box 16 of the 256 cores the kernel shows.
    None when there is no quota."""
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(float(q) / float(period), 2)
    except (OSError, ValueError):
        return None


def source_digest(*names):
    """sha256 over kernel sources: profiles/ files carry it, and a profile is only quoted when it was collected from the sources of this tree"""
    import hashlib
    h = hashlib.sha256()
    for n in names:
        h.update(open(os.path.join(ROOT, "x265_amd", "csrc", n), "rb").read())
    return h.hexdigest()[:16]


def pmc_profile(pattern, kernel, digest):
    """(HBM bytes per launch, file, note) of `kernel` from the newest committed PMC summary matching `pattern` whose `# sources` stamp equals `digest`
    (tools/collect_profiles.sh writes the stamp); (None, file, why) when there is none for these sources."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not files:
        return None, None, "no committed PMC profile"
    f = files[-1]
    stamp, note = None, "from the committed profile, not collected in this run"
    val = None
    for line in open(f):
        if line.startswith("# sources"):
            stamp = line.split()[-1]
        if line.startswith("# fetch_correction"):
            note += "; " + line[1:].strip()
        if kernel in line and not line.startswith("#"):
            cols = line.split()
            try:
                val = int((float(cols[-2]) + float(cols[-1])) * 1024)
            except ValueError:
                continue
    if stamp != digest:
        return None, os.path.relpath(f, ROOT), "the committed profile was collected from other kernel sources (stamp %s, tree %s): not quoted" % (stamp, digest)
    return val, os.path.relpath(f, ROOT), note


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed steps; one step = one %d-frame chunk of the encode" % CHUNK)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cpu-frames", type=int, default=10, help="frame passes of the CPU port sample inside `frame_pass` (0 = skip)")
    ap.add_argument("--no-ref-encoder", action="store_true")
    ap.add_argument("--no-frame-pass", action="store_true", help="(default since round 4; kept for old command lines)")
    ap.add_argument("--frame-pass", action="store_true",
                    help="also run the device-resident frame-pass harness (`frame_pass` block: rounds 1-2's pipeline of the kernels on HBM-resident pictures with its CPU port leg); "
                         "no product consumes it, so the default run spends the driver's time on the encode only (VERDICT r03)")
    ap.add_argument("--frame-pass-only", action="store_true", help="profiling aid (tools/collect_profiles.sh): only the frame-pass block, printed as the line")
    ap.add_argument("--quick", action="store_true", help="skip the multi-process CPU port leg")
    ap.add_argument("--lookahead-probe-only", action="store_true",
                    help="profiling aid (tools/collect_profiles.sh): only the lookahead_p_kernel probe of the roofline block, so that PMC passes see exactly its launches")
    ap.add_argument("--probe-pairs", type=int, default=8, help="--lookahead-probe-only: (frame, reference) pairs per launch")
    ap.add_argument("--frames-in-flight", type=int, default=3,
                    help="frame pass: independent passes per GPU per step, each on its own HIP stream (x265 --frame-threads inside one device)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from x265_amd import hipprim as hp
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d)" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libx265hip has no CPU fallback")
    # X265HIP_BENCH_SAME_DEVICE=1 is a debugging aid for 1-GPU boxes: every rank uses cuda:0 and the exchange runs over gloo
    same_dev = os.environ.get("X265HIP_BENCH_SAME_DEVICE") == "1"
    if same_dev:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    L = hp.lib()
    hp.check(L.x265hip_init(local_rank))
    if args.lookahead_probe_only:
        la_ms, la_blocks = lookahead_kernel_probe(L, hp, np, pairs=args.probe_pairs)
        print(json.dumps({"lookahead_probe": {"launch_ms": round(la_ms, 4), "blocks": la_blocks, "pairs": la_blocks // 8160}}), flush=True)
        return
    if world > 1:
        import datetime
        tmo = datetime.timedelta(seconds=300)              # a wedged collective must end the run with an error, not hang the box
        if same_dev:
            dist.init_process_group("gloo", timeout=tmo)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=tmo)
    dev = torch.device("cuda", local_rank)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.frame_pass_only:
        fpb = frame_pass_bench(args, rank, local_rank, world, 10, 2)
        if rank == 0:
            print(json.dumps({"frame_pass": fpb}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    def allmax(v):
        if world == 1:
            return v
        t = torch.tensor([v], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    enc = encode_bench(args, rank, local_rank, world, fence, allmax)
    dt = allmax(enc["dt"])
    fps = enc["frames"] / dt                                  # ONE encoder at every N: the same job, N GPUs (strong scaling)
    fallback = bool(enc.get("one_encoder_error")) and bool((enc.get("chunk_form") or {}).get("fps"))
    if fallback:
        fps, dt = enc["chunk_form"]["fps"], enc["chunk_form"]["wall_s"]       # the chunk form's own timed region (fences either side, MAX over ranks)

    fpb = None
    if args.frame_pass and not args.no_frame_pass:
        try:
            fpb = frame_pass_bench(args, rank, local_rank, world, max(args.steps, 20), max(args.warmup, 5))
        except Exception as e:  # noqa: BLE001  — the secondary block must never cost the run its headline
            fpb = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}

    if rank == 0:
        served = parse_served(enc["served"])
        clocks = served.get("clocks", {})
        ctu_cols = (W + 63) // 64
        # ---- SAD surfaces: the search-window kernel, live in the timed encode (the library's HIP events around every launch) -------------------------
        ss = clocks.get("SAD surfaces", {})
        ss_digest = source_digest("sadsurf.hip")
        ss_traffic, ss_tfile, ss_tnote = pmc_profile("r*_pmc_sadsurf.txt", "sadsurf_ctu_kernel", ss_digest)
        ss_block = None
        if ss.get("ms") and served.get("surface_launches"):
            secs, n = ss["ms"] * 1e-3, served["surface_launches"]
            ctus = served["ctu_rows"] * ctu_cols
            ach = ss["algorithmic_bytes"] / secs / 1e9
            tad = ctus * 16 * 4096 * 256 / secs / 1e12
            ss_block = {"bound": "valu", "kernel": "sadsurf_ctu_kernel, live in the timed encode: %d launches building %d CTU rows (%.1f CTUs per launch) of %d SAD surfaces; per CTU "
                                                   "16 exhaustive 16x16 block searches over 64 x 64 vectors (the 32x32 / 64x64 surfaces are sums of them), one v_qsad_pk_u16_u8 per "
                                                   "four absolute differences: the kernel is bound by that instruction's issue rate, not by memory"
                                                   % (n, served["ctu_rows"], ctus / n, served["surfaces"]),
                        "achieved": round(tad, 2), "peak": VALU_SAD_CEILING_T, "unit": "T absolute differences/s", "frac": round(tad / VALU_SAD_CEILING_T, 5),
                        "peak_note": "v_qsad_pk_u16_u8 issue rate of the whole chip measured by tools/micro/qsad_rate (profiles/r03_v2_sadsurf_kernel.txt): 92.9 G wave "
                                     "instructions/s x 64 lanes x 16 absolute differences",
                        "traffic": int(ss_traffic * (ctus / n) / 510.0) if ss_traffic else None, "traffic_source": ss_tfile,
                        "traffic_note": (ss_tnote + "; the profile's launches build 510 CTUs each: scaled to this run's CTUs per launch") if ss_traffic else ss_tnote,
                        "launch_ms": round(ss["ms"] / n, 5),
                        "launch_ms_note": "HIP events around every launch on the stream it runs on, inside libx265hip.so, summed over the timed encode (x265hip_device_time)",
                        "hbm": {"achieved": round(ach, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 5),
                                "algorithmic_bytes_per_launch": int(ss["algorithmic_bytes"] / n),
                                "note": "SURVEY 8d unique footprint of what a workgroup stages (one 64x64 source block + one 127x127 window per CTU) + the bytes it emits "
                                        "(16 x 16 windows and origins of 21 blocks): 33 621 B per complete CTU; rounds 1-3 charged 508 437 B (VERDICT r03)"}}
        probe_block, planes_probe = None, None
        try:
            pr = sadsurf_probe(np)
            secs = pr["kernel_ns"] * 1e-9
            ach = pr["ctus"] * pr["algorithmic_bytes_per_ctu"] / secs / 1e9
            tad = pr["ctus"] * 16 * 4096 * 256 / secs / 1e12
            probe_block = {"bound": "valu", "kernel": "sadsurf_ctu_kernel on whole 1080p pictures (tools/sadsurf_bench.py frame mode: %d launches of %d CTUs)" % (pr["launches"], pr["ctus_per_launch"]),
                           "achieved": round(tad, 2), "peak": VALU_SAD_CEILING_T, "unit": "T absolute differences/s", "frac": round(tad / VALU_SAD_CEILING_T, 5),
                           "traffic": ss_traffic, "traffic_source": ss_tfile, "traffic_note": ss_tnote, "launch_ms": round(pr["us_per_launch"] * 1e-3, 5),
                           "hbm": {"achieved": round(ach, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 5),
                                   "algorithmic_bytes_per_launch": int(pr["ctus_per_launch"] * pr["algorithmic_bytes_per_ctu"])}}
            pp = pr.get("planes") or {}
            if pp.get("spans") and pp.get("ns"):
                psecs = pp["ns"] * 1e-9
                planes_probe = {"bound": "hbm", "kernel": "subpel_planes8_kernel on whole padded 1080p pictures (%d launches inside the same probe): bytes = rows x padded width x (1 picture + 15 "
                                                          "phase planes)" % pp["spans"],
                                "achieved": round(pp["bytes"] / psecs / 1e9, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(pp["bytes"] / psecs / 1e9 / HBM_PEAK_GBPS, 5),
                                "traffic": None, "algorithmic_bytes_per_launch": int(pp["bytes"] / pp["spans"]), "launch_ms": round(pp["ns"] / pp["spans"] * 1e-6, 5),
                                "launch_ms_note": "HIP events around each launch (x265hip_device_time), after a warm-up repetition"}
        except Exception as e:  # noqa: BLE001
            probe_block = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        # ---- lookahead searches: live in the encode, and the probe at the encode's launch size ---------------------------------------------------------
        # ALGORITHMIC bytes per 8x8 lowres block of the P cost pass: SURVEY.md §8d per-call figures (sad / satd 2WHB, a quarter-pel candidate
        # adds the two half-pel blocks and the averaged one) over the calls the reference issues per block on this clip geometry, counted
        # with the pinned oracle (tools/count_lookahead_units.py: 19.77 calls, 3978 B per block, seed 4321)
        LA_BYTES_PER_BLOCK, LA_CALLS_PER_BLOCK, LA_BLOCKS = 3978.0, 19.77, 8160
        la = clocks.get("lookahead searches", {})
        la_digest = source_digest("lookahead.hip", "lasession.hip")
        la_traffic, la_tfile, la_tnote = pmc_profile("r*_pmc_lookahead.txt", "lookahead_p_kernel", la_digest)
        la_live = None
        if la.get("ms") and served.get("search_launches"):
            pairs = served["search_launches"] * served["pairs_per_launch"]
            secs = la["ms"] * 1e-3
            ach = pairs * LA_BLOCKS * LA_BYTES_PER_BLOCK / secs / 1e9
            la_live = {"bound": "hbm", "kernel": "lookahead_p_kernel<u8>, live in the timed encode: %d launches of %.1f (frame, reference) pairs of 960x544 lowres = %d 8x8 blocks "
                                                 "x %.0f B each (%.1f reference slot calls per block, SURVEY 8d per-call bytes); a latency-bound dependent chain, see DESIGN.md §5"
                                                 % (served["search_launches"], served["pairs_per_launch"], LA_BLOCKS, LA_BYTES_PER_BLOCK, LA_CALLS_PER_BLOCK),
                       "achieved": round(ach, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 5),
                       "traffic": int(la_traffic * served["pairs_per_launch"] / 35.0) if la_traffic else None, "traffic_source": la_tfile,
                       "traffic_note": (la_tnote + "; the profile's launches hold 35 pairs each: scaled to this run's pairs per launch") if la_traffic else la_tnote,
                       "algorithmic_bytes_per_launch": int(served["pairs_per_launch"] * LA_BLOCKS * LA_BYTES_PER_BLOCK),
                       "launch_ms": round(la["ms"] / served["search_launches"], 4), "launch_ms_note": "HIP events around each launch, summed over the timed encode"}
        probe_pairs = int(min(64, max(1, round(served.get("pairs_per_launch", 8)))))
        la_ms, la_blocks = lookahead_kernel_probe(L, hp, np, pairs=probe_pairs)
        la_bytes = LA_BYTES_PER_BLOCK * la_blocks
        ach = la_bytes / (la_ms * 1e-3) / 1e9
        # what HBM has to deliver when caches work: the frame's plane + the reference's four half-pel planes once per pair, 22 B of results per block
        uniq = (la_blocks // LA_BLOCKS) * 5 * 1024 * 608 + la_blocks * 22
        la_probe = {"bound": "hbm", "kernel": "lookahead_p_kernel<u8> probe: %d identical (frame, reference) pairs per launch (the encode's average launch size), %d blocks x %.0f B"
                                              % (la_blocks // LA_BLOCKS, la_blocks, LA_BYTES_PER_BLOCK),
                    "achieved": round(ach, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 5),
                    "traffic": la_traffic if probe_pairs == 35 else None, "traffic_source": la_tfile, "traffic_note": la_tnote,
                    "algorithmic_bytes_per_launch": int(la_bytes), "launch_ms": round(la_ms, 4),
                    "launch_ms_note": "HIP events on the kernel's own stream, measured in this run",
                    "unique_footprint": {"bytes_per_launch": uniq, "achieved": round(uniq / (la_ms * 1e-3) / 1e9, 2),
                                         "frac": round(uniq / (la_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5)}}
        # ---- sub-pel plane bands: the one kernel of the encode that is bandwidth-shaped ------------------------------------------------------------------
        pl = clocks.get("sub-pel plane bands", {})
        pl_block = None
        if pl.get("ms") and pl.get("launch_groups"):
            secs = pl["ms"] * 1e-3
            ach = pl["algorithmic_bytes"] / secs / 1e9
            pl_block = {"bound": "hbm", "kernel": "subpel_planes8_kernel (+ nothing else in the span), live in the timed encode: %d bands; bytes = rows x padded width x (1 picture + 15 "
                                                  "phase planes)" % pl["launch_groups"],
                        "achieved": round(ach, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 5), "traffic": None,
                        "algorithmic_bytes_per_launch": int(pl["algorithmic_bytes"] / pl["launch_groups"]), "launch_ms": round(pl["ms"] / pl["launch_groups"], 5),
                        "launch_ms_note": "HIP events around each band's launch; bands are one CTU row each (launch-latency sized)"}
        # ---- sub-pel SATD tables: the launch group with the most device time after the job server ------------------------------------------------------------------
        # per CTU of a surface row, three block levels (16 / 32 / 64), 49 quarter-pel vectors around each block's window centre.  Per-call bytes (satd =
        # 2 W H B per call) = 3 x 64 x 64 x 49 x 2 B per CTU = 1 204 224.  What the kernel asks L2 / HBM for ONCE (subpel_satd_kernel_lds, round 6: a block's
        # footprint goes through LDS once for its 49 vectors): per block of Q x Q (Q = 16; 32 for the 32x32 blocks and the four quadrants of a 64x64 block) Q + 1
        # rows of Q + 4 bytes in each of the 16 phase planes, the source block once per level, the tables written:
        # 16 x 16 x 17 x 20 + 2 x 4 x 16 x 33 x 36 + 3 x 4096 + 21 x 49 x 4 = 255 508 B per CTU (the first form's figure was 302 932: rows 4 bytes wider either side).
        # VALU: a lane measures one vector, 62 instructions per 4x4 tile (ISA count), 49 of 64 lanes at work: 3 levels x 256 tiles x 62 wave instructions x 64 lanes
        sp = clocks.get("sub-pel SATD tables", {})
        sp_block = None
        SP_CALL_BYTES_PER_CTU, SP_UNIQUE_BYTES_PER_CTU = 3 * 64 * 64 * 49 * 2, 16 * 16 * 17 * 20 + 2 * 4 * 16 * 33 * 36 + 3 * 4096 + 21 * 49 * 4
        SP_LANE_INSTR_PER_CTU, VALU_PEAK_TLANE = 3 * 256 * 62 * 64, 256 * 4 * 16 * 2.4e9 / 1e12          # lane-instructions; 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz
        sp_traffic, sp_tfile, sp_tnote = pmc_profile("r*_pmc_encode.txt", "subpel_satd_kernel", ss_digest)
        if sp.get("ms") and sp.get("launch_groups"):
            secs = sp["ms"] * 1e-3
            ctus = sp["algorithmic_bytes"] / SP_CALL_BYTES_PER_CTU
            ach = sp["algorithmic_bytes"] / secs / 1e9
            uniq = ctus * SP_UNIQUE_BYTES_PER_CTU
            sp_block = {"bound": "hbm", "kernel": "subpel_satd_kernel_lds (8 bit), live in the timed encode: %d launch groups (one launch per surface of the group) = the 7 x 7 quarter-pel "
                                                  "SATDs around the window centre of every 16x16 / 32x32 / 64x64 block of %.0f CTUs, out of the mirrors' phase planes; a block's footprint "
                                                  "staged in LDS once, a lane per vector, packed 16-bit Hadamard with v_sad_u16 against the source tile's transform"
                                                  % (sp["launch_groups"], ctus),
                        "achieved": round(ach, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 5),
                        "achieved_note": "SURVEY 8d per-call bytes (satd = 2 W H B per (block, vector) call): what the reference's calls would move; most of it is the "
                                         "same phase-plane neighbourhoods read 49 times — served by L2 / MALL, see `unique_footprint`",
                        "unique_footprint": {"bytes_per_ctu": SP_UNIQUE_BYTES_PER_CTU, "achieved": round(uniq / secs / 1e9, 2), "frac": round(uniq / secs / 1e9 / HBM_PEAK_GBPS, 5)},
                        "valu": {"lane_instructions_per_ctu": SP_LANE_INSTR_PER_CTU, "achieved": round(ctus * SP_LANE_INSTR_PER_CTU / secs / 1e12, 2), "peak": round(VALU_PEAK_TLANE, 1),
                                 "unit": "T lane-instructions/s", "frac": round(ctus * SP_LANE_INSTR_PER_CTU / secs / 1e12 / VALU_PEAK_TLANE, 4),
                                 "note": "what binds the kernel: 62 VALU instructions per 4x4 tile and vector (ISA count), 64 lanes issued for 49 vectors; peak = 256 CUs x 4 SIMDs x "
                                         "16 lanes x 2.4 GHz, of which the resident job server's workgroups hold 64 CUs; the time is the launch GROUP's (two or three launches)"},
                        "traffic": sp_traffic, "traffic_source": sp_tfile,
                        "traffic_note": (sp_tnote + "; per LAUNCH of the profile's run (one surface's rows per launch), not per launch group") if sp_traffic else sp_tnote,
                        "algorithmic_bytes_per_launch": int(sp["algorithmic_bytes"] / sp["launch_groups"]), "launch_ms": round(sp["ms"] / sp["launch_groups"], 5),
                        "launch_ms_note": "HIP events around each launch group on its stream (x265hip_device_time), summed over the timed encode"}
        # ---- CU residual quad-tree jobs: a host thread WAITS for each, so the path is built for round trip, not for bytes per second ------------------------------
        cu = clocks.get("CU residual quad-tree jobs", {})
        cu_block = None
        # HBM bytes of ONE 32x32 CU job on the resident server (tools/exp/gpu.sh pmc: FETCH_SIZE and WRITE_SIZE of the server's dispatch / the jobs it served)
        cu_traffic, cu_tfile, cu_tnote = None, None, "no committed PMC profile"
        import glob as _glob
        for f in sorted(_glob.glob(os.path.join(ROOT, "profiles", "r*_cuserve_pmc_per_job.txt")))[-1:]:
            cu_tfile = os.path.relpath(f, ROOT)
            stamp = None
            for line in open(f):
                c = line.split()
                if line.startswith("# sources"):
                    stamp = c[-1]
                if c and c[0] == "srv5" and len(c) >= 7:
                    cu_traffic = int(float(c[3]) + float(c[4]))
                    cu_tnote = ("one 32x32 CU job (4:2:0, 8 bit) on the resident server: corrected FETCH_SIZE %s + WRITE_SIZE %s bytes per job against %s algorithmic bytes in "
                                "(levels, residual and unit records go to page-locked host memory and are not HBM writes); from the committed profile, not collected in this run"
                                % (c[3], c[4], c[5]))
            if stamp != source_digest("cuserve.hip"):
                cu_traffic, cu_tnote = None, "the committed profile was collected from other kernel sources (stamp %s): not quoted" % stamp
        if cu.get("ms") and served.get("cu"):
            c = served["cu"]
            secs = cu["ms"] * 1e-3
            all_jobs = c["jobs"] + (served.get("sao") or {}).get("jobs", 0) + (served.get("intra") or {}).get("jobs", 0)
            ach = cu["algorithmic_bytes"] / secs / 1e9
            cu_block = {"bound": "latency", "kernel": "cu_server_kernel, live in the timed encode (%s): %d jobs = the transform arithmetic (MFMA dct -> quant -> sign-bit hiding -> dequant -> MFMA "
                                                      "idct -> two SSEs) of the residual quad-trees of %d CUs >= %dx%d, %d forward and %d inverse units served to Quant::transformNxN / "
                                                      "invtransformNxN; a pair of workgroups per mailbox slot (luma: the four waves on one 32x32 unit at a time; chroma); data path: the host writes header + pixels into the slot's device-memory half through the large BAR (posted PCIe writes), "
                                                      "the workgroup reads them HBM -> LDS, results go LDS -> page-locked host memory (posted writes again)"
                                                      % (c["handoff"], c["jobs"], c["jobs"], c["min_cu"], c["min_cu"], c["forward_units"], c["inverse_units"]),
                        # SURVEY.md 8d's roofline for this family: fused-chain bytes of the units / the server's wall time (the timed region: the resident
                        # kernel is on the chip for all of it) / HBM peak.  It is of the order of 1e-4 and will stay there: the path is a latency chain
                        "achieved": round(cu["algorithmic_bytes"] / dt / 1e9, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": round(cu["algorithmic_bytes"] / dt / 1e9 / HBM_PEAK_GBPS, 6), "traffic": cu_traffic, "traffic_source": cu_tfile, "traffic_note": cu_tnote,
                        "achieved_note": "fused-chain bytes (SURVEY 8d: source + prediction in, levels + reconstructed residual out) of every unit of every job / wall clock "
                                         "of the timed region; per busy workgroup-second instead of per wall second: see `busy`",
                        "busy": {"achieved": round(ach, 3), "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 6), "note": "the same bytes / the workgroup-seconds spent on them "
                                 "(a CU job keeps its luma and its chroma workgroup busy for different times; both count)"},
                        "peak_note": "HBM3E 8 TB/s (MI355X_MICROARCH.md).  The figure of merit of this kernel is not this fraction: the job is a dependent chain (doorbell seen -> header + pixels from HBM -> two MFMA passes -> quantise -> sign hiding "
                                     "-> levels out -> ready word -> two MFMA passes -> reconstruction, SSE, psy energy -> ready word) — the four waves of a workgroup on a 32x32 luma unit, "
                                     "one wave per chroma tile — and its figure of merit is the round trip, not bytes per second: profiles/r06_v1_cuserve_rt_stamps.txt (6.6 us from submit "
                                     "to the first luma unit's forward half, 3.1 us of it on the device; 8.9 us per 32x32 CU job; round 5: 8.4 / 4.3 / 11.4), against a transport floor of "
                                     "2.4-2.9 us for an empty ping-pong on this box (profiles/r04_v1_bar_mailbox_breakdown.txt)",
                        # the server's slots also carry the SAO statistics jobs (one per CTU plane: SAO::calcSaoStatsCTU's classification of a deblocked plane against
                        # its source; bytes = the two blocks in + 2 x 5 x 32 int32 out): busy time and bytes are over both kinds
                        "sao_statistics_jobs": served.get("sao"),
                        # ... and the intra mode scans of Search::checkIntraInInter (35 predictions + sa8d of one block per job; bytes = the two neighbour lines and
                        # the source block in, 35 costs out)
                        "intra_scan_jobs": served.get("intra"),
                        "busy_us_per_job": round(cu["ms"] * 1e3 / all_jobs, 2),
                        "algorithmic_bytes_per_job": int(cu["algorithmic_bytes"] / all_jobs),
                        "host_waits": {"count": c["waits"], "mean_cycles": c["wait_cycles"]}}
        # the dominant kernel of the timed region = the clock with the most device time
        named = [(ss.get("ms", 0.0), ss_block), (la.get("ms", 0.0), la_live), (cu.get("ms", 0.0), cu_block)]
        named = [b for _, b in sorted(named, key=lambda t: -t[0]) if b]
        dominant = named[0] if named else la_probe
        others = [b for b in (ss_block, probe_block, cu_block, sp_block, la_live, la_probe, pl_block, planes_probe) if b and b is not dominant]
        out = {
            "metric": "encode fps (1080p preset medium)", "value": round(fps, 3), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3 / args.steps, 3),
            "higher_is_better": True, "scaling": "weak" if (world == 1 or fallback) else "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "x265 --preset medium --me hex, 1920x1080 8-bit 4:2:0 (BASELINE configs[1]); ONE synthetic clip = the %d-frame segment of x265_amd/synth.make_clip "
                                   "(96x96 tiles with their own velocities + noise, seed 4321), encoded by ONE encoder (one step = %d frames): the reference encoder's objects + "
                                   "x265_amd/host/*.cpp + libx265hip.so (integration/_build/x265_hip_8bit)%s: lookahead frame-cost "
                                   "estimates batched on the GPU, the residual quad-trees of CUs >= 32x32 (MFMA dct / quant / sign hiding / dequant / idct) as mailbox jobs to a resident GPU "
                                   "server, SAO statistics and the 35-mode intra scans of inter slices as jobs of the same server, integer-pel SADs and sub-pel SATDs of the motion search "
                                   "looked up in GPU-built SAD surfaces, luma sub-pel filter slots served from GPU-built fractional planes of each reference picture, psy-cost source halves "
                                   "from GPU-built energy planes, C slots otherwise"
                                   % (enc["frames"], CHUNK, "" if world == 1 else ", its device work spread over the %d GPUs of the node (X265HIP_DEVICES: places; reconstructed rows of "
                                      "reference pictures pushed device to device with hipMemcpyPeerAsync over xGMI — a copy between two devices of one process, which is what the exchange "
                                      "is; RCCL would need one process per GPU and the encoder is one process)" % world),
                       "frames_per_step": CHUNK, "encoder_cli_fps_rank0": enc["cli_fps"], "served_by_gpu": enc["served"],
                       "build_flags": build_flags(), "host_cores": os.cpu_count(), "host_cpu_quota": host_cpu_quota(), "pool_threads_per_encoder": enc["pools"], "threads": enc["threads_note"], "timed_s": round(dt, 2),
                       "host_note": "host_cpu_quota = CPUs the container's cgroup grants (cpu.max); when it is far below host_cores both encoders are bound by CPU seconds per "
                                    "frame and N encoders share the same budget (DESIGN.md §4c)"},
            "roofline": dominant,
            "rooflines": others,
            "gpu_duty_cycle": {"device_ms": served.get("device_ms"), "timed_s": round(dt, 2),
                               "frac": round(served["device_ms"] * 1e-3 / dt, 4) if served.get("device_ms") else None,
                               "launch_groups_frac": round((served["device_ms"] - cu.get("ms", 0.0)) * 1e-3 / dt, 4) if served.get("device_ms") else None,
                               "cu_job_workgroup_seconds_per_second": round(cu.get("ms", 0.0) * 1e-3 / dt, 4),
                               "by_clock_ms": {k: v["ms"] for k, v in clocks.items()},
                               "note": "rank 0's encoder: sum of the device time of every launch group of the bound modules (x265hip_device_time) / wall clock of the timed "
                                       "region (launch_groups_frac), plus the busy time of the CU-job server's workgroups — each a one-CU job, several at a time, so `frac` can "
                                       "exceed 1: it is workgroup-seconds per second, i.e. CUs kept busy on average; the weight analysis of the lookahead (about 1 % of the "
                                       "device time) is not inside a span"},
        }
        if "reference" in enc:
            r0 = enc["reference"]
            ref_fps = enc["frames"] / r0["wall_s"] if r0["wall_s"] else None
            quota = host_cpu_quota()
            out["cpu_baseline"] = {"value": round(ref_fps, 3) if ref_fps else None, "unit": "frames/s",
                                   "cores": int(min(os.cpu_count() or 1, quota)) if quota else os.cpu_count(), "kind": "reference",
                                   "build_flags": (build_flags() or {}).get("reference_objects"),
                                   "cores_note": "CPUs the process tree can use at once: min(cores the kernel shows = %s, cgroup quota = %s); encoder threads: %s (the same for both encoders)"
                                                 % (os.cpu_count(), quota, enc["threads_note"]),
                                   "sample": "the same %d frames, same arguments, through oracle/_ref/x265_8bit — the unmodified reference, [noasm] C primitives at the reference's "
                                             "own Release flags: no nasm in the image, so the AVX2 / AVX-512 path cannot be built (cpu_baseline_vec: its intrinsics path) — measured "
                                             "like `value`: all frames / wall clock of the run (%.1f s)" % (enc["frames"], r0["wall_s"]),
                                   "cli_fps_rank0": r0["cli_fps"],
                                   "byte_identical_to_gpu_path": r0["byte_identical"], "bitstream_bytes": r0["bitstream_bytes"]}
            if not r0["byte_identical"]:
                out["error"] = "a chunk's GPU-path bitstream differs from the reference encoder's"
            rv = enc.get("reference_vec")
            if rv and rv.get("wall_s") and rv["rc"] == 0:
                out["cpu_baseline_vec"] = {"value": round(enc["frames"] / rv["wall_s"], 3), "unit": "frames/s", "cores": out["cpu_baseline"]["cores"], "kind": "reference+vec",
                                           "build_flags": (build_flags() or {}).get("reference_objects"), "cli_fps": rv["cli_fps"],
                                           "same_bitstream_without_info_sei": rv["same_bitstream_without_info_sei"],
                                           "sample": "the same %d frames through oracle/_ref/x265_vec_8bit --asm SSE4.1: the reference with its own compiler-intrinsics transforms "
                                                     "(source/common/vec: idct8/16/32 SSE3, dct16/32 SSSE3, dequant_scaling SSE4.1 — the reference's SIMD for the MFMA rows that needs "
                                                     "no assembler); every other slot is the C primitive, as in `cpu_baseline` (the .asm kernels need nasm, absent from the image)" % enc["frames"]}
        if enc.get("one_encoder_error"):
            out["one_encoder_error"] = enc["one_encoder_error"]
            out["config"]["fallback"] = ("the one encoder over all %d GPUs failed (see one_encoder_error): `value` is the CHUNK FORM — %d encoders, one per GPU, each a closed-GOP "
                                         "repetition of the segment, weak scaling" % (world, world)) if fallback else "the one encoder over all GPUs failed and so did the chunk form"
            if fallback and not enc["chunk_form"].get("byte_identical_to_reference"):
                out["error"] = "a chunk's GPU-path bitstream differs from the reference encoder's"
            elif not fallback:
                out["error"] = "no bound encode finished"
            elif out.get("error"):
                del out["error"]
        for k in ("chunk_form", "configs4_8k"):
            if enc.get(k):
                out[k] = enc[k]
        if world > 1:
            out["exchange"] = [l for l in enc["served"] if "places:" in l]
        if fpb:
            out["frame_pass"] = fpb
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
