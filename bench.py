#!/usr/bin/env python3
"""bench.py — the contract bench.  One "step" = one frame pass (the P-frame hot path: top-down HEX motion search,
prediction, residual chain, sa8d costs, border extension — include/x265hip.h "frame pass") over one 1920x1080 synthetic
frame per GPU (BASELINE.json configs[1]: 1080p, --me hex, merange 57, subme 2, 8-bit).  Frames shard across ranks the way
x265's frame threads do: rank g owns frames g, g+N, ...; after every step each rank sends its border-extended
reconstruction to rank g+1 (RCCL send/recv ring shift) where it is the reference of that rank's next frame.

  python bench.py --gpus 1 --steps 200 --warmup 20
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line (see DESIGN.md §5 for every field)."""
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
HBM_PEAK_GBPS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
STAGES = ["planes", "me64", "me32", "me16", "me8", "pred8", "chain32", "chain8", "sa8d", "chroma", "border"]


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


def reference_encoder(frames=12):
    """Context only: the REAL reference CLI ([noasm] C path, built into oracle/_ref by oracle/Makefile) encoding the same
    kind of clip at 1080p preset medium --me hex on all host cores.  A full encoder, not the same workload."""
    exe = os.path.join(ROOT, "oracle", "_ref", "x265_8bit")
    if not os.path.exists(exe):
        return None
    import numpy as np
    from x265_amd.synth import make_clip
    path = "/tmp/x265hip_bench_%d.yuv" % os.getpid()
    try:
        make_clip(path, W, H, frames, seed=4321)
        cmd = [exe, "--input", path, "--input-res", "%dx%d" % (W, H), "--fps", "30", "--preset", "medium", "--me", "hex",
               "--frames", str(frames), "-o", "/dev/null"]
        t0 = time.perf_counter()
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
        dt = time.perf_counter() - t0
        fps = None
        for line in (p.stderr + p.stdout).splitlines():
            if "encoded" in line and "fps" in line:
                fps = float(line.split("(")[1].split("fps")[0])
        return {"fps": fps, "cores": os.cpu_count(), "frames": frames, "wall_s": round(dt, 1), "cmd": " ".join(cmd[3:]),
                "note": "full x265 encoder, [noasm] C primitives (no nasm in the image); reported for context, not the same workload"}
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)[:200]}
    finally:
        if os.path.exists(path):
            os.remove(path)


def pmc_traffic(stage):
    """HBM bytes per launch of the stage's kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected
    in separate runs of this same command; FETCH doubled per the gfx950 correction of MI355X_MICROARCH.md).  None if absent."""
    names = {"planes": "subpel_planes", "me64": "motion2_kernel<unsigned char, 64", "me32": "motion3_kernel<unsigned char, 32",
             "me16": "motion3_kernel<unsigned char, 16", "me8": "motion3_kernel<unsigned char, 8", "pred8": "pred_from_planes_kernel",
             "chain32": "residual_chain_kernel<unsigned char, 32", "chain8": "residual_chain_kernel<unsigned char, 8",
             "sa8d": "sa8d_levels_kernel", "border": "extend_border3_kernel", "chroma": "pred_chroma_kernel"}
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_bytes.txt")))
    if not files:
        return None, None
    for line in open(files[-1]):
        if names[stage] in line:
            cols = line.split()
            try:
                fetch2_kib, write_kib = float(cols[-2]), float(cols[-1])
                return int((fetch2_kib + write_kib) * 1024), os.path.relpath(files[-1], ROOT)
            except ValueError:
                continue
    return None, None


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--cpu-frames", type=int, default=30, help="frames of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-ref-encoder", action="store_true")
    ap.add_argument("--frames-in-flight", type=int, default=3,
                    help="independent frame passes per GPU per step, each on its own HIP stream (x265 --frame-threads inside one device)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from x265_amd import hipprim as hp
    from x265_amd.framepass import FramePass, MARGIN, algorithmic_bytes
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
    if world > 1:
        if same_dev:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
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

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    host_dt = time.perf_counter() - t0                     # host time to enqueue the steps (launch-bound check)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt * 1e3 / args.steps
    fps = world * F * args.steps / dt

    # ---- roofline of the dominant kernel: same workload, same stream, HIP events at the stage boundaries
    hp.check(L.x265hip_framepass_set_profiling(fp.h, 1))
    acc = np.zeros(11)
    nprof = max(10, min(args.steps, 50))
    ms9 = (C.c_float * 11)()
    for _ in range(nprof):
        profile_step()
        hp.check(L.x265hip_framepass_stage_ms(fp.h, ms9))
        acc += np.array(list(ms9))
    hp.check(L.x265hip_framepass_set_profiling(fp.h, 0))
    stage_ms = dict(zip(STAGES, (acc / nprof).round(5).tolist()))
    fence()

    if rank == 0:
        ab = algorithmic_bytes(W, H, DEPTH, MERANGE)
        dom = max(STAGES, key=lambda s: stage_ms[s])
        # A motion-search level is a chain of the reference's sad / sad_x3 / sad_x4 / interpolate + satd slot calls.  ALGORITHMIC bytes per
        # launch = SURVEY.md §8d's per-call figures (sad 2WHB, sad_x3 4WHB, sad_x4 5WHB, an interpolated candidate adds the filter's
        # in + out bytes) summed over the calls the reference's search issues per PU on THIS workload, counted with the pinned CPU
        # oracle by tools/count_me_units.py (the search is bit-exact, so the GPU walks the same candidates), x the PUs of the launch.
        # The unique footprint (source + reference window once + 44 B per PU), which is all HBM must deliver when caches work, is
        # reported next to it as `unique_footprint`.
        n_by_stage = {"me64": 480, "me32": 1980, "me16": 8040, "me8": 32400}
        ME_BYTES_PER_PU = {"me64": 283422.6, "me32": 64257.7, "me16": 15408.4, "me8": 4067.6}   # tools/count_me_units.py, seed 4321
        ME_CALLS_PER_PU = {"me64": 17.58, "me32": 16.23, "me16": 15.15, "me8": 15.09}
        S_, R_ = W + 2 * MARGIN, H + 2 * MARGIN
        unique = None
        if dom in n_by_stage:
            n = n_by_stage[dom]
            dom_bytes = int(ME_BYTES_PER_PU[dom] * n)
            ub = W * H + (W + 2 * (MERANGE + 4)) * (H + 2 * (MERANGE + 4)) + n * (12 + 32)
            unique = {"bytes_per_launch": ub, "achieved": round(ub / (stage_ms[dom] * 1e-3) / 1e9, 2),
                      "frac": round(ub / (stage_ms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5)}
            names = {"me64": "motion2_kernel<u8,64,4,planes>", "me32": "motion3_kernel<u8,32,64>", "me16": "motion3_kernel<u8,16,16>",
                     "me8": "motion3_kernel<u8,8,16>"}
            kernel = "%s (%s: %d PUs x %.0f B = %.1f reference slot calls per PU, SURVEY 8d per-call bytes)" % (
                names[dom], dom, n, ME_BYTES_PER_PU[dom], ME_CALLS_PER_PU[dom])
        elif dom == "planes":
            dom_bytes = 17 * S_ * R_                       # read the padded reference once, write 16 planes
            kernel = "subpel_planes_kernel<u8> (16 quarter-pel planes of the padded reference)"
        else:
            dom_bytes = {"pred8": ab["pred"], "chain32": ab["chain"], "chain8": ab["chain"], "sa8d": ab["sa8d"], "border": ab["border"],
                         "chroma": ab["chain"] // 2 + ab["pred"] // 2}[dom]
            kernel = dom
        achieved = dom_bytes / (stage_ms[dom] * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(dom)
        # second roofline that actually binds at F frames in flight: VALU issue.  256 CUs x 4 SIMDs, one wave64 VALU instruction per
        # 4 cycles per SIMD, 2.4 GHz (MI355X_MICROARCH.md) = 614 G wave-instructions / s.
        vpf, vsrc = valu_per_frame()
        valu = None
        if vpf:
            peak = 256 * 4 / 4 * 2.4e9
            valu = {"wave_instr_per_frame": vpf, "source": vsrc, "peak_wave_instr_per_s": peak, "achieved_wave_instr_per_s": round(vpf * fps),
                    "frac": round(vpf * fps / peak, 4)}
        out = {
            "metric": "encode fps (1080p preset medium hot path: frame passes per second)", "value": round(fps, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "1920x1080 8-bit 4:2:0, --me hex --merange 57 --subme 2, qp 28: frame pass = quarter-pel planes + top-down 2Nx2N "
                                   "motion search (64/32/16/8) + luma/chroma prediction + dct/quant/dequant/idct/recon/sse chain (Y, Cb, Cr) + sa8d + borders; "
                                   "F independent frame passes per GPU per step on F streams (x265 frame threads), each referencing the previous chain's recon; "
                                   "the last chain's recon goes to the next rank (RCCL send/recv) when N > 1",
                       "frames_per_step": world * F, "frames_in_flight_per_gpu": F, "pus_per_frame": 42900, "tus_per_frame": 8100},
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": dom_bytes, "launch_ms": stage_ms[dom], "unique_footprint": unique},
            "valu_issue": valu,
            "stage_ms": stage_ms, "host_enqueue_ms_per_step": round(host_dt * 1e3 / args.steps, 4),
        }
        if world == 1 and args.cpu_frames > 0:
            out["cpu_baseline"] = cpu_baseline(args.cpu_frames)
            out["cpu_baseline_parallel"] = cpu_baseline_parallel()
            if not args.no_ref_encoder:
                out["reference_encoder"] = reference_encoder()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
