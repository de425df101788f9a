#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of tools/collect_profiles_r03.sh (gpurun_out/<tag>_*) into the committed summaries under profiles/:
<tag>_pmc_sadsurf.txt, <tag>_pmc_lookahead.txt (both stamped `# sources <sha256 prefix of the kernel sources>`: bench.py only quotes a profile whose
stamp matches its own tree), <tag>_pmc_calibration.txt, <tag>_encode_kernel_stats.csv, <tag>_encode_log.txt.   usage: summarise_profiles_r03.py r03_v3"""
import csv, glob, hashlib, json, os, shutil, subprocess, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def digest(*names):
    h = hashlib.sha256()
    for n in names:
        h.update(open(os.path.join(ROOT, "x265_amd", "csrc", n), "rb").read())
    return h.hexdigest()[:16]


def find(sub, suffix):
    hits = glob.glob(os.path.join(G, f"{tag}_{sub}", "**", f"*{suffix}"), recursive=True)
    return hits[0] if hits else None


def per_kernel(sub):
    f = find(sub, "counter_collection.csv")
    vals, grid = defaultdict(lambda: defaultdict(list)), {}
    if not f:
        return vals, grid
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].strip()
        if "xh::" not in k:
            continue
        vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        grid[k] = r["Grid_Size"]
    return vals, grid


calib = os.path.join(P, f"{tag}_pmc_calibration.txt")
subprocess.call([sys.executable, os.path.join(ROOT, "tools", "pmc_calibrate.py"), "--summarise", G, calib])
ffac, wfac = 2.0, 1.0
if os.path.exists(calib):
    for line in open(calib):
        c = line.split()
        if len(c) == 6 and c[0] == "FETCH_SIZE" and c[1] == "calib_read16" and c[5] != "nan":
            ffac = float(c[5])
        if len(c) == 6 and c[0] == "WRITE_SIZE" and c[1] == "calib_write4" and c[5] != "nan":
            wfac = float(c[5])


def summary(prefix, outname, what, kernel, sources):
    fv, grid = per_kernel(prefix + "fetch")
    wv, _ = per_kernel(prefix + "write")
    sv, _ = per_kernel(prefix + "sq")
    st = find(prefix + "stats", "kernel_stats.csv")
    if not fv:
        print("no counters for", prefix)
        return
    path = os.path.join(P, f"{tag}_{outname}.txt")
    with open(path, "w") as o:
        o.write(f"# rocprofv3 on `{what}`: --kernel-trace --stats, then --pmc FETCH_SIZE, --pmc WRITE_SIZE and the SQ counters, each in its own pass\n")
        o.write(f"# sources {' '.join(sources)} {digest(*sources)}\n")
        o.write(f"# fetch_correction {ffac:.3f} write_correction {wfac:.3f} ({tag}_pmc_calibration.txt); per-launch averages in KiB, raw and corrected\n")
        if st:
            for r in csv.DictReader(open(st)):
                if kernel in r["Name"]:
                    o.write(f"# kernel stats: {r['Calls']} launches, average {float(r['AverageNs']) / 1e3:.1f} us (min {float(r['MinNs']) / 1e3:.1f}, max {float(r['MaxNs']) / 1e3:.1f})\n")
        o.write(f"{'kernel':<72}{'grid':>9}{'launches':>9}{'FETCH_raw':>13}{'WRITE_raw':>13}{'FETCH_KiB':>13}{'WRITE_KiB':>13}\n")
        for k in sorted(fv):
            f = fv[k]["FETCH_SIZE"]
            w = wv.get(k, {}).get("WRITE_SIZE", [0.0])
            fa, wa = sum(f) / len(f), sum(w) / len(w)
            o.write(f"{k:<72}{grid[k]:>9}{len(f):>9}{fa:>13.1f}{wa:>13.1f}{ffac * fa:>13.1f}{wfac * wa:>13.1f}\n")
        for k in sv:
            o.write("# SQ per launch  " + f"{k:<60}" + json.dumps({c: int(sum(v) / len(v)) for c, v in sorted(sv[k].items())}) + "\n")
        log = os.path.join(G, f"{tag}_{prefix}stats.log")
        if os.path.exists(log):
            for l in open(log, errors="replace"):
                if l.startswith("{"):
                    o.write("# run: " + l)
    print(open(path).read())


summary("ss", "pmc_sadsurf", "python tools/sadsurf_bench.py --modes frame --reps 3 (search-window kernel, 510 CTUs of 1920x1080 per launch)", "sadsurf_ctu_kernel", ["sadsurf.hip"])
summary("la", "pmc_lookahead", "python bench.py --lookahead-probe-only --probe-pairs 35 (lookahead_p_kernel, 35 (frame, reference) pairs of 960x544 per launch)",
        "lookahead_p_kernel", ["lookahead.hip", "lasession.hip"])
enc = find("encode", "kernel_stats.csv")
if enc:
    shutil.copy(enc, os.path.join(P, f"{tag}_encode_kernel_stats.csv"))
    log = os.path.join(G, f"{tag}_encode.log")
    if os.path.exists(log):
        keep = [l for l in open(log, errors="replace") if l.startswith(("encoded", "x265hip:"))]
        open(os.path.join(P, f"{tag}_encode_log.txt"), "w").writelines(keep)
# round 4: CU jobs with one launch each (tools/micro/cuserve_rt mode 1): rocprofv3's average duration of cu_job_kernel beside the per-job device time the
# resident server's own busy-time ledger reports (bench.py roofline.busy_us_per_job)
enc1 = find("encode_mode1", "kernel_stats.csv")
if enc1:
    shutil.copy(enc1, os.path.join(P, f"{tag}_encode_mode1_kernel_stats.csv"))
    log = os.path.join(G, f"{tag}_encode_mode1.log")
    if os.path.exists(log):
        keep = [l for l in open(log, errors="replace") if l.startswith(("one launch per job", "resident server")) or " jobs, " in l]
        open(os.path.join(P, f"{tag}_encode_mode1_log.txt"), "w").writelines(keep)
