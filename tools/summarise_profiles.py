#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of tools/collect_profiles.sh (gpurun_out/<tag>_*) into the committed summaries under profiles/:
<tag>_bench_kernel_stats.csv, <tag>_pmc_hbm_bytes.txt, <tag>_pmc_sq.txt, <tag>_bench_line.json.   usage: summarise_profiles.py r01_v4"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
note = sys.argv[2] if len(sys.argv) > 2 else ""


def find(sub, suffix):
    hits = glob.glob(os.path.join(G, f"{tag}_{sub}", "**", f"*{suffix}"), recursive=True)
    return hits[0] if hits else None


def short(name):
    return name.split("(")[0].strip()


def per_kernel(sub):
    """{kernel: {counter: [values]}} and {kernel: grid} for our (xh::) kernels."""
    f = find(sub, "counter_collection.csv")
    vals, grid = defaultdict(lambda: defaultdict(list)), {}
    if not f:
        return vals, grid
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if "xh::" not in k:
            continue
        vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        grid[k] = r["Grid_Size"]
    return vals, grid


stats = find("stats", "kernel_stats.csv")
if stats:
    shutil.copy(stats, os.path.join(P, f"{tag}_bench_kernel_stats.csv"))

la = find("lookahead", "kernel_stats.csv")
if la:
    shutil.copy(la, os.path.join(P, f"{tag}_lookahead_kernel_stats.csv"))
    log = os.path.join(G, f"{tag}_lookahead.log")
    if os.path.exists(log):
        lines = [l for l in open(log) if l.startswith("{")]
        if lines:
            open(os.path.join(P, f"{tag}_lookahead_bench_line.json"), "w").write(lines[-1])

# correction factors of the byte counters in OUR access patterns (tools/pmc_calibrate.py; 4-byte unaligned loads for the 8-bit kernels)
calib = os.path.join(P, f"{tag}_pmc_calibration.txt")
import subprocess
subprocess.call([sys.executable, os.path.join(ROOT, "tools", "pmc_calibrate.py"), "--summarise", G, calib])
ffac, wfac = 2.0, 1.0
if os.path.exists(calib):
    for line in open(calib):
        c = line.split()
        # the counter's own scale comes from the ALIGNED pattern, where bytes fetched == bytes asked for (factor 2.000: the counter tallies
        # 128-byte requests as 64); the unaligned 4-byte pattern of the motion kernels then shows 1.125x over-fetch (each wave's 256-byte
        # span straddles one extra 32-byte sector), which is real traffic and stays in the corrected figure
        if len(c) == 6 and c[0] == "FETCH_SIZE" and c[1] == "calib_read16" and c[5] != "nan":
            ffac = float(c[5])
        if len(c) == 6 and c[0] == "WRITE_SIZE" and c[1] == "calib_write4" and c[5] != "nan":
            wfac = float(c[5])

fv, grid = per_kernel("fetch")
wv, _ = per_kernel("write")
with open(os.path.join(P, f"{tag}_pmc_hbm_bytes.txt"), "w") as o:
    o.write(f"# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on the frame pass (bench.py --frame-pass-only)   {note}\n")
    o.write(f"# fetch_correction {ffac:.3f} write_correction {wfac:.3f}: measured on known byte counts ({tag}_pmc_calibration.txt: the counter reports half the fetched bytes; 4-byte unaligned loads over-fetch 1.125x on top); "
            "per-launch averages in KiB, raw and corrected\n")
    o.write(f"{'kernel':<72}{'grid':>9}{'launches':>9}{'FETCH_raw':>13}{'WRITE_raw':>13}{'FETCH_KiB':>13}{'WRITE_KiB':>13}\n")
    for k in sorted(fv):
        f = fv[k]["FETCH_SIZE"]
        w = wv.get(k, {}).get("WRITE_SIZE", [0.0])
        fa, wa = sum(f) / len(f), sum(w) / len(w)
        o.write(f"{k:<72}{grid[k]:>9}{len(f):>9}{fa:>13.1f}{wa:>13.1f}{ffac * fa:>13.1f}{wfac * wa:>13.1f}\n")

# bench.py's lookahead_p_kernel probe (python bench.py --lookahead-probe-only): duration, byte counters, SQ counters of exactly those launches
lfv, lgrid = per_kernel("lafetch")
lwv, _ = per_kernel("lawrite")
lsv, _ = per_kernel("lasq")
lst = find("lastats", "kernel_stats.csv")
if lfv:
    with open(os.path.join(P, f"{tag}_pmc_lookahead.txt"), "w") as o:
        o.write(f"# rocprofv3 on `python bench.py --lookahead-probe-only` (lookahead_p_kernel, 8 (frame, reference) pairs of 960x544 per launch): --kernel-trace --stats, then --pmc FETCH_SIZE, --pmc WRITE_SIZE and the SQ counters, each in its own pass   {note}\n")
        o.write(f"# fetch_correction {ffac:.3f} write_correction {wfac:.3f} ({tag}_pmc_calibration.txt); per-launch averages in KiB, raw and corrected\n")
        if lst:
            for r in csv.DictReader(open(lst)):
                if "lookahead_p_kernel" in r["Name"]:
                    o.write(f"# kernel stats: {r['Calls']} launches, average {float(r['AverageNs']) / 1e3:.1f} us (min {float(r['MinNs']) / 1e3:.1f}, max {float(r['MaxNs']) / 1e3:.1f})\n")
        o.write(f"{'kernel':<72}{'grid':>9}{'launches':>9}{'FETCH_raw':>13}{'WRITE_raw':>13}{'FETCH_KiB':>13}{'WRITE_KiB':>13}\n")
        for k in sorted(lfv):
            f = lfv[k]["FETCH_SIZE"]
            w = lwv.get(k, {}).get("WRITE_SIZE", [0.0])
            fa, wa = sum(f) / len(f), sum(w) / len(w)
            o.write(f"{k:<72}{lgrid[k]:>9}{len(f):>9}{fa:>13.1f}{wa:>13.1f}{ffac * fa:>13.1f}{wfac * wa:>13.1f}\n")
        for k in lsv:
            o.write("# SQ per launch  " + f"{k:<60}" + json.dumps({c: int(sum(v) / len(v)) for c, v in sorted(lsv[k].items())}) + "\n")
    print(open(os.path.join(P, f"{tag}_pmc_lookahead.txt")).read())

enc = find("encode", "kernel_stats.csv")
if enc:
    shutil.copy(enc, os.path.join(P, f"{tag}_encode_kernel_stats.csv"))
    log = os.path.join(G, f"{tag}_encode.log")
    if os.path.exists(log):
        keep = [l for l in open(log, errors="replace") if l.startswith(("encoded", "x265hip:"))]
        open(os.path.join(P, f"{tag}_encode_log.txt"), "w").writelines(keep)

sv, _ = per_kernel("sq")
with open(os.path.join(P, f"{tag}_pmc_sq.txt"), "w") as o:
    o.write(f"# rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES; per-launch averages {note}\n")
    for k in sv:
        o.write(f"{k:<67}" + json.dumps({c: int(sum(v) / len(v)) for c, v in sorted(sv[k].items())}) + "\n")

line = os.path.join(G, f"{tag}_bench_line.json")
if os.path.exists(line):
    shutil.copy(line, os.path.join(P, f"{tag}_bench_line.json"))
print(open(os.path.join(P, f"{tag}_pmc_hbm_bytes.txt")).read())
