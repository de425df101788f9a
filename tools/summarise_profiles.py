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

fv, grid = per_kernel("fetch")
wv, _ = per_kernel("write")
with open(os.path.join(P, f"{tag}_pmc_hbm_bytes.txt"), "w") as o:
    o.write(f"# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on: python bench.py --steps 10 --warmup 2 --cpu-frames 0   {note}\n")
    o.write("# per-launch averages in KiB as reported; gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md), see doubled column\n")
    o.write(f"{'kernel':<72}{'grid':>9}{'launches':>9}{'FETCH_KiB':>13}{'FETCHx2_KiB':>13}{'WRITE_KiB':>13}\n")
    for k in sorted(fv):
        f = fv[k]["FETCH_SIZE"]
        w = wv.get(k, {}).get("WRITE_SIZE", [0.0])
        fa, wa = sum(f) / len(f), sum(w) / len(w)
        o.write(f"{k:<72}{grid[k]:>9}{len(f):>9}{fa:>13.1f}{2 * fa:>13.1f}{wa:>13.1f}\n")

sv, _ = per_kernel("sq")
with open(os.path.join(P, f"{tag}_pmc_sq.txt"), "w") as o:
    o.write(f"# rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES; per-launch averages {note}\n")
    for k in sv:
        o.write(f"{k:<67}" + json.dumps({c: int(sum(v) / len(v)) for c, v in sorted(sv[k].items())}) + "\n")

line = os.path.join(G, f"{tag}_bench_line.json")
if os.path.exists(line):
    shutil.copy(line, os.path.join(P, f"{tag}_bench_line.json"))
print(open(os.path.join(P, f"{tag}_pmc_hbm_bytes.txt")).read())
