"""Per-launch HBM bytes of the CU-job kernels from rocprofv3's counter passes (tools/exp/gpu.sh pmc): FETCH_SIZE and WRITE_SIZE, each collected in its own
--pmc run of tools/micro/cuserve_rt in launch mode (one kernel launch per job, so that a dispatch IS a job; the resident server's single dispatch spans
the whole run).  Corrections as MI355X_MICROARCH.md's HBM section prescribes and as profiles/r04_v1_pmc_calibration.txt measured them on this pool:
FETCH_SIZE reports KiB and under-reports 16-byte reads by 2.0x; WRITE_SIZE reports KiB, factor 1.0.

    python tools/prof/pmc_launches.py <dir-of-passes> <shape:algorithmic_bytes_in:algorithmic_bytes_out[:jobs]> ...
`jobs`: the pass ran the RESIDENT server (one dispatch serves them all): the dispatches' sums are divided by it instead of by the launch count.
"""
import csv
import re
import glob
import os
import sys
from collections import defaultdict

FETCH_FACTOR, WRITE_FACTOR = 2.0, 1.0


def launches(d, counter):
    out = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                names = re.findall(r"(\w+)\s*\(", r["Kernel_Name"])
                out[names[-1] if names else r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return out


def main():
    root = sys.argv[1]
    import hashlib
    here = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    print("# sources %s" % hashlib.sha256(open(os.path.join(here, "x265_amd", "csrc", "cuserve.hip"), "rb").read()).hexdigest()[:16])
    print("# per-launch HBM traffic of one CU job / one SAO statistics job (launch mode: one dispatch = one job), rocprofv3 --pmc, separate passes")
    print("# corrected = raw KiB x 1024 x factor (FETCH_SIZE x%.1f, WRITE_SIZE x%.1f: profiles/r04_v1_pmc_calibration.txt)" % (FETCH_FACTOR, WRITE_FACTOR))
    print("%-6s %-34s %9s %14s %14s %16s %16s" % ("shape", "kernel", "launches", "fetch_B/job", "write_B/job", "algorithmic_in_B", "algorithmic_out_B"))
    for spec in sys.argv[2:]:
        shape, bin_, bout, *rest = spec.split(":")
        jobs = int(rest[0]) if rest else 0
        f = launches(os.path.join(root, shape + "_fetch"), "FETCH_SIZE")
        w = launches(os.path.join(root, shape + "_write"), "WRITE_SIZE")
        for k in sorted(f):
            if not k.startswith("cu_"):
                continue
            n = jobs or len(f[k])
            fb = sum(f[k]) / n * 1024 * FETCH_FACTOR
            wb = sum(w[k]) / (jobs or len(w[k])) * 1024 * WRITE_FACTOR if w.get(k) else float("nan")
            print("%-6s %-34s %9s %14.0f %14.0f %16s %16s" % (shape, k[:34], ("%d" % len(f[k])) + ("/%dj" % jobs if jobs else ""), fb, wb, bin_, bout))


if __name__ == "__main__":
    main()
