#!/usr/bin/env python3
"""Run the known-byte-count micro-kernels of tools/pmc_calib.hip (tools/libpmccalib.so) once; meant to be run UNDER rocprofv3:

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/calib_fetch -o c -- python tools/pmc_calibrate.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/calib_write -o c -- python tools/pmc_calibrate.py

and then summarised (python tools/pmc_calibrate.py --summarise gpurun_out profiles/r02_pmc_calibration.txt): for every kernel,
reported KiB per launch against the bytes the kernel is known to move -> the correction factor for that access pattern."""
import csv
import ctypes as C
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BYTES = 1 << 30            # 1 GiB: four times the Infinity Cache
REPS = 3


def summarise(gdir, out):
    rows = []
    for sub, counter in (("calib_fetch", "FETCH_SIZE"), ("calib_write", "WRITE_SIZE")):
        hits = glob.glob(os.path.join(gdir, sub, "**", "*counter_collection.csv"), recursive=True)
        if not hits:
            continue
        per = {}
        for r in csv.DictReader(open(hits[0])):
            if r["Counter_Name"] == counter and r["Kernel_Name"].startswith("calib_"):
                per.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
        for k, v in sorted(per.items()):
            moved = BYTES if (counter == "FETCH_SIZE") == k.startswith("calib_read") else 0
            avg = sum(v) / len(v)
            rows.append((counter, k, len(v), avg, moved / 1024.0, (moved / 1024.0 / avg) if avg and moved else float("nan")))
    with open(out, "w") as f:
        f.write("# tools/pmc_calibrate.py: rocprofv3 counters against known byte counts (1 GiB per launch, each byte touched once, buffer 4x the Infinity Cache)\n")
        f.write("# factor = known KiB / reported KiB for the kernel's own direction (reads for FETCH_SIZE, writes for WRITE_SIZE); the other rows show cross-talk\n")
        f.write("%-12s %-16s %9s %16s %14s %8s\n" % ("counter", "kernel", "launches", "reported_KiB", "known_KiB", "factor"))
        for r in rows:
            f.write("%-12s %-16s %9d %16.1f %14.1f %8.3f\n" % r)
    print(open(out).read())


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--summarise":
        return summarise(sys.argv[2], sys.argv[3])
    lib = C.CDLL(os.path.join(ROOT, "tools", "libpmccalib.so"))
    lib.pmc_calib_run.argtypes = [C.c_size_t, C.c_int]
    rc = lib.pmc_calib_run(BYTES, REPS)
    print("pmc_calib_run rc", rc)
    sys.exit(rc)


if __name__ == "__main__":
    main()
