"""Per-launch HBM traffic of every kernel of a run from rocprofv3's counter passes (FETCH_SIZE and WRITE_SIZE, each in its own --pmc pass; tools/exp/gpu.sh
pmc_encode), in the format bench.py's pmc_profile() reads: kernel, grid of the first launch, launches, raw and corrected KiB per launch.  Corrections as
MI355X_MICROARCH.md's HBM section prescribes and as profiles/r04_v1_pmc_calibration.txt measured them on this pool (FETCH_SIZE x 2.0, WRITE_SIZE x 1.0).

    python tools/prof/pmc_kernels.py <dir with fetch/ and write/ passes> <what was run> <source file whose hash stamps the profile> ...
"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_launches import launches, FETCH_FACTOR, WRITE_FACTOR      # noqa: E402


def main():
    root, what, srcs = sys.argv[1], sys.argv[2], sys.argv[3:]
    here = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    h = hashlib.sha256()
    for n in srcs:
        h.update(open(os.path.join(here, "x265_amd", "csrc", n), "rb").read())
    print("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) on `%s`" % what)
    print("# sources %s %s" % (" ".join(srcs), h.hexdigest()[:16]))
    print("# fetch_correction %.3f write_correction %.3f (profiles/r04_v1_pmc_calibration.txt); per-launch averages in KiB, raw and corrected" % (FETCH_FACTOR, WRITE_FACTOR))
    f, w = launches(os.path.join(root, "fetch"), "FETCH_SIZE"), launches(os.path.join(root, "write"), "WRITE_SIZE")
    print("%-44s %9s %12s %12s %12s %12s" % ("kernel", "launches", "FETCH_raw", "WRITE_raw", "FETCH_KiB", "WRITE_KiB"))
    for k in sorted(f, key=lambda k: -sum(f[k])):
        fr = sum(f[k]) / len(f[k])
        wr = sum(w[k]) / len(w[k]) if w.get(k) else 0.0
        print("%-44s %9d %12.1f %12.1f %12.1f %12.1f" % ("xh::" + k[:40], len(f[k]), fr, wr, fr * FETCH_FACTOR, wr * WRITE_FACTOR))


if __name__ == "__main__":
    main()
