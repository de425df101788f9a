#!/usr/bin/env python3
"""Resolve a tools/prof/cpusample.c dump: samples per function and per category.   usage: resolve.py dump.bin [top]"""
import bisect
import collections
import re
import struct
import subprocess
import sys

data = open(sys.argv[1], "rb").read()
top = int(sys.argv[2]) if len(sys.argv) > 2 else 50
i = data.index(b"PCS\n")
maps = data[:i].decode(errors="replace").splitlines()
pcs = struct.unpack("<%dQ" % ((len(data) - i - 4) // 8), data[i + 4:i + 4 + ((len(data) - i - 4) // 8) * 8])
regions = []
for l in maps:
    p = l.split()
    if len(p) >= 6 and "x" in p[1]:
        a, b = [int(x, 16) for x in p[0].split("-")]
        regions.append((a, b, int(p[2], 16), p[5]))
regions.sort()
syms = {}


def table(path):
    if path in syms:
        return syms[path]
    t = []
    for flag in ("", "-D"):
        try:
            out = subprocess.run(["nm", "-C", "--defined-only"] + ([flag] if flag else []) + [path], capture_output=True, text=True).stdout
        except Exception:
            out = ""
        for line in out.splitlines():
            q = line.split(None, 2)
            if len(q) == 3 and q[1] in "TtWw":
                t.append((int(q[0], 16), q[2]))
    t.sort()
    syms[path] = ([a for a, _ in t], [n for _, n in t])
    return syms[path]


def is_pie(path):
    try:
        return b"DYN" in subprocess.run(["readelf", "-h", path], capture_output=True).stdout
    except Exception:
        return True


count = collections.Counter()
for pc in pcs:
    k = bisect.bisect_right(regions, (pc, 1 << 62, 0, "")) - 1
    if k < 0 or not (regions[k][0] <= pc < regions[k][1]):
        count["[unmapped]"] += 1
        continue
    a, b, off, path = regions[k]
    if not path.startswith("/"):
        count[path or "[anon]"] += 1
        continue
    addrs, names = table(path)
    va = pc - a + off if is_pie(path) else pc
    j = bisect.bisect_right(addrs, va) - 1
    name = names[j] if j >= 0 else "?"
    count["%s  [%s]" % (re.sub(r"\(.*", "", name.replace("(anonymous namespace)", "anon"))[:110], path.rsplit("/", 1)[-1])] += 1

n = len(pcs)
CATS = [
    ("integer-pel SAD (C slots)", r"\bsad<|sad_x[34]<|\bsad_x"),
    ("SAD lookups (binding)", r"sad_lookup|sad_x[34]_lookup|locate"),
    ("satd", r"satd"),
    ("sa8d", r"sa8d"),
    ("transforms", r"dct|dst|partialButterfly|inversedst|fastForwardDst|idct"),
    ("quant / dequant / rdoq", r"quant|Quant"),
    ("coefficient coding / entropy", r"Entropy|codeCoeff|costCoeff|scanPosLast|findPosFirstLast|bitsIntraMode|estBit|Cabac|encodeBin|costC1C2|coeffRemain"),
    ("interpolation (C)", r"interp_|filterPixelToShort|interp<"),
    ("plane lookups (binding)", r"lookup|serve|hpp_from|from_plane|plane_copy"),
    ("copies / add / sub / blockfill", r"blockcopy|copy_|cpy|sub_ps|add_ps|calcresidual|blockfill|getResidual|pixel_add|pixeladd|addAvg|pixelavg|memcpy|memmove|memset|copyPart|copyFrom|copyTo"),
    ("sse / ssd", r"sse<|ssd_s|sse_"),
    ("psy", r"psy|Psy"),
    ("intra prediction", r"intra_pred|planar_pred|dc_pred|all_angs|intraFilter|IntraNeighbors|initAdiPattern|intra_ang"),
    ("SAO", r"sao|SAO|Sao"),
    ("deblock", r"deblock|Deblock|pelFilter"),
    ("motion search control", r"MotionEstimate|StarPatternSearch|subpelCompare|mvcost|bitcost|BitCost"),
    ("analysis / search control", r"Analysis|Search::|Predict|CUData|TComDataCU|Yuv::|ShortYuv"),
    ("lookahead / slicetype / ratecontrol", r"Lookahead|CostEstimateGroup|LookaheadTLD|slicetype|RateControl|Lowres|lowres|frame_init_lowres|cuTree|estimateCU"),
    ("thread pool / sync", r"pthread|futex|ThreadPool|WorkerThread|JobProvider|BondedTaskGroup|Event|WaveFront|__lll|sched_yield|syscall"),
    ("HIP runtime / driver libs", r"\[libamdhip64|\[libhsa|\[libx265hip|\[libdrm|\[librocprofiler"),
]
cat = collections.Counter()
for name, c in count.items():
    for label, rx in CATS:
        if re.search(rx, name):
            cat[label] += c
            break
    else:
        cat["other"] += c
print("%d samples" % n)
for label, c in cat.most_common():
    print("  %5.1f %%  %s" % (100.0 * c / n, label))
print()
for name, c in count.most_common(top):
    print("  %5.2f %%  %s" % (100.0 * c / n, name))
