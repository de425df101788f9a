#!/usr/bin/env python3
"""Reads an X265HIP_DEBUG_LA_TIMELINE file (x265_amd/host/x265_hip_lookahead.cpp) and prints where the lookahead's wall time goes: inside the two
seams (device batches, cached reads) and in the gaps between consecutive seam calls (the reference's own host work between them), grouped by the
size of the gap."""
import sys

rows = []
for line in open(sys.argv[1]):
    p = line.split()
    if len(p) < 9:
        continue
    rows.append((int(p[0]), int(p[1]), p[2], p[3], int(p[4]), int(p[5]), int(p[6]), int(p[7]), int(p[8])))
rows.sort()
# top-level calls only (estimates inside a batch run on other threads, nested in the batch's interval)
top = []
end = -1
for r in rows:
    if r[0] >= end:
        top.append(r)
        end = r[1]
total = top[-1][1] - top[0][0]
inside = sum(r[1] - r[0] for r in top)
gaps = [(top[i + 1][0] - top[i][1], top[i], top[i + 1]) for i in range(len(top) - 1)]
print("span %.3f s, %d top-level seam calls: inside the seams %.3f s, between them %.3f s" % (total * 1e-9, len(top), inside * 1e-9, sum(g[0] for g in gaps) * 1e-9))
for what in ("pre", "batch", "est", "estB"):
    sel = [r for r in top if r[3] == what]
    if sel:
        c = [r for r in sel if r[7]]
        print("  %-5s %6d calls %.3f s (%d cached: %.3f s)" % (what, len(sel), sum(r[1] - r[0] for r in sel) * 1e-9, len(c), sum(r[1] - r[0] for r in c) * 1e-9))
for lo, hi in ((0, 1e4), (1e4, 1e5), (1e5, 1e6), (1e6, 5e6), (5e6, 2e7), (2e7, 1e12)):
    sel = [g for g in gaps if lo <= g[0] < hi]
    print("  gaps %8.0f..%-10.0f us: %6d, %.3f s" % (lo / 1e3, hi / 1e3, len(sel), sum(g[0] for g in sel) * 1e-9))
big = sorted(gaps, key=lambda g: -g[0])[:25]
print("largest gaps (ms): after -> before")
for g in big:
    print("  %7.2f  after %s(%d,%d,f%d)%s  before %s(%d,%d,f%d)" % (g[0] / 1e6, g[1][3], g[1][4], g[1][5], g[1][6], " cached" if g[1][7] else "", g[2][3], g[2][4], g[2][5], g[2][6]))
