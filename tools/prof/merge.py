#!/usr/bin/env python3
"""Merge several tools/prof/resolve.py reports (one per encoder run: the address maps differ between runs, the function names do not) into one,
weighted by each run's sample count.   usage: merge.py report1.txt report2.txt ... [--top N]"""
import collections
import re
import sys

top = 60
files = [a for a in sys.argv[1:] if not a.startswith("--")]
if "--top" in sys.argv:
    top = int(sys.argv[sys.argv.index("--top") + 1])
    files = [f for f in files if f != str(top)]
cats, funcs, total = collections.Counter(), collections.Counter(), 0
for f in files:
    lines = open(f).read().splitlines()
    n = int(re.match(r"(\d+) samples", lines[0]).group(1))
    total += n
    blank = lines.index("")
    for l in lines[1:blank]:
        m = re.match(r"\s*([\d.]+) %  (.*)", l)
        if m:
            cats[m.group(2)] += float(m.group(1)) * n / 100.0
    for l in lines[blank + 1:]:
        m = re.match(r"\s*([\d.]+) %  (.*)", l)
        if m:
            funcs[m.group(2)] += float(m.group(1)) * n / 100.0
print("%d samples (%d runs; functions below each run's own top list are missing from their sums)" % (total, len(files)))
for label, c in cats.most_common():
    print("  %5.1f %%  %s" % (100.0 * c / total, label))
print()
for name, c in funcs.most_common(top):
    print("  %5.2f %%  %s" % (100.0 * c / total, name))
