#!/usr/bin/env python3
"""Who calls what: the call chains of the samples whose innermost function matches a pattern (tools/prof/cpusample.c with X265HIP_CPUSAMPLE_STACK=1).
   usage: callers.py dump.bin 'ioctl|munmap' [top]"""
import bisect
import collections
import re
import struct
import subprocess
import sys

data = open(sys.argv[1], "rb").read()
pat = re.compile(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 12
i = data.index(b"PCS\n")
maps = data[:i].decode(errors="replace").splitlines()
n = (len(data) - i - 4) // 8
pcs = struct.unpack("<%dQ" % n, data[i + 4:i + 4 + n * 8])
st = open(sys.argv[1] + ".stacks", "rb").read()
DEPTH = 24
ns = len(st) // (8 * DEPTH)
stacks = struct.unpack("<%dQ" % (ns * DEPTH), st[:ns * DEPTH * 8])
regions = []
for l in maps:
    p = l.split()
    if len(p) >= 6 and "x" in p[1]:
        a, b = [int(x, 16) for x in p[0].split("-")]
        regions.append((a, b, int(p[2], 16), p[5]))
regions.sort()
syms = {}


def table(path):
    if path not in syms:
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
        syms[path] = ([a for a, _ in t], [x for _, x in t])
    return syms[path]


pie = {}


def name_of(pc):
    k = bisect.bisect_right(regions, (pc, 1 << 62, 0, "")) - 1
    if k < 0 or not (regions[k][0] <= pc < regions[k][1]):
        return "?"
    a, b, off, path = regions[k]
    if not path.startswith("/"):
        return path or "[anon]"
    if path not in pie:
        try:
            pie[path] = b"DYN" in subprocess.run(["readelf", "-h", path], capture_output=True).stdout
        except Exception:
            pie[path] = True
    addrs, names = table(path)
    va = pc - a + off if pie[path] else pc
    j = bisect.bisect_right(addrs, va) - 1
    nm = names[j] if j >= 0 else "?"
    return "%s[%s]" % (re.sub(r"\(.*", "", nm.replace("(anonymous namespace)", "anon"))[:60], path.rsplit("/", 1)[-1][:24])


chains = collections.Counter()
hits = 0
for k in range(min(n, ns)):
    if not pat.search(name_of(pcs[k])):
        continue
    hits += 1
    fr = [name_of(x) for x in stacks[k * DEPTH:(k + 1) * DEPTH] if x]
    chains[" <- ".join(fr[:8])] += 1
print("%d of %d samples match %s" % (hits, min(n, ns), sys.argv[2]))
# by the first frame that does not itself match: the caller the matched leaf works for
first = collections.Counter()
for c, v in chains.items():
    fr = c.split(" <- ")
    who = next((f for f in fr if not pat.search(f)), "?")
    first[who] += v
print("by first caller outside the pattern:")
for c, v in first.most_common(20):
    print("%5d  %5.1f %%  %s" % (v, 100.0 * v / max(1, hits), c))
# by the first frame outside the runtime's own libraries (libhsa, libamdhip, libc, the thunk): whose call the matched leaf works for; "(runtime thread)" when the
# whole stack is the runtime's (its event / signal threads)
rt = re.compile(r"libhsa|libamdhip|libc\.so|libhsakmt|libdrm|\[anon\]|\?$|^\?")
owner = collections.Counter()
for k in range(min(n, ns)):
    if not pat.search(name_of(pcs[k])):
        continue
    fr = [name_of(x) for x in stacks[k * DEPTH:(k + 1) * DEPTH] if x]
    who = next((f for f in fr if not rt.search(f)), "(runtime thread)")
    owner[name_of(pcs[k]).split("[")[0][:40] + "  <=  " + who] += 1
print("leaf  <=  first frame outside the runtime libraries:")
for c, v in owner.most_common(25):
    print("%5d  %5.1f %%  %s" % (v, 100.0 * v / max(1, hits), c))
print("call chains:")
for c, v in chains.most_common(top):
    print("%5d  %s" % (v, c))
