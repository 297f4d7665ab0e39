"""Static instruction counts of one kernel by source line, from an ISA listing compiled with -gline-tables-only -S.
usage: tools/isa_lines.py <file.s> <kernel-name-substring> [top-n]"""
import collections
import re
import sys

path, want = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 50
lines = open(path).read().split("\n")
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
start = end = None
for i, l in enumerate(lines):
    if start is None and want in l and l.rstrip().split(" ;")[0].endswith(":") and not l.startswith("\t"):
        start = i
    elif start is not None and l.startswith(".Lfunc_end"):
        end = i
        break
cur = None
valu = collections.Counter()
salu = collections.Counter()
other = collections.Counter()
for l in lines[start:end]:
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    t = l.strip()
    if not t or t.startswith(".") or t.startswith(";") or t.split(" ;")[0].rstrip().endswith(":"):
        continue
    op = t.split()[0]
    (valu if op.startswith("v_") else salu if op.startswith("s_") else other)[cur] += 1
print("kernel lines %d..%d: VALU %d, scalar %d, other %d" % (start, end, sum(valu.values()), sum(salu.values()), sum(other.values())))
byfile = collections.Counter()
for (f, ln), c in valu.items():
    byfile[f] += c
print("VALU by file:", dict(byfile))
for (f, ln), c in sorted(valu.items(), key=lambda x: -x[1])[:top]:
    print("%-26s %5d  valu %4d  scalar %4d" % (f, ln, c, salu.get((f, ln), 0)))
