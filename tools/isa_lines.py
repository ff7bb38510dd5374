#!/usr/bin/env python3
"""Static instruction counts per source line of one kernel, from an ISA listing made with
-gline-tables-only (.loc directives).  usage: isa_lines.py file.s <mangled-name-prefix> [top N]
Prints: instructions by class for the kernel, then the source lines with the most instructions
(file:line  total  valu salu vmem lds other)."""
import collections
import re
import sys

path, prefix = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
files = {}
inside = False
cur = ("?", 0)
per_line = collections.defaultdict(lambda: collections.Counter())
classes = collections.Counter()
ops = collections.Counter()


def klass(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("s_load", "s_buffer", "s_store")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    return "other"


for line in open(path):
    m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', line)
    if m:
        files[int(m.group(1))] = m.group(2)
        continue
    if not inside:
        if line.startswith(prefix) and line.rstrip().split(":")[0].startswith(prefix) and ":" in line:
            inside = True
        continue
    if line.startswith("\t.end_amdhsa_kernel") or "; -- End function" in line:
        break
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    s = line.strip()
    if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
        continue
    op = s.split()[0]
    c = klass(op)
    classes[c] += 1
    ops[op] += 1
    per_line[cur][c] += 1
    per_line[cur]["total"] += 1

print("kernel", prefix, dict(classes), "total", sum(classes.values()))
print("top opcodes:", ops.most_common(40))
rows = sorted(per_line.items(), key=lambda kv: -kv[1]["total"])[:top]
for (f, ln), c in rows:
    print("%-22s %5d  total %5d  valu %5d salu %5d vmem %4d lds %4d" % (f, ln, c["total"], c["valu"], c["salu"], c["vmem"], c["lds"]))
