"""Developer tool: which SOURCE LINES produce a given opcode in one kernel (listing built with -gline-tables-only:
tools/dev_asm.sh out.s Ant -gline-tables-only).  usage: asm_opcode_lines.py <listing.s> <kernel-substring> [opcode-prefix] [top]"""
import collections
import os
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
op = sys.argv[3] if len(sys.argv) > 3 else "v_cndmask"
top = int(sys.argv[4]) if len(sys.argv) > 4 else 25
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lines = open(path).read().split("\n")
files = {}
for l in lines:
    m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
src = {f: open(os.path.join(root, "diffrl_amd", "csrc", f)).read().split("\n") for f in ("dsim_core.hpp", "dsim_hip.hip", "dsim_math.hpp")}
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and pat in l)
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
cur, cnt = ("?", 0), collections.Counter()
for l in lines[start:end]:
    m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
        continue
    if re.match(r"^\s+" + re.escape(op), l):
        cnt[cur] += 1
print(op, "total", sum(cnt.values()))
for (f, ln), n in cnt.most_common(top):
    t = src[f][ln - 1].strip()[:130] if f in src and 0 < ln <= len(src[f]) else ""
    print("%4d %s:%d | %s" % (n, f, ln, t))
