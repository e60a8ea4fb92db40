"""Developer tool: instruction counts between phase boundaries of one kernel (listing built with
-DDSIM_WAVE_SYNC_ASM='"; dsim_sync"', so that every phase boundary leaves a comment in the assembly).
usage: asm_phases.py <listing.s> <kernel-substring>"""
import collections
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and pat in l)
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
seg = collections.Counter()
n = 0
print("seg  total  valu  lds  salu branch wait vmem   (label lines seen)")
labels = []
for l in lines[start:end]:
    if "dsim_sync" in l:
        print("%3d  %5d %5d %4d %5d %5d %4d %4d   %s" % (n, sum(seg.values()), seg["valu"], seg["lds"], seg["salu"], seg["branch"], seg["wait"], seg["vmem"], " ".join(labels[:6])))
        seg = collections.Counter()
        labels = []
        n += 1
        continue
    m = re.match(r"^(\.LBB\w+):(.*)", l)
    if m:
        labels.append(m.group(1) + ("*" if "Loop Header" in m.group(2) else ""))
        continue
    m = re.match(r"^\s+([a-z][a-z0-9_]+)\s", l)
    if m and not l.strip().startswith("."):
        op = m.group(1)
        k = ("valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else "wait" if op == "s_waitcnt" else
             "branch" if op.startswith("s_cbranch") or op == "s_branch" else "salu" if op.startswith("s_") else "vmem")
        seg[k] += 1
print("end  %5d %5d %4d %5d %5d %4d %4d" % (sum(seg.values()), seg["valu"], seg["lds"], seg["salu"], seg["branch"], seg["wait"], seg["vmem"]))
