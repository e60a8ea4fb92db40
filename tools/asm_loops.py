"""Developer tool: instruction census of the loops of one kernel in a hipcc -S listing.
usage: asm_loops.py <listing.s> <kernel-name-substring> [min_span]"""
import collections
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
min_span = int(sys.argv[3]) if len(sys.argv) > 3 else 300
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and pat in l)
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels, insts = {}, []
for l in body:
    m = re.match(r"^(\.LBB\w+):", l)
    if m:
        labels[m.group(1)] = len(insts)
        continue
    m = re.match(r"^\s+([a-z][a-z0-9_]+)\s*(.*)$", l)
    if m and not l.strip().startswith("."):
        insts.append((m.group(1), m.group(2)))


def cat(op):
    if op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"):
        return "lane"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("scratch_") or op.startswith("flat_"):
        return "vmem"
    if op == "s_waitcnt":
        return "waitcnt"
    if op.startswith("s_cbranch") or op == "s_branch":
        return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer"):
        return "smem"
    return "salu"


print("kernel insts:", len(insts), dict(collections.Counter(cat(o) for o, _ in insts)))
loops = []
for i, (op, args) in enumerate(insts):
    if op.startswith("s_cbranch") or op == "s_branch":
        t = args.split()[-1]
        if t in labels and labels[t] <= i and i - labels[t] >= min_span:
            loops.append((labels[t], i))
for a, b in sorted(set(loops)):
    c = collections.Counter(cat(o) for o, _ in insts[a:b + 1])
    ops = collections.Counter(re.sub(r"_e32|_e64|_dpp|_sdwa", "", o) for o, _ in insts[a:b + 1])
    print("loop insts %d..%d (%d):" % (a, b, b - a + 1), dict(c))
    print("   ", " ".join("%s:%d" % kv for kv in ops.most_common(28)))
