"""Developer tool: static instruction census per SOURCE LINE of one kernel, from a `hipcc -S -gline-tables-only` listing
(tools/dev_asm.sh out.s Ant -gline-tables-only).  usage: asm_lines.py <listing.s> <kernel-substring> [top]"""
import collections
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
lines = open(path).read().split("\n")
files = {}
for l in lines:
    m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and pat in l)
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
cur = ("?", 0)
per = collections.Counter()
kinds = collections.defaultdict(collections.Counter)
for l in lines[start:end]:
    m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    m = re.match(r"^\s+([a-z][a-z0-9_]+)\s", l)
    if m and not l.strip().startswith("."):
        op = m.group(1)
        per[cur] += 1
        k = "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else "wait" if op == "s_waitcnt" else "salu" if op.startswith("s_") else "vmem"
        kinds[cur][k] += 1
tot = sum(per.values())
print("instructions:", tot)
src = {}
for (f, ln), n in per.most_common(top):
    if f not in src:
        try:
            import glob
            cand = glob.glob("diffrl_amd/csrc/" + f) + glob.glob("/root/repo/diffrl_amd/csrc/" + f)
            src[f] = open(cand[0]).read().split("\n") if cand else []
        except Exception:
            src[f] = []
    text = src[f][ln - 1].strip()[:110] if 0 < ln <= len(src[f]) else ""
    print("%5d %4.1f%%  %s:%d  %s | %s" % (n, 100.0 * n / tot, f, ln, dict(kinds[(f, ln)]), text))
