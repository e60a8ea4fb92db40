"""Developer tool (GPU box): where the time of ONE graph replay of the bench rollout goes -- kernel by kernel, with the gaps between
consecutive dispatches -- from a rocprofv3 --kernel-trace of `bench.py --steps K`.
usage: python tools/replay_timeline.py <dir with *kernel_trace.csv> [launches per replay to print]"""
import csv
import glob
import os
import sys

rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# rollouts: a forward launch whose previous step kernel was an adjoint launch (or none) starts one; it ends with its 32nd adjoint
# launch.  The LAST complete rollout that is followed by another rollout is a timed graph replay (behind them come the 50 + 50
# launches of the kernel timing).
H = 32
step = [i for i, n in enumerate(names) if "dsim_env_fwd_kernel" in n or "dsim_env_bwd_kernel" in n]
starts = [i for k, i in enumerate(step) if "fwd" in names[i] and (k == 0 or "bwd" in names[step[k - 1]])]
rolls = []
for s0 in starts:
    ks = [i for i in step if i >= s0][:2 * H]
    if len(ks) == 2 * H and all("fwd" in names[i] for i in ks[:H]) and all("bwd" in names[i] for i in ks[H:]):
        rolls.append((s0, ks[-1]))
assert len(rolls) >= 2, "no complete rollouts in the trace"
pick = len(rolls) - 2   # (the last one may be followed by the kernel timing's launches; the one before it is mid-sequence)
# one PERIOD of the replay sequence: from this replay's first forward launch to the dispatch in front of the next replay's first
# forward launch (the torch kernels between two replays -- loss gradient tail, state copies of the next replay -- counted once)
lo, hi = rolls[pick][0], rolls[pick + 1][0] - 1
period_us = (int(rows[rolls[pick + 1][0]]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e3
print("period between the first forward launches of two consecutive replays: %.1f us" % period_us)
seg = rows[lo:hi + 1]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
dsim = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg if "dsim_env_" in r["Kernel_Name"])
print("one replay: %d dispatches, span %.1f us, kernels busy %.1f us (dsim step kernels %.1f us, others %.1f us), gaps %.1f us"
      % (len(seg), (t1 - t0) / 1e3, busy / 1e3, dsim / 1e3, (busy - dsim) / 1e3, (t1 - t0 - busy) / 1e3))
prev = None
agg = {}
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    nm = r["Kernel_Name"][:60]
    a = agg.setdefault(nm, [0, 0.0, 0.0])
    a[0] += 1; a[1] += (e - s) / 1e3; a[2] += gap
    prev = e
for nm, (c, d, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%4d x %-60s busy %8.1f us   gap in front %7.1f us" % (c, nm, d, g))
if len(sys.argv) > 2:
    prev = None
    for r in seg[:int(sys.argv[2])] + seg[-int(sys.argv[2]):]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%-50s gap %6.2f us  dur %7.2f us" % (r["Kernel_Name"][:50], (s - prev) / 1e3 if prev else 0.0, (e - s) / 1e3))
        prev = e
