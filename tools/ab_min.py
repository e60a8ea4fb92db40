"""Developer tool (GPU box): A/B kernel timings that survive box-to-box and run-to-run noise -- the libraries are measured in
alternation (several rounds, each library in its own process), the MINIMUM per library and kernel is reported.
usage: python tools/ab_min.py <env> <N> <rounds> lib1.so lib2.so ..."""
import os
import re
import subprocess
import sys

env, n, rounds, libs = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
best = {l: [1e9, 1e9] for l in libs}
for r in range(rounds):
    for l in libs:
        e = dict(os.environ, DSIM_LIB=os.path.abspath(l))
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "gpu_quick.py"), env, n], env=e, capture_output=True, text=True).stdout
        m = re.search(r"fwd ([\d.]+) ms\s+bwd ([\d.]+) ms", out)
        if m:
            best[l][0] = min(best[l][0], float(m.group(1)))
            best[l][1] = min(best[l][1], float(m.group(2)))
for l in libs:
    f, b = best[l]
    print("%-44s %s N=%s  min of %d: fwd %.4f ms  bwd %.4f ms  -> %.3f M env-steps/s" % (l, env, n, rounds, f, b, int(n) / (f + b) / 1e3))
