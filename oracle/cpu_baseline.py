"""TEST/BENCH INFRASTRUCTURE: times the scalar CPU oracle (forward + taped adjoint) on a bounded sample of the
bench workload, on `workers` host cores (one process each).  Called by bench.py's `cpu_baseline` leg as a
subprocess so that no GPU runtime state is forked.  Prints one JSON object."""
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _work(args):
    name, seconds = args
    from oracle_lib import golden, oracle_backward, template_from_golden
    t = template_from_golden(name)
    g = golden(name + "_step")
    S, mm, dt = int(g["substeps"]), int(g["mm_freq"]), float(g["dt"])
    a = (g["q_in"], g["qd_in"], g["act_in"], g.get("muscle_act_in"), dt, S, mm, g["gq_out"], g["gqd_out"])
    oracle_backward(t, *a)
    n, done, t0 = g["q_in"].shape[0], 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        oracle_backward(t, *a)
        done += n
    return done, time.perf_counter() - t0


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "ant"
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
    workers = int(sys.argv[3]) if len(sys.argv) > 3 else min(os.cpu_count() or 1, 32)
    from oracle_lib import oracle
    oracle()  # build once before the workers race for it
    with mp.get_context("spawn").Pool(workers) as pool:
        res = pool.map(_work, [(name, seconds)] * workers)
    steps = sum(r[0] for r in res)
    el = max(r[1] for r in res)
    print(json.dumps({"value": steps / el, "unit": "env-steps/s", "cores": workers, "kind": "port",
                      "sample": "%d %s env-steps (fwd + taped adjoint) in %.1f s on %d of %d host cores, one process per core"
                                % (steps, name, el, workers, os.cpu_count())}))


if __name__ == "__main__":
    main()
