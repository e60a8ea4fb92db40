"""Cost of one mass-matrix refresh of the SNUHumanoid kernels WITHOUT stamps: the env-step launches with MM_caching_frequency 8
(6 refreshes per step) against 48 (1 refresh), difference / 5.  (The stamped builds of tools/stamps.py allocate registers
differently -- 229 instead of 165 VGPRs for the forward kernel -- and show the Gauss-Jordan phase at 7.2 k or 11.4 k cycles
depending on the build; this is the figure of the shipped kernels.)   python tools/snu_refresh_probe.py"""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
res = {}
for mm in (8, 48):
    bench.MM_FREQ["snu"] = mm
    env = bench.make_env("snu", 512, "cuda:0")
    best = [1e9, 1e9]
    for _ in range(3):
        rf = bench.roofline_record(env, "snu", 512, 32, mm, dev, 20, counters=False)
        best = [min(best[0], rf["fwd_kernel_ms"]), min(best[1], rf["kernel_ms"])]
    res[mm] = best
    print("snu 512 mm_freq %d: fwd %.4f ms adj %.4f ms" % (mm, best[0], best[1]))
    del env
d = [(res[8][i] - res[48][i]) / 5 for i in range(2)]
print("per refresh: forward %.5f ms = %.0f cycles at 2.4 GHz; adjoint %.5f ms = %.0f cycles" % (d[0], d[0] * 2.4e6, d[1], d[1] * 2.4e6))
