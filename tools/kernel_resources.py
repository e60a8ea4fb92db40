"""Developer tool: VGPR / SGPR / spill / occupancy figures of every kernel variant, from the compiler's
-Rpass-analysis=kernel-resource-usage remarks (no GPU needed)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "diffrl_amd", "csrc", "dsim_hip.hip")
out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-Wno-unused-value",
                      "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur, rows = None, []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = m.group(1)
        k = re.search(r"(dsim_[a-z_]+kernel)I\d+DsimOff(\w+?)\d+DsimDims", name) or re.search(r"(dsim_[a-z_]+kernel)I(7)DsimOff", name)
        cur = {"name": (k.group(1) + "<" + (k.group(2) if k.group(2) != "7" else "generic") + ">") if k else name[:60]}
        rows.append(cur)
        continue
    m = re.search(r"remark: [^:]+:\d+:\d+:\s+([A-Za-z ]+[A-Za-z])\s*(?:\[bytes/lane\])?: (\d+)", line) or re.search(r"\s+([A-Za-z ]+[A-Za-z])(?: \[[^\]]+\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
pat = sys.argv[1] if len(sys.argv) > 1 else ""
print("%-34s %5s %5s %6s %6s %5s %7s" % ("kernel", "VGPR", "AGPR", "SGPR", "spillV", "occ", "LDS"))
for r in rows:
    if pat in r["name"]:
        print("%-34s %5s %5s %6s %6s %5s %7s" % (r["name"], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"),
                                                 r.get("VGPRs Spill", r.get("ScratchSize")), r.get("Occupancy"),
                                                 r.get("LDS Size")))
