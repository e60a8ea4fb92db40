"""Developer tool: turns gpurun_out/prof_<tag>_<env>/ (written on the GPU box by tools/profile.sh) into the tracked evidence
files profiles/<name>_<env>_rocprofv3_summary.txt and profiles/<name>_<env>_pmc.json (the HBM traffic of one adjoint launch,
with the hash of the kernel sources it was measured at: bench.py only quotes it for matching sources / env / N).

usage: python tools/make_profile_record.py <tag> <env> <n_envs> <name> ["free text for the header"]"""
import ast
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, env, n, name = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
note = sys.argv[5] if len(sys.argv) > 5 else ""
mm_arg = int(sys.argv[6]) if len(sys.argv) > 6 else 0   # MM_caching_frequency of the profiled run (0: the environment's default)
suffix = ("_n%d" % n if (env == "ant" and n != 1024) else "") + ("_mm%d" % mm_arg if mm_arg else "")
src = os.path.join(ROOT, "gpurun_out", "prof_%s_%s%s" % (tag, env, suffix))
summ = open(os.path.join(src, "summary.txt")).read()
cmd = open(os.path.join(src, "command.txt")).read().strip().replace(os.environ.get("GRAFT_REPO_ROOT", "/nonexistent"), ".")
cmd = re.sub(r"python \S*/bench.py", "python bench.py", cmd)
h = open(os.path.join(src, "csrc_hash.txt")).read().strip()
MM = {"ant": 16, "humanoid": 48, "snu": 8, "cartpole": 4, "hopper": 16, "cheetah": 16}


def counters(section, kernel):
    body = summ[summ.index("== pmc pass %s" % section):]
    body = body[:body.index("== pmc pass", 5)] if "== pmc pass" in body[5:] else body
    for line in body.splitlines():
        if kernel in line:
            return ast.literal_eval(line[line.index("{"):line.rindex("}") + 1])
    return {}


fetch_b = counters("fetch", "dsim_env_bwd_kernel").get("FETCH_SIZE")
write_b = counters("write", "dsim_env_bwd_kernel").get("WRITE_SIZE")
fetch_f = counters("fetch", "dsim_env_fwd_kernel").get("FETCH_SIZE")
write_f = counters("write", "dsim_env_fwd_kernel").get("WRITE_SIZE")
sq = counters("sq ", "dsim_env_bwd_kernel")
sqf = counters("sq ", "dsim_env_fwd_kernel")
for sect in ("sq3", "sq2"):   # wave count and lane-cycles of the VALU; LDS bank conflicts, VMEM instruction counts (own passes)
    try:
        for dst, k in ((sq, "dsim_env_bwd_kernel"), (sqf, "dsim_env_fwd_kernel")):
            for name_, val in counters(sect, k).items():
                if name_ != "SQ_INSTS_VALU":
                    dst[name_] = val
    except ValueError:
        pass
header = [
    "# rocprofv3 summary, %s, MI355X, ROCm 7.2, tools/profile.sh %s %s %d%s" % (name, tag, env, n, ("  -- " + note) if note else ""),
    "# command profiled: %s   (kernel sources: csrc hash %s)" % (cmd, h),
    "# passes: (1) --kernel-trace --stats  (2) --pmc FETCH_SIZE  (3) --pmc WRITE_SIZE  (4,5) --pmc SQ_* / GRBM_*  (each its own run)",
    "# FETCH_SIZE / WRITE_SIZE in KiB per launch; on gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes (MI355X_MICROARCH.md, HBM",
    "# section): bytes read = FETCH_SIZE x 2 x 1024 (checked against the known checkpoint size of the adjoint launch); every",
    "# forward launch of the profiled command writes a checkpoint (bench.py --no-extras: no forward-only leg).",
    "# SQ_*_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* in quad-cycles summed over the wavefronts of a launch.",
]
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
open(os.path.join(ROOT, "profiles", "%s_%s%s_rocprofv3_summary.txt" % (name, env, suffix)), "w").write("\n".join(header) + "\n" + summ)
rec = {"env": env, "n_envs": n, "mm_freq": mm_arg or MM[env], "kernel": "dsim_env_bwd_kernel", "csrc_hash": h,
       "fetch_size_kib_per_launch": fetch_b, "write_size_kib_per_launch": write_b,
       "traffic_bytes_per_launch": int(fetch_b * 2 * 1024 + write_b * 1024) if fetch_b is not None and write_b is not None else None,
       "forward_kernel": {"fetch_size_kib_per_launch": fetch_f, "write_size_kib_per_launch": write_f,
                          "traffic_bytes_per_launch": int(fetch_f * 2 * 1024 + write_f * 1024) if fetch_f is not None and write_f is not None else None},
       "sq_adjoint": sq, "sq_forward": sqf, "source": "profiles/%s_%s%s_rocprofv3_summary.txt" % (name, env, suffix)}
json.dump(rec, open(os.path.join(ROOT, "profiles", "%s_%s%s_pmc.json" % (name, env, suffix)), "w"), indent=1)
print(json.dumps(rec)[:400])
