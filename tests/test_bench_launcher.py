"""bench.py as its own launcher: `python bench.py --gpus N` run directly must produce N ranks
(one process per GPU through torch.distributed.run, rendezvous on 127.0.0.1) or fail loudly -- never report a
smaller job under the requested label.  CPU: the launcher path with `--dry-run` (gloo, stops before the GPU engine).
GPU: the real benchmark through the launcher with an RCCL group of one rank."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, timeout=600, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)


def _json_line(out):
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line expected, got: %r / stderr %r" % (out.stdout[-500:], out.stderr[-500:])
    return json.loads(lines[0])


def test_dry_run_launches_two_ranks_on_cpu():
    out = _run(["--gpus", "2", "--dry-run", "--envs-per-gpu", "1024"], env={"CUDA_VISIBLE_DEVICES": "", "HIP_VISIBLE_DEVICES": ""})
    assert out.returncode == 0, out.stderr[-800:]
    j = _json_line(out)
    assert j["dry_run"] is True and j["n_gpus"] == 2 and j["rccl_ranks"] == 2 and j["backend"] == "gloo"
    assert j["envs_total"] == 2048 and j["scaling"] == "weak"          # every env owned by exactly one rank
    assert j["slowest_rank_s"] >= 0.02                                     # MAX over ranks (rank 1 sleeps longer)
    # every rank's own figures (straggler diagnosis): two ranks, each with its shard, the slower one is rank 1
    pr = j["per_rank"]
    assert [r["rank"] for r in pr] == [0, 1] and [r["envs"] for r in pr] == [1024, 1024] and all(r["device"] == "cpu" for r in pr)
    assert pr[1]["elapsed_s"] > pr[0]["elapsed_s"] + 5e-3 and max(r["elapsed_s"] for r in pr) <= j["slowest_rank_s"]


def test_dry_run_oversubscribed_keeps_the_gpu_label():
    """--oversubscribe R: R ranks per GPU over gloo; the line keeps n_gpus = --gpus and says oversubscribed"""
    out = _run(["--gpus", "1", "--oversubscribe", "2", "--dry-run", "--envs-per-gpu", "512"],
               env={"CUDA_VISIBLE_DEVICES": "", "HIP_VISIBLE_DEVICES": ""})
    assert out.returncode == 0, out.stderr[-800:]
    j = _json_line(out)
    assert j["n_gpus"] == 1 and j["oversubscribed"] is True and j["rccl_ranks"] == 2 and j["backend"] == "gloo"
    assert j["envs_total"] == 1024


def test_more_gpus_than_visible_fails_loudly():
    """a `--gpus 2` invocation on a box with fewer GPUs must not print `n_gpus: 1`"""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    out = _run(["--gpus", str(max(have + 1, 2)), "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert out.returncode != 0
    assert "GPU(s) are visible" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_world_size_must_match_label():
    out = _run(["--gpus", "2", "--dry-run"], env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1",
                                                  "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533"})
    assert out.returncode != 0 and "label must match" in out.stderr


@pytest.mark.gpu
def test_bench_through_the_launcher_rccl_group_of_one():
    out = _run(["--gpus", "1", "--launcher", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--strict"], timeout=900)
    assert out.returncode == 0, out.stderr[-1500:]
    j = _json_line(out)
    assert j["n_gpus"] == 1 and j["rccl_ranks"] == 1 and j["steps"] == 2
    # the other BASELINE.json configurations ride along in the default single-GPU line
    def key(o):
        k = o["workload"].split()[0]
        return (k + ("_mm1" if o["workload"].endswith("frequency 1") else "") + ("_generic" if o["kernels"].startswith("generic") else "") +
                ("_8192" if " 8192 envs" in o["workload"] else ""))
    oc = {key(o): o for o in j["other_configs"]}
    assert set(oc) == {"humanoid", "snu", "ant_mm1", "ant_generic", "ant_8192"}, oc
    for k, o in oc.items():   # each measured like the headline (>= 10 timed replays) and with its own roofline object
        assert o["value"] and o["value"] > 1e5 and o["steps"] >= (5 if k == "ant_8192" else 10), o
        r = o["roofline"]
        assert r["bound"] == "valu-issue" and r["traffic"] > r["alg_bytes_per_launch"] and 0 < r["hbm_measured_frac"] < 1, r
    assert oc["ant_generic"]["value"] < j["value"] < oc["ant_8192"]["value"]
    assert j["config"]["submission_fallback"] is False
    assert j["value"] > 1e5 and j["roofline"]["traffic"] > j["roofline"]["alg_bytes_per_launch"]
    assert j["roofline"]["bound"] == "valu-issue" and "valu_issue_frac" in j["roofline"] and "hbm_measured_frac" in j["roofline"]
    # round 5: the group really is RCCL, per-rank figures are in the line, the roofline object names the LONGER launch
    assert j["backend"] == "nccl" and len(j["per_rank"]) == 1
    r0 = j["per_rank"][0]
    assert r0["rank"] == 0 and "cuda:0" in r0["device"] and j["value"] <= r0["value"] < 1.05 * j["value"]   # (own time: before the closing barrier)
    assert 0.95 * j["ms_per_step"] < r0["ms_per_step"] <= j["ms_per_step"] and r0["ms_per_step_max"] >= r0["ms_per_step_median"] > 0
    rf = j["roofline"]
    longer = "dsim_env_fwd_kernel" if rf["fwd_kernel_ms"] > rf["kernel_ms"] else "dsim_env_bwd_kernel"
    assert rf["kernel"] == longer and rf["dominant_launch"] == ("forward" if longer.endswith("fwd_kernel") else "adjoint")
    assert abs(rf["alg_frac_step"] - 748 * j["value"] / 8e12) < 1e-9 and 0 < rf["alg_frac_step"] < 0.01
    if rf.get("counters"):   # counter file at these kernel sources: the flop view comes from the opcode census, not "every VALU op is an FMA"
        assert rf["fp32_valu_frac_est"] is None or rf["fp32_valu_frac_est"] < rf["fp32_valu_frac_upper_bound"]


@pytest.mark.gpu
def test_dry_run_through_the_launcher_initialises_rccl():
    """what the driver's multi-GPU launch exercises before any kernel runs: torch.distributed.run -> ranks -> RCCL process group
    (backend nccl with a device per rank) -> barrier -> timing all-reduce / all_gather -> one JSON line, here with one rank"""
    out = _run(["--gpus", "1", "--dry-run", "--envs-per-gpu", "1024"], timeout=600)
    assert out.returncode == 0, out.stderr[-1500:]
    j = _json_line(out)
    assert j["dry_run"] is True and j["backend"] == "nccl" and j["rccl_ranks"] == 1 and j["n_gpus"] == 1
    assert j["per_rank"][0]["envs"] == 1024 and j["per_rank"][0]["device"] != "cpu"


@pytest.mark.gpu
def test_bench_two_ranks_oversubscribed_on_one_device():
    """SURVEY 8(e) on a 1-GPU lease: the launcher -> two ranks -> rank-local Engine + shard -> timing all-reduce (MAX over ranks)
    path executes with world_size 2 against the real kernels.  Both ranks share cuda:0 (gloo; RCCL refuses two ranks per device),
    so this is a functional run, never a scaling figure: the line says so and keeps n_gpus = 1."""
    out = _run(["--gpus", "1", "--oversubscribe", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-other-configs",
                "--no-extras", "--strict"], timeout=900)
    assert out.returncode == 0, out.stderr[-1500:]
    j = _json_line(out)
    assert j["n_gpus"] == 1 and j["oversubscribed"] is True and j["rccl_ranks"] == 2 and j["backend"] == "gloo"
    assert j["config"]["envs_total"] == 2048 and j["config"]["envs_per_gpu"] == 1024
    assert j["value"] > 1e5 and j["config"]["submission_fallback"] is False
    assert "NOT a multi-GPU figure" in j["note"]
