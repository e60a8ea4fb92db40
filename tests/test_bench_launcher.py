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
    # the driver reads the LAST stdout line out of a bounded tail (round 5's 21 KB line came back `parsed: null`)
    assert out.stdout.rstrip().splitlines()[-1] == lines[0] and len(lines[0]) < 4096, len(lines[0])
    return json.loads(lines[0])


def _full_record(ranks):
    """a full bench record of the shape main() builds (every per-kernel view, note and sample string populated)"""
    view = {"valu_insts_per_env_step": 32778.32294921875, "lds_insts_per_env_step": 5884.69873046875, "salu_insts_per_env_step": 5542.7,
            "waves_per_env": 2.0, "valu_issue_frac": 0.5020936554990025, "valu_simd_frac": 0.25209494299590024,
            "stall_frac": 0.26903840578677873, "lds_bank_conflict_frac": 0.1199660130845999, "valu_active_lanes_avg": 31.94}
    rf = {"bound": "valu-issue", "kernel": "dsim_env_fwd_kernel", "dominant_launch": "forward", "achieved": 5.6283746, "peak": 8000.0,
          "unit": "GB/s", "frac": 0.000703546, "alg_bytes_per_launch": 294912, "kernel_ms": 0.0523974, "fwd_kernel_ms": 0.0523974,
          "bwd_kernel_ms": 0.0529123, "fwd_alg_bytes_per_launch": 294912, "bwd_alg_bytes_per_launch": 471040,
          "fwd_alg_frac": 0.000703546, "bwd_alg_frac": 0.00111283, "csrc_hash": "3bb1ae48c285", "ckpt_bytes_per_env_step": 30096,
          "ckpt_bytes_per_rollout": 986185728, "valu_issue_frac": 0.5, "fwd_valu_issue_frac": 0.53, "valu_simd_frac": 0.2520949,
          "fwd_valu_simd_frac": 0.2503285, "stall_frac": 0.269, "fwd_stall_frac": 0.307, "valu_insts_per_env_step": 25482.9,
          "counters": "profiles/r05_final_ant_pmc.json", "adjoint": view, "forward": view, "fp32_valu_frac_upper_bound": 0.12,
          "flops_per_valu_inst": {"adjoint": 1.049, "forward": 0.861, "source": "profiles/opcode_census.json"},
          "fp32_valu_frac_est": 0.0581825, "flop_per_env_step": 1791864.9, "bwd_traffic": 32300000, "fwd_traffic": 32700000,
          "traffic_source": "rocprofv3 PMC (FETCH_SIZE x2 + WRITE_SIZE), profiles/r05_final_ant_pmc.json", "traffic": 32700000,
          "hbm_measured_frac": 0.078, "fwd_hbm_measured_frac": 0.078, "bwd_hbm_measured_frac": 0.0763, "alg_frac_step": 0.0008976}
    oc = [{"workload": "%s %d envs x H=32, MM_caching_frequency %d" % (nm, n, mm), "value": 2823456.789, "unit": "env-steps/s", "steps": 10,
           "ms_per_rollout": 11.6, "kernel_ms": 0.18, "fwd_kernel_ms": 0.17, "bwd_kernel_ms": 0.18, "ckpt_bytes_per_env_step": 201000,
           "kernels": k, "roofline": dict(rf)}
          for nm, n, mm, k in (("humanoid", 1024, 48, "specialised (compile-time layout)"), ("snu", 512, 8, "specialised (compile-time layout)"),
                               ("ant", 1024, 1, "specialised (compile-time layout)"), ("ant", 8192, 16, "specialised (compile-time layout)"),
                               ("ant", 1024, 16, "generic (run-time layout)"))]
    return {"metric": "fwd+adjoint env-steps/sec", "value": 9603456.789 * ranks, "unit": "env-steps/s", "n_gpus": ranks, "rccl_ranks": ranks,
            "oversubscribed": False, "backend": "nccl",
            "per_rank": [{"rank": r, "value": 9603456.789, "ms_per_step": 3.4123456, "ms_per_step_median": 3.41, "ms_per_step_max": 3.52,
                          "device": "AMD Instinct MI355X (cuda:%d)" % r} for r in range(ranks)],
            "steps": 20, "warmup": 5, "ms_per_step": 3.4123456, "ms_per_step_min": 3.40, "ms_per_step_median": 3.41, "ms_per_step_max": 3.52,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "ant 1024 envs/GPU x H=32 through DFlexEnv.step, loss=-sum(rew), 1 backward", "envs_per_gpu": 1024,
                       "envs_total": 1024 * ranks, "horizon": 32, "substeps": 16, "mm_freq": 16, "sharding": "envs by index, no collective",
                       "submission": "one HIP graph per rollout: the 32 forward + 32 adjoint launches captured through DFlexEnv.step",
                       "submission_fallback": False},
            "eager_env_steps_per_s": 6543210.123, "roofline": rf, "fp32_valu_frac_est": 0.058, "no_grad_forward_env_steps_per_s": 21e6,
            "other_configs": oc,
            "cpu_baseline": {"value": 3922.413606967296, "unit": "env-steps/s", "cores": 64, "kind": "port",
                             "sample": "47104 ant env-steps (forward + taped reverse sweep, oracle/dsim_oracle.cpp) in 12.0 s on 64 host "
                                       "threads of 256 cores, environments split over the threads",
                             "single_thread": {"value": 355.09, "unit": "env-steps/s", "cores": 1, "sample": "x" * 150},
                             "reference_recorded": {"value": 696.5, "unit": "env-steps/s", "cores": 8, "kernel_threads": 1, "source": "y" * 100}}}


@pytest.mark.parametrize("ranks", [1, 8])
def test_printed_line_is_compact_and_complete(ranks):
    """the printed line stays under 4 KB whatever the full record holds, and carries the contract's keys, a
    numeric roofline, the CPU baseline and one short object per other configuration"""
    sys.path.insert(0, ROOT)
    import bench
    full = _full_record(ranks)
    assert len(json.dumps(full)) > 8000   # (the record main() builds is far over the limit)
    line = bench.compact_line(full, "bench_full.json")
    assert len(line) < 4096 and "\n" not in line
    j = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "per_rank"):
        assert k in j, k
    assert j["dtype"] == "f32" and j["config"]["workload"].startswith("ant 1024 envs") and j["config"]["submission"] == "hip-graph"
    rf = j["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "fwd_alg_frac", "bwd_alg_frac", "alg_frac_step", "kernel_ms",
              "fwd_kernel_ms", "bwd_kernel_ms", "hbm_measured_frac", "valu_simd_frac", "fwd_valu_simd_frac", "fp32_valu_frac_est", "csrc_hash"):
        assert k in rf, k
    assert all(not isinstance(v, (dict, list)) for v in rf.values()) and "note" not in rf
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-6
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 64 and abs(cb["value"] - 3922.4) < 0.1 and cb["single_thread"] > 0 and cb["reference_recorded"] == 696.5
    assert len(j["per_rank"]) == ranks and len(j["other_configs"]) == 5
    assert sum(1 for o in j["other_configs"] if o["workload"].endswith("generic")) == 1
    for o in j["other_configs"]:
        assert set(o) >= {"workload", "value", "fwd_ms", "bwd_ms", "frac"} and o["value"] > 0
    assert abs(j["value"] / full["value"] - 1) < 1e-4   # (5 significant digits in the printed line, every bit in the full record)


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
    full = json.load(open(os.path.join(ROOT, j["full_record"])))   # the full record (per-kernel views, per-config rooflines)
    assert abs(full["value"] / j["value"] - 1) < 1e-4 and full["steps"] == 2
    oc = {key(o): o for o in full["other_configs"]}
    assert set(oc) == {"humanoid", "snu", "ant_mm1", "ant_generic", "ant_8192"}, oc
    for k, o in oc.items():   # each measured like the headline (>= 10 timed replays) and with its own roofline object
        assert o["value"] and o["value"] > 1e5 and o["steps"] >= (5 if k == "ant_8192" else 10), o
        r = o["roofline"]
        assert r["bound"] == "valu-issue" and r["traffic"] > r["alg_bytes_per_launch"] and 0 < r["hbm_measured_frac"] < 1, r
        assert r["kernel_ms"] == max(r["fwd_kernel_ms"], r["bwd_kernel_ms"])   # the triple (kernel, time, traffic) is ONE kernel's
    assert len(j["other_configs"]) == 5 and all(o["value"] > 1e5 and o["fwd_ms"] > 0 and o["bwd_ms"] > 0 for o in j["other_configs"])
    assert "AccumulateGrad" not in out.stderr, out.stderr[-600:]   # no autograd node of an earlier stream inside the timed replays
    assert oc["ant_generic"]["value"] < j["value"] < oc["ant_8192"]["value"]
    assert j["config"]["submission_fallback"] is False
    assert j["value"] > 1e5 and j["roofline"]["traffic"] > j["roofline"]["alg_bytes_per_launch"]
    assert j["roofline"]["bound"] == "valu-issue" and "valu_simd_frac" in j["roofline"] and "hbm_measured_frac" in j["roofline"]
    # round 5: the group really is RCCL, per-rank figures are in the line, the roofline object names the LONGER launch
    assert j["backend"] == "nccl" and len(j["per_rank"]) == 1
    r0 = j["per_rank"][0]
    assert r0["rank"] == 0 and "cuda:0" in r0["device"] and j["value"] <= r0["value"] < 1.05 * j["value"]   # (own time: before the closing barrier)
    assert 0.95 * j["ms_per_step"] < r0["ms_per_step"] <= j["ms_per_step"] and r0["ms_per_step_max"] >= r0["ms_per_step_median"] > 0
    rf = full["roofline"]
    longer = "dsim_env_fwd_kernel" if rf["fwd_kernel_ms"] > rf["bwd_kernel_ms"] else "dsim_env_bwd_kernel"
    assert rf["kernel"] == longer and rf["dominant_launch"] == ("forward" if longer.endswith("fwd_kernel") else "adjoint")
    assert j["roofline"]["kernel"] == longer
    assert abs(rf["alg_frac_step"] - 748 * full["value"] / 8e12) < 1e-9 and 0 < rf["alg_frac_step"] < 0.01
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
