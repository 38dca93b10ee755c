"""bench.py's multi-process plumbing on the CPU: `python bench.py --gpus 2` launches itself (torch.distributed.run, one
process per rank, gloo), runs the headline and every sharded leg on the oracle stand-in, and prints one JSON line.
Nothing it prints is a measurement; this checks that the N > 1 command the driver runs cannot die in the plumbing."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=600):
    """-> the full record (the side file bench.py names in its line) with the parsed stdout line under `_line`"""
    import tempfile
    detail = os.path.join(tempfile.mkdtemp(prefix="pplie_bench_"), "bench_legs.json")
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1", PPLIE_BENCH_DETAIL=detail)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, env=env, timeout=timeout,
                       cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    # the driver parses this line into its record: it must stay small and carry the contract + roofline + summary
    assert len(lines[0]) < 12_000, len(lines[0])
    line = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "summary"):
        assert key in line, key
    assert line["roofline"]["frac"] > 0 and line["roofline"]["bound"] == "hbm" and line["config"]["workload"].startswith("se3_explog")
    assert list(line)[-1] == "summary" and line["detail"] == detail
    with open(detail) as f:
        out = json.load(f)
    for key in ("metric", "value", "n_gpus", "steps", "warmup", "config", "roofline", "summary"):
        assert out[key] == line[key], key
    out["_line"] = line
    out["_stderr"] = r.stderr
    return out


def test_two_ranks_self_launched_dry_run():
    out = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--rows", "1000", "--backend", "gloo", "--standin"])
    assert out["n_gpus"] == 2 and out["config"]["ranks"] == 2 and out["config"]["collective_backend"] == "gloo"
    assert out["config"]["launch"].startswith("self-launched")
    assert out["value"] > 0 and out["scaling"] == "weak"
    assert out["ranks_seen"] == 2
    for leg in ("lm_invnet_sharded", "imu_sharded", "lm_pgo_replicated", "lm_pgo_node_sharded"):
        assert "error" not in out[leg], (leg, out[leg])
        assert out[leg]["n_gpus"] == 2 and out[leg]["value"] > 0
    assert out["lm_pgo_replicated"]["losses"][-1] < out["lm_pgo_replicated"]["losses"][0]
    assert out["lm_pgo_node_sharded"]["mode"].startswith("node-sharded solve")
    assert out["lm_pgo_node_sharded"]["losses"] == pytest.approx(out["lm_pgo_replicated"]["losses"], rel=1e-4)
    assert "deferred" in out["lm_pgo_sharded"]                     # the opt-in peer-exchange leg runs after the line is out
    post = [l for l in out["_stderr"].splitlines() if l.startswith("PPLIE_BENCH_POSTLINE ")]
    assert len(post) == 1
    leg = json.loads(post[0].split(" ", 1)[1])["lm_pgo_sharded"]
    # (asked for explicitly: node shards; on a gloo group the peer exchange itself does not apply and the solve takes the collectives)
    assert "error" not in leg and leg["ranks_seen"] == 2 and leg["effective"]["shard"] == "nodes"


def test_eight_ranks_self_launched_dry_run():
    """the command the driver runs on an 8-GPU node, on the CPU: every sharded leg with 8 ranks, each with `ranks_seen`, the
    effective shard / exchange mode, microseconds per iteration and its one-GPU equivalent"""
    out = _run(["--gpus", "8", "--steps", "2", "--warmup", "1", "--rows", "500", "--backend", "gloo", "--standin"], timeout=900)
    assert out["n_gpus"] == 8 and out["ranks_seen"] == 8 and out["config"]["ranks"] == 8
    for leg in ("lm_invnet_sharded", "imu_sharded", "lm_pgo_replicated", "lm_pgo_node_sharded"):
        assert "error" not in out[leg], (leg, out[leg])
        assert out[leg]["n_gpus"] == 8 and out[leg]["value"] > 0 and out[leg]["one_gpu_equivalent"]["value"] > 0
    for leg in ("lm_pgo_replicated", "lm_pgo_node_sharded"):
        assert out[leg]["ranks_seen"] == 8 and out[leg]["effective"]["shard"] in ("edges", "nodes") and out[leg]["speedup_vs_one_gpu"] > 0
        assert out[leg]["us_per_pcg_iteration_incl_step_overheads"] > 0
    assert out["lm_pgo_100k_one_gpu"]["value"] > 0
    summ = out["_line"]["summary"]
    for leg in ("lm_invnet_sharded", "imu_sharded", "lm_pgo_replicated", "lm_pgo_node_sharded", "lm_pgo_100k_one_gpu"):
        assert leg in summ, leg
    assert summ["lm_pgo_node_sharded"]["ranks"] == 8 and summ["lm_pgo_node_sharded"]["x"] > 0 and summ["lm_pgo_node_sharded"]["one_gpu"] > 0


def test_single_process_dry_run_has_every_config():
    out = _run(["--gpus", "1", "--steps", "2", "--warmup", "1", "--rows", "1000", "--backend", "gloo", "--standin"])
    assert out["n_gpus"] == 1 and out["config"]["launch"] == "single process"
    line = out["_line"]
    assert line["cpu_baseline"]["cores"] == 1 and line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["value"] > 0
    assert line["config"]["metric_second_half"]["value"] > 0 and line["value_lm_pgo_10k"] > 0
    assert "loss_rel" not in json.dumps(line["summary"])           # the floor-aware figure only
    for leg in ("c1", "ops_10m", "lm_invnet", "lm_pgo", "lm_pgo_100k", "imu", "imu_train", "ba_reproj"):
        assert leg in line["summary"], leg
    for leg in ("c1", "ops_10m", "lm_invnet", "lm_pgo", "lm_pgo_100k", "imu", "imu_train"):
        assert "error" not in out[leg], (leg, out[leg])
    assert out["lm_invnet"]["algorithmic_bytes_per_step"] == 84 * out["lm_invnet"]["problems_per_gpu"]
    assert set(out["imu"]) >= {"states_only", "with_covariance"}
