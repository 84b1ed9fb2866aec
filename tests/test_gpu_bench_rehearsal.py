"""GPU (-m gpu): bench.py's N > 1 plumbing, rehearsed on one card so that the first real RCCL run cannot die on arguments.

GS_BENCH_REHEARSAL=1 puts every rank on cuda:0 and runs the collectives over gloo (a one-GPU box has no second card and
RCCL wants one per rank); the NUMBERS of such a run mean nothing, the code path -- pose dealing, gradient all-reduce,
schedule flags, the JSON contract -- is the one `python -m torch.distributed.run ... bench.py --gpus N` takes on an
8-GPU node.  Fresh child processes, a small frame (synthetic.SMALL), three schedules:
  default            one view per rank per step, one all-reduce per step = BASELINE config 4
  --views-per-rank 2 --reduce view      gradient accumulation with one asynchronous all-reduce per view
  --scheme gaussian  Gaussian-parallel (records out, sums back, no all-reduce)
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--steps", "3", "--warmup", "2", "--breakdown-steps", "1", "--workload", "tiny_rehearsal", "--no-cpu-baseline"]


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _run(n, extra):
    env = dict(os.environ, GS_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    bench = os.path.join(ROOT, "bench.py")
    if n == 1:
        cmd = [sys.executable, bench, "--gpus", "1"] + COMMON + extra
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), bench, "--gpus", str(n)] + COMMON + extra
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]            # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def _contract(d, n):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == n and d["steps"] == 3 and d["warmup"] == 2 and d["unit"] == "frames/s" and d["scaling"] == "weak"
    assert d["value"] > 0 and d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["bound_kind"] in ("hbm", "fp32_vector") and r["peak"] > 0 and 0 < r["frac"] < 1.5


def test_single_gpu_line_and_pose_cycling():
    d = _run(1, [])
    _contract(d, 1)
    c = d["config"]
    assert c["views_per_rank"] == 1 and c["collectives_per_step"] == 0 and c["poses_cycled"] == 8
    assert "NOT BASELINE" not in d["metric"]


def test_two_ranks_default_is_baseline_config_4():
    d = _run(2, [])
    _contract(d, 2)
    c = d["config"]
    assert c["rehearsal"] is True
    assert c["views_per_rank"] == 1 and c["views_per_step"] == 2 and c["collectives_per_step"] == 1, c
    assert c["exchange_bytes_sent_per_rank_per_step"] == 236 * c["points"]
    assert "NOT BASELINE" not in d["metric"] and "all-reduce per step" in c["parallelism"]


def test_two_ranks_accumulating_schedule_is_labelled_as_another_workload():
    d = _run(2, ["--views-per-rank", "2", "--reduce", "view"])
    _contract(d, 2)
    c = d["config"]
    assert c["views_per_rank"] == 2 and c["views_per_step"] == 4 and c["collectives_per_step"] == 2, c
    assert "NOT BASELINE config 4" in d["metric"]


def test_two_ranks_gaussian_parallel_schedule():
    d = _run(2, ["--scheme", "gaussian"])
    _contract(d, 2)
    c = d["config"]
    assert c["collectives_per_step"] == 3 and c["exchange_bytes_sent_per_rank_per_step"] > 0 and "gaussian-parallel" in c["parallelism"]
