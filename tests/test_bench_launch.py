"""bench.py's own launch logic, end to end on a box without a GPU: `python bench.py --gpus 2` with no launcher
around it must start two ranks itself (a parent that never touches the GPU -> `python -m
torch.distributed.run` as a child), run BOTH multi-GPU schemes and print ONE line with n_gpus == 2.  The
compute is the oracle-backed stand-in of tests/bench_cpu_engine.py under gloo (DVS_BENCH_TEST_ENGINE, a hook
that exists for this test only); what is under test is bench.py: self-launch, rank plumbing, the exchange
steps of diverseseq_amd.parallel, the max-over-ranks timing, the line's fields and the exit codes."""
import json
import os
import pathlib
import subprocess
import sys

import numpy as np

from conftest import ROOT

ARGS = ["--steps", "1", "--warmup", "0", "--nseq", "240", "--length", "200", "-k", "3", "-n", "5", "--no-cpu-baseline"]


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["DVS_BENCH_TEST_ENGINE"] = "bench_cpu_engine"
    env["PYTHONPATH"] = os.pathsep.join([str(ROOT), str(pathlib.Path(__file__).resolve().parent),
                                         env.get("PYTHONPATH", "")])
    return env


def test_bench_gpus_2_launches_two_ranks_itself():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", *ARGS], env=_env(),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2
    assert out["config"]["launched_by"].startswith("bench.py itself")
    assert out["engine_module"] == "bench_cpu_engine"
    # both schemes in one line, each with its exchange step's time
    assert out["value"] > 0 and out["value_exact"] > 0
    assert out["config"]["chunk_mode_collective"]["samples"] >= 1
    assert out["config"]["exact_mode"]["collective"] is None or out["config"]["exact_mode"]["collective"]["samples"] >= 1
    assert out["config"]["exact_mode"]["accepts_per_step"] > 0
    assert out["scaling"] == "weak" and out["unit"] == "sequences/s"


def test_bench_refuses_a_world_that_is_not_gpus():
    """under a launcher (WORLD_SIZE set) the line must not claim another n_gpus: exit non-zero"""
    env = _env()
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29611")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", *ARGS], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "WORLD_SIZE=1" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_exact_scheme_of_the_stand_in_gives_the_one_process_answer():
    """the stand-in's exact mode at world 1 (no launcher, --gpus 1 --mode exact) selects what the oracle selects
    from the same stream: the hook is a faithful stand-in, not a stub"""
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--mode", "exact", *ARGS], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-4000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["config"]["accepts_per_step"] > 0
