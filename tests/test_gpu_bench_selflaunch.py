"""bench.py --gpus N started BARE (no torchrun around it, the form the driver uses for --gpus 1) must become N ranks by itself:
the parent spawns the launcher as a child before anything touches the GPU.  On a one-GPU box the 2-rank path is rehearsed with
RADAD_BENCH_REHEARSE=1 (both ranks on cuda:0, gloo); without that flag fewer devices than ranks is an error, never a silent
one-rank run.  (The reference is single-GPU: vector_database.py:23 `device_id = 0`; sharding is this build's.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *args, timeout=420):
    env = dict(os.environ, **extra_env)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True,
                          timeout=timeout, cwd=ROOT)


def test_more_ranks_than_gpus_is_an_error():
    """(runs on the CPU too: the device count is 0 there)"""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("a multi-GPU box runs the real thing")
    r = _run({"RADAD_BENCH_REHEARSE": "0"}, "--gpus", "2", "--steps", "1", timeout=120)
    assert r.returncode != 0 and "refusing to fall back" in r.stderr and not r.stdout.strip()


@pytest.mark.gpu
def test_bare_bench_with_two_ranks_launches_itself(gpu):
    import torch
    env = {} if torch.cuda.device_count() >= 2 else {"RADAD_BENCH_REHEARSE": "1"}
    r = _run(env, "--gpus", "2", "--steps", "4", "--warmup", "1", "--db-rows", "200000", "--clips", "256", "--sustain", "0",
             "--pcie", "0", "--unstructured", "0", "--cpu-sample", "0", "--cpu-baseline-clips", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "shard2" and out["config"]["planted_neighbours_found"]
    assert "sharded" in out and out["sharded"]["collective_ms"] >= 0
