"""`python bench.py --gpus N` must start N ranks itself (the driver's command has no launcher in front of it).
No GPU here: --dry-run makes the ranks rendezvous over gloo and count themselves; without it the ranks fail on
the missing GPU and the parent must report that with a non-zero exit code instead of printing a one-rank line."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=e, capture_output=True,
                          text=True, timeout=240)


@pytest.mark.timeout(300)
def test_gpus_n_spawns_n_ranks():
    p = _run("--gpus", "2", "--dry-run", "--workload", "tiny-int4sym-g128")
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["placement"] == "lpt"
    assert sum(line["tensors_per_rank"]) == 6 and min(line["tensors_per_rank"]) > 0
    assert line["launcher"].startswith("bench.py --gpus N")


@pytest.mark.timeout(300)
def test_expert_placement_for_mixtral_workloads():
    p = _run("--gpus", "4", "--dry-run", "--workload", "mixtral-mxfp4-g32", "--layers", "2")
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["ranks_seen"] == 4 and line["placement"] == "expert"
    assert line["tensors_per_rank"] == [12, 12, 12, 12]  # 2 experts x 3 matrices x 2 layers per rank


@pytest.mark.timeout(300)
def test_external_launcher_env_is_respected():
    """Under torchrun the ranks exist already: WORLD_SIZE in the environment must suppress the own launcher."""
    p = _run("--gpus", "2", "--dry-run", "--workload", "tiny-int4sym-g128", env={"WORLD_SIZE": "1", "RANK": "0"})
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["launcher"] == "external (torchrun)"


@pytest.mark.timeout(300)
@pytest.mark.skipif(torch.cuda.device_count() > 0, reason="needs a host without a GPU")
def test_failing_ranks_fail_the_parent():
    p = _run("--gpus", "2", "--workload", "tiny-int4sym-g128", "--steps", "1", "--warmup", "0")
    assert p.returncode != 0 and "ranks failed" in p.stderr
    assert p.stdout.strip() == ""
