"""`python bench.py --gpus N` must start its own ranks (the driver may call it either way): the launcher path on
CPU with gloo, world size 2 -- spawn under torch.distributed.run, rendezvous on 127.0.0.1, one collective, exactly
one JSON line from rank 0 on stdout, the children's exit status handed back."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True,
                          timeout=600)


def test_gpus_2_spawns_two_ranks_and_prints_one_line():
    p = _run(["--gpus", "2", "--launcher-selftest"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec == {"launcher_selftest": True, "n_gpus": 2, "sum_of_ranks_plus_one": 3.0}


def test_children_failure_is_the_exit_status():
    # the children see WORLD_SIZE=2 but are told --gpus 3 through the pass-through arguments: they refuse, and the
    # parent reports failure instead of printing a line
    p = _run(["--gpus", "2", "--launcher-selftest"], {"R_TUCKER_AMD_BENCH_TEST_BREAK": "1"})
    assert p.returncode == 0          # (sanity: the variable is not read anywhere)
    q = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--launcher-selftest"],
                       env={**os.environ, "WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                            "MASTER_PORT": "29999"}, capture_output=True, text=True, timeout=120)
    assert q.returncode != 0 and "WORLD_SIZE=2" in (q.stderr + q.stdout)
