"""`python bench.py --gpus 2` on real GPUs: the self-spawned two-rank form (one process per GPU, RCCL all-gather of the
score shards) -- runs wherever two devices are visible, skipped on a one-GPU box.  The CPU counterpart of the launcher
is tests/test_bench_launcher.py (gloo, world size 2)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
@pytest.mark.parametrize("workload", ["wn18rr_asym_r10x200_b512_f32", "fb15k_asym_r200x200_b512_f32"])
def test_bench_gpus_2(workload):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"          # dmabuf IPC (RCCL across processes on this pool)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", workload, "--steps", "20",
                        "--warmup", "5", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 20 and rec["value"] > 0
    assert rec["config"]["sharding"] != "none"
    assert 0 < rec["roofline"]["frac"] < 1
