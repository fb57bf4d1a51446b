"""GPU: the multi-GPU path, self-checking for the day the test box has more than one device (the driver's 8-GPU node, or any >= 2).

One process per GPU over RCCL, launched exactly as `bench.py --gpus N` launches its ranks (mage_amd.utils.dist.launch_ranks ->
python -m torch.distributed.run, rendezvous on 127.0.0.1; main_mage.py:76-77,93-95,279-295).  On a one-GPU box the same rank script
runs as ONE rank (a 1-rank nccl group still goes through RCCL), so its logic is exercised at every round; the N >= 2 tests skip."""
import json
import os
import subprocess
import sys

import pytest
import torch

from mage_amd.utils import dist as D

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "tests", "rank_multi_gpu.py")


def _run_ranks(n):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(D.free_port()), SCRIPT]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert f"MULTI_GPU_OK world={n} ranks_seen={n}" in p.stdout, p.stdout[-2000:]


def test_rank_script_as_one_rank_over_rccl():
    _run_ranks(1)


@pytest.mark.parametrize("n", [2, 4, 8])
def test_sharded_generation_and_training_over_rccl(n):
    if torch.cuda.device_count() < n:
        pytest.skip(f"needs {n} GPUs, this box has {torch.cuda.device_count()}")
    _run_ranks(n)


def test_bench_line_on_two_gpus_counts_two_ranks():
    """bench.py --gpus 2 launches its own ranks; the line's whole-job value covers both and `ranks_seen` comes from an RCCL all-reduce."""
    if torch.cuda.device_count() < 2:
        pytest.skip(f"needs 2 GPUs, this box has {torch.cuda.device_count()}")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-other-mode",
           "--no-parity-mode", "--no-decode-roofline", "--no-train-step", "--no-latency-b1"]
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=1800, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["ranks_seen"] == 2 and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 2 * 64
