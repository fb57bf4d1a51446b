"""CPU, world_size 2, gloo: the multi-GPU contract of the generation path.  Clips are independent, so sharding the
batch across ranks and gathering the per-shard results must reproduce the single-process result exactly.  The GPU
kernels cannot run here; the per-rank generator is the CPU oracle (test infrastructure), which is enough to exercise
mage_amd.utils.dist (rendezvous from env, shard ranges, barrier, max-over-ranks, gather in rank order)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from mage_amd.utils import dist as D, synth
    from oracle import mage_oracle as O
    from tests.helpers import build_mage, cpu_sd
    r, w = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    L = 3
    sd = cpu_sd(build_mage(synth.mnist_model_config(frames_length=L, width=64, layers=2, vq_dim=32, K=32), 9))
    full = synth.synth_batch_mnist(4, L, seed=9)
    mine = D.shard_batch(full, r, w)
    assert mine["images"].shape[0] == 2
    D.barrier()
    _, tok, _, _ = O.mage_generate(sd, mine, L, return_trace=True)
    allt = D.gather_clips(tok)
    tmax = D.max_over_ranks(1.0 + r, torch.device("cpu"))
    _, want, _, _ = O.mage_generate(sd, full, L, return_trace=True)
    q.put((rank, bool(torch.equal(allt, want)), tmax))
    D.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_clip_sharding_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True, 2.0), (1, True, 2.0)]


def test_shard_range_rejects_ragged_split():
    sys.path.insert(0, ROOT)
    from mage_amd.utils.dist import shard_range
    assert shard_range(256, 3, 8) == (96, 128)
    with pytest.raises(ValueError):
        shard_range(10, 0, 4)


def test_launch_ranks_starts_world_and_all_reduce_sees_every_rank(tmp_path):
    """bench.py --gpus N (outside torchrun) starts its ranks with mage_amd.utils.dist.launch_ranks; the same launcher, world
    size 2, gloo: both ranks come up, the all-reduce of ones (bench.py's `ranks_seen`) counts 2, the strong-scaling split of a
    global batch of 256 is the contiguous halves."""
    import json
    sys.path.insert(0, ROOT)
    from mage_amd.utils.dist import launch_ranks
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["OMP_NUM_THREADS"] = "1"
    rc = launch_ranks([os.path.join(ROOT, "tests", "rank_probe.py"), str(tmp_path), "256"], 2, env=env)
    assert rc == 0
    got = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    assert [g["ranks_seen"] for g in got] == [2, 2] and [g["world"] for g in got] == [2, 2]
    assert [g["shard"] for g in got] == [[0, 128], [128, 256]] and [g["tmax"] for g in got] == [1.0, 1.0]


def test_cfg3_captions_have_the_double_mnist_lengths():
    sys.path.insert(0, ROOT)
    from mage_amd.utils import synth
    b = synth.synth_batch_mnist(64, 4, seed=1, digits=2, caption_lengths=(16, 18, 20))
    text = b["text"]
    assert tuple(text.shape) == (64, 20)
    lens = (text != 0).sum(1)
    assert set(lens.tolist()) == {16, 18, 20}                      # ragged: right-padded with 0 (the padded-text quirk of a9/a10)
    assert (text[:, 0] == 1).all() and all(text[i, lens[i] - 1] == 2 for i in range(64))
