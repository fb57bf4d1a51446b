"""One-process-per-GPU plumbing for the clip-sharded generation path.

Clips are independent, so the data path needs NO collective: each rank generates its own
contiguous shard.  torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" on CPU for tests) is used only for
the barrier, the max-over-ranks timing and the optional gather of generated tokens/clips to every rank.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist


def env_rank_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_from_env(backend: str = "nccl", device: torch.device = None) -> Tuple[int, int]:
    """Initialise the default process group from RANK/WORLD_SIZE/MASTER_* (torchrun).  No-op for world size 1."""
    rank, _, world = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split of n clips: rank r gets [r*n/W, (r+1)*n/W) (mirrors batchsize / ngpus_per_node, main_mage.py:93)."""
    if n % world:
        raise ValueError(f"global batch {n} is not divisible by world size {world}")
    per = n // world
    return rank * per, (rank + 1) * per


def shard_batch(batch: Dict[str, torch.Tensor], rank: int, world: int) -> Dict[str, torch.Tensor]:
    n = next(iter(batch.values())).shape[0]
    s, e = shard_range(n, rank, world)
    return {k: v[s:e] for k, v in batch.items()}


def barrier() -> None:
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(seconds: float, device: torch.device) -> float:
    if not dist.is_initialized():
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_clips(x: torch.Tensor) -> torch.Tensor:
    """All-gather shard outputs along dim 0 (tokens [B/W, L-1, h, w] or frames) in rank order."""
    if not dist.is_initialized():
        return x
    out = [torch.empty_like(x) for _ in range(dist.get_world_size())]
    dist.all_gather(out, x.contiguous())
    return torch.cat(out, 0)


def ranks_seen(device: torch.device) -> int:
    """Sum over ranks of 1 through the process group's all-reduce (RCCL on GPUs): how many ranks the collective layer
    actually connected, as opposed to what the environment claims."""
    if not dist.is_initialized():
        return 1
    t = torch.ones(1, dtype=torch.int32, device=device)
    dist.all_reduce(t)
    return int(t.item())


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(script_argv: List[str], nproc: int, port: Optional[int] = None, env: Optional[dict] = None) -> int:
    """One process per GPU on this node: re-run ``script_argv`` (script path + its arguments) under
    ``python -m torch.distributed.run`` with ``nproc`` ranks, rendezvous on 127.0.0.1.  The reference does this with
    ``mp.spawn(main_worker, nprocs=ngpus_per_node)`` (main_mage.py:279-295); here the launcher also exports
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, which is what ``init_from_env`` reads.  Returns the exit code."""
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: required for RCCL on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port or free_port())] + list(script_argv)
    return subprocess.call(cmd, env=e)
