"""Launched by tests/test_dist_cpu.py through mage_amd.utils.dist.launch_ranks (the launcher bench.py --gpus N uses):
each rank initialises the process group from the environment (gloo on CPU), counts the ranks the collective layer connected,
takes its shard of a global batch and writes what it saw."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from mage_amd.utils import dist as D  # noqa: E402


def main():
    out_dir, global_batch = sys.argv[1], int(sys.argv[2])
    rank, local_rank, world = D.env_rank_world()
    D.init_from_env("gloo")
    seen = D.ranks_seen(torch.device("cpu"))
    s, e = D.shard_range(global_batch, rank, world)
    tmax = D.max_over_ranks(float(rank), torch.device("cpu"))
    D.barrier()
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump({"rank": rank, "local_rank": local_rank, "world": world, "ranks_seen": seen, "shard": [s, e], "tmax": tmax}, f)
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
