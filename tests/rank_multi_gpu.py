"""One rank of the multi-GPU self-check (tests/test_gpu_multi.py launches N of these through mage_amd.utils.dist.launch_ranks, the
launcher `bench.py --gpus N` uses; backend nccl = RCCL, one process per GPU -- main_mage.py:76-77,93-95).

Checks, every one through the real collective layer:
  * ranks_seen == WORLD_SIZE (an RCCL all-reduce of ones);
  * generation: each rank generates its contiguous shard of a global batch; the all-gathered token sequences equal, bit for bit, the
    sequences of the SAME global batch generated on one GPU (clips are independent: SURVEY 8e);
  * training: W ranks, each on its shard, with the sharded FlatAdam step (reduce-scatter -> Adam on the shard -> all-gather) reach the
    parameters of ONE process stepping torch.optim.Adam on the mean of the shards' losses (= the gradient DistributedDataParallel averages);
  * the sharded optimizer's checkpoint protocol: consolidate_state_dict() on every rank, state_dict() on rank 0 only.
Prints MULTI_GPU_OK from rank 0."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from mage_amd.optim import FlatAdam  # noqa: E402
from mage_amd.utils import dist as D  # noqa: E402
from mage_amd.utils import synth  # noqa: E402
from tests.helpers import build_mage  # noqa: E402


def main():
    rank, local_rank, world = D.env_rank_world()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)       # also for ONE rank: the path must be RCCL's
    assert dist.get_backend() == "nccl"
    seen = D.ranks_seen(dev)
    assert seen == world, (seen, world)

    L, per = 4, 2
    cfg = synth.mnist_model_config(frames_length=L, width=64, layers=3, vq_dim=32, K=64)
    gbatch = synth.synth_batch_mnist(per * world, L, seed=77)
    # ---- generation: shard == slice of the one-GPU run
    m = build_mage(cfg, 77, dev)
    mine = {k: v.to(dev) for k, v in D.shard_batch(gbatch, rank, world).items()}
    m.autoregressive_generate(mine)
    toks = D.gather_clips(m.last_tokens)
    m.autoregressive_generate({k: v.to(dev) for k, v in gbatch.items()})
    assert torch.equal(toks, m.last_tokens), "sharded generation differs from the one-GPU generation of the same batch"

    # ---- training: sharded FlatAdam over RCCL == torch.optim.Adam on the mean of the shards' losses
    m = build_mage(cfg, 78, dev)
    opt = FlatAdam(m.parameters(), lr=1e-3, betas=(0.9, 0.98), eps=1e-6, shard=True)
    assert opt.sharded and opt.shard_n * world == opt.n_pad
    ref = build_mage(cfg, 78, dev)
    ropt = torch.optim.Adam([p for p in ref.parameters() if p.requires_grad], lr=1e-3, betas=(0.9, 0.98), eps=1e-6)
    for _ in range(2):
        opt.zero_grad()
        loss, _ = m(mine)
        loss.backward()
        opt.step()
        ropt.zero_grad()
        total = 0.0
        for r in range(world):
            lr_, _ = ref({k: v.to(dev) for k, v in D.shard_batch(gbatch, r, world).items()})
            total = total + lr_ / world
        total.backward()
        ropt.step()
    got, want = m.state_dict(), ref.state_dict()
    for k in want:            # not bitwise: the embedding scatters use fp32 atomics, and the sum over ranks is a different order
        assert torch.allclose(got[k].float(), want[k].float(), atol=5e-6, rtol=1e-4), k
    # every rank holds the same parameters after the all-gather
    flat = torch.cat([p.detach().reshape(-1) for p in m.parameters() if p.requires_grad])
    others = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(others, flat)
    assert all(torch.equal(o, others[0]) for o in others)
    # ---- checkpoint protocol
    if world > 1:
        try:
            opt.state_dict()
            raise AssertionError("state_dict() on sharded state must ask for consolidate_state_dict()")
        except RuntimeError:
            pass
    opt.consolidate_state_dict(to=0)
    if rank == 0:
        sd = opt.state_dict()
        assert len(sd["param_groups"][0]["params"]) == len(list(m.parameters())) and len(sd["state"]) == len(opt.params)
    D.barrier()
    if rank == 0:
        print(f"MULTI_GPU_OK world={world} ranks_seen={seen}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
