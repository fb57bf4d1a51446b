"""CPU side of the training path: the flat-arena optimizer's bookkeeping and its data-parallel step
(reduce-scatter of the gradient arena -> Adam on the local shard -> all-gather of the parameter arena), world size 2 over gloo.
The fused Adam kernel is HIP-only (FlatAdam raises on CPU tensors); here a test double with torch arithmetic stands in for that
ONE launch so that the arena layout, the sharding and the collectives are what is tested."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _adam_double(self, p, g, m, v, lr, b1, b2, eps, step, grad_scale):
    g = g * grad_scale
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    p.sub_((lr / (1 - b1 ** step)) * m / (v.sqrt() / (1 - b2 ** step) ** 0.5 + eps))


def _net(seed):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Linear(7, 13), torch.nn.Tanh(), torch.nn.Linear(13, 5), torch.nn.LayerNorm(5))


def test_flat_adam_refuses_cpu_tensors_and_keeps_arena_views():
    from mage_amd.optim import FlatAdam
    net = _net(0)
    opt = FlatAdam(net.parameters(), lr=1e-2)
    n = sum(p.numel() for p in net.parameters())
    assert opt.n == n and opt.n_pad % 4 == 0
    for p, off in zip(opt.params, opt.offsets):
        assert p.data_ptr() == opt.flat_p.data_ptr() + 4 * off and p.grad.data_ptr() == opt.flat_g.data_ptr() + 4 * off
    net(torch.randn(3, 7)).sum().backward()
    assert opt.flat_g.abs().sum() > 0                       # autograd accumulated INTO the arena
    with pytest.raises(RuntimeError, match="mage_adam"):
        opt.step()                                          # no CPU path
    opt.zero_grad(set_to_none=False)
    assert opt.flat_g.abs().sum() == 0 and all(p.grad.data_ptr() == opt.flat_g.data_ptr() + 4 * off for p, off in zip(opt.params, opt.offsets))
    # the default (torch.optim's set_to_none=True): autograd hands the gradients over by reference, step() gathers them into the arena
    opt.zero_grad()
    assert all(p.grad is None for p in opt.params)
    net(torch.randn(3, 7)).sum().backward()
    want = torch.cat([p.grad.reshape(-1) for p in opt.params])
    opt._collect_grads()
    assert torch.equal(opt.flat_g[:n], want) and all(p.grad.data_ptr() == opt.flat_g.data_ptr() + 4 * off for p, off in zip(opt.params, opt.offsets))


def test_flat_adam_matches_torch_adam_single_process(monkeypatch):
    from mage_amd.optim import FlatAdam
    monkeypatch.setattr(FlatAdam, "_adam", _adam_double)
    a, b = _net(1), _net(1)
    ref = torch.optim.Adam(a.parameters(), lr=1e-2, betas=(0.9, 0.98), eps=1e-6)
    opt = FlatAdam(b.parameters(), lr=1e-2, betas=(0.9, 0.98), eps=1e-6)
    for i in range(4):
        x = torch.randn(6, 7, generator=torch.Generator().manual_seed(i))
        for net, o in ((a, ref), (b, opt)):
            o.zero_grad()
            net(x).pow(2).mean().backward()
            o.step()
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.allclose(pa, pb, atol=1e-6)
    sd = opt.state_dict()
    opt2 = FlatAdam(_net(1).parameters(), lr=1e-2, betas=(0.9, 0.98), eps=1e-6)
    opt2.load_state_dict(sd)
    assert opt2.steps == 4 and torch.equal(opt2.m, opt.m) and opt2.param_groups[0]["lr"] == 1e-2
    # the checkpoint entry has torch.optim.Adam's layout: interchangeable with the reference's optimizer in both directions
    rsd = ref.state_dict()
    assert set(sd["state"]) == set(rsd["state"]) and sd["param_groups"][0]["params"] == rsd["param_groups"][0]["params"]
    for i in rsd["state"]:
        assert torch.allclose(sd["state"][i]["exp_avg"], rsd["state"][i]["exp_avg"], atol=1e-7)
        assert torch.allclose(sd["state"][i]["exp_avg_sq"], rsd["state"][i]["exp_avg_sq"], atol=1e-9)
        assert float(sd["state"][i]["step"]) == float(rsd["state"][i]["step"])
    opt3 = FlatAdam(_net(1).parameters(), lr=1e-2, betas=(0.9, 0.98), eps=1e-6)
    opt3.load_state_dict(rsd)                               # a reference checkpoint's 'optimizer' entry
    assert opt3.steps == 4 and torch.allclose(opt3.m, opt.m, atol=1e-7)
    torch.optim.Adam(_net(1).parameters(), lr=1e-2).load_state_dict(sd)     # and the other way round


def test_flat_adam_accepts_the_layout_numbered_over_trainable_parameters(monkeypatch):
    """Checkpoints this class wrote before it counted frozen parameters (entry j = the j-th TRAINABLE parameter, param_groups.params of that
    length) load into the right slots; a length that is neither layout is still an error."""
    from mage_amd.optim import FlatAdam
    monkeypatch.setattr(FlatAdam, "_adam", _adam_double)

    def frozen_first(seed):
        net = _net(seed)
        net[0].weight.requires_grad_(False)                  # parameter 0 frozen: trainable indices are 1..5
        return net
    a = frozen_first(3)
    opt = FlatAdam(a.parameters(), lr=1e-2)
    for i in range(2):
        opt.zero_grad()
        a(torch.randn(4, 7, generator=torch.Generator().manual_seed(i))).pow(2).mean().backward()
        opt.step()
    sd = opt.state_dict()
    assert sorted(sd["state"]) == opt.index == [1, 2, 3, 4, 5] and len(sd["param_groups"][0]["params"]) == 6
    assert sd["param_groups"][0]["flat_adam_numbering"] == "all"                  # the explicit marker of the index space
    g_old = {k: v for k, v in sd["param_groups"][0].items() if k != "flat_adam_numbering"}      # a checkpoint older than the marker
    legacy = {"state": {j: sd["state"][i] for j, i in enumerate(opt.index)}, "param_groups": [dict(g_old, params=list(range(len(opt.index))))]}
    for ck in (legacy, {"state": legacy["state"], "param_groups": [dict(legacy["param_groups"][0], flat_adam_numbering="trainable")]}):
        opt2 = FlatAdam(frozen_first(3).parameters(), lr=1e-2)
        opt2.load_state_dict(ck)
        assert opt2.steps == 2 and torch.equal(opt2.m, opt.m) and torch.equal(opt2.v, opt.v)
    # a foreign checkpoint that merely has the legacy LENGTH is refused, not remapped: an out-of-range key, a shape that does not fit
    foreign = {"state": {**legacy["state"], 7: legacy["state"][0]}, "param_groups": legacy["param_groups"]}
    with pytest.raises(ValueError, match="outside the 5 trainable"):
        FlatAdam(frozen_first(3).parameters(), lr=1e-2).load_state_dict(foreign)
    swapped = {"state": {0: legacy["state"][2], **{j: legacy["state"][j] for j in (1, 2, 3, 4)}}, "param_groups": legacy["param_groups"]}
    if tuple(legacy["state"][0]["exp_avg"].shape) != tuple(legacy["state"][2]["exp_avg"].shape):
        with pytest.raises(ValueError, match="not a checkpoint of this parameter list"):
            FlatAdam(frozen_first(3).parameters(), lr=1e-2).load_state_dict(swapped)
    bad = {"state": {}, "param_groups": [dict(g_old, params=list(range(4)))]}
    with pytest.raises(ValueError, match="built over 4 parameters"):
        FlatAdam(frozen_first(3).parameters(), lr=1e-2).load_state_dict(bad)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    import torch.distributed as dist
    from mage_amd.optim import FlatAdam
    from mage_amd.utils import dist as D
    FlatAdam._adam = _adam_double
    D.init_from_env("gloo")
    net = _net(2 + 10 * rank)                                # replicas initialised DIFFERENTLY: FlatAdam broadcasts rank 0's (as DDP does)
    opt = FlatAdam(net.parameters(), lr=1e-2, betas=(0.9, 0.98), eps=1e-6)
    assert opt.sharded and opt.shard_n * world == opt.n_pad and opt.m.numel() == opt.shard_n      # optimizer state is divided by W
    assert all(torch.equal(a, b) for a, b in zip(net.parameters(), _net(2).parameters()))
    data = torch.randn(8, 7, generator=torch.Generator().manual_seed(5))
    mine = data[rank * 4:(rank + 1) * 4]                     # each rank sees its half of the global batch
    for _ in range(3):
        opt.zero_grad()
        net(mine).pow(2).mean().backward()
        opt.step()
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    # checkpoint (main_mage.py:186-193 saves from rank 0 only): state_dict() is LOCAL and refuses sharded state that was not gathered;
    # consolidate_state_dict() is the collective every rank makes first; the saving rank then holds the full moments
    try:
        opt.state_dict()
        refused = False
    except RuntimeError as e:
        refused = "consolidate_state_dict" in str(e)
    opt.consolidate_state_dict(to=0)
    mom = []
    if rank == 0:
        sd = opt.state_dict()
        mom = torch.cat([sd["state"][i]["exp_avg"].reshape(-1) for i in range(len(opt.params))]).tolist()
    q.put((rank, flat.tolist(), bool(torch.equal(gathered[0], gathered[1])) and refused, mom))    # plain lists: no shared-memory handles outlive the worker
    D.barrier()
    dist.destroy_process_group()


def test_sharded_step_over_two_ranks_equals_the_global_batch_step():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][2] and res[1][2]                           # replicas stay identical after the all-gather
    # single process on the whole batch with torch's own Adam: mean over 8 samples == mean of the two half-batch means
    net = _net(2)
    ref = torch.optim.Adam(net.parameters(), lr=1e-2, betas=(0.9, 0.98), eps=1e-6)
    data = torch.randn(8, 7, generator=torch.Generator().manual_seed(5))
    for _ in range(3):
        ref.zero_grad()
        (0.5 * (net(data[:4]).pow(2).mean() + net(data[4:]).pow(2).mean())).backward()
        ref.step()
    want = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    assert torch.allclose(torch.tensor(res[0][1]), want, atol=2e-6)
    want_m = torch.cat([ref.state_dict()["state"][i]["exp_avg"].reshape(-1) for i in range(len(list(net.parameters())))])
    assert res[1][3] == [] and torch.allclose(torch.tensor(res[0][3]), want_m, atol=2e-6)     # sharded moments, gathered onto rank 0


def test_flat_adam_checkpoint_numbers_frozen_parameters_like_torch_adam(monkeypatch):
    """The reference builds optim.Adam over model.parameters() with the frozen first stage inside (main_mage.py:121): its checkpoint's
    state indices count the frozen parameters.  FlatAdam keeps that numbering, so the two 'optimizer' entries load into each other; a
    parameter without a state entry (never received a gradient) starts from zero moments; an entry that would land on a frozen parameter
    or a different shape raises instead of being reassigned."""
    from mage_amd.optim import FlatAdam
    monkeypatch.setattr(FlatAdam, "_adam", _adam_double)

    def build(seed):
        torch.manual_seed(seed)
        net = torch.nn.ModuleDict({"top": torch.nn.Linear(7, 7), "frozen": torch.nn.Linear(7, 7), "body": _net(seed), "unused": torch.nn.Linear(3, 3)})
        for p in net["frozen"].parameters():                # registered BEFORE the trainable body, same shapes as `top`
            p.requires_grad_(False)
        return net

    def fwd(net, x):
        return net["body"](net["frozen"](net["top"](x))).pow(2).mean()

    a, b = build(3), build(3)
    ref = torch.optim.Adam(a.parameters(), lr=1e-2, betas=(0.9, 0.98), eps=1e-6)
    opt = FlatAdam(b.parameters(), lr=1e-2, betas=(0.9, 0.98), eps=1e-6)
    for i in range(3):
        x = torch.randn(6, 7, generator=torch.Generator().manual_seed(i))
        for net, o in ((a, ref), (b, opt)):
            o.zero_grad()
            fwd(net, x).backward()
            o.step()
    rsd, sd = ref.state_dict(), opt.state_dict()
    n_all = len(list(a.parameters()))
    assert rsd["param_groups"][0]["params"] == sd["param_groups"][0]["params"] == list(range(n_all))
    assert set(rsd["state"]) <= set(sd["state"]) and 2 not in sd["state"] and 3 not in sd["state"]       # 2, 3 = the frozen Linear
    for i in rsd["state"]:                                  # (`unused` has no entry in torch's dict: it never received a gradient)
        assert torch.allclose(sd["state"][i]["exp_avg"], rsd["state"][i]["exp_avg"], atol=1e-7)
    opt2 = FlatAdam(build(3).parameters(), lr=1e-2, betas=(0.9, 0.98), eps=1e-6)
    opt2.load_state_dict(rsd)                               # the reference's checkpoint entry, unused parameters absent
    assert opt2.steps == 3 and torch.allclose(opt2.m, opt.m, atol=1e-7)
    torch.optim.Adam(build(3).parameters(), lr=1e-2).load_state_dict(sd)
    shifted = {"state": {i + 2: e for i, e in rsd["state"].items() if i < 2}, "param_groups": rsd["param_groups"]}     # top's moments on `frozen`
    with pytest.raises(ValueError, match="frozen"):
        FlatAdam(build(3).parameters()).load_state_dict(shifted)
    with pytest.raises(ValueError, match="built over"):
        FlatAdam(_net(3).parameters()).load_state_dict(rsd)
