"""Optimizer of the training path: Adam over flat fp32 arenas, sharded across the data-parallel ranks.

The reference trains with ``optim.Adam(model.parameters(), lr, betas=(0.9, 0.98), eps=1e-6)`` under
``DistributedDataParallel`` (main_mage.py:95,121,150-153): every rank all-reduces every gradient and then repeats the same
Adam step on all parameters.  On one MI355X node (xGMI is point-to-point: ring collectives are per-link bound, so fewer and
larger transfers win) the same update is done as

    reduce-scatter(flat gradient arena)  ->  Adam on this rank's 1/W shard  ->  all-gather(flat parameter arena)

over RCCL: the gradient crosses the links once (an all-reduce is a reduce-scatter plus an all-gather of the GRADIENT; here the
second half moves the updated PARAMETERS instead and the optimizer state and arithmetic are divided by W).  All parameters
live in one contiguous fp32 arena (``p.data`` are views into it), all gradients in another (``p.grad`` are views), so each
collective is ONE call on a 16-byte aligned buffer (148 MB at the MNIST config, 539 MB at caterv1) and the step is ONE
``mage_adam`` launch per shard.

``FlatAdam`` keeps ``torch.optim.Optimizer``'s surface (``param_groups[...]['lr']`` for the reference's
``adjust_learning_rate``, ``zero_grad``, ``step``, ``state_dict`` / ``load_state_dict`` for its checkpoints).
"""
from __future__ import annotations

from typing import Iterable, Optional

import torch
import torch.distributed as dist

from . import ops
from .modules import vqvae_model as _vq

__all__ = ["FlatAdam"]


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.98), eps: float = 1e-6,
                 process_group: Optional[dist.ProcessGroup] = None, shard: Optional[bool] = None):
        # The reference hands EVERY parameter to optim.Adam (main_mage.py:121: model.parameters(), the frozen first stage included -- its
        # parameters follow MAGE's three top-level nn.Parameters), so the indices of its checkpoint's 'state' and 'param_groups' count the
        # frozen ones too.  Keep that numbering: `all_params` is the caller's order, `index[i]` the position of trainable parameter i in it.
        all_params = list(params)
        if any(isinstance(p, dict) for p in all_params):
            raise ValueError("FlatAdam: one parameter group (an iterable of tensors), as the reference's optim.Adam call")
        plist = [p for p in all_params if p.requires_grad]
        if not plist:
            raise ValueError("FlatAdam: no trainable parameters")
        super().__init__(all_params, dict(lr=lr, betas=tuple(betas), eps=eps))
        self.all_params = all_params
        self.index = [i for i, p in enumerate(all_params) if p.requires_grad]
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        # shard=None: shard whenever there is more than one rank; shard=True forces the reduce-scatter / all-gather path on any
        # initialised process group (a 1-rank group still goes through RCCL: used by the single-GPU test of the collective path)
        self.sharded = (self.world > 1) if shard is None else (bool(shard) and dist.is_initialized())
        dev = plist[0].device
        if any(p.device != dev or p.dtype != torch.float32 for p in plist):
            raise ValueError("FlatAdam: parameters must be fp32 tensors on one device")
        self.params = plist
        n = sum(p.numel() for p in plist)
        q = 4 * self.world                                       # every shard starts 16-byte aligned
        self.n, self.n_pad = n, (n + q - 1) // q * q
        self.flat_p = torch.zeros(self.n_pad, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(self.n_pad, device=dev, dtype=torch.float32)
        self.offsets = []
        off = 0
        with torch.no_grad():
            for p in plist:
                k = p.numel()
                self.flat_p[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[off:off + k].view(p.shape)             # parameters become views into the arena
                p.grad = self.flat_g[off:off + k].view(p.shape)
                self.offsets.append(off)
                off += k
        self.shard_n = self.n_pad // self.world if self.sharded else self.n_pad
        self.shard_off = self.rank * self.shard_n if self.sharded else 0
        self.m = torch.zeros(self.shard_n, device=dev, dtype=torch.float32)
        self.v = torch.zeros(self.shard_n, device=dev, dtype=torch.float32)
        self.shard_g = torch.empty(self.shard_n, device=dev, dtype=torch.float32) if self.sharded else None
        self.steps = 0
        self._consolidated = None                                # (steps, exp_avg, exp_avg_sq) gathered by consolidate_state_dict
        if self.world > 1:
            # DistributedDataParallel broadcasts rank 0's parameters when it wraps a model (main_mage.py:95); a bare model + FlatAdam
            # gets the same guarantee here: replicas that were initialised differently start from rank 0's weights
            dist.broadcast(self.flat_p, src=dist.get_global_rank(self.pg, 0) if self.pg is not None else 0, group=self.pg)
            _vq.bump_weights_epoch()

    # ------------------------------------------------------------------ gradients
    def zero_grad(self, set_to_none: bool = True):
        """``set_to_none=True`` (torch.optim's default, what the reference's bare ``optimizer.zero_grad()`` gets, main_mage.py:150): drop the
        ``.grad`` views -- autograd then hands each gradient over by reference instead of launching one in-place add per parameter into a
        zeroed arena (149 launches per step at the MNIST config), and ``step`` gathers them into the arena with one multi-tensor copy.
        ``set_to_none=False``: zero the arena in one launch; ``.grad`` stays a view into it (autograd accumulates in place)."""
        if set_to_none:
            for p in self.params:
                p.grad = None
            return
        self.flat_g.zero_()
        for p, off in zip(self.params, self.offsets):
            view = self.flat_g[off:off + p.numel()].view(p.shape)
            if p.grad is None or p.grad.data_ptr() != view.data_ptr():
                p.grad = view

    def _collect_grads(self):
        """Bring gradients that live outside the arena (set_to_none, or a .grad autograd REPLACED) back into it: one multi-tensor copy."""
        src, dst = [], []
        for p, off in zip(self.params, self.offsets):
            view = self.flat_g[off:off + p.numel()].view(p.shape)
            if p.grad is None:
                view.zero_()
            elif p.grad.data_ptr() != view.data_ptr():
                if p.grad.dtype == view.dtype and p.grad.device == view.device:
                    src.append(p.grad.detach())
                    dst.append(view)
                else:
                    view.copy_(p.grad)
            p.grad = view
        if src:
            torch._foreach_copy_(dst, src)

    # ------------------------------------------------------------------ the update
    def _adam(self, p, g, m, v, lr, b1, b2, eps, step, grad_scale):
        """One fused launch over a flat shard (libmage_hip.so mage_adam).  No CPU path."""
        if not p.is_cuda:
            raise RuntimeError("FlatAdam runs on libmage_hip.so (mage_adam): the parameters must live on a ROCm GPU")
        ops.adam(p, g, m, v, lr=lr, beta1=b1, beta2=b2, eps=eps, step=step, grad_scale=grad_scale)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self._collect_grads()
        self.steps += 1
        g0 = self.param_groups[0]
        lr, (b1, b2), eps = g0["lr"], g0["betas"], g0["eps"]
        if self.sharded:
            # gradient: summed over ranks, each rank keeps its shard (the mean's 1/W is folded into the Adam kernel)
            dist.reduce_scatter_tensor(self.shard_g, self.flat_g, op=dist.ReduceOp.SUM, group=self.pg)
            p_shard = self.flat_p[self.shard_off:self.shard_off + self.shard_n]
            self._adam(p_shard, self.shard_g, self.m, self.v, lr, b1, b2, eps, self.steps, 1.0 / self.world)
            dist.all_gather_into_tensor(self.flat_p, p_shard.clone(), group=self.pg)
        else:
            if self.world > 1:                                   # replicated state: plain all-reduce of the arena
                dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM, group=self.pg)
            self._adam(self.flat_p, self.flat_g, self.m, self.v, lr, b1, b2, eps, self.steps, 1.0 / self.world)
        _vq.bump_weights_epoch()                                 # the kernels' derived weight copies are stale now
        return loss

    # ------------------------------------------------------------------ checkpoints (main_mage.py:189-199)
    def consolidate_state_dict(self, to: int = 0):
        """COLLECTIVE (every rank calls it, like torch's ZeroRedundancyOptimizer.consolidate_state_dict): gather the sharded moments so that
        rank `to` can build the checkpoint.  The reference saves from inside its rank-0 branch (main_mage.py:186-193); a sharded optimizer
        cannot gather there without hanging the other ranks in RCCL, so the loop calls this on all ranks first:

            optimizer.consolidate_state_dict()          # all ranks
            if rank == 0: torch.save({... 'optimizer': optimizer.state_dict()}, path)

        A no-op when the state is not sharded (one rank, or shard=False)."""
        if not self.sharded or self.world == 1:
            return
        m = torch.empty(self.n_pad, device=self.m.device, dtype=torch.float32)
        v = torch.empty(self.n_pad, device=self.m.device, dtype=torch.float32)
        dist.all_gather_into_tensor(m, self.m, group=self.pg)
        dist.all_gather_into_tensor(v, self.v, group=self.pg)
        self._consolidated = (self.steps, m[:self.n], v[:self.n]) if self.rank == to else None

    def _full_moments(self):
        if not self.sharded or self.world == 1:
            return self.m[:self.n].clone(), self.v[:self.n].clone()
        if self._consolidated is None or self._consolidated[0] != self.steps:
            raise RuntimeError("FlatAdam.state_dict(): the optimizer state is sharded over the ranks; call optimizer.consolidate_state_dict() on "
                               "EVERY rank first (a collective), then state_dict() on the rank that saves -- state_dict() itself is local")
        return self._consolidated[1], self._consolidated[2]

    def state_dict(self):
        """torch.optim.Adam's layout, numbered over ALL parameters the optimizer was given (frozen ones included, as the reference's
        optim.Adam(model.parameters()) numbers them): state[i] = {step, exp_avg, exp_avg_sq} for every TRAINABLE parameter i, and
        param_groups[0]['params'] = range(len(all parameters)) -- so the 'optimizer' entry of a checkpoint is interchangeable with the
        reference's (main_mage.py:121,189-199,210-228) and independent of the world size it was written with.  LOCAL: with sharded state
        call consolidate_state_dict() on every rank first."""
        m, v = self._full_moments()
        state = {}
        for i, p, off in zip(self.index, self.params, self.offsets):
            k = p.numel()
            state[i] = {"step": torch.tensor(float(self.steps)), "exp_avg": m[off:off + k].view(p.shape).clone(),
                        "exp_avg_sq": v[off:off + k].view(p.shape).clone()}
        # 'flat_adam_numbering': an explicit marker of the index space; torch.optim.Adam ignores unknown group keys on load
        groups = [dict({k: v_ for k, v_ in g.items() if k != "params"}, params=list(range(len(self.all_params))), flat_adam_numbering="all")
                  for g in self.param_groups]
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        """Accepts torch.optim.Adam's per-parameter layout (written by this class at ANY world size, or by the reference's optim.Adam over
        model.parameters()): entries are matched by their index in the full parameter list (a checkpoint numbered over the TRAINABLE
        parameters only -- this class's layout before it counted frozen ones -- is recognised by its length and re-indexed); a parameter without an entry (one that never
        received a gradient, e.g. ln_q / ln_kv under find_unused_parameters=True) starts from zero moments; an entry for a frozen parameter
        or with a different shape is an error, never a silent reassignment."""
        st = {int(k): e for k, e in sd["state"].items()}
        groups = sd.get("param_groups") or []
        n_ckpt = len(groups[0]["params"]) if groups and "params" in groups[0] else None
        numbering = groups[0].get("flat_adam_numbering") if groups else None
        if numbering not in (None, "all", "trainable"):
            raise ValueError(f"FlatAdam.load_state_dict: unknown flat_adam_numbering {numbering!r}")
        legacy = numbering == "trainable" or (numbering is None and n_ckpt is not None and n_ckpt == len(self.index) and n_ckpt != len(self.all_params))
        if legacy:
            # the layout this class wrote before it numbered frozen parameters too: entry j belongs to the j-th TRAINABLE parameter.  Recognised
            # by the marker, or (checkpoints older than the marker) by its length -- then EVERY entry must be in range and match its
            # parameter's shape, so that a foreign checkpoint of the same length is refused instead of remapped
            bad = [j for j in st if not 0 <= j < len(self.index)]
            if bad:
                raise ValueError(f"FlatAdam.load_state_dict: state entries {bad[:5]} are outside the {len(self.index)} trainable parameters of the "
                                 f"legacy (trainable-only) numbering this checkpoint appears to use")
            for j, e in st.items():
                pj = self.params[j]
                if tuple(e["exp_avg"].shape) != tuple(pj.shape):
                    raise ValueError(f"FlatAdam.load_state_dict: legacy-numbered state {j} has shape {tuple(e['exp_avg'].shape)}, trainable parameter {j} "
                                     f"{tuple(pj.shape)}: not a checkpoint of this parameter list")
            st = {self.index[j]: e for j, e in st.items()}
        elif n_ckpt is not None and n_ckpt != len(self.all_params):
            raise ValueError(f"FlatAdam.load_state_dict: the checkpoint's optimizer was built over {n_ckpt} parameters, "
                             f"this one over {len(self.all_params)} (pass the same model.parameters())")
        trainable = set(self.index)
        for i in st:
            if i not in trainable:
                raise ValueError(f"FlatAdam.load_state_dict: state entry {i} belongs to a parameter that is frozen (or absent) here")
        m = torch.zeros(self.n_pad, device=self.m.device, dtype=torch.float32)
        v = torch.zeros(self.n_pad, device=self.m.device, dtype=torch.float32)
        steps = 0
        for i, p, off in zip(self.index, self.params, self.offsets):
            e = st.get(i)
            if e is None:
                continue
            if tuple(e["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"FlatAdam.load_state_dict: state {i} has shape {tuple(e['exp_avg'].shape)}, parameter {tuple(p.shape)}")
            k = p.numel()
            m[off:off + k].copy_(e["exp_avg"].reshape(-1))
            v[off:off + k].copy_(e["exp_avg_sq"].reshape(-1))
            steps = max(steps, int(float(e["step"])))
        self.steps = steps
        self._consolidated = None
        self.m.copy_(m[self.shard_off:self.shard_off + self.shard_n])
        self.v.copy_(v[self.shard_off:self.shard_off + self.shard_n])
        for g, sg in zip(self.param_groups, groups):
            g.update({k: v_ for k, v_ in sg.items() if k not in ("params", "flat_adam_numbering")})
