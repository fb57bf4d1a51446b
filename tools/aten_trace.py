import os, sys, collections, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mage_amd.utils import synth
from mage_amd.utils.util import instantiate_from_config
dev = "cuda:0"
m = instantiate_from_config(synth.mnist_model_config(frames_length=16)).eval()
synth.fill_state_dict(m, 0)
m = m.to(dev).set_precision("bf16"); m.ar_mode = "incremental"; m.use_graph = False
b = {k: v.to(dev) for k, v in synth.synth_batch_mnist(1, 16, seed=100).items()}
for _ in range(3): m.autoregressive_generate(b)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    m.autoregressive_generate(b)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.name in ("aten::copy_", "aten::to", "aten::_to_copy", "aten::clone", "aten::contiguous", "aten::cat", "aten::fill_", "aten::zero_", "aten::zeros", "aten::index", "aten::flip", "aten::sum", "aten::add", "aten::mul", "aten::repeat", "aten::index_put_", "aten::eq", "aten::lt", "aten::cumsum", "aten::ones", "aten::any", "aten::max", "aten::item", "aten::_local_scalar_dense"):
        st = [s for s in (ev.stack or []) if "mage_amd" in s or "bench" in s]
        cnt[(ev.name, st[0].split("/root/repo/")[-1].split(os.environ.get("GRAFT_REPO_ROOT","/x")+"/")[-1] if st else "?")] += 1
for (n, s), c in cnt.most_common(45):
    print(f"{c:4d}  {n:28s} {s}")
