import sys, time, torch
sys.path.insert(0, "/root/repo")
from mage_amd.optim import FlatAdam
from mage_amd.utils import glue, synth
from mage_amd.utils.util import instantiate_from_config
B, L = 64, 16
model = instantiate_from_config(synth.mnist_model_config(frames_length=L))
synth.fill_state_dict(model, 0)
model = model.to("cuda:0").set_precision("bf16").train()
opt = FlatAdam(model.parameters(), lr=1e-4, betas=(0.9, 0.98), eps=1e-6)
batches = [{k: v.to("cuda:0") for k, v in synth.synth_batch_mnist(B, L, seed=100 + i).items()} for i in range(2)]
for it in range(8):
    batch = batches[it % 2]
    torch.cuda.synchronize(); st0 = torch.cuda.memory_stats()
    t0 = time.perf_counter()
    opt.zero_grad()
    loss, ld = model(batch)
    t1 = time.perf_counter()
    loss.backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    opt.step(); torch.cuda.synchronize()
    st1 = torch.cuda.memory_stats()
    print(f"iter {it}: fwd {1e3*(t1-t0):.1f} bwd {1e3*(t2-t1):.1f} ms; new segments {st1['segment.all.allocated']-st0['segment.all.allocated']}, freed segments {st1['segment.all.freed']-st0['segment.all.freed']}, "
          f"retries {st1['num_alloc_retries']-st0['num_alloc_retries']}, reserved {st1['reserved_bytes.all.current']/2**30:.1f} GiB", flush=True)
