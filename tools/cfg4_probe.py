import sys, time, torch
sys.path.insert(0, '/root/repo')
from mage_amd.utils import synth
from mage_amd.utils.util import instantiate_from_config
B, L = int(sys.argv[1]), int(sys.argv[2])
cfg = synth.cater_model_config(frames_length=L)
m = instantiate_from_config(cfg).eval()
synth.fill_state_dict(m, 0)
m = m.to('cuda:0').set_precision('bf16')
batch = {k: v.to('cuda:0') for k, v in synth.synth_batch_cater(B, L, seed=1).items()}
batch['video_noise'] = torch.randn(B, 64, 16, 16, generator=torch.Generator().manual_seed(5)).to('cuda:0')   # same ADAIN noise in both modes
for mode in ('incremental', 'full'):
    m.ar_mode = mode
    v = m.autoregressive_generate(batch); torch.cuda.synchronize()
    t0 = time.perf_counter(); v = m.autoregressive_generate(batch); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(mode, tuple(v.shape), f"{dt*1e3:.1f} ms  {B*L/dt:.1f} frames/s  max|v| {v.abs().max().item():.3f} finite {torch.isfinite(v).all().item()}  mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
    tok = m.last_tokens.clone() if mode == 'incremental' else tok
print("tokens identical across modes:", torch.equal(tok, m.last_tokens))
