import sys, time, torch
sys.path.insert(0, '/root/repo')
from mage_amd.utils import synth
from mage_amd.utils.util import instantiate_from_config
B, L = 32, 32
m = instantiate_from_config(synth.cater_model_config(frames_length=L)).eval()
synth.fill_state_dict(m, 0)
m = m.to('cuda:0').set_precision('bf16')
batch = {k: v.to('cuda:0') for k, v in synth.synth_batch_cater(B, L, seed=1).items()}
m.ar_mode = sys.argv[1] if len(sys.argv) > 1 else 'incremental'
for _ in range(3):
    m.autoregressive_generate(batch)
torch.cuda.synchronize()
