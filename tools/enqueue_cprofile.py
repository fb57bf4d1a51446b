"""Where does the host spend its time while it enqueues one incremental autoregressive_generate call (cfg2, B = 64)?  cProfile by internal time.
Tuning only."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mage_amd.utils import synth
from mage_amd.utils.util import instantiate_from_config
dev = torch.device("cuda", 0)
B, L = (int(sys.argv[1]) if len(sys.argv) > 1 else 64), 16
m = instantiate_from_config(synth.mnist_model_config(frames_length=L)).eval()
synth.fill_state_dict(m, 0)
m = m.to(dev).set_precision("bf16")
m.use_graph = False
batch = {k: v.to(dev) for k, v in synth.synth_batch_mnist(B, L, seed=100).items()}
m.ar_mode = "incremental"
for _ in range(3): m.autoregressive_generate(batch)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5): m.autoregressive_generate(batch)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
