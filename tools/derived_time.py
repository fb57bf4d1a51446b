import sys, time, torch
sys.path.insert(0, "/root/repo")
from mage_amd.utils import synth
from mage_amd.utils.util import instantiate_from_config
from mage_amd.modules import vqvae_model as V
m = instantiate_from_config(synth.mnist_model_config(frames_length=16))
synth.fill_state_dict(m, 0)
m = m.to("cuda:0").set_precision("bf16").train()
mods = [(n, mod) for n, mod in m.named_modules() if hasattr(mod, "_derived")] + [("MAGE", m)]
seen = set()
for rep in range(3):
    V.bump_weights_epoch()
    for n, mod in mods:
        if id(mod) in seen and rep == 0: continue
        seen.add(id(mod))
        b = getattr(mod, "_build", None) or getattr(mod, "_build_weights", None)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        try:
            if hasattr(mod, "_weights") and not hasattr(mod, "_build"):
                mod._weights()
            else:
                mod._derived.get(mod._build)
        except Exception as e:
            print(n, "ERR", e); continue
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"rep {rep} {n or 'MAGE'}: host {1e3*(t1-t0):.2f} ms, +gpu drain {1e3*(t2-t1):.2f} ms")
