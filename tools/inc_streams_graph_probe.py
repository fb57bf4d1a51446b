"""the clip-group probe with NO host in the loop.  The incremental AR call with the clips as n groups on n HIP streams
(MAGE.streams), captured as ONE HIP graph with fork / join edges (MAGE.use_graph) and replayed: do independent groups' dependency chains fill each
other's dispatch latency and ramps at the step's sizes?  (Round 5 measured this eager: the time followed the host's enqueue cost.)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mage_amd.utils import synth
from mage_amd.utils.util import instantiate_from_config

dev = "cuda:0"
for name, cfg, mk, B, L in (("cfg2", synth.mnist_model_config(frames_length=16), synth.synth_batch_mnist, 64, 16),
                            ("cfg4", synth.cater_model_config(frames_length=32), synth.synth_batch_cater, 32, 32)):
    m = instantiate_from_config(cfg).eval()
    synth.fill_state_dict(m, 0)
    m = m.to(dev).set_precision("bf16")
    b = mk(B, L, seed=3)
    if name == "cfg4":
        b["video_noise"] = torch.randn(B, 64, 16, 16, generator=torch.Generator().manual_seed(5))
    batch = {k: v.to(dev) for k, v in b.items()}
    m.ar_mode = "incremental"
    ref = None
    for n in (1, 2, 4):
        for graph in (False, True):
            m.streams, m.use_graph, m.graph_multistream = n, graph, True
            try:
                for _ in range(3):
                    m.autoregressive_generate(batch)                # eager, capture, first replay
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(5): m.autoregressive_generate(batch)
                torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
                if ref is None: ref = m.last_tokens.clone()
                print(f"{name} incremental bf16 B={B}: groups={n} graph={int(graph)}: {ms:7.2f} ms per call, tokens identical to one stream: {torch.equal(ref, m.last_tokens)}, mode {m.last_call_mode}", flush=True)
            except Exception as e:
                print(f"{name} groups={n} graph={int(graph)}: {type(e).__name__}: {str(e)[:200]}", flush=True)
    del m
    torch.cuda.empty_cache()
