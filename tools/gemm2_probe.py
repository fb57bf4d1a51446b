"""(Runs only on a library built WITH the experiment: see the header of tools/probes/gemm2_experiment.hip.)
The two-workgroups-per-CU GEMM experiment (library option gemm_2wg) against the shipped kernels on the decoder's LayerNorm-consuming
GEMMs at full-loop size: bitwise comparison and interleaved timing.  Tuning only."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mage_amd import ops, config
M = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
OPT = sys.argv[2] if len(sys.argv) > 2 else "gemm_2wg"
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
def timeit(f, n=10):
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, N, K, act, ln in (("c_fc", 2048, 512, ops.ACT_QUICKGELU, True), ("QKV", 1536, 512, ops.ACT_NONE, True), ("plain bias", 2048, 512, ops.ACT_NONE, False),
                            ("c_fc at 16 k rows", 2048, 512, ops.ACT_QUICKGELU, True)):
    m = 16384 if "16 k" in name else M
    a = torch.randn(m, K, device=dev, generator=g).bfloat16()
    w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).bfloat16()
    b = torch.randn(N, device=dev, generator=g)
    kw = dict(M=m, N=N, K=K, lda=K, ldy=N, bias=b, act=act)
    if ln:
        stats = torch.stack([torch.randn(m, device=dev, generator=g) * 0.1, torch.rand(m, device=dev, generator=g) + 0.5], 1).contiguous()
        kw.update(ln_stats=stats, ln_colsum=w.float().sum(1).contiguous())
    y0 = torch.empty(m, N, device=dev, dtype=torch.bfloat16)
    y1 = torch.empty_like(y0)
    ops.gemm(a, w, y0, **kw)
    with config.lib_option(OPT, 1):
        ops.gemm(a, w, y1, **kw)
    same = torch.equal(y0, y1)
    res = []
    for rep in range(3):
        t0 = timeit(lambda: ops.gemm(a, w, y0, **kw))
        with config.lib_option(OPT, 1):
            t1 = timeit(lambda: ops.gemm(a, w, y1, **kw))
        res.append((t0, t1))
    fl = 2.0 * m * N * K
    print(f"{name:18s} M={m} N={N} K={K}: same bits {same} | shipped " + " ".join(f"{t0:7.1f}" for t0, _ in res) + " us | two workgroups per CU " +
          " ".join(f"{t1:7.1f}" for _, t1 in res) + f" us  ({fl / min(t0 for t0, _ in res) / 1e6:.0f} vs {fl / min(t1 for _, t1 in res) / 1e6:.0f} TFLOP/s)")
