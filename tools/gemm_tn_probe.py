#!/usr/bin/env python
"""mage_gemm_tn (weight gradient from row-major operands) against the transposes + split-K route, per decoder-Linear shape (GPU box)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mage_amd import config, ops
from mage_amd.modules import mage_train as T

DEV = "cuda:0"


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


M = 262144
for N, K, tag in ((1536, 512, "in_proj"), (512, 512, "out_proj"), (2048, 512, "c_fc"), (512, 2048, "c_proj")):
    dy = (torch.randn(M, N, device=DEV) * 0.1).bfloat16()
    x = torch.randn(M, K, device=DEV).bfloat16()
    fl = 2.0 * M * N * K
    t_tn = timeit(lambda: ops.gemm_tn(dy, x, T=M, N=N, K=K, ld_dy=N, ld_x=K, want_bias=False))
    t_tnb = timeit(lambda: ops.gemm_tn(dy, x, T=M, N=N, K=K, ld_dy=N, ld_x=K, want_bias=True))
    with config.override(train_wgrad_transpose=True):
        t_old = timeit(lambda: T._wgrad(dy, x, M=M, N=N, K=K, ld_dy=N, ld_x=K))
    print(f"{tag:9s} N={N:5d} K={K:5d}: gemm_tn {t_tn:6.3f} ms ({fl / t_tn / 1e9:6.0f} TF/s)  + colsum {t_tnb:6.3f} ms   transposes + split-K gemm8 + db {t_old:6.3f} ms", flush=True)
