#!/usr/bin/env python
"""Fixed cost of a GEMM launch at the incremental step's size (M = 16384 rows): time vs K for the epilogue kinds (GPU box)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mage_amd import ops

DEV = "cuda:0"
M, N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 512


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for K in (64, 128, 256, 512, 1024, 2048):
    a = torch.randn(M, K, device=DEV).bfloat16()
    w = torch.randn(N, K, device=DEV).bfloat16()
    bias = torch.randn(N, device=DEV)
    x = torch.randn(M, N, device=DEV)
    y16 = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    xb = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    part = torch.empty(N // 64, M, 2, device=DEV)
    t_plain = timeit(lambda: ops.gemm(a, w, y16, M=M, N=N, K=K, lda=K, ldy=N, bias=bias))
    t_res = timeit(lambda: ops.gemm(a, w, x, M=M, N=N, K=K, lda=K, ldy=N, bias=bias, residual=x, ldr=N))
    t_ln = timeit(lambda: ops.gemm(a, w, x, M=M, N=N, K=K, lda=K, ldy=N, bias=bias, residual=x, ldr=N, y2=xb, ldy2=N, ln_part=part))
    print(f"M={M} N={N} K={K:5d}: bias->bf16 {t_plain:6.1f} us   x+Linear fp32 {t_res:6.1f} us   + bf16 copy & LN partials {t_ln:6.1f} us", flush=True)
e = torch.empty(1 << 20, device=DEV)
print(f"empty-ish elementwise launch (4 MB add_): {timeit(lambda: e.add_(1.0)):.1f} us")
