#!/usr/bin/env python
"""Split-precision GEMM (MAGE_BF16X3 / MAGE_F16X3) against an fp64 product: error and time per shape, beside the exact-fp32 MFMA
and the bf16 GEMM of the same shape (GPU box).

    python tools/split_probe.py > gpurun_out/split_probe.txt
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mage_amd import config, ops  # noqa: E402

DEV = "cuda:0"


def unsplit(y, kind):
    rows, c2 = y.shape
    v = y.view(rows, c2 // 128, 2, 64).float()
    lo = v[:, :, 1] / (2048.0 if kind == ops.F16X3 else 1.0)
    return (v[:, :, 0].double() + lo.double()).reshape(rows, c2 // 2)


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    g = torch.Generator(device="cpu").manual_seed(0)
    # representation check
    x = (torch.randn(256, 512, generator=g) * torch.logspace(-6, 3, 512)[None, :]).to(DEV)
    for kind, nm in ((ops.BF16X3, "bf16x3"), (ops.F16X3, "f16x3")):
        s = ops.split(x, kind)
        rel = ((unsplit(s, kind) - x.double()).abs() / x.double().abs().clamp_min(1e-30)).max().item()
        print(f"split {nm}: max relative representation error {rel:.3e}", flush=True)
    for (M, N, K, tag) in ((2048, 512, 512, "small/lockstep"), (4096, 1536, 512, "qkv small"), (262144, 1536, 512, "QKV"), (262144, 512, 512, "out_proj"),
                           (262144, 2048, 512, "c_fc"), (262144, 512, 2048, "c_proj")):
        a = torch.randn(M, K, generator=g).to(DEV)
        w = (torch.randn(N, K, generator=g) * K ** -0.5).to(DEV)
        b = torch.randn(N, generator=g).to(DEV)
        r = torch.randn(M, N, generator=g).to(DEV)
        rows = slice(0, min(M, 4096))
        want = a[rows].double() @ w.double().t() + b.double() + r[rows].double()
        fl = 2.0 * M * N * K
        # fp32 MFMA
        y = torch.empty(M, N, device=DEV)
        f = lambda: ops.gemm(a, w, y, M=M, N=N, K=K, lda=K, ldy=N, bias=b, residual=r, ldr=N)
        ms = timeit(f, 2)
        print(f"{tag:16s} M={M} N={N} K={K}  fp32     err {(y[rows].double() - want).abs().max().item():.3e}  {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TF/s", flush=True)
        ab, wb = a.bfloat16(), w.bfloat16()
        f = lambda: ops.gemm(ab, wb, y, M=M, N=N, K=K, lda=K, ldy=N, bias=b, residual=r, ldr=N)
        ms = timeit(f)
        print(f"{'':16s} {'':28s} bf16     err {(y[rows].double() - want).abs().max().item():.3e}  {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TF/s", flush=True)
        for kind, nm in ((ops.BF16X3, "bf16x3"), (ops.F16X3, "f16x3")):
            a_s, w_s = ops.split(a, kind), ops.split(w, kind)
            f = lambda: ops.gemm(a_s, w_s, y, M=M, N=N, K=K, lda=2 * K, ldy=N, bias=b, residual=r, ldr=N, split_kind=kind)
            ms = timeit(f)
            err = (y[rows].double() - want).abs().max().item()
            y2 = torch.empty_like(y)
            with config.lib_option("gemm_no_8phase", 1):
                ops.gemm(a_s, w_s, y2, M=M, N=N, K=K, lda=2 * K, ldy=N, bias=b, residual=r, ldr=N, split_kind=kind)
            print(f"{'':16s} {'':28s} {nm:8s} err {err:.3e}  {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TF/s (x3 = {3 * fl / ms / 1e9:7.1f} MFMA TF/s)", flush=True)
            # split output with QuickGELU (no residual)
            ys = ops.split_empty(M, N, kind, DEV)
            ops.gemm(a_s, w_s, ys, M=M, N=N, K=K, lda=2 * K, ldy=2 * N, bias=b, act=ops.ACT_QUICKGELU, split_kind=kind, y_split=True)
            t_ = a[rows].double() @ w.double().t() + b.double()
            wantg = t_ * torch.sigmoid(1.702 * t_)
            print(f"{'':16s} {'':28s} {nm:8s} gelu->split err {(unsplit(ys[rows], kind) - wantg).abs().max().item():.3e}", flush=True)
    # small magnitudes: are f16 denormal pieces kept by the MFMA?
    M, N, K = 512, 256, 512
    a = (torch.randn(M, K, generator=g) * 1e-6).to(DEV)
    w = torch.randn(N, K, generator=g).to(DEV)
    want = a.double() @ w.double().t()
    for kind, nm in ((ops.BF16X3, "bf16x3"), (ops.F16X3, "f16x3")):
        y = torch.empty(M, N, device=DEV)
        ops.gemm(ops.split(a, kind), ops.split(w, kind), y, M=M, N=N, K=K, lda=2 * K, ldy=N, split_kind=kind)
        print(f"tiny activations (1e-6) {nm}: err {(y.double() - want).abs().max().item():.3e} of |y| ~ {want.abs().max().item():.3e}", flush=True)


if __name__ == "__main__":
    main()
