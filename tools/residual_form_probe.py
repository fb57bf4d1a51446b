"""How much of the x + Linear(.) producer's time is the residual tile's HBM fetch?  The same GEMM (bf16 stream form: residual bf16 rows in,
bf16 rows + LayerNorm partial sums out) with the residual read (a) in place from the stream, (b) from one cache-resident row (ldr = 0):
an upper bound on what hiding the fetch could buy.  usage: res_prefetch_probe.py [M=262144]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mage_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
for N, K in ((512, 512), (512, 2048)):
    a = torch.randn(M, K, device=dev, generator=g).bfloat16()
    w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).bfloat16()
    b = torch.randn(N, device=dev, generator=g)
    x = torch.randn(M, N, device=dev, generator=g).bfloat16()
    y = torch.empty_like(x)
    part = torch.empty(N // 64, M, 2, device=dev)
    row = torch.randn(8, N, device=dev, generator=g).bfloat16()
    def timeit(f, n=10):
        f(); f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    t_in = timeit(lambda: ops.gemm(a, w, x, M=M, N=N, K=K, lda=K, ldy=N, bias=b, residual=x, ldr=N, ln_part=part))
    t_out = timeit(lambda: ops.gemm(a, w, y, M=M, N=N, K=K, lda=K, ldy=N, bias=b, residual=x, ldr=N, ln_part=part))
    t_hot = timeit(lambda: ops.gemm(a, w, y, M=M, N=N, K=K, lda=K, ldy=N, bias=b, residual=row, ldr=0, ln_part=part))
    t_nores = timeit(lambda: ops.gemm(a, w, y, M=M, N=N, K=K, lda=K, ldy=N, bias=b))
    from mage_amd import config
    t_noln = timeit(lambda: ops.gemm(a, w, y, M=M, N=N, K=K, lda=K, ldy=N, bias=b, residual=x, ldr=N))
    with config.lib_option("gemm_stagger_groups", 0):
        t_nostag = timeit(lambda: ops.gemm(a, w, x, M=M, N=N, K=K, lda=K, ldy=N, bias=b, residual=x, ldr=N, ln_part=part))
    with config.lib_option("gemm_stagger_forced", 1):
        t_plain_stag = timeit(lambda: ops.gemm(a, w, y, M=M, N=N, K=K, lda=K, ldy=N, bias=b))
    x32 = x.float()
    t_f32res = timeit(lambda: ops.gemm(a, w, x32, M=M, N=N, K=K, lda=K, ldy=N, bias=b, residual=x32, ldr=N))
    print(f"   residual, no partial sums {t_noln:7.1f} | no stagger {t_nostag:7.1f} | plain with stagger {t_plain_stag:7.1f} | fp32 residual in place, fp32 out {t_f32res:7.1f}")
    fl = 2.0 * M * N * K
    print(f"N={N} K={K}: in place {t_in:7.1f} us ({fl / t_in / 1e6:6.0f} TF/s) | out of place {t_out:7.1f} | residual from one hot row {t_hot:7.1f} | "
          f"no residual, no partial sums {t_nores:7.1f}")
