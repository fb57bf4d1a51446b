"""x + Linear(.) producer GEMMs (bf16 residual stream + LayerNorm partial sums: out_proj K = 512, c_proj K = 2048) and QKV (gemm4_kernel) at cfg2's
full-loop size under the staggered-start options of the persistent kernels: interleaved rounds, median / min / max per setting (tuning only;
profiles/r06_producer_stagger.txt).  Note: since round 6 the library applies the 8-wave kernels' stagger only up to 16 K slabs, so the c_proj rows
show the option's effect only when it is forced through MAGE_GEMM_STAGGER."""
import contextlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mage_amd import config, ops
DEV, M = "cuda:0", 262144
g = torch.Generator(device=DEV).manual_seed(0)
rn = lambda *s, dtype=torch.float32, scale=1.0: (torch.randn(*s, device=DEV, generator=g) * scale).to(dtype)


def timed(f, opts):
    with contextlib.ExitStack() as es:
        for k, v in opts.items():
            es.enter_context(config.lib_option(k, v))
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            e0.record()
            for _ in range(6):
                f()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 6)
    return best * 1e3


def producer(N, K):
    a, w, b = rn(M, K, dtype=torch.bfloat16), rn(N, K, dtype=torch.bfloat16, scale=K ** -0.5), rn(N, scale=0.1)
    x, part = rn(M, N, dtype=torch.bfloat16), torch.empty(N // 64, M, 2, device=DEV)
    return lambda: ops.gemm(a, w, x, M=M, N=N, K=K, lda=K, ldy=N, bias=b, residual=x, ldr=N, ln_part=part)


def qkv():
    N, K = 1536, 512
    a, w, b = rn(M, K, dtype=torch.bfloat16), rn(N, K, dtype=torch.bfloat16, scale=K ** -0.5), rn(N, scale=0.1)
    st, cs = torch.stack([0.05 * rn(M), 1.0 + 0.2 * rn(M).abs()], 1).contiguous(), 0.3 * rn(N)
    y = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    return lambda: ops.gemm(a, w, y, M=M, N=N, K=K, lda=K, ldy=N, bias=b, ln_stats=st, ln_colsum=cs)


def sweep(name, f, combos, rounds=8):
    res = {str(o): [] for o in combos}
    for _ in range(rounds):
        for o in combos:
            res[str(o)].append(timed(f, o))
    for k, v in sorted(res.items(), key=lambda kv: sorted(kv[1])[len(kv[1]) // 2]):
        print(f"{name:10s} {k:62s} median {sorted(v)[len(v) // 2]:7.1f} min {min(v):7.1f} max {max(v):7.1f}")


S8 = [{"gemm_stagger_groups": 8, "gemm_stagger_percent": 60}, {"gemm_stagger_groups": 8, "gemm_stagger_percent": 30},
      {"gemm_stagger_groups": 4, "gemm_stagger_percent": 30}, {"gemm_stagger_groups": 8, "gemm_stagger_percent": 40}, {"gemm_stagger_groups": 0}]
sweep("c_proj", producer(512, 2048), S8)
sweep("out_proj", producer(512, 512), S8)
sweep("qkv(gemm4)", qkv(), [{"gemm4_stagger_groups": 8, "gemm4_stagger_percent": 100}, {"gemm4_stagger_groups": 0}, {"gemm4_stagger_groups": 8, "gemm4_stagger_percent": 50},
                            {"gemm4_stagger_groups": 4, "gemm4_stagger_percent": 100}, {"gemm4_stagger_groups": 16, "gemm4_stagger_percent": 100}])
