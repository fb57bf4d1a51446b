"""Race screen of gemm4h_kernel beyond the unit test: N launches of the c_fc / QKV forms at full-loop and incremental sizes while a second stream
keeps the HBM busy with copies of varying length; every output compared bit for bit with gemm4_kernel's / the 8-wave kernels' (option gemm_no_4h)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mage_amd import config, ops
DEV = "cuda:0"
n_rep = int(sys.argv[1]) if len(sys.argv) > 1 else 40
g = torch.Generator(device=DEV).manual_seed(0)
def rn(*s, dtype=torch.float32, scale=1.0):
    return (torch.randn(*s, device=DEV, generator=g) * scale).to(dtype)
side = torch.cuda.Stream()
big, big2 = torch.empty(1 << 28, device=DEV, dtype=torch.uint8), torch.empty(1 << 28, device=DEV, dtype=torch.uint8)
bad_total = 0
for dt in (torch.bfloat16, torch.float16):
    for M, N, act in ((262144, 2048, ops.ACT_QUICKGELU), (16384, 2048, ops.ACT_QUICKGELU), (16384, 1536, 0), (8192, 1536, 0), (65792, 2048, ops.ACT_QUICKGELU)):
        K = 512
        a, w, b = rn(M, K, dtype=dt), rn(N, K, dtype=dt, scale=K ** -0.5), rn(N, scale=0.1)
        st = torch.stack([0.05 * rn(M), 1.0 + 0.2 * rn(M).abs()], 1).contiguous()
        cs = 0.3 * rn(N)
        kw = dict(M=M, N=N, K=K, lda=K, ldy=N, bias=b, ln_stats=st, ln_colsum=cs, act=act)
        ref = torch.empty(M, N, device=DEV, dtype=dt)
        with config.lib_option("gemm_no_4h", 1):
            ops.gemm(a, w, ref, **kw)
        torch.cuda.synchronize()
        outs = [torch.empty(M, N, device=DEV, dtype=dt) for _ in range(4)]
        bad = 0
        with config.lib_option("gemm_4h_plain", 1):
            for rep in range(n_rep):
                with torch.cuda.stream(side):
                    for _ in range(rep % 7):
                        big2.copy_(big, non_blocking=True)
                for y in outs:
                    y.fill_(float("nan"))
                    ops.gemm(a, w, y, **kw)
                torch.cuda.synchronize()
                bad += sum(int(not torch.equal(y, ref)) for y in outs)
        bad_total += bad
        print(f"{str(dt)[6:]:9s} M={M:6d} N={N} act={act}: {bad} of {4 * n_rep} launches differ")
print("TOTAL mismatching launches:", bad_total)
sys.exit(1 if bad_total else 0)
