"""Fused MLP kernel vs the two-GEMM path (+LayerNorm excluded), decoder shape.  Tuning harness."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mage_amd import ops
M, Cc = int(sys.argv[1]) if len(sys.argv) > 1 else 262144, 512
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s, sc=1.0: torch.randn(*s, device=dev, generator=g) * sc
xn = rn(M, Cc).bfloat16(); w1 = rn(4 * Cc, Cc, sc=(2 * Cc) ** -0.5).bfloat16(); w2 = rn(Cc, 4 * Cc, sc=(4 * Cc) ** -0.5).bfloat16()
b1, b2, x = rn(4 * Cc, sc=0.1), rn(Cc, sc=0.1), rn(M, Cc)
h = torch.empty(M, 4 * Cc, device=dev, dtype=torch.bfloat16)
def unfused():
    ops.gemm(xn, w1, h, M=M, N=4 * Cc, K=Cc, lda=Cc, ldy=4 * Cc, bias=b1, act=ops.ACT_QUICKGELU)
    ops.gemm(h, w2, x, M=M, N=Cc, K=4 * Cc, lda=4 * Cc, ldy=Cc, bias=b2, residual=x, ldr=Cc)
def fused():
    ops.mlp_fused(xn, w1, b1, w2, b2, x)
for f in (unfused, fused): f()
torch.cuda.synchronize()
for name, f in (("unfused c_fc + c_proj", unfused), ("fused mlp", fused), ("unfused c_fc + c_proj", unfused), ("fused mlp", fused)):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{name:24s} {ms:8.3f} ms  {16.0 * M * Cc * Cc / ms / 1e9:8.1f} TFLOP/s")
