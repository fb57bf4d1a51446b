"""Race screen for conv3x3_c64_kernel (csrc/conv_tile.hip): many tiles per workgroup, 20 repeats per shape, every run compared bitwise with the implicit-GEMM kernel.  Tuning / CI aid."""
import sys, torch
sys.path.insert(0, "/root/repo")
from mage_amd import ops as o, config
DEV = "cuda:0"
g = torch.Generator().manual_seed(1)
for (n_img, R, half) in ((200, 128, False), (400, 64, True), (120, 128, True)):
    Ri = R // 2 if half else R
    x = torch.randn(n_img * Ri * Ri, 64, generator=g).bfloat16().to(DEV)
    w = (torch.randn(64, 576, generator=g) / 24).bfloat16().to(DEV)
    b = torch.randn(64, generator=g).to(DEV)
    kw = dict(M=n_img * R * R, N=64, K=576, lda=64, ldy=64, out_h=R, out_w=R, in_h=R, in_w=R, taps_h=3, taps_w=3, cin=64, stride=1, dy0=-1, dx0=-1, bias=b, act=o.ACT_RELU)
    if half: kw.update(a_half=True, a_img_stride=Ri * Ri)
    ref = torch.empty(n_img * R * R, 64, device=DEV, dtype=torch.bfloat16)
    with config.lib_option("conv_no_tile", 1):
        o.gemm(x, w, ref, **kw)
    bad = 0
    for it in range(20):
        y = torch.full_like(ref, 7.0)
        o.gemm(x, w, y, **kw)
        bad += int(not torch.equal(y, ref))
    print(n_img, R, half, "mismatching runs of 20:", bad)
