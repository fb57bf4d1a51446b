import sys, torch
sys.path.insert(0, "/root/repo")
from mage_amd import ops
from mage_amd.modules.vqvae_model import VectorQuantizedVAE as V
torch.manual_seed(0)
dev = "cuda:0"
for dt in (torch.float32, torch.bfloat16):
    for (N, H, W, cin, cout) in ((2, 8, 8, 16, 32), (3, 16, 16, 64, 256), (2, 32, 32, 64, 64)):
        x = torch.randn(N * H * W, cin, device=dev).to(dt)
        w = (torch.randn(cout, 9 * cin, device=dev) * 0.05).to(dt)
        b = torch.randn(cout, device=dev)
        r_low = torch.randn(N * (H // 2) * (W // 2), cout, device=dev).to(dt)
        y = torch.empty(N * H * W, cout, device=dev, dtype=dt)
        V._conv(x, w, y, n_img=N, H=H, W=W, cin=cin, cout=cout, k=3, bias=b, residual=r_low, ldr=cout, post_relu=True, res_half=True)
        r_up = torch.empty(N * H * W, cout, device=dev, dtype=dt)
        ops.upsample2(r_low, r_up, N=N, H=H // 2, W=W // 2, Cc=cout)
        y2 = torch.empty_like(y)
        V._conv(x, w, y2, n_img=N, H=H, W=W, cin=cin, cout=cout, k=3, bias=b, residual=r_up, ldr=cout, post_relu=True)
        print(dt, (N, H, W, cin, cout), "max diff", (y.float() - y2.float()).abs().max().item(), "equal", torch.equal(y, y2))
