import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from mage_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
for (M, D, K) in ((40960, 1024, 512), (262144, 256, 512), (5000, 32, 64), (777, 128, 1000), (33, 64, 300)):
    z = torch.randn(M, D, device=dev)
    cb = torch.randn(K, D, device=dev) * 0.8
    z[:K] = cb[torch.randperm(K, device=dev)[: min(K, M)]] if M >= K else z[:K]        # exact hits
    cbt, c2 = ops.vq_prepare(cb)
    res = {}
    for mode in ("mfma", "valu"):
        if mode == "valu":
            continue
        for _ in range(2):
            ids, mg = ops.vq_nearest(z, cbt, c2, want_margin=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            ids, mg = ops.vq_nearest(z, cbt, c2, want_margin=True)
        torch.cuda.synchronize(); res[mode] = (ids.clone(), mg.clone(), (time.perf_counter() - t0) / 5 * 1e3)
    # reference in float64 on the GPU via torch
    d64 = (cb.double() ** 2).sum(1)[None, :].float().double() * 0   # placeholder
    dots = z.double() @ cb.double().t()
    s = (c2[None, :] + (z.double() ** 2).sum(1).float()[:, None])          # fl32(|c|^2 + |z|^2)
    dist = torch.addcmul(s, dots.float(), torch.tensor(-2.0, device=dev)) if False else (s.double() - 2.0 * dots.float().double()).float()
    # fmaf(-2, dot32, s): single rounding of s - 2*dot32 computed exactly in double then rounded
    ref = dist.argmin(1)
    ids, mg, ms = res["mfma"]
    srt = dist.sort(1).values
    margin_ref = srt[:, 1] - srt[:, 0]
    bad = (ids != ref)
    print(f"M={M} D={D} K={K}: {ms:.3f} ms; mismatches vs fp64 formula {bad.sum().item()} (all inside margin <= {margin_ref[bad].max().item() if bad.any() else 0:.2e}); "
          f"margin max diff {(mg - margin_ref).abs().max().item():.2e}")
