"""Per-shape timing of the decoder-stack GEMMs (HIP events, interleaved rounds) -- kernel tuning harness."""
import os, sys, argparse
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mage_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=262144)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--check", action="store_true")
ap.add_argument("--vendor", action="store_true", help="also time torch.nn.functional.linear (hipBLASLt / rocBLAS) on the same operands: a bench-only comparator, never used by mage_amd")
ap.add_argument("--custom", nargs="*", default=None, help="N,K,act,res,outf32 ...")
args = ap.parse_args()
dev = "cuda:0"
dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
M = args.M
g = torch.Generator(device=dev).manual_seed(0)
def rn(*s, dtype=torch.float32, scale=1.0):
    return (torch.randn(*s, device=dev, generator=g) * scale).to(dtype)

shapes = [("qkv   N1536 K512  bias->bf16", 1536, 512, dict()),
          ("proj  N512  K512  +res(f32)", 512, 512, dict(res=True)),
          ("fc1   N2048 K512  gelu->bf16", 2048, 512, dict(act=ops.ACT_QUICKGELU)),
          ("fc2   N512  K2048 +res(f32)", 512, 2048, dict(res=True))]
if args.custom:
    shapes = []
    for c in args.custom:
        N, K, act, res, of32 = (int(v) for v in c.split(","))
        shapes.append((f"N{N} K{K} act{act} res{res} f32out{of32}", N, K, dict(act=act, res=bool(res), of32=bool(of32))))
bufs = {}
for name, N, K, o in shapes:
    a, w, b = rn(M, K, dtype=dt), rn(N, K, dtype=dt, scale=K ** -0.5), rn(N)
    y = rn(M, N) if (o.get("res") or o.get("of32")) else torch.empty(M, N, device=dev, dtype=dt)
    bufs[name] = (a, w, b, y)

def run(name, N, K, o):
    a, w, b, y = bufs[name]
    kw = dict(M=M, N=N, K=K, lda=K, ldy=N, bias=b, act=o.get("act", 0))
    if o.get("res"):
        kw.update(residual=y, ldr=N)
    ops.gemm(a, w, y, **kw)

if args.check:
    for name, N, K, o in shapes:
        a, w, b, y = bufs[name]
        y0 = y.clone()
        run(name, N, K, o)
        ref = a[:512].float() @ w.float().t() + b
        if o.get("act"): ref = ref * torch.sigmoid(1.702 * ref)
        if o.get("res"): ref = ref + y0[:512]
        print(name, "max err", (y[:512].float() - ref).abs().max().item())
for s in shapes: run(*s)
torch.cuda.synchronize()
tot = {s[0]: 0.0 for s in shapes}
for r in range(args.rounds):
    for s in shapes:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(*s); run(*s); e1.record()
        torch.cuda.synchronize()
        tot[s[0]] += e0.elapsed_time(e1) / 2
vend = {}
if args.vendor and args.dtype == "bf16":
    import torch.nn.functional as F
    for name, N, K, o in shapes:
        a, w, b, y = bufs[name]
        bb = b.to(torch.bfloat16)
        for _ in range(2): F.linear(a, w, bb)
        torch.cuda.synchronize()
        t = 0.0
        for r in range(args.rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); F.linear(a, w, bb); F.linear(a, w, bb); e1.record()
            torch.cuda.synchronize()
            t += e0.elapsed_time(e1) / 2
        vend[name] = t / args.rounds
allms, allfl = 0.0, 0.0
for name, N, K, o in shapes:
    ms = tot[name] / args.rounds
    fl = 2.0 * M * N * K
    allms += ms; allfl += fl
    extra = f"   | vendor F.linear(bias, bf16 out, no activation / residual) {vend[name]:8.3f} ms {fl / vend[name] / 1e9:8.1f} TFLOP/s" if name in vend else ""
    print(f"{name:32s} {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s{extra}")
print(f"{'block total (4 GEMMs)':32s} {allms:8.3f} ms  {allfl / allms / 1e9:8.1f} TFLOP/s")
