"""Per-workgroup phase timestamps (s_memtime) of one GEMM launch of a -DMAGE_PROBE build: K loop / epilogue issue /
first-slab wait / barrier, per tile.  Tuning only.  usage: MAGE_HIP_LIB=<probe .so> python tools/gemm_phase_probe.py N,K,act,res"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mage_amd import ops, _lib
N, K, act, res = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1536,512,0,0").split(","))
M = 262144
dev = "cuda:0"
a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16(); b = torch.randn(N, device=dev)
# res: 0 = plain, 1 = x + Linear(.) on the fp32 stream, 2 = on the bf16 stream with the LayerNorm partial sums (the bf16 mode's producer)
y = torch.randn(M, N, device=dev) if res == 1 else torch.randn(M, N, device=dev).bfloat16()
kw = dict(M=M, N=N, K=K, lda=K, ldy=N, bias=b, act=act)
if res: kw.update(residual=y, ldr=N)
if res == 2: kw.update(ln_part=torch.empty(N // 64, M, 2, device=dev))
for _ in range(3): ops.gemm(a, w, y, **kw)
torch.cuda.synchronize()
l = _lib.load()
buf = np.zeros(256 * 64 * 8, np.uint64)
l.mage_debug_read.argtypes = [C.c_void_p, C.c_size_t]
assert l.mage_debug_read(buf.ctypes.data, buf.nbytes) == 0
t8 = buf.reshape(256, 64, 8).astype(np.int64)
t, rt = t8[:, :, :4], t8[:, :, 4:]
ntile = int(((M // 256) * (N // 256)) // 256)
t = t[:, :ntile]; rt = rt[:, :ntile]
# segments per tile i (i >= 1): K loop = t[i,0]-t[i,3]; epilogue issue = t[i,1]-t[i,0]; wait = t[i+1,2]-t[i,1]; barrier = t[i+1,3]-t[i+1,2]
kl = (t[:, 1:, 0] - t[:, 1:, 3]); ep = (t[:, :, 1] - t[:, :, 0]); wt = (t[:, 1:, 2] - t[:, :-1, 1]); br = (t[:, 1:, 3] - t[:, 1:, 2])
def st(x): return f"mean {x.mean():8.0f}  p10 {np.percentile(x, 10):8.0f}  p90 {np.percentile(x, 90):8.0f}"
print(f"N={N} K={K} act={act} res={res}: {ntile} tiles per workgroup; s_memtime ticks")
print("K loop (after barrier -> last MFMA issued) ", st(kl))
print("epilogue (compute + store issue)           ", st(ep))
print("first-slab vmcnt(0) wait (store acks, DMA) ", st(wt))
print("first-slab barrier wait                    ", st(br))
tot = (t[:, -1, 1] - t[:, 0, 3])
print(f"shader clock over the K loops (s_memtime ticks per s_memrealtime 10 ns): {100.0 * kl.sum() / (rt[:, 1:, 0] - rt[:, 1:, 3]).sum():.0f} MHz; "
      f"over whole tile periods: {100.0 * (t[:, -1, 1] - t[:, 0, 3]).sum() / (rt[:, -1, 1] - rt[:, 0, 3]).sum():.0f} MHz")
print("per-tile period                            ", f"{(tot / (ntile - 1)).mean():.0f}", " kernel span", int(t[:, :, 1].max() - t[:, :, 3].min()))
# phase spread across workgroups: where in its period each workgroup's epilogue k starts, relative to workgroup 0
k = ntile // 2
print(f"epilogue start of tile {k} across workgroups: spread p10..p90 = {np.percentile(t[:, k, 0], 90) - np.percentile(t[:, k, 0], 10):.0f} ticks")
# de-phasing check on the chip-wide 100 MHz clock: epilogue start per start group (li & 3, li = blockIdx >> 3) relative to
# group 0 (10 ns ticks), and how many workgroups are inside their epilogue at the moment group 0's median workgroup starts its own
grp = (np.arange(256) >> 3) & 3
for k in (1, ntile // 2, ntile - 2):
    base = np.median(rt[grp == 0, k, 0])
    inside = int(((rt[:, k, 0] <= base) & (rt[:, k + 1, 3] > base)).sum())
    print(f"tile {k}: epilogue start by group (10 ns ticks rel. group 0):", [int(np.median(rt[grp == j, k, 0]) - base) for j in range(4)],
          f" epilogue..barrier (10 ns): {int((rt[:, k + 1, 3] - rt[:, k, 0]).mean())}  period {int((rt[:, k + 1, 0] - rt[:, k, 0]).mean())}"
          f"  workgroups in epilogue at that instant: {inside}")
# per-wave epilogue begin/end inside one workgroup (shader clock of that CU), tiles 2..: relative to the earliest wave's begin
wb = np.zeros(256 * 16 * 8 * 2, np.uint64)
l.mage_debug_read_waves.argtypes = [C.c_void_p, C.c_size_t]
assert l.mage_debug_read_waves(wb.ctypes.data, wb.nbytes) == 0
w = wb.reshape(256, 16, 8, 2).astype(np.int64)[:, 2:min(ntile, 16)]
rel = w - w[:, :, :, 0].min(axis=2)[:, :, None, None]
print("per-wave epilogue begin (mean ticks after the first wave's begin):", [int(rel[:, :, i, 0].mean()) for i in range(8)])
print("per-wave epilogue end                                            :", [int(rel[:, :, i, 1].mean()) for i in range(8)])
# 8-phase kernel, workgroup 8, third tile (shader clock), per wave and slab: per phase  load / wait-A / mfma / wait-B  clocks
sb = np.zeros(8 * 160, np.uint64)
l.mage_debug_read_seg.argtypes = [C.c_void_p, C.c_size_t]
if l.mage_debug_read_seg(sb.ctypes.data, sb.nbytes) == 0 and sb.any():
    sg = sb.reshape(8, 160).astype(np.int64)
    ns = min((K + 63) // 64, 9)
    for wv in (0, 4):
        for k in range(1, min(ns - 1, 5)):
            parts = []
            for ph in range(4):
                st = sg[wv, (k * 4 + ph) * 4: (k * 4 + ph) * 4 + 5]
                parts.append(f"p{ph + 1} {st[1] - st[0]}/{st[2] - st[1]}/{st[3] - st[2]}/{st[4] - st[3]}")
            print(f"wave {wv} slab {k}: " + "   ".join(parts))
