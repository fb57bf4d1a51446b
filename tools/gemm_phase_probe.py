"""Per-workgroup phase timestamps (s_memtime) of one GEMM launch: main loop vs epilogue, overlap per CU. Tuning only."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mage_amd import ops, _lib
N, K, act, res = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1536,512,0,0").split(","))
M = 262144 // 4
dev = "cuda:0"
a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16(); b = torch.randn(N, device=dev)
y = torch.randn(M, N, device=dev) if res else torch.empty(M, N, device=dev, dtype=torch.bfloat16)
kw = dict(M=M, N=N, K=K, lda=K, ldy=N, bias=b, act=act)
if res: kw.update(residual=y, ldr=N)
for _ in range(2): ops.gemm(a, w, y, **kw)
torch.cuda.synchronize()
l = _lib.load()
nblk = (M // 128) * (N // 128)
buf = np.zeros(8 * 65536, np.uint64)
l.mage_debug_read.argtypes = [C.c_void_p, C.c_size_t]
assert l.mage_debug_read(buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(-1, 8)[:nblk].astype(np.int64)
t0 = t[:, 0].min()
ml, ep = (t[:, 1] - t[:, 0]), (t[:, 2] - t[:, 1])
print(f"blocks {nblk}; s_memtime ticks (100 MHz?) total span {t[:, 2].max() - t0}")
print(f"main loop ticks: mean {ml.mean():.0f} p10 {np.percentile(ml, 10):.0f} p90 {np.percentile(ml, 90):.0f}")
print(f"epilogue  ticks: mean {ep.mean():.0f} p10 {np.percentile(ep, 10):.0f} p90 {np.percentile(ep, 90):.0f}")
hw = t[:, 3]
cu = ((hw >> 32) << 16) | (hw & 0xffff & ~0xf)   # xcc + (se, sh, cu) bits, wave slot masked
for nm, a, b in (("prefetch(yrow,loads issue)", 1, 4), ("barrier wait", 4, 5), ("stage+math+store issue", 5, 6), ("final store drain", 6, 2)):
    dd = t[:, b] - t[:, a]
    print(f"  {nm:28s} mean {dd.mean():.0f} p10 {np.percentile(dd, 10):.0f} p90 {np.percentile(dd, 90):.0f}")
print("distinct CU ids", len(np.unique(cu)))
# timeline of the workgroups that ran on one CU (same XCC + SE/CU bits of HW_ID), to see how co-resident WGs phase
hwid = hw & 0xffffffff
xcc = hw >> 32
cu_id = (xcc << 12) | ((hwid >> 8) & 0xf) << 4 | ((hwid >> 13) & 0x7) << 8 | ((hwid >> 12) & 1)   # cu_id[11:8], se_id[15:13], sh_id[12]
for target in np.unique(cu_id)[:2]:
    sel = np.flatnonzero(cu_id == target)
    order = sel[np.argsort(t[sel, 0])][:10]
    base = t[order[0], 0]
    print(f"CU {target:#x}: {len(sel)} workgroups; (bid, start, ml_end, ep_end) relative ticks")
    for b in order:
        print(f"   bid {b:6d}  {t[b,0]-base:8d} {t[b,1]-base:8d} {t[b,2]-base:8d}   simd/wave bits {hwid[b] & 0xff:#x}")
