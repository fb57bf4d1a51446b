"""x + Linear(.) producers at the incremental loop's step sizes (16 k / 8 k rows) under the kernel-selection options (tuning only)."""
import os, sys, torch, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mage_amd import config, ops
DEV="cuda:0"
g=torch.Generator(device=DEV).manual_seed(0)
rn=lambda *s, dtype=torch.float32, scale=1.0: (torch.randn(*s, device=DEV, generator=g)*scale).to(dtype)
def run(M,N,K,opts):
    a=rn(M,K,dtype=torch.bfloat16); w=rn(N,K,dtype=torch.bfloat16,scale=K**-0.5); b=rn(N,scale=0.1)
    x=rn(M,N,dtype=torch.bfloat16); part=torch.empty(N//64,M,2,device=DEV)
    with contextlib.ExitStack() as es:
        for k,v in opts.items(): es.enter_context(config.lib_option(k,v))
        f=lambda: ops.gemm(a,w,x,M=M,N=N,K=K,lda=K,ldy=N,bias=b,residual=x,ldr=N,ln_part=part)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        best=1e9
        for r in range(5):
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize(); best=min(best,e0.elapsed_time(e1)/10)
    return best*1e3
combos=[{}, {"gemm_no_narrow_few":1}, {"gemm_no_narrow":1}, {"gemm_no_8phase":1}]
for M in (16384, 8192):
    for name,N,K in (("out_proj",512,512),("c_proj",512,2048)):
        res={str(o):[] for o in combos}
        for rnd in range(5):
            for o in combos: res[str(o)].append(run(M,N,K,o))
        for k,v in sorted(res.items(), key=lambda kv: sorted(kv[1])[len(kv[1])//2]):
            print(f"M={M:6d} {name:9s} {k:32s} median {sorted(v)[len(v)//2]:7.1f} min {min(v):7.1f}")
