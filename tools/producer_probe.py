"""x + Linear(.) producer GEMMs (bf16 residual stream, LayerNorm partial sums: out_proj K = 512, c_proj K = 2048 at cfg2 full-loop size) under the
staggered-start options of the 8-wave kernels: timing sweep (tuning only)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mage_amd import config, ops
DEV="cuda:0"; M=262144
g=torch.Generator(device=DEV).manual_seed(0)
rn=lambda *s, dtype=torch.float32, scale=1.0: (torch.randn(*s, device=DEV, generator=g)*scale).to(dtype)
def run(N,K,opts):
    a=rn(M,K,dtype=torch.bfloat16); w=rn(N,K,dtype=torch.bfloat16,scale=K**-0.5); b=rn(N,scale=0.1)
    x=rn(M,N,dtype=torch.bfloat16); part=torch.empty(N//64,M,2,device=DEV)
    import contextlib
    with contextlib.ExitStack() as es:
        for k,v in opts.items(): es.enter_context(config.lib_option(k,v))
        f=lambda: ops.gemm(a,w,x,M=M,N=N,K=K,lda=K,ldy=N,bias=b,residual=x,ldr=N,ln_part=part)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        best=1e9
        for r in range(5):
            e0.record()
            for _ in range(6): f()
            e1.record(); torch.cuda.synchronize(); best=min(best,e0.elapsed_time(e1)/6)
    return best*1e3
import sys as _s
combos=[{"gemm_stagger_groups":8,"gemm_stagger_percent":60},{"gemm_stagger_groups":8,"gemm_stagger_percent":30},{"gemm_stagger_groups":4,"gemm_stagger_percent":30},
        {"gemm_stagger_groups":8,"gemm_stagger_percent":40},{"gemm_stagger_groups":0}]
for name,N,K in (("c_proj",512,2048),("out_proj",512,512)):
    res={str(o):[] for o in combos}
    for rnd in range(8):
        for o in combos:
            res[str(o)].append(run(N,K,o))
    for k,v in sorted(res.items(), key=lambda kv: sorted(kv[1])[len(kv[1])//2]):
        print(f"{name:9s} {k:62s} median {sorted(v)[len(v)//2]:7.1f} min {min(v):7.1f} max {max(v):7.1f}")

def run_qkv(opts):
    N,K=1536,512
    a=rn(M,K,dtype=torch.bfloat16); w=rn(N,K,dtype=torch.bfloat16,scale=K**-0.5); b=rn(N,scale=0.1)
    st=torch.stack([0.05*rn(M),1.0+0.2*rn(M).abs()],1).contiguous(); cs=0.3*rn(N); y=torch.empty(M,N,device=DEV,dtype=torch.bfloat16)
    import contextlib
    with contextlib.ExitStack() as es:
        for k,v in opts.items(): es.enter_context(config.lib_option(k,v))
        f=lambda: ops.gemm(a,w,y,M=M,N=N,K=K,lda=K,ldy=N,bias=b,ln_stats=st,ln_colsum=cs)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        best=1e9
        for r in range(5):
            e0.record()
            for _ in range(6): f()
            e1.record(); torch.cuda.synchronize(); best=min(best,e0.elapsed_time(e1)/6)
    return best*1e3
combos=[{"gemm4_stagger_groups":8,"gemm4_stagger_percent":100},{"gemm4_stagger_groups":0},{"gemm4_stagger_groups":8,"gemm4_stagger_percent":50},{"gemm4_stagger_groups":4,"gemm4_stagger_percent":100},{"gemm4_stagger_groups":16,"gemm4_stagger_percent":100}]
res={str(o):[] for o in combos}
for rnd in range(8):
    for o in combos: res[str(o)].append(run_qkv(o))
for k,v in sorted(res.items(), key=lambda kv: sorted(kv[1])[len(kv[1])//2]):
    print(f"qkv(gemm4) {k:62s} median {sorted(v)[len(v)//2]:7.1f} min {min(v):7.1f} max {max(v):7.1f}")
