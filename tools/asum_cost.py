import sys, torch
sys.path.insert(0, ".")
from multimodaldiscussiontransformer_amd import ops
from tools.kbench import timeit
M=100864
g=torch.Generator(device="cuda").manual_seed(1)
r=lambda *s: torch.randn(*s, device="cuda", dtype=torch.bfloat16, generator=g)
x768,x2304,x3072=r(M,768),r(M,2304),r(M,3072)
for nm,dy,x in (("qkv",x2304,x768),("o",x768,x768),("fc1",x3072,x768),("fc2",x768,x3072)):
    n_out,k_in=dy.shape[1],x.shape[1]
    tiles=((n_out+127)//128)*((k_in+127)//128); split=max(1,min(1024//tiles,M//1024))
    out=torch.zeros(n_out,k_in,device="cuda"); gb=torch.zeros(n_out,device="cuda")
    res=[]
    for rep in range(2):
        for use in (True,False):
            f=lambda: ops.gemm(dy,x,trans_a=True,trans_b=True,out=out,epilogue=ops.EPI_ATOMIC,split_k=split,asum=gb if use else None)
            timeit(f,iters=5); t=timeit(f,iters=20); res.append((use,t))
    print(nm, " | ".join(f"{'asum' if u else 'plain'} {t*1e6:7.1f} us" for u,t in res), flush=True)
