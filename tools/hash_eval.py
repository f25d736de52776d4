import numpy as np
M32=np.uint64(0xFFFFFFFF)
def rotl(x,r): return ((x<<np.uint64(r))|(x>>np.uint64(32-r)))&M32
def mad24(x,c,add): return (((x&np.uint64(0xFFFFFF))*np.uint64(c&0xFFFFFF))+add)&M32
def make(rounds):
    def h(x):
        x=x.astype(np.uint64)&M32
        for (c,r,s) in rounds:
            x=mad24(x,c,rotl(x,r))
            x^=x>>np.uint64(s)
        return x
    return h
def lowbias2(x):
    x=x.astype(np.uint64)&M32
    for (a,b,s1,s2,s3) in [(0x7feb352d,0x846ca68b,16,15,16),(0x2c1b3c6d,0x297a2d39,15,12,15)]:
        x^=x>>np.uint64(s1); x=(x*np.uint64(a))&M32; x^=x>>np.uint64(s2); x=(x*np.uint64(b))&M32; x^=x>>np.uint64(s3)
    return x
def avalanche(h, xs):
    base=h(xs)
    worst=0; 
    mat=np.zeros((32,32))
    for i in range(32):
        d=base^h(xs^np.uint64(1<<i))
        for o in range(32):
            mat[i,o]=((d>>np.uint64(o))&np.uint64(1)).mean()
    return np.abs(mat-0.5).max(), np.abs(mat-0.5).mean()
def seqtests(h,key=0x1234567):
    n=1<<22
    xs=(np.arange(n,dtype=np.uint64)^np.uint64(key))&M32
    v=h(xs)
    lo=(v&np.uint64(0xFFFF)).astype(np.float64)/65536; hi=(v>>np.uint64(16)).astype(np.float64)/65536
    out={}
    for p in (0.1,0.3,0.4):
        k=np.stack([lo>=p,hi>=p],1).reshape(-1).astype(np.float64)  # element order
        out[f'keep{p}']=k.mean()-(1-p)
        kc=k-k.mean()
        out[f'ac1_{p}']=(kc[:-1]*kc[1:]).mean()/kc.var()
        out[f'ac2_{p}']=(kc[:-2]*kc[2:]).mean()/kc.var()
        for stride in (768,3072,104,208):
            out[f'acS{stride}_{p}']=(kc[:-stride]*kc[stride:]).mean()/kc.var()
    return out
rng=np.random.default_rng(0)
rand=rng.integers(0,1<<32,size=200000,dtype=np.uint64)
seq=np.arange(200000,dtype=np.uint64)
cands={
 'lowbias2':lowbias2,
 'm3a':make([(0xD3B54B,16,13),(0xA54FF5,11,15),(0x9E3779,7,16)]),
 'm3b':make([(0x95F24D,15,15),(0xC2B2AE,13,13),(0x85EBCB,17,16)]),
 'm4':make([(0xD3B54B,16,13),(0xA54FF5,11,15),(0x9E3779,7,14),(0xC2B2AD,13,16)]),
 'm2':make([(0xD3B54B,16,13),(0xA54FF5,11,16)]),
}
for n,h in cands.items():
    print(n,'aval rand max/mean %.3f %.4f'%avalanche(h,rand),' seq %.3f %.4f'%avalanche(h,seq))
    st=seqtests(h)
    print('   worst |stat| %.5f'%max(abs(v) for v in st.values()), {k:round(v,5) for k,v in st.items() if abs(v)>2e-3})
