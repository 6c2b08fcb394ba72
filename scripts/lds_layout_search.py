"""Per-pass LDS layout search for the Stockham plans of muse_psfr_amd/csrc/fft_lds.h: for every
boundary between two passes, rank the x + c*floor(x/S) images by modelled LDS-array cycles
(writes of the pass + reads of the next).  Usage: python scripts/lds_layout_search.py N ESIZE"""
import sys
import lds_bank_model as model
N=int(sys.argv[1]); esize=int(sys.argv[2])
plans = {64:([8,8],8),128:([8,4,4],16),256:([8,8,4],32),512:([8,8,8],64),1024:([8,8,4,4],128),1280:([4,4,4,20],64)}
radix,TPR0=plans[N]; TPR=min(TPR0,64)
rk = 'r64' if esize == 8 else 'r128'; wk = 'w64' if esize == 8 else 'w128'
SLOTS = 64//TPR
def lanes():
    for lane in range(64):
        slot,t=divmod(lane,TPR)
        yield slot,t
def wcost(p,f,padlen):
    ns=1
    for i in range(p): ns*=radix[i]
    R=radix[p]; NB=N//R; NBT=NB//TPR0; tot=0
    for b in range(NBT):
        for q in range(R):
            ad=[]
            for slot,t in lanes():
                j=t+b*TPR0; k=j%ns; base=(j-k)*R+k
                ad.append((slot*padlen+f(base+q*ns))*esize)
            c,n=model.cycles(wk,ad,esize); tot+=c
    return tot
def rcost(p,f,padlen):   # reads of pass p (image after pass p-1)
    R=radix[p]; NB=N//R; NBT=NB//TPR0; tot=0
    for b in range(NBT):
        for q in range(R):
            ad=[(slot*padlen+f(t+b*TPR0+q*NB))*esize for slot,t in lanes()]
            c,n=model.cycles(rk,ad,esize); tot+=c
    return tot
cands={'ident':lambda x:x}
for S in (2,4,8,16,32,64,128):
    for c in (1,2,3,4):
        cands['x+%d(x/%d)'%(c,S)]=(lambda x,S=S,c=c: x+c*(x//S))
for S1,S2 in ((8,64),(4,16),(4,64),(16,64),(16,256),(4,32),(8,32)):
    for c1 in (1,2):
        for c2 in (1,2,4,8):
            cands['x+%d(x/%d)+%d(x/%d)'%(c1,S1,c2,S2)]=(lambda x,S1=S1,S2=S2,c1=c1,c2=c2: x+c1*(x//S1)+c2*(x//S2))
for p in range(len(radix)):
    best=[]
    for name,f in cands.items():
        vals=[f(x) for x in range(N)]
        if len(set(vals))!=N or max(vals)>=N+N//8: continue      # must fit the NPAD buffer
        padlen=N+N//8
        w=wcost(p,f,padlen); r=rcost(p+1,f,padlen) if p+1<len(radix) else 0
        best.append((w+r,w,r,name))
    best.sort()
    print('after pass',p,best[:6])
