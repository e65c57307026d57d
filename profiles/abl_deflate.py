import sys,os,ctypes as C,time
sys.path.insert(0,"python-zlib-ng_amd")
from zlib_ng_amd import _lib, corpus, devmem
ctx=_lib.Context(0); L,h=ctx.L,ctx.h
n=1<<30; B=131072
host=corpus.text(32<<20)
d=devmem.empty(ctx,n+64); d[0:host.size]=host
k=host.size
while k<n:
    m=min(k,n-k); d[k:k+m]=d[0:m]; k+=m
d[n:n+64]=0
nb=n//B
blocks=(_lib.Block*nb)()
for b in range(nb): blocks[b]=_lib.Block(b*B,B,32768 if b else 0,0,0)
slots=devmem.empty(ctx,nb*_lib.SLOT_STRIDE); ul=devmem.empty(ctx,4*nb); uc=devmem.empty(ctx,4*nb)
for it in range(2):
    ctx.profiling(True); ctx.kernel_times(True)
    r=L.zngamd_deflate_blocks_dev(h,d.vp(),n,blocks,nb,int(os.environ.get("LEVEL","6")),slots.vp(),ul.vp(),uc.vp(),None)
    kt=ctx.kernel_times(True)
print("ablate",os.environ.get("ZNGAMD_ABLATE"),"rc",r,{k:round(v[0],2) for k,v in kt.items() if v[1]})
if hasattr(L, "zngamd_debug_ch_stats"):
    o=(C.c_ulonglong*32)(); L.zngamd_debug_ch_stats(o)
    for t in range(3):
        print(f"ablate chstats table {'ABC'[t]} (both iterations): " + " ".join(f"role{r}: waits {o[8*t+2*r]/max(1,o[8*t+2*r+1]):.2f} of {o[8*t+2*r+1]/1e9:.2f} Gclk" for r in range(4)))
    if o[27]: print(f"ablate chstats table A atomic wave per group: reads {o[24]/o[27]:.0f} atomics {o[25]/o[27]:.0f} writes {o[26]/o[27]:.0f} clocks")
