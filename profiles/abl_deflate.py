import sys,os,ctypes as C,time
sys.path.insert(0,"python-zlib-ng_amd")
import torch
from zlib_ng_amd import _lib, corpus
ctx=_lib.Context(0); L,h=ctx.L,ctx.h
n=1<<30; B=131072
host=corpus.text(32<<20)
d=torch.from_numpy(host).cuda().repeat(n//(32<<20)); d=torch.cat([d,torch.zeros(64,dtype=torch.uint8,device="cuda")])
nb=n//B
blocks=(_lib.Block*nb)()
for b in range(nb): blocks[b]=_lib.Block(b*B,B,32768 if b else 0,0,0)
slots=torch.empty(nb*_lib.SLOT_STRIDE,dtype=torch.uint8,device="cuda"); ul=torch.empty(nb,dtype=torch.int32,device="cuda"); uc=torch.empty(nb,dtype=torch.int32,device="cuda")
p=lambda t:C.c_void_p(t.data_ptr())
for it in range(2):
    ctx.profiling(True); ctx.kernel_times(True)
    r=L.zngamd_deflate_blocks_dev(h,p(d),n,blocks,nb,int(os.environ.get("LEVEL","6")),p(slots),p(ul),p(uc),None)
    kt=ctx.kernel_times(True)
print("ablate",os.environ.get("ZNGAMD_ABLATE"),"rc",r,{k:round(v[0],2) for k,v in kt.items() if v[1]})
