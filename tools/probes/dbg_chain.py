import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch, golden_util as gu
from slotformer_amd import engine, _lib
from slotformer_amd.base_slots import build_model
lib=_lib.lib(); dev=torch.device('cuda:0')
cfg=gu.C2_SAVI; torch.manual_seed(47)
m=build_model(gu.ParamsView(cfg)).eval().to(dev); m.testing=True
N,D=7,128
B,T=2,2
img=gu.seeded_img(B,T,128,seed=131).to(dev)
noise=torch.zeros(B,T,N,D,device=dev)
outs={}
with torch.no_grad():
    for mode in (0,1):
        lib.sf_set_slot_chain(mode)
        outs[mode]=engine.savi_encode(m,img,noise=noise,want_attn=True,ws_slot=('d',mode),side_stream=None)
        torch.cuda.synchronize()
p0,k0,a0=outs[0]; p1,k1,a1=outs[1]
print('post err per (t,slot):'); print((p1-p0).abs().amax(dim=(0,3)))
d=(a1-a0).abs()   # [B,T,N,HW]
print('attn err per (t,slot):'); print(d.amax(dim=(0,3)))
dd=d[0,0]  # [N,HW]
bad=(dd>1e-3).nonzero()
print('bad count',bad.shape[0],'of',dd.numel())
print(bad[:40].tolist())
px=bad[:,1]
print('bad pixel mod 32 histogram', torch.bincount(px%32, minlength=32).tolist())
print('bad pixel // 512 hist', torch.bincount(px//512, minlength=8).tolist())
print('sum over slots chain', a1[0,0].sum(0)[:8].tolist(), 'ref', a0[0,0].sum(0)[:8].tolist())
