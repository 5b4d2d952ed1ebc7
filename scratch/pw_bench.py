import torch, time, sys
sys.path.insert(0,'.')
import torch.nn.functional as F
from amcontrast3d_amd import ops
DEV='cuda:0'
shapes=[(8,32,64,(6000,32)),(8,64,128,(1500,32)),(8,128,256,(375,32)),(8,256,512,(93,32)),(8,4,32,(24000,)),(8,96,32,(24000,)),(8,32,32,(24000,)),(8,32,13,(24000,)),(8,192,64,(6000,)),(8,64,64,(6000,)),
        (8,384,128,(1500,)),(8,128,128,(1500,)),(8,768,256,(375,)),(8,256,256,(375,)),(8,131,256,(375,32)),(8,259,512,(93,32))]
def tm(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n*1000
for B,Ci,Co,sp in shapes:
    x=torch.randn(B,Ci,*sp,device=DEV,requires_grad=True); w=torch.randn(Co,Ci,*([1]*len(sp)),device=DEV,requires_grad=True)
    go=torch.randn(B,Co,*sp,device=DEV)
    conv=F.conv1d if len(sp)==1 else F.conv2d
    P=x[0,0].numel()
    t_f=tm(lambda: ops.pointwise_conv(x,w)); 
    def bw():
        y=ops.pointwise_conv(x,w); y.backward(go)
    t_fb=tm(bw)
    r_f=tm(lambda: conv(x,w))
    def rbw():
        y=conv(x,w); y.backward(go)
    r_fb=tm(rbw)
    gb=4*B*P*(Ci+Co)/1e9; gf=2*B*P*Ci*Co/1e9
    print(f"{Ci:4d}->{Co:4d} P={P:7d}: fwd {t_f:7.1f} us ({gb/t_f*1e6:6.0f} GB/s {gf/t_f*1e3:5.1f} TF)  fwd+bwd {t_fb:7.1f} | torch fwd {r_f:7.1f} fwd+bwd {r_fb:7.1f}")
