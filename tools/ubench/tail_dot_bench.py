import torch, time, sys
sys.path.insert(0,'/root/repo')
from spi_amd.torch_utils.ops import bias_act
from spi_amd import hip
def t(fn, r=20):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(r): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/r*1e3
for (n,c,h) in [(2,128,512),(2,256,256),(2,512,64),(2,512,16),(1,128,512)]:
    dy=torch.randn(n,c,h,h,device='cuda'); y=torch.randn(n,c,h,h,device='cuda')
    b=torch.randn(c,device='cuda'); nz=torch.randn(h,h,device='cuda'); ng=torch.ones(1,device='cuda')
    zo=torch.zeros(n*c,device='cuda')
    a=t(lambda: bias_act.tail_backward(dy,y,nz,ng,3,0.2,1.414,256.0,False,False,False))
    bb=t(lambda: bias_act.tail_backward(dy,y,nz,ng,3,0.2,1.414,256.0,False,False,False,zdot=(zo,b,nz,ng)))
    def cd():
        dz=bias_act.tail_backward(dy,y,nz,ng,3,0.2,1.414,256.0,False,False,False)[0]
        hip.call('spi_chan_dot', hip.ptr(dz), hip.ptr(y), hip.ptr(zo), n*c, c, h*h, hip.ptr(b), hip.ptr(nz), hip.ptr(ng), 3, 0.2, 1.414, hip.stream())
    cc=t(cd)
    print(n,c,h,'tail %.1f us  tail+zdot %.1f us  tail+chan_dot %.1f us'%(a,bb,cc))
