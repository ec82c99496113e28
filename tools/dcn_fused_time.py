import torch, sys
sys.path.insert(0,".")
from simpledet_amd import ops
from simpledet_amd._lib import lib
def t(fn,it=40):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/it
N,C,H,W,F=16,256,50,84,256
for sc in (2.0, 1.0, 0.5):
    x=torch.randn(N,C,H,W,device="cuda"); off=torch.randn(N,72,H,W,device="cuda")*sc; wt=torch.randn(F,C,3,3,device="cuda")*0.05
    print("off sigma",sc,"fused fwd %.3f ms  unfused %.3f ms"%(t(lambda:ops.deform_conv_forward(x,off,wt,1,1,1,4)), t(lambda:ops.deform_conv_forward(x,off,wt,1,1,1,4,keep_col=True))))
for tw in (96,88,80,64):
    lib().set_tuning("dcn_fused_tile",tw)
    print("tile",tw,"%.3f ms"%t(lambda:ops.deform_conv_forward(x,off,wt,1,1,1,4)))
