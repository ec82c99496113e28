import sys; sys.path.insert(0,".")
import numpy as np, torch
from simpledet_amd import ops
from oracle import pyoracle as orc
N,C,H,W,F=16,256,50,84,256
x=torch.randn((N,C,H,W),device="cuda"); off=torch.randn((N,72,H,W),device="cuda")*2; wt=torch.randn((F,C,3,3),device="cuda")*0.05
y=ops.deform_conv_forward(x,off,wt,1,1,1,4)
yu,_=ops.deform_conv_forward(x,off,wt,1,1,1,4,keep_col=True)
print("fused vs unfused max", float((y-yu).abs().max()), "y absmax", float(y.abs().max()))
wy=orc.deform_conv_fwd(x[:2].cpu().numpy(),off[:2].cpu().numpy(),wt.cpu().numpy(),1,1,1,4)
print("oracle vs fused (2 img)", float(np.abs(y[:2].cpu().numpy()-wy).max()), "oracle absmax", float(np.abs(wy).max()))
wy16=orc.deform_conv_fwd(x.cpu().numpy(),off.cpu().numpy(),wt.cpu().numpy(),1,1,1,4)
e=np.abs(y.cpu().numpy()-wy16).reshape(N,-1).max(1); print("per image", e)
