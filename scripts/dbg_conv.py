import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stmask_amd import ops
dev = "cuda"
C = 32
x = torch.arange(1 * 2 * 4 * C, dtype=torch.float32).reshape(1, 2, 4, C) 
w = torch.eye(C).reshape(C, C, 1, 1)
y = ops.conv2d_nhwc(x.to(dev), ops.conv_pack_weights(w.to(dev)), (C, C, 1, 1)).cpu()
print("identity: equal", torch.equal(y, x))
print(y[0, 0, 0]); print(y[0, 0, 1]); print(y[0, 1, 3])
# single nonzero weight
w = torch.zeros(C, C, 1, 1); w[3, 5] = 1.0
y = ops.conv2d_nhwc(x.to(dev), ops.conv_pack_weights(w.to(dev)), (C, C, 1, 1)).cpu()
print("w[3,5]=1: y[...,3] =", y[0, :, :, 3].flatten(), "expected", x[0, :, :, 5].flatten())
print("nonzero cols", y.abs().sum(dim=(0, 1, 2)).nonzero().flatten())
pk = ops.conv_pack_weights(w.to(dev)).cpu()
v = pk.view(torch.int16)
nz = v.nonzero().flatten()
print("packed nonzero int16 idx", nz, v[nz])
