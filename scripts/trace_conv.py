"""Per-slab clock stamps of workgroup 0 of the planar conv kernel (debug aid)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stmask_amd import ops, _lib
H, W, C, O, k = 96, 160, 256, 256, 3
x = torch.randn(8, H, W, C, device="cuda"); w = torch.randn(O, C, k, k, device="cuda") * 0.02
pk = ops.conv_pack_weights(w)
xp = ops.split_planes(x)
for _ in range(3):
    ops.conv2d_planar(xp, pk, (O, C, k, k), (8, H, W), None, None, padding=1)
tr = torch.zeros(2 * 64 * 8, dtype=torch.int64, device="cuda")
_lib.lib().stm_debug_conv_set_trace(ctypes.c_void_p(tr.data_ptr()))
ops.conv2d_planar(xp, pk, (O, C, k, k), (8, H, W), None, None, padding=1)
torch.cuda.synchronize()
_lib.lib().stm_debug_conv_set_trace(ctypes.c_void_p(0))
t = tr.cpu().view(2, 64, 8)
t0 = int(t[0, 0, 0])
print("slab wave | top  dma_landed  after_barrier  dma_issued  mfma_done   (clock ticks rel. to slab top)")
for s in range(8, 20):
    for g in range(2):
        b = int(t[g, s, 0])
        r = [int(v) - b for v in t[g, s, :5]]
        print(s, g, r, " slab len", int(t[g, s + 1, 0]) - b)
print("loop end -> epilogue end:", int(t[0, 1, 7]) - int(t[0, 0, 7]), " whole loop", int(t[0, 0, 7]) - t0)
