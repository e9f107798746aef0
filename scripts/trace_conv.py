"""Per-phase clock stamps of workgroup 0 of the ping-pong conv kernel (debug aid)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stmask_amd import ops, _lib
os.environ["STM_CONV_KERNEL"] = "2"
H, W, C, O, k = 96, 160, 256, 256, 3
x = torch.randn(8, H, W, C, device="cuda"); w = torch.randn(O, C, k, k, device="cuda") * 0.02
pk = ops.conv_pack_weights(w)
for _ in range(3):
    ops.conv2d_nhwc(x, pk, (O, C, k, k), None, None, padding=1)
tr = torch.zeros(2 * 64 * 8, dtype=torch.int64, device="cuda")
_lib.lib().stm_debug_conv_set_trace(ctypes.c_void_p(tr.data_ptr()))
ops.conv2d_nhwc(x, pk, (O, C, k, k), None, None, padding=1)
torch.cuda.synchronize()
_lib.lib().stm_debug_conv_set_trace(ctypes.c_void_p(0))
t = tr.cpu().view(2, 64, 8)
t0 = int(t[0, 0, 0])
print("phase grp | start  loads_done  staged  fetched  mfma_done  at_barrier  after_barrier   (clock ticks rel. to start)")
for p in range(8, 24):
    for g in range(2):
        r = [int(v) - t0 if int(v) else 0 for v in t[g, p, :7]]
        print(p, g, r, " phase len", int(t[g, p + 1, 0]) - int(t[g, p, 0]))
