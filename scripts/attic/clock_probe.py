"""Sample rocm-smi (clocks, power) while the planar conv runs back to back (is the matrix pipe clock/power limited?)."""
import sys, os, subprocess, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stmask_amd import ops
mode = sys.argv[1] if len(sys.argv) > 1 else "conv"
B = 32 if mode == "conv16" else 8
x = torch.randn(B, 96, 160, 256, device="cuda"); w = torch.randn(256, 256, 3, 3, device="cuda") * 0.02
if mode == "conv16":      # the fp16x2 ring kernel on 256 x 128 tiles (the benchmark's format; STM_CONV_ABL ablations apply to it)
    (pk, osc), xp = ops.conv_pack_weights(w, fmt=1), ops.split_planes(x, 1)
else:
    pk, xp = ops.conv_pack_weights(w), ops.split_planes(x)
stop = False
def sampler():
    time.sleep(1.5)
    for _ in range(3):
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        keep = [l for l in out.splitlines() if any(k in l for k in ("sclk", "mclk", "Power", "fclk"))]
        print("\n".join(keep[:8]), flush=True)
        print("--", flush=True)
        time.sleep(1.0)
t = threading.Thread(target=sampler); t.start()
t0 = time.time(); n = 0
while time.time() - t0 < 6.0:
    for _ in range(20):
        if mode == "conv16":
            ops.conv2d_planar(xp, pk, (256, 256, 3, 3), (B, 96, 160), None, None, padding=1, fmt=1, out_scale=osc)
        elif mode == "conv":
            ops.conv2d_planar(xp, pk, (256, 256, 3, 3), (8, 96, 160), None, None, padding=1)
        else:
            x.mul_(1.0)
    torch.cuda.synchronize(); n += 20
t.join()
us = (time.time() - t0) / n * 1e6
gf = 2.0 * B * 96 * 160 * 256 * 256 * 9 / 1e9
print(mode, "launches", n, "avg us %.1f" % us, ("= %.1f TFLOP/s fp32-equivalent, %.0f TFLOP/s of issued fp16 MFMA (%.3f of 2500)" % (gf / us * 1e3, 3 * gf / us * 1e3, 3 * gf / us * 1e3 / 2500)) if mode == "conv16" else "")
