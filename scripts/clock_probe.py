"""Sample rocm-smi (clocks, power) while the planar conv runs back to back (is the matrix pipe clock/power limited?)."""
import sys, os, subprocess, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stmask_amd import ops
mode = sys.argv[1] if len(sys.argv) > 1 else "conv"
x = torch.randn(8, 96, 160, 256, device="cuda"); w = torch.randn(256, 256, 3, 3, device="cuda") * 0.02
pk = ops.conv_pack_weights(w); xp = ops.split_planes(x)
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
        if mode == "conv":
            ops.conv2d_planar(xp, pk, (256, 256, 3, 3), (8, 96, 160), None, None, padding=1)
        else:
            x.mul_(1.0)
    torch.cuda.synchronize(); n += 20
t.join()
print(mode, "launches", n, "avg us", (time.time() - t0) / n * 1e6)
