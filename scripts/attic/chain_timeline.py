"""Per-stage timeline of conv_chain_kernel (build with EXTRA=-DCH_TIMING=1): consumer wave 0 of workgroup 0."""
import ctypes, sys
import torch
sys.path.insert(0, ".")
import scripts.bench_chain as bc      # runs the benchmark once (sets everything up)
from stmask_amd import _lib
dbg = torch.zeros(4096, device="cuda", dtype=torch.int64)
_lib.lib().stm_debug_chain_timing(ctypes.c_void_p(dbg.data_ptr()))
bc.chain()
torch.cuda.synchronize()
t = dbg.cpu().numpy()
n = int((t != 0).sum())
t = t[:n]
per_tile = 1 + 2 * 15
tiles = (n - 1) // per_tile
print("samples", n, "tiles", tiles)
import numpy as np
rows = []
for k in range(tiles):
    s = t[k * per_tile:(k + 1) * per_tile + 1]
    start = s[0]
    arrive, leave = s[1:31:2], s[2:32:2]
    end = s[31] if len(s) > 31 else leave[-1]
    wait = leave - arrive                       # cycles parked at each barrier
    work = np.append(arrive[1:], end) - leave  # cycles from leaving barrier k to arriving at barrier k + 1
    rows.append((arrive[0] - start, wait, work))
clk = 100e6   # s_memtime / readcyclecounter ticks at 100 MHz on gfx9
sel = rows[2:-1] if len(rows) > 4 else rows
w = np.mean([r[1] for r in sel], 0) / clk * 1e6
c = np.mean([r[2] for r in sel], 0) / clk * 1e6
print("setup us %.2f" % (np.mean([r[0] for r in sel]) / clk * 1e6))
print("stage :", " ".join(f"{i:5d}" for i in range(15)))
print("wait  :", " ".join(f"{x:5.2f}" for x in w))
print("work  :", " ".join(f"{x:5.2f}" for x in c))
print("tile total us %.2f (wait %.2f, work %.2f)" % ((w.sum() + c.sum()), w.sum(), c.sum()))
