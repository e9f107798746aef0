"""Join bench.py --layer-table outputs: python scripts/join_layers.py base.txt other1.txt ...  (us/call per layer shape)."""
import sys


def load(path):
    rows = {}
    for line in open(path):
        f = line.split()
        if len(f) == 11 and f[0].isdigit():
            rows[tuple(int(v) for v in f[:7])] = (int(f[7]), float(f[8]), float(f[10]))
    return rows


tabs = [load(p) for p in sys.argv[1:]]
print("%9s %5s %5s %2s %2s %2s %4s %6s" % ("M", "C", "O", "k", "s", "g", "tile", "calls"), *["%9s" % p.split("/")[-1][-9:] for p in sys.argv[1:]])
tot = [0.0] * len(tabs)
for key, (n, us, ms) in sorted(tabs[0].items(), key=lambda kv: -kv[1][2]):
    cells = []
    for i, t in enumerate(tabs):
        v = t.get(key)
        cells.append("%9.1f" % v[1] if v else "%9s" % "-")
        tot[i] += v[2] if v else 0.0
    print("%9d %5d %5d %2d %2d %2d %4d %6d" % (*key, n), *cells)
print("ms/step:", *["%9.3f" % t for t in tot])
