"""Compact view of a rocprofv3 kernel_stats.csv: python scripts/short_stats.py stats.csv steps [top]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.3f ms (%.3f ms/step over %g steps)" % (tot / 1e6, tot / 1e6 / steps, steps))
for r in rows[:top]:
    name = r["Name"]
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"at::native::", "", name)
    print("%-90s %6d %9.3f ms %8.1f us/call %7.3f ms/step" % (name[:90], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                                                       float(r["TotalDurationNs"]) / 1e6 / steps))
