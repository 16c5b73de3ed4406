"""Summarise the LAST `frac` of a rocprofv3 kernel_trace.csv (steady-state sweeps of bench.py)."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.15
rows = list(csv.DictReader(open(path)))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
t1 = max(int(r["End_Timestamp"]) for r in rows)
cut = t1 - frac * (t1 - t0)
agg = defaultdict(lambda: [0, 0.0, 0.0])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < cut:
        continue
    name = r["Kernel_Name"].split("(")[0][:40]
    a = agg[name]
    a[0] += 1
    a[1] += (e - s) / 1e3
    a[2] = max(a[2], (e - s) / 1e3)
tot = sum(a[1] for a in agg.values())
print(f"window {frac * (t1 - t0) / 1e9:.2f} s, GPU busy {tot / 1e6:.3f} s")
for name, (n, us, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{name:42s} n={n:7d} total={us / 1e3:9.1f} ms avg={us / n:9.1f} us max={mx:9.1f} us")
