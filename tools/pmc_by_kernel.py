"""Per-kernel means of rocprofv3 --pmc counters over the steady-state tail of a bench run.
usage: pmc_by_kernel.py counter_collection.csv [tail_fraction]"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
rows = list(csv.DictReader(open(path)))
ids = sorted({int(r["Dispatch_Id"]) for r in rows})
cut = ids[int(len(ids) * (1 - frac))]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for r in rows:
    if int(r["Dispatch_Id"]) < cut:
        continue
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:28]
    a = acc[k][r["Counter_Name"]]
    a[0] += float(r["Counter_Value"])
    a[1] += 1
names = sorted({c for k in acc for c in acc[k]})
print("kernel".ljust(30) + "".join(n[-22:].rjust(24) for n in names) + "   launches")
for k, d in sorted(acc.items(), key=lambda kv: -sum(v[0] for v in kv[1].values())):
    n = max(v[1] for v in d.values())
    print(k.ljust(30) + "".join(f"{d[c][0] / max(d[c][1], 1):24.1f}" for c in names) + f"   {n}")
