"""Average a rocprofv3 --pmc counter over the steady-state launches of one kernel.
usage: pmc_summary.py counter_collection.csv COUNTER [tail_fraction]"""
import csv
import sys

path, counter = sys.argv[1], sys.argv[2]
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
tail = vals[int(len(vals) * (1 - frac)):]
print(f"{counter}: {len(vals)} dispatches, steady-state tail {len(tail)}: mean {sum(tail) / len(tail):.3f} (raw counter units, KB) "
      f"min {min(tail):.3f} max {max(tail):.3f}")
