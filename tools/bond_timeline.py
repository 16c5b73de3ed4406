"""Timeline of ONE steady-state bond update from a rocprofv3 kernel_trace.csv: every kernel between two consecutive
k_qr_large launches near the end of the trace, consecutive launches of the same kernel merged, with gaps."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")[:34]
qr = [i for i, r in enumerate(rows) if name(r).startswith("k_qr_large")]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 40
i0, i1 = qr[-skip - 1], qr[-skip]
t0 = int(rows[i0]["Start_Timestamp"])
# start the window at the first Lanczos kernel after the previous SVD: walk back from i0 to the end of the previous SVD
# (k_jacobi_ring; k_jacobi_finish on the multi-launch path)
j = i0 - 1
while j > 0 and not (name(rows[j]).startswith("k_jacobi_finish") or name(rows[j]).startswith("k_jacobi_ring")):
    j -= 1
seq = rows[j:i1]
t_base = int(seq[0]["Start_Timestamp"])
out = []
for r in seq:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = name(r)
    if out and out[-1][0] == n:
        out[-1][2] = e
        out[-1][3] += 1
        out[-1][4] += e - s
    else:
        out.append([n, s, e, 1, e - s])
prev_end = t_base
for n, s, e, cnt, busy in out:
    print(f"{(s - t_base) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:7.1f}  {n:36s} x{cnt:4d}  span {(e - s) / 1e3:8.1f} us  busy {busy / 1e3:8.1f} us")
    prev_end = e
print(f"total {(prev_end - t_base) / 1e3:.1f} us")
