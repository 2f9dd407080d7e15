"""Summarise a rocprofv3 kernel_trace.csv: per (kernel, grid) average duration of the last N dispatches."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
last = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows)
agg = collections.OrderedDict()
for r in rows[-last:]:
    k = (r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:44], r["Grid_Size_X"], r["Grid_Size_Y"], r["VGPR_Count"], r["Scratch_Size"])
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(k, [0, 0.0, 1e30, 0.0])
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
for k, a in agg.items():
    print(f"{k[0]:44s} grid=({k[1]},{k[2]}) vgpr={k[3]} scratch={k[4]} n={a[0]:4d} avg={a[1]/a[0]:9.1f}us min={a[2]:9.1f} max={a[3]:9.1f}")
