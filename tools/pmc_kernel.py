"""Print counters of the LAST dispatch of kernels whose name contains argv[1], from rocprofv3 counter_collection dirs argv[2:]."""
import csv, glob, sys
pat = sys.argv[1]
for d in sys.argv[2:]:
    f = glob.glob(d + "/*/*_counter_collection.csv")[0]
    rows = [r for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
    last = max(int(r["Dispatch_Id"]) for r in rows)
    agg = {}
    for r in rows:
        if int(r["Dispatch_Id"]) == last:
            agg[r["Counter_Name"]] = agg.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for k, v in agg.items():
        print(f"{k:45s} {v:.5g}")
