"""Join rocprofv3 counter_collection.csv files: per dispatch of the conv kernels in the last forward, counter values."""
import collections
import csv
import glob
import sys

def load(d):
    f = glob.glob(d + "/*/*_counter_collection.csv")[0]
    rows = list(csv.DictReader(open(f)))
    out = collections.OrderedDict()
    for r in rows:
        k = int(r["Dispatch_Id"])
        e = out.setdefault(k, {"name": r["Kernel_Name"], "grid": r["Grid_Size"], "vgpr": r.get("VGPR_Count"), "lds": r.get("LDS_Block_Size")})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return out

dirs = sys.argv[1:]
tabs = [load(d) for d in dirs]
base = tabs[0]
keys = sorted(base)
names = []
for t in tabs:
    for e in t.values():
        for c in e:
            if c not in ("name", "grid", "vgpr", "lds") and c not in names:
                names.append(c)
print("disp kernel grid", " ".join(names))
for k in keys[-42:]:
    e = base[k]
    nm = e["name"].replace("(anonymous namespace)::", "").replace("void ", "")[:28]
    vals = []
    for c in names:
        v = None
        for t in tabs:
            if k in t and c in t[k]:
                v = t[k][c]
        vals.append("-" if v is None else f"{v:.4g}")
    print(k, nm, e["grid"], " ".join(vals))
