"""Summarise a rocprofv3 --pmc counter_collection CSV: per kernel (name + grid), mean of each counter over dispatches."""
import csv, glob, sys, collections
path = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob("gpurun_out/pmc*/**/*_counter_collection.csv", recursive=True))[-1]
acc = collections.OrderedDict()
for r in csv.DictReader(open(path)):
    name = r["Kernel_Name"]
    if "suave" not in name:
        continue
    short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void suave::", "").replace("suave::", "")
    key = (short, r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"])
    d = acc.setdefault(key, collections.defaultdict(list))
    d[r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, d in acc.items():
    print(f"{key[0][:44]:44s} blocks={int(key[1]) // max(int(key[2]), 1)} lds={key[3]}")
    for c, v in d.items():
        print(f"    {c:32s} {sum(v)/len(v):16.1f}  (n={len(v)})")
