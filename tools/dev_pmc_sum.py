"""dev only: per-dispatch average of each PMC counter for kernels whose name contains argv[2]"""
import csv, glob, sys, collections
d, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float)
disp = collections.defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
            disp[r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(acc):
    print(f"{pat} {k} = {acc[k] / max(1, len(disp[k])):.4g} per dispatch ({len(disp[k])} dispatches)")
