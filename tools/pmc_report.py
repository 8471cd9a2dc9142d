import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = "step" if ("eh_step_kernel" in k or "eh_wide_kernel" in k or "eh_widebf_kernel" in k or "eh_bfs_kernel" in k) else "reduce" if "eh_reduce" in k else None
        if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d2 in acc.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in sorted(d2.items())})
