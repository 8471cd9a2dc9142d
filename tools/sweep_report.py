import csv, glob, json, sys
d = sys.argv[1]
plan = json.load(open("gpurun_out/sweep_plan.json"))
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
step = [r for r in rows if "eh_step_kernel" in r["Kernel_Name"]]
red = [r for r in rows if "eh_reduce_kernel" in r["Kernel_Name"]]
i = 0
for p in plan:
    n = p["steps"]
    s = step[i:i + n]; r = red[i:i + n] if len(red) >= i + n else s; i += n
    ds = sorted(int(x["End_Timestamp"]) - int(x["Start_Timestamp"]) for x in s)
    dr = sorted(int(x["End_Timestamp"]) - int(x["Start_Timestamp"]) for x in r)
    wall = (int(r[-1]["End_Timestamp"]) - int(s[10]["Start_Timestamp"])) / (n - 10)
    print(f"B={p['batch']:8d} max_blocks={p['max_blocks']:4d} variant={p.get('variant',0)} wg={s[0].get('Workgroup_Size_X', s[0].get('Workgroup_Size','?'))} step_kernel median {ds[len(ds)//2]/1e3:7.2f} us  reduce median {dr[len(dr)//2]/1e3:6.2f} us  wall/step {wall/1e3:7.2f} us  -> {p['batch']/wall*1e3:8.1f} Msamples/s")
