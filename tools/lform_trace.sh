#!/bin/bash
# Where a small-batch step of the layer-wise form spends its time: rocprofv3 kernel trace of tools/bench_lform.py at one batch size,
# then per kernel the mean duration and, over the steady-state steps, busy time against wall time (the rest are the gaps between
# dependent launches).   usage (repo root, GPU box): bash tools/lform_trace.sh OUTDIR BATCH
set -u
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/$1; B=$2
mkdir -p $OUT; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$B -- python3 $ROOT/tools/bench_lform.py $B > $OUT/bench_$B.json 2> $OUT/trace_$B.err
python3 - "$OUT/trace_$B" "$B" <<'PY'
import csv, glob, sys, collections
d, B = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
rows.sort()
rows = rows[len(rows) // 3:]                     # steady state: the last two thirds
busy = sum(e - s for s, e, _ in rows); wall = rows[-1][1] - rows[0][0]
gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
per = collections.defaultdict(list)
for s, e, k in rows: per[k].append(e - s)
nstep = sum(1 for _, _, k in rows if "eh_lform_mech_kernel<true" in k)
print(f"batch {B}: {len(rows)} dispatches over {nstep} steps = {len(rows)/max(nstep,1):.1f} per step; wall {wall/1e3/max(nstep,1):.1f} us/step, kernels busy {busy/1e3/max(nstep,1):.1f} us/step, "
      f"gaps {sum(gaps)/1e3/max(nstep,1):.1f} us/step (mean gap {sum(gaps)/len(gaps)/1e3:.2f} us)")
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {k:62s} n/step {len(v)/max(nstep,1):5.1f}  mean {sum(v)/len(v)/1e3:7.2f} us  total/step {sum(v)/1e3/max(nstep,1):7.1f} us")
PY
