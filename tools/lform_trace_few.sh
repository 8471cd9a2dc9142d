#!/bin/bash
# per-kernel durations of the layer-wise few-rows step (batch 64 by default): rocprofv3 kernel trace of tools/bench_lform.py, the stats table
# usage (GPU box, repo root): bash tools/lform_trace_few.sh [batch] [out_dir]
B=${1:-64}
OUT=${2:-gpurun_out/lform_few}
PROG=${3:-tools/bench_lform.py}      # (tools/bench_lform_epoch.py: the same through the epoch driver)
mkdir -p $OUT
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_few
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_few -- python3 $root/$PROG $B > $root/$OUT/bench_B$B.json 2> $root/$OUT/bench_B$B.err
f=$(find /tmp/tr_few -name "*kernel_stats.csv" | head -1)
cp $f $root/$OUT/kernel_stats_B$B.csv
t=$(find /tmp/tr_few -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-90s calls %6s  avg %8.2f us  %5s %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
# the timeline of a step from the dispatch records: for the kernel at each position of the repeating sequence, its duration and the gap to the previous end
tr = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0][-60:] for r in tr]
last = len(names) - 1
while last > 0 and "apply" not in names[last] and "reduce" not in names[last]: last -= 1
end = last
per = 1
while per < 40 and names[end - per] != names[end]: per += 1
seq = names[end - per + 1:end + 1]
nrep = 0; dur = [0.0] * per; gap = [0.0] * per
i = end
while i - per + 1 > 0 and names[i - per + 1:i + 1] == seq and nrep < 200:
    for j in range(per):
        r = tr[i - per + 1 + j]; prev = tr[i - per + j]
        dur[j] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); gap[j] += int(r["Start_Timestamp"]) - int(prev["End_Timestamp"])
    nrep += 1; i -= per
print("step timeline over %d steps (us): gap before | duration" % nrep)
for j in range(per): print("   %-60s %6.2f | %6.2f" % (seq[j], gap[j] / nrep / 1e3, dur[j] / nrep / 1e3))
print("   sum of durations %.2f, sum of gaps %.2f" % (sum(dur) / nrep / 1e3, sum(gap) / nrep / 1e3))
PY
