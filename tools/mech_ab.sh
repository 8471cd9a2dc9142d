#!/bin/bash
# A/B of the stand-alone mechanistic stage on ONE box: per configuration the un-profiled call time (tools/bench_mech.py) and the
# rocprofv3 kernel-trace durations of the streaming and the finish kernel.   usage (repo root, on the GPU box):
#   bash tools/mech_ab.sh OUTDIR "tag|ENV=... |bench_mech arguments" ...
set -u
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/$1; shift
mkdir -p $OUT
cd /tmp
for spec in "$@"; do
  IFS='|' read -r tag envs args <<< "$spec"
  env $envs python3 $ROOT/tools/bench_mech.py $args > $OUT/$tag.json 2> $OUT/$tag.err
  # (the program itself after --: an `env` hop under the profiler would be an exec from a process that has initialised the GPU)
  ( export $envs >/dev/null 2>&1; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$tag -- python3 $ROOT/tools/bench_mech.py $args > /dev/null 2> $OUT/trace_$tag.err )
  python3 - "$OUT" "$tag" <<'PY'
import csv, glob, json, sys
out, tag = sys.argv[1:3]
d = json.loads(open(f"{out}/{tag}.json").read().strip().splitlines()[-1])
row = f"{tag:28s} call {d['ms_per_call']*1e3:7.1f} us  frac {d['frac']:.4f}  with count {d['ms_per_call_with_counting_pass']*1e3:7.1f} us |"
for f in glob.glob(f"{out}/trace_{tag}/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "eh_mech" in r["Name"] or "eh_count" in r["Name"]:
            row += f" {r['Name'].split('(')[0][:34]} n={r['Calls']} avg {float(r['AverageNs'])/1e3:.1f} us;"
print(row)
PY
done
