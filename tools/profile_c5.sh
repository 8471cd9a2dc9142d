#!/bin/bash
# rocprofv3 kernel stats of BASELINE config 5 (1e7 resident, B = 65 536) in both bf16 modes; run on the GPU box from the repo root:
#   tools/profile_c5.sh [out dir]
set -u
export TMPDIR=/tmp
R=$PWD; OUT=$R/${1:-gpurun_out/prof_c5}
rm -rf $OUT; mkdir -p $OUT
cd /tmp
for p in bf16_fwd bf16; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$p -- python3 $R/tools/bench_config.py c5 --precision $p --steps 200 --specialize 1 > $OUT/c5_$p.json 2>/dev/null
  f=$(find $OUT/trace_$p -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/c5_${p}_kernel_stats.csv
  rm -rf $OUT/trace_$p
  cut -c1-200 $OUT/c5_$p.json; head -5 $OUT/c5_${p}_kernel_stats.csv | cut -c1-200
done
