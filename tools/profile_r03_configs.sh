#!/bin/bash
# Round-3 evidence, the part re-collected after the last kernel change of the round (the workgroup reduction of the per-wave kernel):
# rocprofv3 kernel stats of config 3 and the un-profiled lines of the other BASELINE configs.  Run on the GPU box from the repo root.
set -u
export TMPDIR=/tmp
R=$PWD; OUT=$R/gpurun_out/prof_r03c
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c3 -- python3 $R/tools/bench_config.py c3 --steps 200 --fused 0 --specialize 1 > $OUT/c3.json 2>/dev/null
cd $R
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/c3_kernel_stats.csv \;
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete; find $OUT -name "*domain_stats.csv" -delete
{
  timeout -k 10 300 python3 tools/bench_config.py c5 --steps 300 --specialize 1
  timeout -k 10 300 python3 tools/bench_config.py c5 --precision f32 --steps 200 --n 2000000 --specialize 1
  timeout -k 10 300 python3 tools/bench_config.py c5 --steps 100 --batch 262144 --n 2097152 --specialize 1
  timeout -k 10 300 python3 tools/bench_config.py c3 --steps 200 --fused 0 --specialize 1
  timeout -k 10 300 python3 tools/bench_config.py c2 --steps 2000 --specialize 1
  timeout -k 10 300 python3 tools/bench_config.py c1 --steps 2000 --specialize 1
} > $OUT/bench_config_all.jsonl 2>/dev/null
cut -c1-150 $OUT/bench_config_all.jsonl
head -3 $OUT/c3_kernel_stats.csv | cut -c1-160
