#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace stats of the bench command plus separate
# PMC passes for HBM traffic.  Writes under gpurun_out/prof_$1/ ; copy the summaries into profiles/.
set -u
export TMPDIR=/tmp
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 2000 --warmup 200 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > /dev/null 2> $OUT/pmc_sq.err
timeout 300 python3 bench.py --steps 3000 --warmup 300 > $OUT/bench.json 2> $OUT/bench.err
# the other BASELINE shapes: kernel trace of the front-door step loop (tools/bench_config.py), run-time specialised kernels like the bench;
# bench_config_<cfg>_aot.json = the kernels built ahead of time, for comparison
for cfg in c3 c5; do timeout 120 python3 tools/bench_config.py $cfg --steps 200 --fused 0 --specialize 0 > $OUT/bench_config_${cfg}_aot.json 2>/dev/null; done
timeout 120 python3 bench.py --steps 3000 --warmup 300 --no-cpu-baseline --no-specialize > $OUT/bench_aot.json 2>/dev/null
for cfg in c3 c5; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$cfg -- python3 tools/bench_config.py $cfg --steps 200 --fused 0 --specialize 1 > $OUT/bench_config_$cfg.json 2> $OUT/trace_$cfg.err
done
# MFMA utilisation and HBM traffic of the MFMA-bound shapes (separate PMC passes)
for cfg in c3 c5; do
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_mfma_$cfg -- python3 tools/bench_config.py $cfg --steps 20 --fused 0 --specialize 1 > /dev/null 2> $OUT/pmc_mfma_$cfg.err
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$cfg -- python3 tools/bench_config.py $cfg --steps 20 --fused 0 --specialize 1 > /dev/null 2> $OUT/pmc_fetch_$cfg.err
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$cfg -- python3 tools/bench_config.py $cfg --steps 20 --fused 0 --specialize 1 > /dev/null 2> $OUT/pmc_write_$cfg.err
done
timeout 120 python3 tools/bench_config.py c5 --steps 100 --batch 262144 --specialize 1 > $OUT/bench_config_c5_b262144.json 2>/dev/null
timeout 120 python3 tools/bench_config.py c5 --steps 30 --batch 1048576 --nbatches 2 --specialize 1 > $OUT/bench_config_c5_b1048576.json 2>/dev/null
timeout 120 python3 tools/bench_config.py c3 --steps 100 --batch 1048576 --nbatches 2 --fused 0 --specialize 1 > $OUT/bench_config_c3_b1048576.json 2>/dev/null
timeout 120 python3 tools/bench_config.py c2 --steps 300 --batch 1048576 --nbatches 4 --specialize 1 > $OUT/bench_config_c2_b1048576.json 2>/dev/null
timeout 120 python3 tools/bench_config.py c1 --steps 1000 --specialize 1 > $OUT/bench_config_c1.json 2>/dev/null
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/trace*/**/*kernel_stats.csv", recursive=True)):
    print(f); print(open(f).read())
for name in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_mfma_c3", "pmc_fetch_c3", "pmc_write_c3", "pmc_mfma_c5", "pmc_fetch_c5", "pmc_write_c5"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "eh_" in k:
                acc[k.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(name, k, {c: (round(sum(v) / len(v), 2), len(v)) for c, v in sorted(d.items())})
PY
# keep only small summaries for merging back
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +2M -delete
