#!/bin/bash
# Round-3 evidence, run on the GPU box through gpurun from the repo root: rocprofv3 kernel-trace stats of the bench command, of the
# stand-alone mechanistic stage and of the other BASELINE shapes; separate PMC passes (HBM traffic; matrix-pipe utilisation);
# un-profiled numbers next to them.  Writes under gpurun_out/prof_r03/ ; the summaries are copied into profiles/r03/.
set -u
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_r03
rm -rf $OUT; mkdir -p $OUT
cd /tmp
prof() { tag=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$tag -- "$@" > $OUT/$tag.json 2> $OUT/$tag.err; }
pmc() { tag=$1; ctr=$2; shift 2; timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_$tag -- "$@" > /dev/null 2> $OUT/pmc_$tag.err; }
B="--no-cpu-baseline --no-mech-stage --no-epoch --no-layerwise"
# headline
prof bench python3 $ROOT/bench.py --steps 2000 --warmup 200 $B
pmc bench_fetch FETCH_SIZE python3 $ROOT/bench.py --steps 200 --warmup 20 $B
pmc bench_write WRITE_SIZE python3 $ROOT/bench.py --steps 200 --warmup 20 $B
pmc bench_sq "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" python3 $ROOT/bench.py --steps 200 --warmup 20 $B
# the path train() runs: shuffled epochs (records gathered through the device-side permutation), traffic next to the contiguous form
pmc epoch_shuffled_fetch FETCH_SIZE python3 $ROOT/tools/bench_epoch.py --mode shuffled --epochs 2
pmc epoch_contiguous_fetch FETCH_SIZE python3 $ROOT/tools/bench_epoch.py --mode contiguous --epochs 2
timeout -k 10 300 python3 $ROOT/tools/bench_epoch.py > $OUT/bench_epoch.json 2>/dev/null
timeout -k 10 300 python3 $ROOT/bench.py --steps 3000 --warmup 300 > $OUT/bench_3000steps.json 2> $OUT/bench_3000.err
timeout -k 10 300 python3 $ROOT/bench.py --steps 20 --warmup 5 > $OUT/bench_driver_sized_20steps.json 2>/dev/null
timeout -k 10 300 python3 $ROOT/bench.py --steps 3000 --warmup 300 $B --no-specialize > $OUT/bench_3000steps_ahead_of_time_kernels.json 2>/dev/null
# the stand-alone mechanistic stage (north_star: >= 60 % of HBM for the mechanistic + VJP kernel), 1.07 GB per call
prof mech_stage python3 $ROOT/tools/bench_mech.py --batch 67108864 --steps 40
pmc mech_stage_fetch FETCH_SIZE python3 $ROOT/tools/bench_mech.py --batch 67108864 --steps 10
pmc mech_stage_write WRITE_SIZE python3 $ROOT/tools/bench_mech.py --batch 67108864 --steps 10
for m in expo2pool fluxpart; do timeout -k 10 300 python3 $ROOT/tools/bench_mech.py --batch 67108864 --steps 20 --mech $m >> $OUT/mech_stage_other_models.jsonl 2>/dev/null; done
timeout -k 10 300 python3 $ROOT/tools/bench_mech.py --batch 16777216 --steps 40 >> $OUT/mech_stage_other_models.jsonl 2>/dev/null
# config 5 as BASELINE states it and its fp32 twin; config 3; the tutorial net
for p in bf16_fwd f32; do
  prof c5_$p python3 $ROOT/tools/bench_config.py c5 --precision $p --steps 200 --n 2000000 --specialize 1
  pmc c5_${p}_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES" python3 $ROOT/tools/bench_config.py c5 --precision $p --steps 20 --n 2000000 --specialize 1
  pmc c5_${p}_fetch FETCH_SIZE python3 $ROOT/tools/bench_config.py c5 --precision $p --steps 20 --n 2000000 --specialize 1
  pmc c5_${p}_write WRITE_SIZE python3 $ROOT/tools/bench_config.py c5 --precision $p --steps 20 --n 2000000 --specialize 1
done
prof c3 python3 $ROOT/tools/bench_config.py c3 --steps 200 --fused 0 --specialize 1
pmc c3_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES" python3 $ROOT/tools/bench_config.py c3 --steps 20 --fused 0 --specialize 1
prof lform python3 $ROOT/tools/bench_lform.py 65536
cd $ROOT
{
  timeout -k 10 300 python3 tools/bench_config.py c5 --steps 300 --specialize 1                       # N = 1e7 resident (152 batches)
  timeout -k 10 300 python3 tools/bench_config.py c5 --precision f32 --steps 200 --n 2000000 --specialize 1
  timeout -k 10 300 python3 tools/bench_config.py c5 --steps 100 --batch 262144 --n 2097152 --specialize 1
  timeout -k 10 300 python3 tools/bench_config.py c3 --steps 200 --fused 0 --specialize 1
  timeout -k 10 300 python3 tools/bench_config.py c2 --steps 2000 --specialize 1
  timeout -k 10 300 python3 tools/bench_config.py c1 --steps 2000 --specialize 1
} > $OUT/bench_config_all.jsonl 2> $OUT/bench_config_all.err
timeout -k 10 300 python3 tools/bench_lform.py > $OUT/bench_lform.jsonl 2>&1
for b in 64 1024 4096; do bash tools/lform_trace.sh gpurun_out/prof_r03/lform_trace $b > $OUT/lform_trace_B$b.txt 2>&1; done
hipcc --offload-arch=gfx950 -O3 -w tools/ubench/stream31.hip -o /tmp/ubench_stream31 && timeout -k 5 120 /tmp/ubench_stream31 > $OUT/ubench_stream31.txt 2>&1
python3 - "$OUT" > $OUT/pmc_summary.txt <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
for d in sorted(glob.glob(out + "/pmc_*")):
    if not os.path.isdir(d): continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "eh_" in k:
                acc[k.split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, dd in acc.items():
        print(os.path.basename(d), k, {c: (round(sum(v) / len(v), 2), len(v)) for c, v in sorted(dd.items())})
PY
for f in $(find $OUT -name "*kernel_stats.csv"); do cp $f $OUT/$(basename $(dirname $(dirname $f)))_kernel_stats.csv; done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete; find $OUT -name "*domain_stats.csv" -delete
ls $OUT
