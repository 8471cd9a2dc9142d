#!/bin/bash
# Round-4 evidence, run on the GPU box through gpurun from the repo root: rocprofv3 kernel-trace stats of the bench command, of config 5
# in both bf16 modes, of config 3 and of the evaluation kernel; separate PMC passes (HBM traffic; matrix-pipe utilisation); the
# un-profiled lines next to them; the primitives of the weak-scaling prediction.  Writes under gpurun_out/prof_r04/ ; the summaries are
# copied into profiles/r04/.
set -u
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_r04
rm -rf $OUT; mkdir -p $OUT
cd /tmp
prof() { tag=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$tag -- "$@" > $OUT/$tag.json 2> $OUT/$tag.err; }
pmc() { tag=$1; ctr=$2; shift 2; timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_$tag -- "$@" > /dev/null 2> $OUT/pmc_$tag.err; }
B="--no-cpu-baseline --no-mech-stage --no-epoch --no-layerwise --no-train-e2e"
# headline (the kernel specialised ahead of time runs by default)
prof bench python3 $ROOT/bench.py --steps 2000 --warmup 200 $B
pmc bench_fetch FETCH_SIZE python3 $ROOT/bench.py --steps 200 --warmup 20 $B
pmc bench_write WRITE_SIZE python3 $ROOT/bench.py --steps 200 --warmup 20 $B
echo "progress: headline profiled"
# config 5 as BASELINE states it (1e7 resident, B = 65 536) in both bf16 modes; config 3
for p in bf16_fwd bf16; do
  prof c5_$p python3 $ROOT/tools/bench_config.py c5 --precision $p --steps 200
  pmc c5_${p}_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES" python3 $ROOT/tools/bench_config.py c5 --precision $p --steps 20 --n 2000000
  pmc c5_${p}_fetch FETCH_SIZE python3 $ROOT/tools/bench_config.py c5 --precision $p --steps 20 --n 2000000
  pmc c5_${p}_write WRITE_SIZE python3 $ROOT/tools/bench_config.py c5 --precision $p --steps 20 --n 2000000
  echo "progress: config 5 $p profiled"
done
prof c3 python3 $ROOT/tools/bench_config.py c3 --steps 200 --fused 0
pmc c3_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES" python3 $ROOT/tools/bench_config.py c3 --steps 20 --fused 0
# what a user runs: eh.train end to end (and eh_eval's own kernel under the profiler)
prof train_e2e python3 $ROOT/tools/bench_train_e2e.py
echo "progress: end-to-end profiled"
cd $ROOT
timeout -k 10 400 python3 bench.py --steps 3000 --warmup 300 > $OUT/bench_3000steps.json 2> $OUT/bench_3000.err
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 $B > $OUT/bench_driver_sized_20steps.json 2>/dev/null
timeout -k 10 300 python3 bench.py --steps 3000 --warmup 300 $B --no-specialize > $OUT/bench_3000steps_no_specialize.json 2>/dev/null
EH_NO_AOT_SPEC=1 timeout -k 10 300 python3 bench.py --steps 3000 --warmup 300 $B > $OUT/bench_3000steps_hiprtc_kernel.json 2>/dev/null
EH_NO_AOT_SPEC=1 timeout -k 10 300 python3 bench.py --steps 3000 --warmup 300 $B --no-specialize > $OUT/bench_3000steps_generic_kernel.json 2>/dev/null
echo "progress: headline variants done"
{
  for p in bf16_fwd bf16 f32; do timeout -k 10 300 python3 tools/bench_config.py c5 --precision $p --steps 300; done
  for b in 16384 32768 131072 262144; do timeout -k 10 300 python3 tools/bench_config.py c5 --precision bf16 --batch $b --n 2097152 --steps 200; done
  timeout -k 10 300 python3 tools/bench_config.py c3 --steps 200 --fused 0
  timeout -k 10 300 python3 tools/bench_config.py c2 --steps 2000
  timeout -k 10 300 python3 tools/bench_config.py c1 --steps 2000
} > $OUT/bench_config_all.jsonl 2> $OUT/bench_config_all.err
timeout -k 10 300 python3 tools/dp_primitives.py > $OUT/dp_primitives.txt 2>&1
timeout -k 10 300 python3 tools/p2p_local_group.py 8 > $OUT/p2p_local_group_8.txt 2>&1
echo "progress: configs and primitives done"
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
                acc[k.split("(")[0][:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, dd in acc.items():
        print(os.path.basename(d), k, {c: (round(sum(v) / len(v), 2), len(v)) for c, v in sorted(dd.items())})
PY
for f in $(find $OUT -name "*kernel_stats.csv"); do cp $f $OUT/$(basename $(dirname $(dirname $f)) | sed s/trace_//)_kernel_stats.csv; done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete; find $OUT -name "*domain_stats.csv" -delete
find $OUT -type d -empty -delete
ls $OUT
