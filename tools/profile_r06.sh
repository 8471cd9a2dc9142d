#!/bin/bash
# Round-6 evidence (tools/profile_r05.sh + the layer-wise few-rows step, train() end to end part by part, the two forms of the headline step)
# Round-5 evidence, run on the GPU box through gpurun from the repo root (one lease): rocprofv3 kernel-trace stats of every kernel a fraction
# on the bench line is quoted for -- headline, config 5 in both bf16 modes (the sample-owned kernel and the row-split one), config 3, the
# mechanistic stage on 64 M samples, the layer-wise tutorial net at B = 65 536, eh_eval over the headline split, the multi-step launches --,
# separate PMC passes (HBM traffic; VALU / MFMA instruction and busy counters of config 5 and of the evaluation kernel), the un-profiled
# lines next to them, the primitives of the weak-scaling prediction in both publishing modes.  Writes under gpurun_out/prof_r06/; the
# summaries are copied into profiles/r06/.  Every stage prints a progress line (nothing silent for minutes).
set -u
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_r06
rm -rf $OUT; mkdir -p $OUT
cd /tmp
prof() { tag=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$tag -- "$@" > $OUT/$tag.json 2> $OUT/$tag.err; echo "progress: traced $tag"; }
pmc() { tag=$1; ctr=$2; shift 2; timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_$tag -- "$@" > /dev/null 2> $OUT/pmc_$tag.err; echo "progress: counters $tag"; }
B="--no-cpu-baseline --no-mech-stage --no-epoch --no-layerwise --no-train-e2e"
C5="python3 $ROOT/tools/bench_config.py c5"
SQ1="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_VMEM"
SQ2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
SQ3="SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
PART=${PART:-all}      # PART=1: traces and counters, PART=2: the un-profiled lines, stamps, primitives, soak (a gpurun call is 20 minutes at most)
if [ "$PART" != "2" ]; then
# headline
prof bench python3 $ROOT/bench.py --steps 2000 --warmup 200 $B
pmc bench_fetch FETCH_SIZE python3 $ROOT/bench.py --steps 200 --warmup 20 $B
pmc bench_write WRITE_SIZE python3 $ROOT/bench.py --steps 200 --warmup 20 $B
# config 5 as BASELINE states it (1e7 resident, B = 65 536): "bf16" on the sample-owned kernel (default) and on the row-split one, "bf16_fwd"
prof c5_bf16 $C5 --precision bf16 --steps 200
EH_NO_SAMPLE_OWNED=1 prof c5_bf16_rowsplit $C5 --precision bf16 --steps 200
prof c5_bf16_fwd $C5 --precision bf16_fwd --steps 200
for set in "$SQ1" "$SQ2" "$SQ3" FETCH_SIZE WRITE_SIZE; do
  tag=$(echo $set | cut -c1-14 | tr " " _)
  pmc c5_bf16_$tag "$set" $C5 --precision bf16 --steps 10 --n 2000000
done
EH_NO_SAMPLE_OWNED=1 pmc c5_bf16_rowsplit_insts "$SQ1" $C5 --precision bf16 --steps 10 --n 2000000
pmc c5_bf16_fwd_insts "$SQ1" $C5 --precision bf16_fwd --steps 10 --n 2000000
# config 3
prof c3 python3 $ROOT/tools/bench_config.py c3 --steps 200 --fused 0
# mechanistic stage on 64 M samples (1.07 GB per call), layer-wise tutorial net at B = 65 536, evaluation over the headline split
prof mech_stage_64M python3 $ROOT/tools/bench_mech.py --batch 67108864 --steps 20
prof lform_B65536 python3 $ROOT/tools/bench_lform.py 65536
prof lform_B64 python3 $ROOT/tools/bench_lform.py 64
prof eval python3 $ROOT/tools/bench_eval.py --config c2
for set in "$SQ1" "$SQ2" FETCH_SIZE WRITE_SIZE; do
  tag=$(echo $set | cut -c1-14 | tr " " _)
  pmc eval_$tag "$set" python3 $ROOT/tools/bench_eval.py --config c2
done
# what a user runs: eh.train end to end; epochs of one-workgroup minibatches as single and as multi-step launches
prof train_e2e python3 $ROOT/tools/bench_train_e2e.py
prof multistep python3 $ROOT/tools/multistep_probe.py
fi
cd $ROOT
if [ "$PART" != "1" ]; then
timeout -k 10 400 python3 bench.py --steps 3000 --warmup 300 > $OUT/bench_3000steps.json 2> $OUT/bench_3000.err; echo "progress: bench line"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 $B > $OUT/bench_driver_sized_20steps.json 2>/dev/null
{
  for p in bf16_fwd bf16 f32; do timeout -k 10 300 python3 tools/bench_config.py c5 --precision $p --steps 300; done
  EH_NO_SAMPLE_OWNED=1 timeout -k 10 300 python3 tools/bench_config.py c5 --precision bf16 --steps 300
  for b in 16384 32768 131072 262144; do timeout -k 10 300 python3 tools/bench_config.py c5 --precision bf16 --batch $b --n 2097152 --steps 200; done
  timeout -k 10 300 python3 tools/bench_config.py c3 --steps 200 --fused 0
  timeout -k 10 300 python3 tools/bench_config.py c2 --steps 2000
  timeout -k 10 300 python3 tools/bench_config.py c2 --batch 1048576 --steps 500
  timeout -k 10 300 python3 tools/bench_config.py c1 --steps 2000
} > $OUT/bench_config_all.jsonl 2> $OUT/bench_config_all.err; echo "progress: configs"
timeout -k 10 300 python3 tools/multistep_probe.py > $OUT/multistep_probe.txt 2>&1
{ timeout -k 10 200 python3 tools/bench_lform.py 16 64 128 256 300 1024
  for v in EH_LFORM_NOFIRST EH_LFORM_NOKEEP EH_LFORM_NOAPPLY64 EH_LFORM_NOAPPLY EH_LFORM_NOTAIL; do
    echo "== $v=1 (first launch = prep + first layer + second layer | chain kernel with the weights in registers | 64-row weight-gradient + optimiser launch | optimiser in the weight-gradient launch at all | chain kernel at all: round 5)"
    env $v=1 timeout -k 10 200 python3 tools/bench_lform.py 64 256
  done; } > $OUT/lform_few_rows_ab.txt 2>&1
bash tools/lform_trace_few.sh 64 gpurun_out/prof_r06/lform_few > $OUT/lform_few_timeline_B64.txt 2>&1
bash tools/lform_pmc_few.sh gpurun_out/prof_r06/lform_pmc > $OUT/lform_few_counters_B64.txt 2>&1
{ S=easyhybrid.jl_amd/libeasyhybrid_hip_stamps.so
  echo "== first launch (workgroup (0, 0)): 0 entry | 1 weights requested | 2 statistics read | 3 barrier | 4 statistics done | 5 rows normalised | 6 first layer on the fly (sigmoid) | 7 MFMAs | 8 barrier | 9 end"
  EH_STAMP_FIRST=1 EASYHYBRID_HIP_LIB=$S timeout -k 10 200 python3 tools/stamps_lform.py 64 | tail -11
  echo "== chain kernel (workgroup 0): 0 entry | 1 arguments | 2 row in LDS | 12-14, 6 inside the first forward layer | 3 4 forward layers | 5 output layer | 8 mechanistic stage | 9 delta across the output layer | 10 15 delta products"
  EASYHYBRID_HIP_LIB=$S timeout -k 10 200 python3 tools/stamps_lform.py 64 | tail -15
  for w in 8 100 400 680; do echo "== weight gradients + optimiser, workgroup $w: 12 tables | 13 everything requested | 7-11 arrivals | 1 MFMAs | 2 3 fold | 4 5 update | 6 stores"; EH_STAMP_DW=$w EASYHYBRID_HIP_LIB=$S timeout -k 10 200 python3 tools/stamps_lform.py 64 | tail -16; done; } > $OUT/stamps_lform.txt 2>&1
timeout -k 10 300 python3 tools/e2e_spikes.py 2>&1 | grep -v amdgpu.ids > $OUT/train_e2e_calls.txt
timeout -k 10 300 python3 tools/e2e_breakdown.py > $OUT/train_e2e_breakdown.txt 2>&1
timeout -k 10 300 python3 tools/bench_step_modes.py > $OUT/headline_step_modes.json 2>/dev/null
echo "progress: round-6 probes"
for p in 2; do EH_NO_AOT_SPEC=1 EH_SPECIALIZE=1 EH_JIT_DEFINES=EH_STAMPS EH_PRECISION=$p timeout -k 10 200 python3 tools/stamps_bfs.py 65536; done > $OUT/stamps_c5_bf16.txt 2>&1
timeout -k 10 300 python3 tools/dp_primitives.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" > $OUT/dp_primitives.txt
for m in 0 1; do EH_TOOL_P2P_MODE=$m timeout -k 10 300 python3 tools/p2p_local_group.py 8; done > $OUT/p2p_local_group_8.txt 2>&1
EH_JIT_DEFINES="EH_STAMPS EH_STAMPS_PROLOGUE" EH_JIT_CACHE=0 timeout -k 10 300 python3 tools/stamps_p2p.py 2>&1 | grep -v amdgpu.ids > $OUT/stamps_prologue_p2p.txt
for bn in "" 1; do echo "== input BatchNorm + sigmoid: ${bn:-0}"; EH_TOOL_BN=$bn EH_JIT_DEFINES="EH_STAMPS EH_STAMPS_FINE" EH_SPECIALIZE=1 EH_JIT_CACHE=0 EH_NO_AOT_SPEC=1 timeout -k 10 200 python3 tools/stamps_multistep.py 2>&1 | grep -v amdgpu.ids; done > $OUT/stamps_multistep.txt
{ echo "== sums stored straight into pinned host memory (default)"; timeout -k 10 300 python3 tools/bench_eval.py --config c2; echo "== EH_EVAL_COPY=1: copied out of the slab behind the kernel (rounds 1-4)"; EH_EVAL_COPY=1 timeout -k 10 300 python3 tools/bench_eval.py --config c2; } > $OUT/eval_zero_copy_ab.txt 2>&1
echo "progress: primitives"
for n in 2 4; do
  EH_DP_P2P_MODE=1 EH_SOAK_STEPS=200000 timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29570 + n)) tools/p2p_two_ranks.py 2>&1 | sed "s/rank /\nrank /g" | grep "^rank "
  echo "progress: soak $n ranks" >&2
done > $OUT/p2p_soak_mode1.txt
fi
if [ "$PART" != "2" ]; then
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
                acc[k.split("(")[0][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, dd in acc.items():
        print(os.path.basename(d), k, {c: (round(sum(v) / len(v), 2), len(v)) for c, v in sorted(dd.items())})
PY
fi
for f in $(find $OUT -name "*kernel_stats.csv"); do cp $f $OUT/$(basename $(dirname $(dirname $f)) | sed s/trace_//)_kernel_stats.csv; done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete; find $OUT -name "*domain_stats.csv" -delete
find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete
ls $OUT
