#!/bin/bash
# kernel trace + PMC passes for config 5 ([32,128,128,6], tools/bench_config.py c5): tools/pmc_c5.sh <precision bf16|bf16_fwd> [batch] [out dir]
# run from the repo root on the GPU box; every pass prints its summary (nothing silent for minutes)
export TMPDIR=/tmp
PREC=${1:-bf16}; B=${2:-65536}; OUT=${3:-gpurun_out/pmc_c5_$PREC}
rm -rf $OUT; mkdir -p $OUT
ARGS="tools/bench_config.py c5 --precision $PREC --batch $B --n 2000000"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS --steps 200 > $OUT/trace.log 2>&1
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -6 "$f" && cp "$f" $OUT/kernel_stats.csv
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_VMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -c1-14 | tr " " _)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -- python3 $ARGS --steps 10 > $OUT/$tag.log 2>&1
  python3 tools/pmc_report.py $OUT/$tag | tee -a $OUT/pmc_summary.txt
done
