#!/bin/bash
# kernel trace + PMC passes for the [32,128,128,6] shape (tools/bench_config.py c5 [--batch $1]); run from the repo root on the GPU box
export TMPDIR=/tmp
OUT=gpurun_out/pmc_wide
rm -rf $OUT; mkdir -p $OUT
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_config.py c5 --batch ${1:-65536} --steps 50 --fused 0 --specialize 1 --n 2000000 > $OUT/trace.log 2>&1
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -6 "$f"
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_VMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -c1-14 | tr " " _)
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -- python3 tools/bench_config.py c5 --batch ${1:-65536} --steps 10 --fused 0 --specialize 1 --n 2000000 > /dev/null 2>&1
  python3 tools/pmc_report.py $OUT/$tag
done
