#!/bin/bash
# PMC passes over the config 3 step kernel.  usage: bash tools/pmc_c3.sh [variant] [specialize]
export TMPDIR=/tmp
V=${1:-0}; S=${2:-1}
OUT=gpurun_out/pmc_c3_v$V
mkdir -p $OUT
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_VMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS"; do
  tag=$(echo $set | cut -c1-14 | tr " " _)
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -- python3 tools/bench_config.py c3 --steps 20 --fused 0 --variant $V --specialize $S > /dev/null 2>&1
  python3 tools/pmc_report.py $OUT/$tag
done
