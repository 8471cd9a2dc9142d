#!/bin/bash
# counters of the layer-wise few-rows step's kernels (batch 64): cache / translation / memory-side requests per launch
# usage (GPU box, repo root): bash tools/lform_pmc_few.sh [out_dir]
OUT=${1:-gpurun_out/lform_pmc}
root=$(pwd)
mkdir -p $root/$OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQC_DCACHE_REQ SQC_DCACHE_MISSES SQC_ICACHE_REQ SQC_ICACHE_MISSES"; do
  i=$((i+1))
  rm -rf /tmp/pmc_few
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_few -- python3 $root/tools/bench_lform.py 64 > /dev/null 2> $root/$OUT/pass$i.err
  f=$(find /tmp/pmc_few -name "*counter_collection.csv" | head -1)
  echo "== pass $i: $set"
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][-50:]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in acc:
    if "rocclr" in k or "image" in k: continue
    print("   %-52s %s" % (k, "  ".join("%s=%.0f" % (c, acc[k][c] / n[k][c]) for c in sorted(acc[k]))))
PY
done
