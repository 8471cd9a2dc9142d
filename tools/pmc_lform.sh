#!/bin/bash
# PMC passes over the layer-wise form's kernels (tutorial net, B = 65536)
export TMPDIR=/tmp
OUT=gpurun_out/pmc_lform
rm -rf $OUT; mkdir -p $OUT
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $set | cut -c1-14 | tr " " _)
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -- python3 tools/bench_lform.py 65536 > /dev/null 2>&1
  python3 - $OUT/$tag <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm" in k: acc[k.split("(")[0][-50:] + " grid=" + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(d.items())}, len(next(iter(d.values()))))
PY
done
