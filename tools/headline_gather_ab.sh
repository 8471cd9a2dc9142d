#!/bin/bash
# A/B on one lease: the headline step with the round-5 gather (live waves only) against the form of rounds 1-4 (EH_AB_FULL_GATHER), both as
# run-time compiled kernels of the same source, alternating; then the kernel built ahead of time
B="--no-cpu-baseline --no-mech-stage --no-epoch --no-layerwise --no-train-e2e --steps 3000 --warmup 300"
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(1e3*d["ms_per_step"],3), "us/step; kernel", round(1e3*d["roofline"]["kernel_ms"],3))'
for i in 1 2 3; do
  for d in "" "EH_AB_FULL_GATHER"; do
    echo -n "jit defines [$d]: "; EH_NO_AOT_SPEC=1 EH_JIT_DEFINES="$d" timeout -k 10 200 python3 bench.py $B 2>/dev/null | tail -1 | python3 -c "$P"
  done
done
echo -n "ahead of time: "; timeout -k 10 200 python3 bench.py $B 2>/dev/null | tail -1 | python3 -c "$P"
