#!/bin/bash
# A/B of the headline step on ONE lease: the round-1 final library + bench (a copy under dbg/r01, built from commit a6913d7) against
# the current tree, alternating, long runs (3 000 steps) and the driver-sized form (20 steps).
# (needs a built copy of the round-1 tree next to this one, which is NOT tracked: git worktree add dbg/r01 a6913d7 && make -C dbg/r01/easyhybrid.jl_amd/csrc -j8)
if [ ! -d "$(dirname "$0")/../dbg/r01" ]; then echo "$0: dbg/r01 is missing (a worktree of commit a6913d7, built): see the comment at the top" >&2; exit 2; fi
set -u
ROOT=$PWD
OUT=$ROOT/$1; mkdir -p $OUT
pick() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get('roofline') or {}
print(sys.argv[2], 'ms_per_step', round(d['ms_per_step']*1e3,3), 'us; kernel', round(r.get('kernel_ms',0)*1e3,3), 'us;', r.get('kernel_build'))" $1 $2; }
for rep in 1 2 3; do
  ( cd $ROOT/dbg/r01 && python3 bench.py --steps 3000 --warmup 300 --no-cpu-baseline --no-mech-stage > $OUT/r01_long_$rep.json 2>$OUT/r01_long_$rep.err ); pick $OUT/r01_long_$rep.json r01_long_$rep
  python3 bench.py --steps 3000 --warmup 300 --no-cpu-baseline --no-mech-stage --no-epoch > $OUT/cur_long_$rep.json 2>/dev/null; pick $OUT/cur_long_$rep.json cur_long_$rep
  ( cd $ROOT/dbg/r01 && python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mech-stage > $OUT/r01_20_$rep.json 2>/dev/null ); pick $OUT/r01_20_$rep.json r01_20_$rep
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mech-stage --no-epoch > $OUT/cur_20_$rep.json 2>/dev/null; pick $OUT/cur_20_$rep.json cur_20_$rep
done
