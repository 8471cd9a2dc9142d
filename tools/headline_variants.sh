#!/bin/bash
# Which change since round 1 costs the headline step its 0.45 us?  One lease, the current library, the step kernel recompiled at run
# time with single changes switched back (EH_JIT_DEFINES / EH_JIT_SLP), 3 000-step bench each, two rounds; dbg/r01 alongside.
# (needs a built copy of the round-1 tree next to this one, which is NOT tracked: git worktree add dbg/r01 a6913d7 && make -C dbg/r01/easyhybrid.jl_amd/csrc -j8)
if [ ! -d "$(dirname "$0")/../dbg/r01" ]; then echo "$0: dbg/r01 is missing (a worktree of commit a6913d7, built): see the comment at the top" >&2; exit 2; fi
set -u
ROOT=$PWD
OUT=$ROOT/$1; mkdir -p $OUT
run() { tag=$1; shift; env "$@" python3 bench.py --steps 3000 --warmup 300 --no-cpu-baseline --no-mech-stage --no-epoch > $OUT/$tag.json 2>$OUT/$tag.err
  python3 -c "
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get('roofline') or {}
    print('%-34s ms_per_step %.3f us; kernel %.3f us; %s' % (sys.argv[2], d['ms_per_step']*1e3, r.get('kernel_ms',0)*1e3, r.get('kernel_build')))
except Exception as e: print(sys.argv[2], 'FAILED', e)" $OUT/$tag.json $tag; }
for rep in 1 2; do
  ( cd $ROOT/dbg/r01 && python3 bench.py --steps 3000 --warmup 300 --no-cpu-baseline --no-mech-stage > $OUT/r01_$rep.json 2>/dev/null; python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('%-34s ms_per_step %.3f us' % ('r01_'+sys.argv[2], d['ms_per_step']*1e3))" $OUT/r01_$rep.json $rep )
  run base_$rep A=1
  run no_sched_fence_$rep EH_JIT_DEFINES=EH_NO_SCHED_FENCE
  run release_only_wave_sync_$rep EH_JIT_DEFINES=EH_SYNC_ORDER=__ATOMIC_RELEASE
  run vector_wave_index_$rep EH_JIT_DEFINES=EH_AB_VECTOR_WAVE_INDEX
  run slp_on_$rep EH_JIT_SLP=1
  run all_four_$rep EH_JIT_SLP=1 "EH_JIT_DEFINES=EH_NO_SCHED_FENCE EH_SYNC_ORDER=__ATOMIC_RELEASE EH_AB_VECTOR_WAVE_INDEX"
done
