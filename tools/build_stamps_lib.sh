#!/bin/bash
# A second build of the library with in-kernel phase stamps (-DEH_STAMPS) in ONE translation unit, next to the normal one:
#   tools/build_stamps_lib.sh wide 2_8_2 [extra flags]     -> easyhybrid.jl_amd/libeasyhybrid_hip_stamps.so
# (EASYHYBRID_HIP_LIB=easyhybrid.jl_amd/libeasyhybrid_hip_stamps.so python tools/stamps_wide.py ...).  Every other object is the
# normal build's: the stamps pointer is part of EhStepArgs in every build, so the objects agree on the layout.
set -e
kind=$1; shape=$2; shift 2
here=$(cd "$(dirname "$0")/.." && pwd)
src=$here/easyhybrid.jl_amd/csrc
make -C $src -j8 >/dev/null
mkdir -p $src/build_stamps
flags=$(make -s -C $src print-cxxflags)
IFS=_ read nbi nbh nl <<< "$shape"
if [ "$kind" = wide ]; then file=eh_arch_wide.hip; else file=eh_arch.hip; fi
extra=""
if [ "$kind" = arch ] && [ "$nbi" = 1 ]; then extra="-DEH_FAST_PATHS"; fi
/opt/rocm/bin/hipcc $flags -DEH_STAMPS "$@" -DEH_NBI=$nbi -DEH_NBH=$nbh -DEH_NL=$nl $extra -c $src/$file -o $src/build_stamps/eh_${kind}_${shape}.o
objs=""
for o in $src/build/*.o; do
  b=$(basename $o)
  if [ "$b" = "eh_${kind}_${shape}.o" ]; then objs="$objs $src/build_stamps/$b"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $here/easyhybrid.jl_amd/libeasyhybrid_hip_stamps.so $objs -L/opt/rocm/lib -lhiprtc -ldl -Wl,-rpath,/opt/rocm/lib -Wl,-soname,libeasyhybrid_hip.so
echo built $here/easyhybrid.jl_amd/libeasyhybrid_hip_stamps.so
