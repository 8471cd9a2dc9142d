#!/bin/bash
# The layer-wise form's few-rows chain kernel (eh_lform_tailchain_kernel) with phase stamps: a second library whose eh_api.o is built with
# -DEH_STAMPS, then tools/stamps_lform.py prints the segments.   usage (repo root): bash tools/stamps_lform.sh   (build here, run on the GPU box)
set -e
here=$(cd "$(dirname "$0")/.." && pwd)
src=$here/easyhybrid.jl_amd/csrc
make -C $src -j8 >/dev/null
mkdir -p $src/build_stamps
flags=$(make -s -C $src print-cxxflags)
/opt/rocm/bin/hipcc $flags -DEH_STAMPS "$@" -c $src/eh_api.hip -o $src/build_stamps/eh_api.o
objs=""
for o in $src/build/*.o; do
  b=$(basename $o)
  if [ "$b" = "eh_api.o" ]; then objs="$objs $src/build_stamps/$b"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $here/easyhybrid.jl_amd/libeasyhybrid_hip_stamps.so $objs -L/opt/rocm/lib -lhiprtc -ldl -Wl,-rpath,/opt/rocm/lib -Wl,-soname,libeasyhybrid_hip.so
echo built $here/easyhybrid.jl_amd/libeasyhybrid_hip_stamps.so
