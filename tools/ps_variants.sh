#!/bin/bash
# Diagnostic builds for the P <= 4 weight-gradient path on shapes wider than one block (tools/ps_relu_repro.py):
# the two translation units the reproducer uses, rebuilt with -DEH_PS_WIDE plus one experimental flag set each, linked with the
# objects of the normal build into dbg/lib_<name>.so.   usage: tools/ps_variants.sh name "flags" [name "flags" ...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/easyhybrid.jl_amd/csrc
make -C $CS -j8 >/dev/null
# the flags of the normal build (incl. -fno-slp-vectorize, the fix these variants were written to find) come from the Makefile: pass
# e.g. "-fslp-vectorize" as a variant's flags to get the failing build back
BASE=$(make -C $CS -s --no-print-directory print-cxxflags)
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  out=$ROOT/dbg/$name; mkdir -p $out
  for s in 1_4_2 1_4_3; do
    IFS=_ read a b c <<< "$s"
    ( /opt/rocm/bin/hipcc $BASE -DEH_NBI=$a -DEH_NBH=$b -DEH_NL=$c -DEH_FAST_PATHS $flags \
        -c $CS/eh_arch.hip -o $out/eh_arch_$s.o ) &
  done
  wait
  objs=$(ls $CS/build/*.o | grep -v -e eh_arch_1_4_2.o -e eh_arch_1_4_3.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $ROOT/dbg/lib_$name.so $objs $out/eh_arch_1_4_2.o $out/eh_arch_1_4_3.o -L/opt/rocm/lib -lhiprtc -ldl -Wl,-rpath,/opt/rocm/lib
  echo built dbg/lib_$name.so
done
