#!/bin/bash
# A/B of two library builds on the haplotype bench line, alternating, in one GPU-box call.
# usage: tools/ab_hap.sh <variant in build_variants/> [bench args...]   -> gpurun_out/ab_hap.txt
v=$1; shift
out=gpurun_out/ab_hap.txt
: > $out
for rep in 1 2 3; do
  echo "== A default" >> $out; python bench.py "$@" >> $out 2>&1
  echo "== B $v" >> $out; JK_HIP_LIB=$PWD/build_variants/lib_$v.so python bench.py "$@" >> $out 2>&1
done
