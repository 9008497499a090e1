#!/bin/bash
# Probe: generator launches of fewer than 256 workgroups leave whole CUs free, where the compaction of the
# previous batch can run beside the generator (it cannot share a CU with a 128-VGPR generator workgroup).
# usage: tools/partition_probe.sh [variant names in build_variants/ ...]  (on the GPU box; writes gpurun_out/partition_probe.txt)
out=gpurun_out/partition_probe.txt
: > $out
variants=${@:-default}
for v in $variants; do
  if [ "$v" = default ]; then unset JK_HIP_LIB; else export JK_HIP_LIB=$PWD/build_variants/lib_$v.so; fi
  for wg in 256 240 232 224 216; do
    bl=$((wg*1024))
    lanes=$((bl*4))
    echo "== $v: generator workgroups per launch: $wg (batch lanes $bl, lanes $lanes)" >> $out
    JK_BATCH_LANES=$bl python bench.py --steps 10 --warmup 2 --lanes $lanes --no-cpu-baseline --no-extras >> $out 2>&1
  done
done
