#!/bin/bash
# Packed reference on / off (JK_PACKED_REF) at three genome sizes and on the haplotype workload: bench lines into gpurun_out/packed_probe.txt
out=gpurun_out/packed_probe.txt
: > $out
for mode in 1 0; do
  for mbp in 100 1000 3000; do
    echo "== JK_PACKED_REF=$mode genome $mbp Mbp" >> $out
    JK_PACKED_REF=$mode timeout -k 10 300 python bench.py --genome-mbp $mbp --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | cut -c1-330 >> $out || exit 1
  done
  echo "== JK_PACKED_REF=$mode haplotype workload" >> $out
  JK_PACKED_REF=$mode timeout -k 10 300 python bench.py --workload hap --no-extras 2>/dev/null | cut -c1-420 >> $out || exit 1
done
