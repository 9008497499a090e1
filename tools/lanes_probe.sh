#!/bin/bash
# Probe: at a fixed generator launch size (JK_BATCH_LANES = wg * 1024), more lanes = more and shorter launches of the
# same job: the generator total stays (pairs / lanes-per-launch pair slots) while the un-overlapped tail (the last
# launch's compaction) shrinks.   usage: tools/lanes_probe.sh [wg]   (GPU box; writes gpurun_out/lanes_probe.txt)
out=gpurun_out/lanes_probe.txt
: > $out
wg=${1:-224}
bl=$((wg*1024))
for n in 4 6 8 12 16; do
  lanes=$((bl*n))
  echo "== wg $wg launches $n lanes $lanes" >> $out
  JK_BATCH_LANES=$bl python bench.py --steps 10 --warmup 2 --lanes $lanes --no-cpu-baseline --no-extras >> $out 2>&1
done
