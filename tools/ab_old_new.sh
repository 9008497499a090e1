# usage: tools/ab_old_new.sh <workload> [rounds]   (GPU box) -- the library in variants/libjk_old.so against the one in the tree, alternating
w=${1:-illumina}; n=${2:-3}
for i in $(seq $n); do
for v in old new; do
  if [ $v = old ]; then export JK_HIP_LIB=$PWD/variants/libjk_old.so; else unset JK_HIP_LIB; fi
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-extras --steps 4 --warmup 1 2>/dev/null | tail -1 > gpurun_out/ab_$v.json && python -c "
import json; d=json.load(open('gpurun_out/ab_$v.json')); print('$v', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done; done
