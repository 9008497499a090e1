for i in 1 2 3; do
for v in old new; do
  if [ $v = old ]; then export JK_HIP_LIB=$PWD/variants/libjk_old.so; else unset JK_HIP_LIB; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > gpurun_out/ab_$v.json && python -c "
import json; d=json.load(open('gpurun_out/ab_$v.json')); print('$v', d['value'], d['step_ms']['median'], d['roofline']['kernel_ms'])"
done; done
