# PacBio: waves of the plan kernel per CU (16 = every SIMD four deep, no room for the emit kernel's waves beside them)
for w in 16 12 8 16 12; do
  JK_PB_WAVES_PER_CU=$w timeout -k 10 300 python bench.py --workload pacbio --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | tail -1 > gpurun_out/pb_wpc_$w.json && python -c "
import json; d=json.load(open('gpurun_out/pb_wpc_$w.json')); print('waves/CU $w:', d['value'], 'M reads/s', d['ms_per_step'], 'ms/step', d['roofline']['launches_per_step'], 'launches', d['roofline']['kernel_ms'], d['roofline'].get('plan_kernel_ms'))"
done
