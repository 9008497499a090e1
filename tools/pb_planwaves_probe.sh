# PacBio: the plan kernel of launch b + 1 beside the emit kernel of launch b (JK_PB_SERIAL=0) or after it (=1), default heuristic
for v in default 0 1; do
  if [ $v = default ]; then unset JK_PB_SERIAL; else export JK_PB_SERIAL=$v; fi
  echo "== JK_PB_SERIAL: $v"
  timeout -k 10 400 python tools/pb_workloads_probe.py 2>&1 | grep -v amdgpu.ids
done
