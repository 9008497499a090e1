#!/bin/bash
# usage: tools/pmc_sq.sh <workload> <kernel-substring> [<kernel-substring> ...]   (GPU box) -- SQ counter passes for one bench workload
set -e
w=$1; shift; k="$*"
root=$(pwd); out=$root/gpurun_out/sq_$w; rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d "$out/p$i" -o p --output-format csv -- python3 "$root/bench.py" --workload $w --steps 1 --warmup 1 --no-cpu-baseline > "$out/p$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$out/p$i.log"; }
done
cd "$root"
python3 - "$out" $k <<'PY'
import csv, glob, sys, collections
out, ks = sys.argv[1], sys.argv[2:]
rows = [r for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))]
for k in ks:
    agg = collections.defaultdict(list)
    for r in rows:
        if k in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("== kernels matching %r" % k)
    for c, v in sorted(agg.items()):
        print("%-24s n=%3d mean=%.4g" % (c, len(v), sum(v) / len(v)))
PY
rm -rf "$out"/p*/
