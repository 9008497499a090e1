#!/bin/bash
# The PacBio part of tools/profile_r03.sh alone (bench line, kernel stats + timeline, FETCH/WRITE passes, SQ counters, N = 2 rehearsal)
set -e
root=$(pwd); out=$root/gpurun_out/r03; mkdir -p "$out"
python3 bench.py --workload pacbio --steps 4 > "$out/r03_bench_line_pacbio.json" 2> "$out/bench_pacbio.err"
python3 bench.py --workload pacbio --steps 4 --sync-steps --no-cpu-baseline > "$out/r03_bench_line_pacbio_sync_steps.json" 2>> "$out/bench_pacbio.err"
cd /tmp && export TMPDIR=/tmp
d=$out/prof_pacbio; rm -rf "$d"; mkdir -p "$d"
rocprofv3 --kernel-trace --stats -d "$d" -o pacbio --output-format csv -- python3 "$root/bench.py" --workload pacbio --steps 3 --warmup 1 --no-cpu-baseline --no-extras --sync-steps > "$d/bench.log" 2>&1
cp "$(find "$d" -name "*kernel_stats.csv" | head -1)" "$out/r03_pacbio_bench_kernel_stats.csv"
python3 "$root/tools/ktrace_step.py" "$(find "$d" -name "*kernel_trace.csv" | head -1)" pb_plan_kernel 4 > "$out/r03_ktrace_pacbio_step.txt"
rm -rf "$d"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d "$out/pmc_pb_$c" -o p --output-format csv -- python3 "$root/bench.py" --workload pacbio --steps 1 --warmup 1 --no-cpu-baseline --no-extras --sync-steps > "$out/pmc_pb_$c.log" 2>&1
done
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(out + "/pmc_pb_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            agg[r["Kernel_Name"].split("(")[0][:60]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "pb_" in k: res.setdefault(k, {})[c + "_kb_mean"] = sum(v) / len(v); res[k]["launches"] = len(v)
for k, d in res.items():
    if "FETCH_SIZE_kb_mean" in d and "WRITE_SIZE_kb_mean" in d:
        d["traffic_bytes_per_launch (2*FETCH + WRITE, KB units)"] = int((2 * d["FETCH_SIZE_kb_mean"] + d["WRITE_SIZE_kb_mean"]) * 1024)
json.dump(res, open(out + "/r03_pmc_pacbio.json", "w"), indent=1)
PY
rm -rf "$out"/pmc_pb_*
bash tools/pmc_sq.sh pacbio pb_plan pb_emit > "$out/r03_sq_pacbio.txt" 2>&1
JK_BENCH_ONE_DEVICE=1 JK_BENCH_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --workload pacbio --lanes 262144 --genome-mbp 300 > "$out/r03_dist2_gloo_one_device_pacbio.json" 2> "$out/dist2_pacbio.err"
echo "pacbio profile done"
