#!/bin/bash
# Round-3 evidence, run on the GPU box: bench lines, rocprofv3 --kernel-trace --stats summaries, the generator's and the
# PacBio kernels' FETCH_SIZE / WRITE_SIZE passes, SQ counters, timelines, the N = 2 rehearsals -> gpurun_out/r03/ (copied
# into profiles/ afterwards).   usage: bash tools/profile_r03.sh [part ...]   parts: bench stats pmc sq dist arena
set -e
root=$(pwd)
out=$root/gpurun_out/r03
mkdir -p "$out"
parts=${@:-bench stats pmc sq dist arena}
for part in $parts; do
case $part in
bench)
  python3 bench.py --steps 20 --warmup 3 > "$out/r03_bench_line_illumina.json" 2> "$out/bench_illumina.err"
  python3 bench.py --steps 20 --warmup 3 --sync-steps --no-extras --no-cpu-baseline > "$out/r03_bench_line_illumina_sync_steps.json" 2>> "$out/bench_illumina.err"
  python3 bench.py --workload hap --steps 4 > "$out/r03_bench_line_hap.json" 2> "$out/bench_hap.err"
  JK_HAP_MATERIALISE=0 python3 bench.py --workload hap --steps 4 > "$out/r03_bench_line_hap_tables.json" 2>> "$out/bench_hap.err"
  python3 bench.py --workload pacbio --steps 3 > "$out/r03_bench_line_pacbio.json" 2> "$out/bench_pacbio.err"
  python3 bench.py --workload bgzf --steps 5 > "$out/r03_bench_line_bgzf.json" 2> "$out/bench_bgzf.err"
  JK_BGZF_LZ=0 python3 bench.py --workload bgzf --steps 5 > "$out/r03_bench_line_bgzf_literals_only.json" 2>> "$out/bench_bgzf.err"
  python3 tools/seqsys_perf.py > "$out/r03_seqsys_perf.txt" 2>&1
  echo "bench lines done" ;;
stats)
  cd /tmp && export TMPDIR=/tmp
  for w in illumina hap pacbio bgzf; do
    d=$out/prof_$w; rm -rf "$d"; mkdir -p "$d"
    st=3; wu=1; if [ $w = illumina ]; then st=12; wu=4; fi
    rocprofv3 --kernel-trace --stats -d "$d" -o $w --output-format csv -- python3 "$root/bench.py" --workload $w --steps $st --warmup $wu --no-cpu-baseline --no-extras > "$d/bench.log" 2>&1
    f=$(find "$d" -name "*kernel_stats.csv" | head -1)
    cp "$f" "$out/r03_${w}_bench_kernel_stats.csv"
    t=$(find "$d" -name "*kernel_trace.csv" | head -1)
    if [ $w = illumina ]; then python3 "$root/tools/ktrace_step.py" "$t" illumina_kernel 8 > "$out/r03_ktrace_illumina_steps.txt"; fi
    if [ $w = pacbio ]; then python3 "$root/tools/ktrace_step.py" "$t" pb_plan_kernel 8 > "$out/r03_ktrace_pacbio_step.txt"; fi
    rm -rf "$d"
    echo "stats $w done"
  done
  cd "$root" ;;
pmc)
  cd /tmp && export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -d "$out/pmc_ill_$c" -o p --output-format csv -- python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extras > "$out/pmc_ill_$c.log" 2>&1
    rocprofv3 --pmc $c -d "$out/pmc_pb_$c" -o p --output-format csv -- python3 "$root/bench.py" --workload pacbio --steps 1 --warmup 1 --no-cpu-baseline > "$out/pmc_pb_$c.log" 2>&1
  done
  cd "$root"
  f=$(find "$out/pmc_ill_FETCH_SIZE" -name '*counter_collection.csv' | head -1)
  w=$(find "$out/pmc_ill_WRITE_SIZE" -name '*counter_collection.csv' | head -1)
  python3 tools/summarize_rocprof.py pmc "$f" "$w" 2500000 "$out/r03_pmc_generator.json"
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
  rm -rf "$out"/pmc_ill_* "$out"/pmc_pb_*
  echo "pmc done" ;;
sq)
  bash tools/pmc_sq.sh illumina illumina_kernel > "$out/r03_sq_illumina.txt" 2>&1
  bash tools/pmc_sq.sh pacbio pb_plan pb_emit > "$out/r03_sq_pacbio.txt" 2>&1
  bash tools/pmc_sq.sh bgzf bgzf_deflate_lz > "$out/r03_sq_bgzf.txt" 2>&1
  JK_HAP_MATERIALISE=0 bash tools/pmc_sq.sh hap illumina_kernel > "$out/r03_sq_hap_tables.txt" 2>&1
  echo "sq done" ;;
dist)
  for w in illumina hap pacbio; do
    case $w in illumina) sz="--pairs 2000000 --lanes 262144 --genome-mbp 100";; hap) sz="--lanes 131072 --genome-mbp 10";; pacbio) sz="--lanes 262144 --genome-mbp 300";; esac
    JK_BENCH_ONE_DEVICE=1 JK_BENCH_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --workload $w $sz > "$out/r03_dist2_gloo_one_device_$w.json" 2> "$out/dist2_$w.err"
  done
  JK_BENCH_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29612 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > "$out/r03_dist1_rccl_illumina.json" 2> "$out/dist1_rccl.err"
  echo "dist done" ;;
arena)
  JK_TIMING=1 python3 tools/config3_full.py --jobs 2 > "$out/r03_arena_config3_two_jobs.txt" 2>&1
  JK_ARENA=0 JK_TIMING=1 python3 tools/config3_full.py --jobs 2 > "$out/r03_arena_off_config3_two_jobs.txt" 2>&1
  echo "arena done" ;;
esac
done
