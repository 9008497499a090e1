#!/bin/bash
# Round-2 evidence, run on the GPU box: bench lines, rocprofv3 --kernel-trace --stats summaries, the generator's
# FETCH_SIZE / WRITE_SIZE passes, one kernel timeline -> gpurun_out/r02/ (copied into profiles/ afterwards).
set -e
root=$(pwd)
out=$root/gpurun_out/r02
rm -rf "$out"; mkdir -p "$out"
# 1. bench lines (no profiler)
python3 bench.py --steps 10 --warmup 3 > "$out/r02_bench_line_illumina.json" 2> "$out/bench_illumina.err"
python3 bench.py --workload hap --steps 4 > "$out/r02_bench_line_hap.json" 2> "$out/bench_hap.err"
python3 bench.py --workload pacbio --steps 3 > "$out/r02_bench_line_pacbio.json" 2> "$out/bench_pacbio.err"
python3 bench.py --workload bgzf --steps 5 > "$out/r02_bench_line_bgzf.json" 2> "$out/bench_bgzf.err"
echo "bench lines done"
# 2. kernel stats
cd /tmp && export TMPDIR=/tmp
for w in illumina hap pacbio; do
  d=$out/prof_$w; mkdir -p "$d"
  st=3; wu=1; if [ $w = illumina ]; then st=12; wu=4; fi      # (the first steps of a process run slower: clocks, first touches)
  rocprofv3 --kernel-trace --stats -d "$d" -o $w --output-format csv -- python3 "$root/bench.py" --workload $w --steps $st --warmup $wu --no-cpu-baseline --no-extras > "$d/bench.log" 2>&1
  f=$(find "$d" -name "*kernel_stats.csv" | head -1)
  cp "$f" "$out/r02_${w}_bench_kernel_stats.csv"
  if [ $w = illumina ]; then
    t=$(find "$d" -name "*kernel_trace.csv" | head -1)
    python3 "$root/tools/ktrace_step.py" "$t" illumina_kernel 4 > "$out/r02_ktrace_illumina_step.txt"
  fi
  rm -rf "$d"
  echo "stats $w done"
done
# 3. HBM traffic counters of the generator (separate passes)
rocprofv3 --pmc FETCH_SIZE -d "$out/fetch" -o fetch --output-format csv -- python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extras > "$out/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$out/write" -o write --output-format csv -- python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extras > "$out/write.log" 2>&1
cd "$root"
f=$(find "$out/fetch" -name '*counter_collection.csv' | head -1)
w=$(find "$out/write" -name '*counter_collection.csv' | head -1)
python3 tools/summarize_rocprof.py pmc "$f" "$w" 2500000 "$out/r02_pmc_generator.json"
rm -rf "$out/fetch" "$out/write"
echo "pmc done"
