#!/bin/bash
# Run on the GPU box: rocprofv3 kernel-trace summaries of every bench workload -> gpurun_out/prof_<tag>/
set -e
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
for w in illumina pacbio bgzf create_genome read_fasta; do
  out=$root/gpurun_out/prof_$w
  rm -rf "$out"; mkdir -p "$out"
  rocprofv3 --kernel-trace --stats -d "$out" -o $w --output-format csv -- python3 "$root/bench.py" --workload $w --steps 3 --warmup 1 --no-cpu-baseline > "$out/bench.log" 2>&1
  f=$(find "$out" -name "*kernel_stats.csv" | head -1)
  cp "$f" "$root/gpurun_out/r01_${w}_kernel_stats.csv"
  rm -f $(find "$out" -name "*kernel_trace.csv")
done
