#!/bin/bash
# Kernel timeline of one bench step (rocprofv3 --kernel-trace) -> gpurun_out/ktrace_<tag>.txt
# usage: tools/ktrace_run.sh <tag> <wg per launch> <launches>   (GPU box)
set -e
root=$(pwd)
tag=$1; wg=${2:-256}; n=${3:-4}
bl=$((wg*1024)); lanes=$((bl*n))
out=$root/gpurun_out/kt_$tag
rm -rf "$out"; mkdir -p "$out"
export JK_BATCH_LANES=$bl
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$out" -o t --output-format csv -- python3 "$root/bench.py" --steps 3 --warmup 1 --lanes $lanes --no-cpu-baseline --no-extras > "$out/bench.log" 2>&1
f=$(find "$out" -name "*kernel_trace.csv" | head -1)
python3 "$root/tools/ktrace_step.py" "$f" illumina_kernel $n > "$root/gpurun_out/ktrace_$tag.txt"
rm -rf "$out"
