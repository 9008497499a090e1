#!/bin/bash
# usage: tools/pmc_variant.sh <tag> [path of a variant libjackalope_hip.so]
# Times the default bench and collects FETCH_SIZE / WRITE_SIZE of the generator kernel in two
# separate rocprofv3 --pmc passes (gpurun refuses --pmc combined with tracing).  Run on the GPU box.
set -e
tag=$1
[ -n "$2" ] && export JK_HIP_LIB=$(realpath "$2")
root=$(pwd)
out=$root/gpurun_out/pmc_$tag
mkdir -p "$out"
python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > "$out/bench.json"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d "$out/fetch" -o fetch --output-format csv -- python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$out/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$out/write" -o write --output-format csv -- python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$out/write.log" 2>&1
cd "$root"
f=$(find "$out/fetch" -name '*counter_collection.csv' | head -1)
w=$(find "$out/write" -name '*counter_collection.csv' | head -1)
python3 tools/summarize_rocprof.py pmc "$f" "$w" 2500000 "$out/pmc.json"
python3 - "$out" <<'PY'
import json, sys
o = sys.argv[1]
b = json.loads(open(o + "/bench.json").read().strip().splitlines()[-1])
p = json.load(open(o + "/pmc.json"))
print("value", b["value"], "kernel_ms", b["roofline"]["kernel_ms"], "fetch_KB", p["fetch_size_kb"], "write_KB", p["write_size_kb"])
PY
rm -rf "$out/fetch" "$out/write"
