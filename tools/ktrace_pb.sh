#!/bin/bash
# Generator-launch durations of the last PacBio bench step (rocprofv3 --kernel-trace): tools/ktrace_pb.sh <tag>   (GPU box)
root=$(pwd); tag=$1; out=$root/gpurun_out/kt_pb_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$out" -o t --output-format csv -- python3 "$root/bench.py" --workload pacbio --steps 2 --warmup 1 --no-cpu-baseline --no-extras > "$out/bench.log" 2>&1
f=$(find "$out" -name "*kernel_trace.csv" | head -1)
python3 - "$f" > "$root/gpurun_out/ktrace_pb_$tag.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]) for r in rows)
gen = [k for k in ks if "pacbio_kernel" in k[2]][-8:]
cp = [k for k in ks if "compact_linear" in k[2]][-8:]
print("generator launches (ms):", " ".join("%.1f" % ((k[1] - k[0]) / 1e6) for k in gen))
print("compactions (ms):       ", " ".join("%.1f" % ((k[1] - k[0]) / 1e6) for k in cp))
print("step (first generator start to last compaction end): %.1f ms" % ((cp[-1][1] - gen[0][0]) / 1e6))
t0 = gen[0][0]
print("generator start-end (ms):", " ".join("%.0f-%.0f" % ((k[0] - t0) / 1e6, (k[1] - t0) / 1e6) for k in gen))
print("compaction start-end (ms):", " ".join("%.0f-%.0f" % ((k[0] - t0) / 1e6, (k[1] - t0) / 1e6) for k in cp))
PY
grep -o '"value": [0-9.]*' "$out/bench.log" | head -1 >> "$root/gpurun_out/ktrace_pb_$tag.txt"
rm -rf "$out"
cat "$root/gpurun_out/ktrace_pb_$tag.txt"
