#!/usr/bin/env python3
"""Turn rocprofv3 CSV output (gpurun_out/...) into the small summaries committed under profiles/.

  python tools/summarize_rocprof.py stats  <kernel_stats.csv> <out.csv>
  python tools/summarize_rocprof.py pmc    <fetch counter_collection.csv> <write counter_collection.csv> \
                                           <pairs_per_launch> <out.json>
"""
import collections
import csv
import json
import shutil
import sys


def pmc_mean(path, counter, kernel_substr):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if r["Counter_Name"] == counter and kernel_substr in r["Kernel_Name"]]
    return sum(vals) / len(vals), len(vals)


def per_kernel(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return {"%s :: %s" % k: {"dispatches": len(v), "mean": sum(v) / len(v)} for k, v in agg.items()}


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        shutil.copy(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "pmc":
        f, nf = pmc_mean(sys.argv[2], "FETCH_SIZE", "illumina_kernel")
        w, nw = pmc_mean(sys.argv[3], "WRITE_SIZE", "illumina_kernel")
        out = {"kernel": "illumina_kernel", "pairs_per_launch": float(sys.argv[4]),
               "fetch_size_kb": f, "write_size_kb": w, "dispatches": [nf, nw],
               "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python3 bench.py "
                       "--steps 2 --warmup 1 --no-cpu-baseline`; KB units; on gfx950 FETCH_SIZE counts half of a "
                       "wide coalesced read, so HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 "
                       "(/opt/skills/guides/MI355X_MICROARCH.md, HBM)",
               "all_kernels": {"fetch": per_kernel(sys.argv[2]), "write": per_kernel(sys.argv[3])}}
        json.dump(out, open(sys.argv[5], "w"), indent=1)
