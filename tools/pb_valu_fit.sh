root=$(pwd); cd /tmp && export TMPDIR=/tmp
for L in 4000 16000; do
  per=$((48000 / L)); rm -rf /tmp/vf_$L
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU -d /tmp/vf_$L -o p --output-format csv -- python3 $root/tools/pb_valu_fit.py $L $per 524288 > /tmp/vf_$L.log 2>&1
  tail -4 /tmp/vf_$L.log
  python3 - /tmp/vf_$L <<'PY'
import csv, glob, sys, collections
rows=[r for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))]
agg=collections.defaultdict(float)
for r in rows:
    k=r["Kernel_Name"].split("(")[0]
    if "pb_" in k: agg[(k[-28:], r["Counter_Name"])]+=float(r["Counter_Value"])
for k,v in sorted(agg.items()): print("  ", k, "%.4g" % v)
PY
done
