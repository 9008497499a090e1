"""Print the kernel timeline of the last bench step from a rocprofv3 --kernel-trace CSV (ms, relative)."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']),
             r['Kernel_Name'].split('(')[0].replace('jk::', '').replace('void ', '')[:28], r['Queue_Id']) for r in rows)
key = sys.argv[2] if len(sys.argv) > 2 else 'illumina_kernel'
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4
idx = [i for i, k in enumerate(ks) if key in k[2]]
t0 = ks[idx[-n]][0]
for k in ks[idx[-n]:]:
    print("%9.3f %9.3f %8.3f  %-28s q=%s" % ((k[0] - t0) / 1e6, (k[1] - t0) / 1e6, (k[1] - k[0]) / 1e6, k[2], k[3]))
