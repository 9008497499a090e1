"""Print name, calls and average duration (us) per kernel from a rocprofv3 *_kernel_stats.csv."""
import csv
import signal
import sys
signal.signal(signal.SIGPIPE, signal.SIG_DFL)      # quiet under `| head`
for r in csv.DictReader(open(sys.argv[1])):
    print("%-34s calls %4s  avg %9.1f us  min %9.1f  max %9.1f" % (r["Name"].split("(")[0].replace("void ", "").replace("jk::", "")[:34], r["Calls"],
                                                               float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
