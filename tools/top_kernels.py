"""Print the per-kernel table of a rocprofv3 kernel_stats.csv (calls, average / min / max duration in us)."""
import csv, glob, sys
import os
f = max(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)   # gpurun merges runs: newest
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for r in list(csv.DictReader(open(f)))[:n]:
    print("%-72s %5s  avg %8.2f  min %8.2f  max %8.2f" % (r["Name"].replace("(anonymous namespace)::", "")[:72], r["Calls"],
          float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
