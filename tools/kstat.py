"""Average duration of the kernels whose name contains one of the given fragments, from a rocprofv3 --kernel-trace --stats output directory.
usage: kstat.py <dir> <fragment> [<fragment> ...]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if any(k in n for k in sys.argv[2:]):
        print("   %-70s calls %5s avg %8.1f us" % (n[:70], r["Calls"], float(r["AverageNs"]) / 1e3))
