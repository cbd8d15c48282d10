"""print the fpx kernels of a rocprofv3 kernel_stats.csv: name, calls, average us"""
import csv
import sys

for r in csv.DictReader(open(sys.argv[1])):
    if "fpx::" in r["Name"] or len(sys.argv) > 2:
        print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), "%9.2f us" % (float(r["AverageNs"]) / 1e3))
