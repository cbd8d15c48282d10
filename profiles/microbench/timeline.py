"""Timeline of the last steps of a rocprofv3 --kernel-trace run: python timeline.py <kernel_trace.csv> [how many kernels]
Prints start offset, duration and the gap to the previous kernel, in microseconds."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
last = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-last:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f  dur %7.1f  gap %6.1f  %s" % ((a - t0) / 1e3, (b - a) / 1e3, (a - prev_end) / 1e3, r["Kernel_Name"][:70]))
    prev_end = b
