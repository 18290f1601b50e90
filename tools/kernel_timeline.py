#!/usr/bin/env python3
"""Print the last N kernels of a rocprofv3 kernel trace with start/end relative to the first of them and the idle gap before
each (launch-bound sequences show up as gaps): usage kernel_timeline.py DIR [N=200]"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-70:]))
rows.sort()
rows = rows[-(int(sys.argv[2]) if len(sys.argv) > 2 else 200):]
t0, prev = rows[0][0], rows[0][0]
busy = 0
for s, e, n in rows:
    print("%9.3f  gap %7.1f us  run %8.1f us  %s" % ((s - t0) / 1e6, (s - prev) / 1e3, (e - s) / 1e3, n))
    prev = max(prev, e)
    busy += e - s
print("span %.3f ms, kernels busy %.3f ms over %d launches" % ((rows[-1][1] - t0) / 1e6, busy / 1e6, len(rows)))
