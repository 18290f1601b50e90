#!/usr/bin/env python3
"""Print the kernel timeline (start/end relative to the first kernel, ms) of the LAST MSM in a rocprofv3 kernel trace:
usage: msm_timeline.py DIR   (rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --workload msm ...)"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:], r.get("Queue_Id", "?")))
rows.sort()
last = max(i for i, r in enumerate(rows) if "msm_digits" in r[2])
# include the normalisation that started just before
start = last
while start > 0 and rows[start - 1][0] > rows[last][0] - 2_000_000:
    start -= 1
t0 = rows[start][0]
for s, e, n, q in rows[start:start + 80]:
    print("%8.3f -> %8.3f  (%7.3f ms)  q%s  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, n))
