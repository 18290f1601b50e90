#!/usr/bin/env python3
"""Per-kernel launch count, average, MEDIAN, min and max duration from a rocprofv3 --kernel-trace CSV (the --stats
summary has no median; cold first launches skew the average of kernels with few launches).
usage: kernel_medians.py TRACE_DIR > kernel_medians.csv"""
import csv
import glob
import statistics
import sys


def main():
    d = {}
    for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            dur = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3   # us
            d.setdefault(row["Kernel_Name"].split("(")[0][:110], []).append(dur)
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "launches", "avg_us", "median_us", "min_us", "max_us", "total_us"])
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([k, len(v), "%.2f" % (sum(v) / len(v)), "%.2f" % statistics.median(v), "%.2f" % min(v), "%.2f" % max(v), "%.1f" % sum(v)])


if __name__ == "__main__":
    main()
