#!/usr/bin/env python3
"""Per-kernel mean of every counter found in rocprofv3 --pmc output directories -> one CSV on stdout.
usage: pmc_summary.py DIR [DIR ...]"""
import csv
import glob
import sys


def main():
    acc = {}
    for d in sys.argv[1:]:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                key = (row["Kernel_Name"].split("(")[0][:110], row["Counter_Name"])
                t = acc.setdefault(key, [0.0, 0])
                t[0] += float(row["Counter_Value"])
                t[1] += 1
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "counter", "mean_per_launch", "launches"])
    for (k, c), (tot, n) in sorted(acc.items()):
        w.writerow([k, c, "%.6g" % (tot / n), n])


if __name__ == "__main__":
    main()
