#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc run (reads every *_counter_collection.csv under a directory).
usage: tools/pmc_summary.py <dir> [kernel-substring]"""
import collections
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    dur = collections.defaultdict(dict)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if pat not in k:
                continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
            dur[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k in sorted(acc):
        n = len(disp[k])
        print("%s dispatches=%d avg_ns=%.0f" % (k, n, sum(dur[k].values()) / n))
        for c in sorted(acc[k]):
            print("  %-28s %.4g" % (c, acc[k][c] / n))


if __name__ == "__main__":
    main()
