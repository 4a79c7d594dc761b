#!/usr/bin/env python3
"""Summarise a tools/gpu_profile.sh output directory: per-kernel average duration and PMC means."""
import collections
import csv
import glob
import os
import sys


def main(d):
    for f in glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True):
        print("== kernel stats", os.path.relpath(f, d))
        for row in csv.DictReader(open(f)):
            print("  %-70s calls=%s avg_ns=%s pct=%s" % (row["Name"][:70], row["Calls"], row["AverageNs"], row["Percentage"]))
    for f in sorted(glob.glob(os.path.join(d, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("== counters", os.path.relpath(f, d))
        for k, cs in acc.items():
            print("  ", k)
            for c, v in cs.items():
                print("      %-24s mean=%.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))


if __name__ == "__main__":
    main(sys.argv[1])
