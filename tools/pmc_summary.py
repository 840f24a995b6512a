#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection.csv files: mean counter value per dispatch, per kernel."""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name).replace("void ", "").replace("ppcr::dev::", "")
    if "rocprim" in name:
        return "rocprim"
    return name[:64]


def main(dirs):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                a = acc[k][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    for k in sorted(acc):
        print(k)
        for c in sorted(acc[k]):
            s, n = acc[k][c]
            print(f"    {c:36s} {s / n:16.1f}   (n={n})")


if __name__ == "__main__":
    main(sys.argv[1:])
