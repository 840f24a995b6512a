#!/usr/bin/env python3
"""Trim a rocprofv3 --kernel-trace --stats kernel_stats.csv to a short, committable summary."""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)               # drop the argument list
    name = re.sub(r"^void ", "", name)
    name = name.replace("ppcr::dev::", "")
    if "rocprim" in name:
        m = re.search(r"(radix_sort\w*|merge_sort\w*|scan\w*|onesweep\w*|histogram\w*)", name)
        name = "rocprim::" + (m.group(1) if m else "kernel")
    return name[:70]


def main(path, out):
    rows = list(csv.DictReader(open(path)))
    with open(out, "w") as f:
        f.write("kernel,calls,total_ns,avg_ns,pct,min_ns,max_ns\n")
        for r in rows:
            f.write(f"{short(r['Name'])},{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.0f},"
                    f"{float(r['Percentage']):.2f},{r['MinNs']},{r['MaxNs']}\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
