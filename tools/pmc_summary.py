#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection.csv files: mean counter value per dispatch, per kernel.

--split KERNEL_SUBSTRING KB: the dispatches of the kernels whose name contains the substring are ALSO reported in two
regimes, told apart by their WRITE_SIZE (KB per dispatch; the passes run the same deterministic schedule, so a dispatch
keeps its place in the kernel's sequence from pass to pass): "[list-building launches]" write more than KB, "[answering
launches]" less — the Verlet variant of K1 writes every row's list (64+ MB at 1M rows) only when it builds them."""
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


def main(argv):
    split_sub, split_kb = None, 0.0
    if "--split" in argv:
        k = argv.index("--split")
        split_sub, split_kb = argv[k + 1], float(argv[k + 2])
        argv = argv[:k] + argv[k + 3:]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    seq = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> values in dispatch order (one file = one pass)
    for d in argv:
        for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
            rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r.get("Dispatch_Id", 0)))
            for r in rows:
                k = short(r["Kernel_Name"])
                a = acc[k][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
                if split_sub and split_sub in k:
                    seq[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        print(k)
        for c in sorted(acc[k]):
            s, n = acc[k][c]
            print(f"    {c:36s} {s / n:16.1f}   (n={n})")
        if k in seq and "WRITE_SIZE" in seq[k]:
            w = seq[k]["WRITE_SIZE"]
            for tag, pick in (("[answering launches]", [x <= split_kb for x in w]), ("[list-building launches]", [x > split_kb for x in w])):
                if not any(pick):
                    continue
                print(f"{k} {tag}")
                for c in sorted(seq[k]):
                    v = seq[k][c]
                    if len(v) != len(w):
                        continue            # (a pass that saw another number of dispatches: no regime split for its counters)
                    sel = [x for x, p in zip(v, pick) if p]
                    print(f"    {c:36s} {sum(sel) / len(sel):16.1f}   (n={len(sel)})")


if __name__ == "__main__":
    main(sys.argv[1:])
