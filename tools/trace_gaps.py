"""Where the time between kernels goes: from a rocprofv3 --kernel-trace CSV, the mean idle gap in front of every kernel,
by (previous kernel -> this kernel), over the dispatches of one queue in start order.
usage: python tools/trace_gaps.py <..._kernel_trace.csv> [min_count]"""
import csv, sys, re, collections

def short(name):
    name = re.sub(r"\(.*", "", name).replace("ppcr::dev::", "").replace("void ", "")
    return re.sub(r"rocprim::.*", "rocprim", name)[:60]

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"]))
rows.sort()
min_count = int(sys.argv[2]) if len(sys.argv) > 2 else 50
gaps = collections.defaultdict(list)
dur = collections.defaultdict(list)
prev = None
for s, e, n, q in rows:
    dur[n].append(e - s)
    if prev is not None:
        gaps[(prev[2], n)].append(s - prev[1])
    prev = (s, e, n, q)
print("%-62s -> %-62s %7s %9s %9s" % ("previous", "kernel", "count", "gap_us", "dur_us"))
for (a, b), g in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
    if len(g) < min_count:
        continue
    g2 = sorted(g)
    print("%-62s -> %-62s %7d %9.2f %9.2f   (median gap %.2f)" % (a, b, len(g), sum(g) / len(g) / 1e3, sum(dur[b]) / len(dur[b]) / 1e3, g2[len(g2) // 2] / 1e3))
