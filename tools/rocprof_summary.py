#!/usr/bin/env python3
"""Reduce rocprofv3 CSV output to the small summaries kept under profiles/.

  rocprof_summary.py stats <dir> <out.csv>     copies *_kernel_stats.csv (per-kernel calls/avg/min/max ns)
  rocprof_summary.py pmc <dir> <out.csv>       per kernel: dispatches and mean/sum of each counter
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    if not hits:
        raise SystemExit("no %s under %s: %s" % (suffix, d, os.listdir(d)))
    return hits[0]


def main():
    mode, d, out = sys.argv[1:4]
    if mode == "stats":
        src = find(d, "kernel_stats.csv")
        rows = list(csv.reader(open(src)))
        with open(out, "w", newline="") as f:
            csv.writer(f).writerows(rows)
        for r in rows[:8]:
            print(",".join(r))
    else:
        src = find(d, "counter_collection.csv")
        agg = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(src)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        with open(out, "w", newline="") as f:
            wr = csv.writer(f)
            wr.writerow(["kernel", "counter", "dispatches", "mean", "sum"])
            for k, cs in agg.items():
                for c, v in cs.items():
                    wr.writerow([k[:100], c, len(v), sum(v) / len(v), sum(v)])
                    print(k[:60], c, len(v), sum(v) / len(v))


if __name__ == "__main__":
    main()
