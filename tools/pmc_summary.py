"""Aggregate rocprofv3 --pmc CSV output per kernel: mean counter value per dispatch.

usage: python tools/pmc_summary.py out.csv in1_counter_collection.csv [in2 ...]
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*", "", name)[:48]


def main(out, *files):
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                c = row["Counter_Name"]
                acc[k][c] += float(row["Counter_Value"])
                cnt[k][c] += 1
    counters = sorted({c for k in acc for c in acc[k]})
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "dispatches"] + [f"mean_{c}" for c in counters])
        for k in sorted(acc, key=lambda k: -max(acc[k].values())):
            n = max(cnt[k].values())
            w.writerow([k, n] + [f"{acc[k][c] / max(cnt[k][c], 1):.1f}" if c in acc[k] else "" for c in counters])
    print(open(out).read())


if __name__ == "__main__":
    main(*sys.argv[1:])
