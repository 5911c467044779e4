"""Where does a federated round go? Reads a rocprofv3 kernel trace (CSV) of `bench.py --steps N`, takes the span of the
last `--rounds` rounds (delimited by the FedAvg kernel) and prints per-kernel busy time, the union of busy intervals and
the idle remainder.

usage: python tools/round_timeline.py gpurun_out/x/t_kernel_trace.csv [out.csv]
"""
import csv
import sys


def main(path, out=None):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if "fedavg" in r["Kernel_Name"]]
    if len(ends) < 2:
        raise SystemExit("need at least two FedAvg launches in the trace")
    lo, hi = ends[-2] + 1, ends[-1] + 1            # one whole round: after a FedAvg up to and including the next
    span = rows[lo:hi]
    t0 = int(rows[ends[-2]]["End_Timestamp"])
    t1 = int(span[-1]["End_Timestamp"])
    per = {}
    iv = []
    for r in span:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0][:60]
        d = per.setdefault(name, [0, 0])
        d[0] += 1
        d[1] += e - s
        iv.append((s, e))
    iv.sort()
    busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    total = t1 - t0
    lines = [("kernel", "calls", "total_us", "avg_us", "pct_of_round")]
    for name, (n, ns) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        lines.append((name, n, round(ns / 1e3, 1), round(ns / 1e3 / n, 1), round(100.0 * ns / total, 2)))
    lines.append(("ROUND (FedAvg end -> FedAvg end)", 1, round(total / 1e3, 1), "", 100.0))
    lines.append(("union of kernel intervals", "", round(busy / 1e3, 1), "", round(100.0 * busy / total, 2)))
    lines.append(("idle (no kernel running)", "", round((total - busy) / 1e3, 1), "", round(100.0 * (total - busy) / total, 2)))
    for l in lines:
        print(f"{str(l[0]):62s} {str(l[1]):>6s} {str(l[2]):>10s} {str(l[3]):>9s} {str(l[4]):>7s}")
    if out:
        with open(out, "w", newline="") as f:
            csv.writer(f).writerows(lines)


if __name__ == "__main__":
    main(*sys.argv[1:3])
