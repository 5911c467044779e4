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
    def union(ivs):
        ivs = sorted(ivs)
        busy, cur_s, cur_e = 0, ivs[0][0], ivs[0][1]
        for s, e in ivs[1:]:
            if s > cur_e:
                busy += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        return busy + cur_e - cur_s
    busy = union(iv)
    total = t1 - t0
    # kernels of parallel graph branches / streams run on different hardware queues: busy time of each, and the time
    # during which more than one queue had a kernel running
    qcol = next((c for c in ("Queue_Id", "Stream_Id") if c in span[0]), None)
    per_queue = {}
    if qcol:
        for r in span:
            per_queue.setdefault(r[qcol], []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    lines = [("kernel", "calls", "total_us", "avg_us", "pct_of_round")]
    for name, (n, ns) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        lines.append((name, n, round(ns / 1e3, 1), round(ns / 1e3 / n, 1), round(100.0 * ns / total, 2)))
    lines.append(("ROUND (FedAvg end -> FedAvg end)", 1, round(total / 1e3, 1), "", 100.0))
    lines.append(("union of kernel intervals", "", round(busy / 1e3, 1), "", round(100.0 * busy / total, 2)))
    lines.append(("idle (no kernel running)", "", round((total - busy) / 1e3, 1), "", round(100.0 * (total - busy) / total, 2)))
    if len(per_queue) > 1:
        qbusy = {q: union(v) for q, v in per_queue.items()}
        for q, b in sorted(qbusy.items(), key=lambda kv: -kv[1]):
            lines.append((f"busy on {qcol.lower()} {q}", len(per_queue[q]), round(b / 1e3, 1), "", round(100.0 * b / total, 2)))
        both = sum(qbusy.values()) - busy          # (exact for two queues; an upper bound of the overlap beyond)
        lines.append(("kernels of different queues beside each other", "", round(both / 1e3, 1), "", round(100.0 * both / total, 2)))
    for l in lines:
        print(f"{str(l[0]):62s} {str(l[1]):>6s} {str(l[2]):>10s} {str(l[3]):>9s} {str(l[4]):>7s}")
    if out:
        with open(out, "w", newline="") as f:
            csv.writer(f).writerows(lines)


if __name__ == "__main__":
    main(*sys.argv[1:3])
