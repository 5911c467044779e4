"""Aggregate rocprofv3 --pmc CSV output per kernel (mean counter value per dispatch) and derive the judged figures.

usage: python tools/pmc_summary.py out.csv [kernel_stats.csv] in1_counter_collection.csv [in2 ...]

Derived columns (only when their inputs were collected):
  hbm_read_MB / hbm_write_MB  TCC<->EA request bytes: 32 B * n32 + 64 B * n64 + 128 B * n128 (the packaged FETCH_SIZE
                              expression prices 128 B reads through TCC_BUBBLE and came out ~1.8x low on gfx950)
  avg_us, GBps                mean kernel time of the --kernel-trace pass and (read + write) / time
  mfma_busy_pct               SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)
  mfma_f32_TFLOPs             SQ_INSTS_VALU_MFMA_MOPS_F32 * 512 / time
  lds_conflict_pct            SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
"""
import csv
import re
import sys
from collections import defaultdict

XCDS, SIMDS = 8, 1024


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*", "", name)[:48]


def derive(c, avg_ns):
    d = {}
    g = c.get
    if "TCC_EA0_RDREQ_sum" in c:
        n32, n64, n128 = g("TCC_EA0_RDREQ_32B_sum", 0.0), g("TCC_EA0_RDREQ_64B_sum", 0.0), g("TCC_EA0_RDREQ_128B_sum", 0.0)
        d["hbm_read_MB"] = (32 * n32 + 64 * n64 + 128 * n128) / 1e6
    if "TCC_EA0_WRREQ_sum" in c:
        n64 = g("TCC_EA0_WRREQ_64B_sum", 0.0)
        d["hbm_write_MB"] = (64 * n64 + 32 * (c["TCC_EA0_WRREQ_sum"] - n64)) / 1e6
    if "TCC_HIT_sum" in c and c["TCC_HIT_sum"] + g("TCC_MISS_sum", 0.0) > 0:
        d["l2_hit_pct"] = 100.0 * c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + g("TCC_MISS_sum", 0.0))
    if avg_ns:
        d["avg_us"] = avg_ns / 1e3
        if "hbm_read_MB" in d or "hbm_write_MB" in d:
            d["GBps"] = (d.get("hbm_read_MB", 0.0) + d.get("hbm_write_MB", 0.0)) * 1e6 / avg_ns
        if "SQ_INSTS_VALU_MFMA_MOPS_F32" in c:
            d["mfma_f32_TFLOPs"] = c["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512 / avg_ns / 1e3
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and g("GRBM_GUI_ACTIVE", 0.0) > 0:
        d["mfma_busy_pct"] = 100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / XCDS * SIMDS)
    if g("SQ_LDS_IDX_ACTIVE", 0.0) > 0:
        d["lds_conflict_pct"] = 100.0 * g("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
    return d


def main(out, *files):
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    avg_ns = {}
    for f in files:
        with open(f) as fh:
            rd = csv.DictReader(fh)
            if "AverageNs" in rd.fieldnames:  # the --stats file of the kernel-trace pass
                for row in rd:
                    avg_ns[short(row["Name"])] = float(row["AverageNs"])
                continue
            for row in rd:
                k = short(row["Kernel_Name"])
                c = row["Counter_Name"]
                acc[k][c] += float(row["Counter_Value"])
                cnt[k][c] += 1
    mean = {k: {c: acc[k][c] / max(cnt[k][c], 1) for c in acc[k]} for k in acc}
    der = {k: derive(mean[k], avg_ns.get(k)) for k in mean}
    counters = sorted({c for k in mean for c in mean[k]})
    dcols = sorted({c for k in der for c in der[k]})
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "dispatches"] + dcols + [f"mean_{c}" for c in counters])
        for k in sorted(mean, key=lambda k: -mean[k].get("GRBM_GUI_ACTIVE", max(mean[k].values()))):
            n = max(cnt[k].values())
            w.writerow([k, n] + [f"{der[k][c]:.2f}" if c in der[k] else "" for c in dcols]
                       + [f"{mean[k][c]:.1f}" if c in mean[k] else "" for c in counters])
    print(open(out).read())


if __name__ == "__main__":
    main(*sys.argv[1:])
