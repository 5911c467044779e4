"""Post-processing / planning tools under tools/ (CPU only)."""
import csv
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_round_timeline_splits_a_trace_at_the_fedavg_launches(tmp_path):
    rows = []
    t = 1000

    def k(name, dur, gap=0):
        nonlocal t
        t += gap
        rows.append({"Kernel_Name": name, "Start_Timestamp": t, "End_Timestamp": t + dur})
        t += dur
    for _ in range(3):                                   # three rounds: 4 steps of two kernels, then the reduce
        for _ in range(4):
            k("void (anonymous namespace)::update_kernel<8>(Args)", 3000)
            k("void (anonymous namespace)::fwd_kernel(Args)", 600, gap=10)
        k("void (anonymous namespace)::fedavg_inplace_kernel<64>(Args)", 1200, gap=100)
    trace = tmp_path / "t_kernel_trace.csv"
    with open(trace, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)
    out = tmp_path / "round.csv"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "round_timeline.py"), str(trace), str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = {row["kernel"]: row for row in csv.DictReader(open(out))}
    assert got["update_kernel<8>"]["calls"] == "4" and float(got["update_kernel<8>"]["total_us"]) == 12.0
    assert got["fedavg_inplace_kernel<64>"]["calls"] == "1"
    total = float(got["ROUND (FedAvg end -> FedAvg end)"]["total_us"])
    busy = float(got["union of kernel intervals"]["total_us"])
    idle = float(got["idle (no kernel running)"]["total_us"])
    # (the tool prints microseconds with one decimal)
    assert abs(total - (4 * 3.61 + 0.1 + 1.2)) < 0.06 and abs(total - busy - idle) < 0.11 and abs(idle - 0.14) < 0.06


def test_round_timeline_reports_queues_that_run_beside_each_other(tmp_path):
    rows = []
    for rnd in range(3):
        base = 1000 + rnd * 10000
        for step in range(2):                                # queue 1: 3 us update; queue 2: 2 us convolution inside it
            t = base + step * 4000
            rows.append({"Kernel_Name": "void update_kernel(Args)", "Queue_Id": 1, "Start_Timestamp": t, "End_Timestamp": t + 3000})
            rows.append({"Kernel_Name": "void conv_kernel(Args)", "Queue_Id": 2, "Start_Timestamp": t + 500, "End_Timestamp": t + 2500})
        rows.append({"Kernel_Name": "void fedavg_inplace_kernel<64>(Args)", "Queue_Id": 1, "Start_Timestamp": base + 8000,
                     "End_Timestamp": base + 9000})
    trace = tmp_path / "t_kernel_trace.csv"
    with open(trace, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)
    out = tmp_path / "round.csv"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "round_timeline.py"), str(trace), str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    with open(out) as f:
        got = {row["kernel"]: row for row in csv.DictReader(f)}
    assert float(got["busy on queue_id 1"]["total_us"]) == 7.0 and float(got["busy on queue_id 2"]["total_us"]) == 4.0
    assert float(got["kernels of different queues beside each other"]["total_us"]) == 4.0
    assert float(got["union of kernel intervals"]["total_us"]) == 7.0


def test_scaling_model_plans_the_bench_scenario():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scaling_model.py"), "270"], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = [l.split() for l in r.stdout.strip().splitlines()[2:]]
    assert [l[0] for l in lines] == ["1", "2", "4", "8"]
    one, eight = lines[0], lines[3]
    assert one[2] == "0" and one[-1] == "100.0%"                     # one GPU: nothing crosses a boundary
    assert 100 < int(eight[2]) < 400 and 80.0 < float(eight[-1].rstrip("%")) < 100.0
