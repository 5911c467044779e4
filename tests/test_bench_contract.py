"""bench.py keeps the driver's contract when launched the way the driver launches it: ``torch.distributed.run`` with one
rank per device, ONE JSON line from rank 0 with the whole-job aggregate (here: 2 and 3 gloo ranks on the CPU)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config"}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, extra=(), launcher="torchrun"):
    head = ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
             "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] if launcher == "torchrun"
            else [sys.executable])          # "self": bench.py --gpus N starts its own ranks
    cmd = [*head, os.path.join(ROOT, "bench.py"),
           "--gpus", str(world), "--steps", "2", "--warmup", "1", "--cpu", "--nodes-per-gpu", "3", "--model", "mnist-mlp",
           "--epochs", "1", "--split-nodes", "1000", *extra]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout          # rank 0 only, one line
    return json.loads(lines[0])


@pytest.mark.parametrize("world", [2, 3])
def test_multi_rank_launch_prints_one_aggregate_line(world):
    out = _run(world)
    assert KEYS <= set(out)
    assert out["n_gpus"] == world and out["steps"] == 2 and out["warmup"] == 1
    assert out["higher_is_better"] is True and out["scaling"] == "weak" and out["data"] == "synthetic"
    cfg = out["config"]
    assert cfg["nodes"] == 3 * world and cfg["nodes_per_gpu"] == 3 and cfg["parallelism"] == f"fdp{world}x3"
    # whole-job aggregate: every node of every rank trains its 54-sample shard once per round
    samples = 54 * cfg["nodes"] * out["steps"]
    assert out["value"] == pytest.approx(samples / (out["ms_per_step"] * out["steps"] / 1000.0), rel=1e-6)
    assert cfg["exchanged_rows"] > 0          # models crossed the rank boundary
    assert "mlp" in out["metric"].lower()


def test_strong_scaling_keeps_the_node_count():
    out = _run(2, ["--scaling", "strong", "--nodes-per-gpu", "4"])
    assert out["scaling"] == "strong" and out["config"]["nodes"] == 4 and out["config"]["nodes_per_gpu"] == 2


def test_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no external launcher (how the driver ran round 2) runs TWO ranks."""
    out = _run(2, launcher="self")
    assert out["n_gpus"] == 2 and out["config"]["rccl_ranks"] == 2 and out["config"]["backend"] == "gloo"
    assert out["config"]["nodes"] == 6 and out["config"]["exchanged_rows"] > 0


def test_more_ranks_than_gpus_fails_loudly():
    """Without --cpu / --rehearse-on-one-gpu a rank refuses to share a GPU: the launch exits non-zero and says why
    (this box has 0 GPUs, the GPU box 1 - fewer than 2 either way) instead of silently benchmarking one device."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "2 ranks were asked for" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]
