"""FedAvg over the headline slab (100 rows x 6.5 M, random topology): two-slab kernel vs the in-place one."""
import sys, time
import torch
sys.path.insert(0, ".")
from p2pfl_amd.ops import hip
from p2pfl_amd.runtime.aggregation import build_plan
K, P, dev = 100, 6497216, "cuda"
R = int(sys.argv[1]) if len(sys.argv) > 1 else 0        # staged rows of other GPUs (multi-GPU round: ~350)
gen = torch.Generator().manual_seed(0)
slab = torch.randn(K, P, device=dev)
out = torch.empty_like(slab)
remote = torch.randn(R, P, device=dev) if R else None
groups = []
for i in range(K):
    nb = {i, (i + 1) % K, (i - 1) % K} | set(torch.randint(0, K, (2,), generator=gen).tolist())
    if R:                                                # every staged row is needed by exactly one local node
        nb = {i, (i + 1) % K} | {K + j for j in range(R) if j % K == i}
    groups.append((sorted(nb), [540.0] * len(nb), [i]))
plan = build_plan(groups).to(dev)
nnz = plan.src_row.numel()


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


t = timed(lambda: hip.fedavg(slab, remote, out, plan.src_ptr, plan.src_row, plan.src_w, plan.dst_ptr, plan.dst_row))
print(f"two slabs : {t:.3f} ms  ({(nnz + K) * P * 4 / 1e9:.1f} GB moved, {(nnz + K) * P * 4 / 1e9 / t:.2f} TB/s) + copy back")
t = timed(lambda: hip.fedavg_inplace(slab, remote, plan.src_ptr, plan.src_row, plan.src_w, plan.dst_ptr, plan.dst_row))
print(f"in place  : {t:.3f} ms  ({(2 * K + R) * P * 4 / 1e9:.1f} GB moved, {(2 * K + R) * P * 4 / 1e9 / t:.2f} TB/s)")
