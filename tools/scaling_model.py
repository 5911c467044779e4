"""What does the exchange of `bench.py --gpus N` move? (CPU only - a MODEL of the multi-GPU round, not a measurement.)

For N = 1, 2, 4, 8 ranks the bench scenario (N x 100 nodes, ring + seeded random links, DFL) is planned exactly as
BatchedFederation plans it; printed per N: rows a rank receives / sends, the largest single (src, dst) transfer, the
per-link lower bound of the all-to-all (parallel/planner.estimate_exchange_seconds), the send-buffer gather and the
FedAvg over local + staged rows at the rates measured on one GPU, and the projected round time / weak-scaling
efficiency against the measured 1-GPU round.

usage: python tools/scaling_model.py [measured 1-GPU ms per round]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from p2pfl_amd.models.specs import get_spec  # noqa: E402
from p2pfl_amd.parallel.planner import estimate_exchange_seconds, owner_of, partition_nodes, plan_exchange  # noqa: E402
from p2pfl_amd.runtime.slab import ParamLayout  # noqa: E402
from p2pfl_amd.utils.topologymanager import TopologyManager  # noqa: E402

base_ms = float(sys.argv[1]) if len(sys.argv) > 1 else 253.3
row_bytes = ParamLayout(get_spec("mnist-cnn")).row_stride * 4
GATHER_BPS = 2.5e12          # index_select into the send buffer: read + write (measured class: device copy)
FEDAVG_BPS = 3.6e12          # in-place reduce with staged rows (tools/probe/fedavg_probe.py 350: 14.3 GB in 3.96 ms)
print(f"row = {row_bytes / 1e6:.1f} MB; 1-GPU round = {base_ms} ms (measured); everything below is a model")
print(f"{'GPUs':>4} {'nodes':>5} {'recv rows/rank':>15} {'sent rows/rank':>15} {'largest pair':>12} {'all-to-all ms':>13} "
      f"{'gather ms':>9} {'fedavg ms':>9} {'round ms':>8} {'efficiency':>10}")
for world in (1, 2, 4, 8):
    n = 100 * world
    tm = TopologyManager(n_nodes=n, b_symmetric=True, undirected_neighbor_num=3, seed=0)
    tm.generate_topology(extra_random_links=2.0)
    adj = np.asarray(tm.topology)
    train_sets = {i: sorted(set(np.nonzero(adj[i])[0].tolist()) | {i}) for i in range(n)}
    owner = owner_of(n, world)
    blocks = partition_nodes(n, world)
    weights = [540.0] * n
    recv, sent, worst, t_x = [], [], 0, 0.0
    for r in range(world):
        plan = plan_exchange(train_sets, {}, owner, weights, r, world, blocks[r])
        if plan.mode != "rows":
            continue
        recv.append(plan.n_recv)
        sent.append(sum(len(l) for l in plan.send_lists))
        worst = max(worst, max([len(l) for l in plan.send_lists] + [0]))
        t_x = max(t_x, estimate_exchange_seconds(plan, row_bytes, world))
    r_max, s_max = (max(recv) if recv else 0), (max(sent) if sent else 0)
    t_gather = 2.0 * s_max * row_bytes / GATHER_BPS
    t_fedavg = (2 * 100 + r_max) * row_bytes / FEDAVG_BPS if world > 1 else 0.0     # 1 GPU: already inside base_ms
    extra = (t_x + t_gather) * 1e3 + (t_fedavg * 1e3 - 1.2 if world > 1 else 0.0)
    total = base_ms + extra
    print(f"{world:>4} {n:>5} {r_max:>15} {s_max:>15} {worst:>12} {t_x * 1e3:>13.1f} {t_gather * 1e3:>9.1f} "
          f"{t_fedavg * 1e3:>9.1f} {total:>8.1f} {100 * base_ms / total:>9.1f}%")
