"""Aggregation plans: the fixed point of one round of model gossip, as index lists.

In the reference every node pushes (partial) aggregates to its neighbours at 1 Hz until each
neighbour holds every model of its ``train_set = neighbours U {self}`` and then averages them
weighted by the contributors' training-sample counts (node.py:402-456,692-700,
aggregator.py:106-229, fedavg.py:40-60). Absent failures the end state is deterministic:

    theta_i  <-  sum_{j in train_set(i)} n_j * theta_j / sum_j n_j

An :class:`AggregationPlan` encodes exactly that as CSR "groups" (sources + normalised weights ->
destination rows) which one launch of the fused reduce kernel (``csrc/fedavg.hip``) executes over
the parameter slab. Nodes with identical (sources, weights) share a group, so a fully connected
federation collapses to ONE reduce with N destinations, a star (CFL) to one reduce broadcast to
all rows, and waiting roles (trainer / idle / proxy "adopt the first model received",
aggregator.py:119-123) to copies of their aggregator's group.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch


@dataclass
class AggregationPlan:
    src_ptr: torch.Tensor   # [G+1] int32
    src_row: torch.Tensor   # [nnz] int32 rows of the source pool (local rows, then staged remote rows)
    src_w: torch.Tensor     # [nnz] fp32, normalised per group
    dst_ptr: torch.Tensor   # [G+1] int32
    dst_row: torch.Tensor   # [ndst] int32 rows of the output slab
    n_groups: int

    def to(self, device) -> "AggregationPlan":
        return AggregationPlan(self.src_ptr.to(device), self.src_row.to(device), self.src_w.to(device),
                               self.dst_ptr.to(device), self.dst_row.to(device), self.n_groups)

    def bytes_read(self, row_bytes: int) -> int:
        return int(self.src_row.numel()) * row_bytes

    def bytes_written(self, row_bytes: int) -> int:
        return int(self.dst_row.numel()) * row_bytes


def build_plan(groups: Sequence[Tuple[Sequence[int], Sequence[float], Sequence[int]]]) -> AggregationPlan:
    """groups: iterable of (source rows, raw weights, destination rows)."""
    sp, sr, sw, dp, dr = [0], [], [], [0], []
    for srcs, ws, dsts in groups:
        if len(srcs) == 0 or len(dsts) == 0:
            continue
        tot = float(sum(ws))
        if tot <= 0:
            raise ValueError("aggregation group with non-positive total weight")
        sr.extend(int(s) for s in srcs)
        sw.extend(float(w) / tot for w in ws)
        dr.extend(int(d) for d in dsts)
        sp.append(len(sr))
        dp.append(len(dr))
    i32 = torch.int32
    return AggregationPlan(torch.tensor(sp, dtype=i32), torch.tensor(sr, dtype=i32),
                           torch.tensor(sw, dtype=torch.float32), torch.tensor(dp, dtype=i32),
                           torch.tensor(dr, dtype=i32), len(sp) - 1)


def plan_from_train_sets(train_sets: Dict[int, Sequence[int]], weights: Dict[int, float],
                         row_of: Optional[Dict[int, int]] = None,
                         dest_rows: Optional[Dict[int, int]] = None,
                         adopt: Optional[Dict[int, int]] = None) -> AggregationPlan:
    """Plan for nodes that aggregate their own ``train_set``.

    train_sets: node id -> contributor node ids (must include the node itself when it trains)
    weights:    node id -> FedAvg weight (number of training samples, node.py:447)
    row_of:     node id -> row in the source pool (default: identity)
    dest_rows:  node id -> row of the output slab (default: row_of)
    adopt:      waiting node id -> aggregator node id whose result it adopts
    """
    row_of = row_of or {n: n for n in weights}
    dest_rows = dest_rows or row_of
    merged: Dict[tuple, List[int]] = {}
    key_of: Dict[int, tuple] = {}
    for node, contrib in train_sets.items():
        contrib = sorted(set(contrib))
        key = tuple((row_of[c], float(weights[c])) for c in contrib)
        key_of[node] = key
        merged.setdefault(key, []).append(dest_rows[node])
    for waiter, agg in (adopt or {}).items():
        if agg in key_of:
            merged[key_of[agg]].append(dest_rows[waiter])
    groups = [([r for r, _ in key], [w for _, w in key], dsts) for key, dsts in merged.items()]
    return build_plan(groups)


def fedavg_reference(local: torch.Tensor, remote: Optional[torch.Tensor], out: torch.Tensor,
                     plan: AggregationPlan) -> None:
    """Plain PyTorch implementation of the plan (CPU path and oracle of the HIP kernel)."""
    pool = local if remote is None or remote.numel() == 0 else torch.cat([local, remote], dim=0)
    sp, dp = plan.src_ptr.tolist(), plan.dst_ptr.tolist()
    for g in range(plan.n_groups):
        rows = plan.src_row[sp[g]:sp[g + 1]].long().to(pool.device)
        w = plan.src_w[sp[g]:sp[g + 1]].to(pool.device)
        avg = (pool[rows] * w.unsqueeze(1)).sum(dim=0)
        out[plan.dst_row[dp[g]:dp[g + 1]].long().to(out.device)] = avg


def run_plan(local: torch.Tensor, remote: Optional[torch.Tensor], out: torch.Tensor,
             plan: AggregationPlan) -> None:
    """Execute a plan on whatever device the slab lives on (HIP kernel on the GPU)."""
    if plan.n_groups == 0:
        return
    if local.is_cuda:
        from ..ops import hip
        p = plan if plan.src_ptr.is_cuda else plan.to(local.device)
        hip.fedavg(local, remote, out, p.src_ptr, p.src_row, p.src_w, p.dst_ptr, p.dst_row)
    else:
        fedavg_reference(local, remote, out, plan)


def inplace_supported(slab: torch.Tensor, n_remote: int) -> bool:
    """Can :func:`run_plan_inplace` reduce this slab (+ ``n_remote`` staged rows) in place? (GPU slabs whose rows
    fit the kernel's LDS pool; ``P2PFL_FEDAVG_TWO_SLABS=1`` forces the two-slab path for A/B runs.)"""
    import os
    if not slab.is_cuda or os.environ.get("P2PFL_FEDAVG_TWO_SLABS"):
        return False
    from ..ops import hip
    return int(hip.load().p2pfl_fedavg_inplace_chunk(int(slab.shape[0]) + int(n_remote))) > 0


def single_group_inplace(slab: torch.Tensor, plan: AggregationPlan) -> bool:
    """ONE group (a star / fully connected federation: everybody gets the same average) can always be reduced in
    place by the plain kernel: a thread reads all sources of its columns before it writes any destination, and no
    other thread touches those columns."""
    import os
    return slab.is_cuda and plan.n_groups == 1 and not os.environ.get("P2PFL_FEDAVG_TWO_SLABS")


def run_plan_inplace(slab: torch.Tensor, remote: Optional[torch.Tensor], plan: AggregationPlan,
                     n_remote: int = 0) -> bool:
    """Execute a plan whose destination slab IS the source slab (``csrc/fedavg.hip``: every source row is read once,
    destinations are overwritten in place; rows no group writes keep their content). False: not applicable."""
    if plan.n_groups == 0:
        return True
    rem = remote[:n_remote] if (remote is not None and n_remote > 0) else None
    if single_group_inplace(slab, plan):
        from ..ops import hip
        p = plan if plan.src_ptr.is_cuda else plan.to(slab.device)
        hip.fedavg(slab, rem, slab, p.src_ptr, p.src_row, p.src_w, p.dst_ptr, p.dst_row)
        return True
    if not inplace_supported(slab, n_remote):
        return False
    from ..ops import hip
    p = plan if plan.src_ptr.is_cuda else plan.to(slab.device)
    return hip.fedavg_inplace(slab, rem, p.src_ptr, p.src_row, p.src_w, p.dst_ptr, p.dst_row)
