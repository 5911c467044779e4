"""Topology -> collective planner.

The reference moves models with a per-node TCP gossip (fedstellar/node.py:692-809): O(N^2) payloads
per round on a fully connected federation. Here the nodes of a scenario are split into contiguous
blocks, one block per GPU, and the *fixed point* of that gossip (every aggregating node ends the
round holding the weighted mean of its train set, see ``runtime/aggregation.py``) is mapped onto
the cheapest RCCL pattern for the federation graph:

``allreduce``  every aggregating node shares ONE contributor set (fully connected DFL, CFL star,
               SDFL with a single leader): each GPU reduces its local contributors to one [P] row,
               one all-reduce(sum) combines the partials; payload per GPU is P*4 bytes whatever N is.
``rows``       contributor sets differ (ring, random-k, custom matrices): one variable-split
               all-to-all moves exactly the rows a peer needs (a row crosses a link at most once per
               destination GPU even when many nodes there need it); the grouped reduce kernel then
               reads local rows in place and remote rows from the staging buffer.

Everything in this module is pure Python over node ids, so the 8-GPU schedules are checked
structurally on the CPU (tests/test_parallel_planner.py): what rank a sends to rank b is exactly what
b expects from a, in the same order.

xGMI is a point-to-point mesh (7 links x ~153 GB/s per GPU). :func:`estimate_exchange_seconds`
prices a plan per *link*, which is what bounds both patterns, and is used by the bench / docs to
reason about scaling; it is not a tuning knob.
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

XGMI_LINK_BYTES_PER_S = 153e9      # per direction, per link (MI355X: 7 links per GPU)
XGMI_LINKS = 7


def partition_nodes(n_nodes: int, world: int, sizes: Optional[Sequence[int]] = None) -> List[range]:
    """Contiguous blocks of node ids per rank: balanced, or ``sizes[r]`` nodes on rank r (explicit placement: the
    launcher orders the nodes by ``device_args.placement`` first, see :func:`placement_order`)."""
    if sizes is not None:
        sizes = [int(c) for c in sizes]
        if len(sizes) != world or sum(sizes) != n_nodes or min(sizes) < 1:
            raise ValueError(f"block sizes {sizes} do not place {n_nodes} nodes on {world} ranks (every rank hosts >= 1)")
    else:
        base, rem = divmod(n_nodes, world)
        sizes = [base + (1 if r < rem else 0) for r in range(world)]
    out, start = [], 0
    for cnt in sizes:
        out.append(range(start, start + cnt))
        start += cnt
    return out


def placement_order(placements: Sequence[Optional[int]], world: int):
    """``device_args.placement`` of every participant (rank that hosts the node; ``None`` = anywhere) ->
    ``(order, sizes)``: the node order that makes every rank's nodes one contiguous block, and the block sizes.
    Unplaced nodes fill the emptiest ranks, in index order. ``(identity, None)`` when nobody names a placement."""
    n = len(placements)
    if all(p is None for p in placements):
        return list(range(n)), None
    buckets: List[List[int]] = [[] for _ in range(world)]
    free = []
    for i, p in enumerate(placements):
        if p is None:
            free.append(i)
        elif not 0 <= int(p) < world:
            raise ValueError(f"participant {i}: placement {p} outside the {world} ranks of this launch")
        else:
            buckets[int(p)].append(i)
    for i in free:
        min(buckets, key=len).append(i)
    if any(not b for b in buckets):
        raise ValueError("placement leaves a rank without nodes")
    return [i for b in buckets for i in sorted(b)], [len(b) for b in buckets]


def locality_placement(adj: np.ndarray, world: int, passes: int = 8) -> List[int]:
    """A placement (rank per node) that keeps neighbours together: breadth-first order from a lowest-degree node
    (Cuthill-McKee), cut into balanced blocks, then refined by pairwise swaps. Rows that cross a rank boundary are what the xGMI exchange has to move
    (``plan_exchange``), so fewer cut links = fewer staged rows; a ring stays at 2 cut links per boundary, a random
    graph loses a third to a half of its cut."""
    a = np.asarray(adj) != 0
    a = a | a.T
    n = a.shape[0]
    deg = a.sum(1)
    seen = np.zeros(n, dtype=bool)
    order: List[int] = []
    while len(order) < n:
        rest = np.flatnonzero(~seen)
        root = int(rest[np.argmin(deg[rest])])
        seen[root] = True
        queue = [root]
        while queue:
            u = queue.pop(0)
            order.append(u)
            nb = [int(v) for v in np.flatnonzero(a[u] & ~seen)]
            nb.sort(key=lambda v: (deg[v], v))
            for v in nb:
                seen[v] = True
            queue.extend(nb)
    place = np.zeros(n, dtype=np.int64)
    for r, blk in enumerate(partition_nodes(n, world)):
        place[[order[k] for k in blk]] = r
    # refinement: swap a node with the node of its favourite rank that gains most from moving the other way
    # (Kernighan-Lin moves; block sizes never change), until a pass finds nothing
    ai = a.astype(np.int64)
    cnt = ai @ np.eye(world, dtype=np.int64)[place]              # [n, world]: neighbours of u on rank r
    for _ in range(passes):
        improved = False
        for u in range(n):
            ra = int(place[u])
            rb = int(np.argmax(cnt[u]))
            if rb == ra or cnt[u, rb] <= cnt[u, ra]:
                continue
            cand = np.flatnonzero(place == rb)
            gain = (cnt[u, rb] - cnt[u, ra]) + (cnt[cand, ra] - cnt[cand, rb]) - 2 * ai[u, cand]
            k = int(np.argmax(gain))
            if gain[k] <= 0:
                continue
            v = int(cand[k])
            for w, src, dst in ((u, ra, rb), (v, rb, ra)):
                nb = np.flatnonzero(a[w])
                cnt[nb, src] -= 1
                cnt[nb, dst] += 1
                place[w] = dst
            improved = True
        if not improved:
            break
    return [int(r) for r in place]


def cut_links(adj: np.ndarray, place: Sequence[int]) -> int:
    """Undirected links whose two ends sit on different ranks."""
    a = np.asarray(adj) != 0
    a = np.triu(a | a.T, 1)
    pl = np.asarray(place)
    i, j = np.nonzero(a)
    return int((pl[i] != pl[j]).sum())


def owner_of(n_nodes: int, world: int) -> np.ndarray:
    owner = np.zeros(n_nodes, dtype=np.int64)
    for r, blk in enumerate(partition_nodes(n_nodes, world)):
        owner[list(blk)] = r
    return owner


def link_key(scenario: str, a: int, b: int) -> bytes:
    """Deterministic 16-byte AES key of the (unordered) rank link a<->b."""
    lo, hi = (a, b) if a <= b else (b, a)
    return hashlib.sha256(f"{scenario}|link|{lo}|{hi}".encode()).digest()[:16]


@dataclass
class ExchangePlan:
    """What one rank does in the exchange phase of a round."""
    mode: str                                     # "allreduce" | "rows"
    # allreduce mode
    local_sources: List[int] = field(default_factory=list)      # contributor node ids owned by this rank
    local_weights: List[float] = field(default_factory=list)    # already divided by the global weight sum
    local_dests: List[int] = field(default_factory=list)        # node ids on this rank that take the result
    # rows mode
    send_lists: List[List[int]] = field(default_factory=list)   # per peer rank: node ids whose rows we send
    recv_lists: List[List[int]] = field(default_factory=list)   # per peer rank: node ids whose rows we receive
    row_of: Dict[int, int] = field(default_factory=dict)        # node id -> row of the source pool (local | staged)
    groups: List[tuple] = field(default_factory=list)           # (contributor ids, weights, local destination rows)
    n_recv: int = 0

    def rows_sent(self) -> int:
        return sum(len(l) for l in self.send_lists)


def plan_exchange(train_sets: Dict[int, Sequence[int]], adopt: Dict[int, int], owner: np.ndarray,
                  weights: Sequence[float], rank: int, world: int, local: range,
                  allow_allreduce: bool = True) -> ExchangePlan:
    """Exchange plan of ``rank`` for one round structure.

    train_sets: aggregating node -> contributor node ids; adopt: waiting node -> aggregating node.
    """
    local_set = set(local)
    lo = local.start
    keys = {tuple(ts) for ts in train_sets.values()}
    if world > 1 and len(keys) == 1 and allow_allreduce:
        contributors = list(next(iter(keys)))
        dests = sorted(set(train_sets) | set(adopt))
        tot = float(sum(weights[j] for j in contributors))
        loc_src = [j for j in contributors if j in local_set]
        return ExchangePlan("allreduce", local_sources=loc_src, local_weights=[weights[j] / tot for j in loc_src],
                            local_dests=[d for d in dests if d in local_set])
    adopters: Dict[int, List[int]] = {}
    for w, a in adopt.items():
        adopters.setdefault(a, []).append(w)
    needed_by: List[set] = [set() for _ in range(world)]        # needed_by[r] = node rows rank r must receive
    for i, ts in train_sets.items():
        # the result of i is needed on owner(i) and on the owners of the nodes adopting it
        for r in {int(owner[i])} | {int(owner[w]) for w in adopters.get(i, [])}:
            needed_by[r].update(j for j in ts if owner[j] != r)
    send_lists = [sorted(j for j in needed_by[r] if owner[j] == rank) if r != rank else [] for r in range(world)]
    recv_lists = [sorted(j for j in needed_by[rank] if owner[j] == r) if r != rank else [] for r in range(world)]
    row_of = {j: j - lo for j in local}
    pos = len(local)
    for r in range(world):
        for j in recv_lists[r]:
            row_of[j] = pos
            pos += 1
    merged: Dict[tuple, List[int]] = {}
    for i, ts in train_sets.items():
        dsts = ([i] if i in local_set else []) + [w for w in adopters.get(i, []) if w in local_set]
        if dsts:
            merged.setdefault(tuple(ts), []).extend(d - lo for d in dsts)
    groups = [([row_of[j] for j in key], [weights[j] for j in key], dsts) for key, dsts in merged.items()]
    return ExchangePlan("rows", send_lists=send_lists, recv_lists=recv_lists, row_of=row_of, groups=groups,
                        n_recv=pos - len(local))


def estimate_exchange_seconds(plan: ExchangePlan, row_bytes: int, world: int,
                              link_bytes_per_s: float = XGMI_LINK_BYTES_PER_S) -> float:
    """Per-link lower bound of the exchange time on the xGMI mesh.

    allreduce: direct reduce-scatter + all-gather over the mesh moves 2*(world-1)/world of a row per
    GPU, spread over ``min(world-1, 7)`` links. rows: every peer pair has its own link, so the bound is
    the largest single (src,dst) transfer.
    """
    if world <= 1:
        return 0.0
    if plan.mode == "allreduce":
        links = min(world - 1, XGMI_LINKS)
        return 2.0 * (world - 1) / world * row_bytes / (links * link_bytes_per_s)
    worst = max([len(l) for l in plan.send_lists] + [len(l) for l in plan.recv_lists] + [0])
    return worst * row_bytes / link_bytes_per_s
