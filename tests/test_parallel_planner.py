"""Structural checks of the topology -> collective planner for 2..8 ranks, on the CPU: what rank a
sends to rank b is exactly what b expects from a (same rows, same order), every destination gets all
of its contributors, and the fully connected / star cases collapse to one all-reduce."""
import numpy as np
import pytest

from p2pfl_amd.parallel.planner import (cut_links, estimate_exchange_seconds, locality_placement, owner_of,
                                        partition_nodes, placement_order, plan_exchange)
from p2pfl_amd.utils.topologymanager import TopologyManager


def _structure(adj, roles=None):
    n = adj.shape[0]
    ts = {i: sorted([int(j) for j in np.nonzero(adj[i])[0] if j != i] + [i]) for i in range(n)}
    return ts, {}


def _plans(adj, world):
    n = adj.shape[0]
    owner = owner_of(n, world)
    ts, adopt = _structure(adj)
    w = [100.0 + i for i in range(n)]
    blocks = partition_nodes(n, world)
    return ts, [plan_exchange(ts, adopt, owner, w, r, world, blocks[r]) for r in range(world)], blocks, w


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("kind", ["ring", "random"])
def test_rows_mode_send_recv_symmetry(world, kind):
    n = 50
    tm = TopologyManager(n_nodes=n, b_symmetric=True, undirected_neighbor_num=3)
    tm.generate_ring_topology(increase_convergence=False) if kind == "ring" else tm.generate_topology()
    adj = np.asarray(tm.get_topology())
    ts, plans, blocks, w = _plans(adj, world)
    assert all(p.mode == "rows" for p in plans)
    for a in range(world):
        assert plans[a].send_lists[a] == [] and plans[a].recv_lists[a] == []
        for b in range(world):
            assert plans[a].send_lists[b] == plans[b].recv_lists[a]          # same rows, same order
            assert all(j in blocks[a] for j in plans[a].send_lists[b])        # only rows we own
    for r, p in enumerate(plans):
        assert p.n_recv == sum(len(l) for l in p.recv_lists)
        rows = sorted(p.row_of.values())
        assert rows == list(range(len(blocks[r]) + p.n_recv))                 # dense source pool
        # every local aggregating node finds all of its contributors in the pool, with the right weights
        covered = set()
        inv = {v: k for k, v in p.row_of.items()}
        for srcs, ws, dsts in p.groups:
            contrib = sorted(inv[s] for s in srcs)
            assert ws == [w[j] for j in [inv[s] for s in srcs]]
            for d in dsts:
                node = blocks[r].start + d
                assert ts[node] == contrib
                covered.add(node)
        assert covered == set(blocks[r])
    # a ring only talks to the neighbouring blocks: at most 2 rows per peer and direction... per boundary
    if kind == "ring":
        assert max(p.rows_sent() for p in plans) <= 4


@pytest.mark.parametrize("world", [2, 8])
def test_fully_connected_and_star_collapse_to_allreduce(world):
    n = 40
    full = np.ones((n, n)) - np.eye(n)
    ts, plans, blocks, w = _plans(full, world)
    assert all(p.mode == "allreduce" for p in plans)
    tot = sum(w)
    assert abs(sum(sum(p.local_weights) for p in plans) - 1.0) < 1e-9
    for r, p in enumerate(plans):
        assert p.local_sources == list(blocks[r]) and p.local_dests == list(blocks[r])
        assert np.allclose(p.local_weights, [w[j] / tot for j in blocks[r]])
    # CFL star: only the server aggregates, every other node adopts its result
    owner = owner_of(n, world)
    ts = {0: list(range(n))}
    adopt = {i: 0 for i in range(1, n)}
    plans = [plan_exchange(ts, adopt, owner, w, r, world, blocks[r]) for r in range(world)]
    assert all(p.mode == "allreduce" for p in plans)
    assert sorted(d for p in plans for d in p.local_dests) == list(range(n))


def test_exchange_time_model_is_per_link():
    n, world = 800, 8
    full = np.ones((n, n)) - np.eye(n)
    _, plans, _, _ = _plans(full, world)
    row_bytes = 6_497_162 * 4
    t = estimate_exchange_seconds(plans[0], row_bytes, world)
    assert 0.2e-4 < t < 1e-4            # ~26 MB all-reduce over 7 links: tens of microseconds, not milliseconds
    assert estimate_exchange_seconds(plans[0], row_bytes, 1) == 0.0


def test_inplace_reduce_is_a_gpu_only_path():
    """On the CPU the in-place FedAvg declines (nothing touched) and the two-slab reference path is used."""
    import torch
    from p2pfl_amd.runtime.aggregation import build_plan, fedavg_reference, inplace_supported, run_plan_inplace
    slab = torch.arange(12, dtype=torch.float32).reshape(3, 4)
    plan = build_plan([([0, 1], [1.0, 3.0], [2])])
    assert not inplace_supported(slab, 0)
    before = slab.clone()
    assert run_plan_inplace(slab, None, plan) is False and torch.equal(slab, before)
    out = slab.clone()
    fedavg_reference(slab, None, out, plan)
    assert torch.allclose(out[2], 0.25 * slab[0] + 0.75 * slab[1]) and torch.equal(out[:2], slab[:2])


def test_explicit_placement_orders_nodes_into_blocks():
    assert [list(b) for b in partition_nodes(6, 2, [5, 1])] == [[0, 1, 2, 3, 4], [5]]
    for bad in ([3, 2], [6, 0], [2, 2, 2]):
        with pytest.raises(ValueError):
            partition_nodes(6, 2, bad)
    assert placement_order([None] * 4, 2) == ([0, 1, 2, 3], None)
    order, sizes = placement_order([1, 0, 1, 0, 0], 2)
    assert order == [1, 3, 4, 0, 2] and sizes == [3, 2]
    order, sizes = placement_order([1, None, None, 1, None], 2)        # unplaced nodes fill the emptiest rank
    assert sizes == [3, 2] and sorted(order) == [0, 1, 2, 3, 4] and order[:3] == [1, 2, 4]
    with pytest.raises(ValueError):
        placement_order([0, 2], 2)
    with pytest.raises(ValueError):
        placement_order([0, 0], 2)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_locality_placement_cuts_fewer_links_and_stays_balanced(world):
    n = 25 * world
    rng = np.random.default_rng(3)
    perm = rng.permutation(n)                      # a ring + chords whose node numbering says nothing about adjacency
    adj = np.zeros((n, n), dtype=np.float32)
    for k in range(n):
        a, b = perm[k], perm[(k + 1) % n]
        adj[a, b] = adj[b, a] = 1
    for a, b in rng.integers(0, n, size=(n // 4, 2)):
        if a != b:
            adj[a, b] = adj[b, a] = 1
    naive = owner_of(n, world)
    place = locality_placement(adj, world)
    counts = np.bincount(place, minlength=world)
    assert counts.max() - counts.min() <= 1
    assert cut_links(adj, place) < 0.6 * cut_links(adj, naive)
    # a plain ring numbered in order is already optimal and stays so
    ring = np.zeros((n, n), dtype=np.float32)
    for k in range(n):
        ring[k, (k + 1) % n] = ring[(k + 1) % n, k] = 1
    assert cut_links(ring, locality_placement(ring, world)) <= max(cut_links(ring, owner_of(n, world)), 2 * world)
    # fewer cut links = fewer rows over the fabric
    order, sizes = placement_order(place, world)
    inv = np.argsort(order)
    ts = {int(inv[i]): sorted(int(inv[j]) for j in list(np.nonzero(adj[i])[0]) + [i]) for i in range(n)}
    blocks = partition_nodes(n, world, sizes)
    own = np.zeros(n, dtype=np.int64)
    for r, b in enumerate(blocks):
        own[list(b)] = r
    rows_placed = sum(plan_exchange(ts, {}, own, [1.0] * n, r, world, blocks[r]).rows_sent() for r in range(world))
    ts0 = {i: sorted([int(j) for j in np.nonzero(adj[i])[0]] + [i]) for i in range(n)}
    b0 = partition_nodes(n, world)
    rows_naive = sum(plan_exchange(ts0, {}, naive, [1.0] * n, r, world, b0[r]).rows_sent() for r in range(world))
    assert rows_placed < rows_naive
