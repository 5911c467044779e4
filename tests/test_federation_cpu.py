"""BatchedFederation semantics on the CPU (eager engine) incl. 2-rank gloo exchange."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from p2pfl_amd.datasets.synthetic import synthetic_shards
from p2pfl_amd.models.specs import get_spec
from p2pfl_amd.role import Role
from p2pfl_amd.runtime.federation import BatchedFederation, partition_nodes
from p2pfl_amd.utils.topologymanager import TopologyManager

SPEC = get_spec("mnist-mlp")


def topo(kind, n, seed=0):
    tm = TopologyManager(n_nodes=n, undirected_neighbor_num=n - 1 if kind == "fully" else 3, seed=seed)
    if kind == "fully":
        tm.generate_topology()
    elif kind == "ring":
        tm.generate_ring_topology()
    elif kind == "star":
        tm.generate_server_topology()
    else:
        tm.generate_topology(extra_random_links=1.5)
    return tm.topology


def make_fed(kind, n, federation="DFL", rank=0, world=1, pg=None, encrypt=False, n_train=32, block_sizes=None):
    fed = BatchedFederation(SPEC, topo(kind, n), federation=federation, device="cpu", engine="eager", epochs=1, seed=5,
                            rank=rank, world_size=world, process_group=pg, encrypt=encrypt, block_sizes=block_sizes)
    data = synthetic_shards(fed.n_local, n_train, 8, 16, 10, seed=1, first_client=fed.local.start)
    fed.init_params()
    fed.set_local_data(*data)
    return fed


def perturb(fed):
    """Deterministic stand-in for local training: row of node g becomes base + (g+1)/100."""
    def fit(epochs, active=None):
        for k, g in enumerate(fed.local):
            if active is None or bool(active[k]):
                fed.engine.params[k] += (g + 1) / 100.0
        return {}
    fed.engine.fit = fit


def test_partition():
    assert [list(r) for r in partition_nodes(5, 2)] == [[0, 1, 2], [3, 4]]


def test_dfl_fully_connected_all_nodes_identical_and_learning():
    fed = make_fed("fully", 4)
    reps = fed.run(2)
    p = fed.engine.params
    assert all(torch.equal(p[0], p[k]) for k in range(1, 4))
    assert reps[0].train_samples == 4 * 32 and reps[1].test_loss.mean() < reps[0].test_loss.mean()


def test_ring_each_node_averages_its_train_set():
    fed = make_fed("ring", 5)
    perturb(fed)
    base = fed.engine.params[0, 0].item()
    fed.run_round(evaluate=False)
    for i in range(5):
        ts = [(i - 1) % 5, i, (i + 1) % 5]
        expect = base + np.mean([(g + 1) / 100.0 for g in ts])
        assert abs(fed.engine.params[i, 0].item() - expect) < 1e-6


def test_weighted_average_uses_training_set_sizes():
    fed = make_fed("fully", 3)
    fed.weights[:] = [100, 300, 600]
    perturb(fed)
    base = fed.engine.params[0, 0].item()
    fed.run_round(evaluate=False)
    assert abs(fed.engine.params[2, 0].item() - (base + (0.01 * 100 + 0.02 * 300 + 0.03 * 600) / 1000)) < 1e-6


def test_cfl_server_never_trains_and_trainers_adopt():
    fed = make_fed("star", 4, federation="CFL")
    assert fed.roles == [Role.SERVER] + [Role.TRAINER] * 3
    seen = {}
    orig = fed.engine.fit
    fed.engine.fit = lambda e, active=None: seen.setdefault("a", active.clone()) is None or orig(e, active)
    fed.run_round()
    assert seen["a"].tolist() == [False, True, True, True]
    p = fed.engine.params
    assert all(torch.equal(p[0], p[k]) for k in range(1, 4))     # server aggregate diffused to every trainer


def test_sdfl_leadership_moves_and_stays_unique():
    fed = make_fed("fully", 4, federation="SDFL")
    leaders = []
    for _ in range(4):
        leaders.append(fed.roles.index(Role.AGGREGATOR))
        fed.run_round(evaluate=False)
        assert fed.roles.count(Role.AGGREGATOR) == 1
        p = fed.engine.params
        assert all(torch.equal(p[0], p[k]) for k in range(1, 4))
    assert len(set(leaders)) > 1


def test_idle_and_proxy_nodes_adopt_without_contributing():
    roles = [Role.AGGREGATOR, Role.AGGREGATOR, Role.IDLE, Role.PROXY]
    fed = BatchedFederation(SPEC, topo("fully", 4), roles=roles, device="cpu", engine="eager", seed=5)
    fed.init_params()
    fed.set_local_data(*synthetic_shards(4, 32, 8, 16, 10, seed=1))
    perturb(fed)
    base = fed.engine.params[0, 0].item()
    fed.run_round(evaluate=False)
    expect = base + (0.01 + 0.02) / 2            # only the two aggregators trained and contributed
    assert all(abs(fed.engine.params[k, 0].item() - expect) < 1e-6 for k in range(4))


def test_killed_node_is_dropped_from_train_sets_and_keeps_its_model():
    fed = make_fed("ring", 5)
    perturb(fed)
    base = fed.engine.params[0, 0].item()
    fed.schedule_fault(0, "kill", 2)
    fed.run_round(evaluate=False)
    assert abs(fed.engine.params[2, 0].item() - base) < 1e-7            # dead: neither trained nor aggregated
    assert abs(fed.engine.params[1, 0].item() - (base + (0.01 + 0.02) / 2)) < 1e-6   # neighbours 0,1 only
    fed.schedule_fault(1, "revive", 2)
    fed.run_round(evaluate=False)
    assert fed.alive.all()


def test_virtual_membership_clock_eviction_and_links():
    """Heartbeat semantics on the virtual clock (reference heartbeater.py:66-111: beat every HEARTBEAT_PERIOD,
    evict after NODE_TIMEOUT), without instant detection."""
    from p2pfl_amd.runtime.membership import VirtualMembership
    m = VirtualMembership(4, heartbeat_period=4.0, node_timeout=20.0, instant=False)
    m.schedule(1, "kill", 2)
    m.schedule(1, "drop_link", 0, 1)
    m.schedule(3, "revive", 2)
    m.schedule(3, "restore_link", 1, 0)
    assert m.apply_round(0) == [] and m.advance(8.0) == [] and m.get_nodes() == [0, 1, 2, 3]
    m.apply_round(1)
    assert not m.alive()[2] and m.get_nodes() == [0, 1, 2, 3]          # silent, but not evicted yet
    assert not m.link_up(1, 0) and m.link_up(0, 2)
    assert m.advance(16.0) == []                                         # 16 s of silence < NODE_TIMEOUT
    assert m.advance(8.0) == [2] and m.get_nodes() == [0, 1, 3]          # 24 s > 20 s: evicted
    m.apply_round(3)
    assert m.alive().all() and m.get_nodes() == [0, 1, 2, 3] and m.link_up(0, 1)
    assert [e[1] for e in m.events] == ["kill", "evicted", "revive"]
    with pytest.raises(ValueError):
        m.schedule(0, "explode", 1)
    with pytest.raises(ValueError):
        m.schedule(0, "drop_link", 1)


def test_dropped_link_and_paused_node_change_the_train_sets():
    fed = make_fed("ring", 5)
    perturb(fed)
    base = fed.engine.params[0, 0].item()
    fed.schedule_fault(0, "drop_link", 1, 2)       # ring 0-1-2-3-4-0 loses the edge 1-2
    fed.schedule_fault(0, "pause", 4)
    fed.run_round(evaluate=False)
    assert fed.train_set(1) == [0, 1] and fed.train_set(2) == [2, 3] and fed.train_set(0) == [0, 1]
    assert abs(fed.engine.params[1, 0].item() - (base + (0.01 + 0.02) / 2)) < 1e-6      # nodes 0, 1
    assert abs(fed.engine.params[2, 0].item() - (base + (0.03 + 0.04) / 2)) < 1e-6      # nodes 2, 3
    assert abs(fed.engine.params[4, 0].item() - base) < 1e-7                            # paused: untouched
    fed.schedule_fault(1, "resume", 4)
    fed.schedule_fault(1, "restore_link", 2, 1)
    fed.run_round(evaluate=False)
    assert fed.alive.all() and fed.train_set(1) == [0, 1, 2]
    assert [e[1] for e in fed.membership.events] == ["pause", "resume"]


def test_fast_forward_replays_the_membership_clock_of_the_finished_rounds():
    """--resume under churn with the reference's detection (NODE_TIMEOUT after the last beat): the replayed rounds
    charge AGGREGATION_TIMEOUT where an aggregator waited in vain, so evictions - and with them the train sets of the
    first resumed round - are those of the run that was never interrupted."""
    def build():
        fed = BatchedFederation(SPEC, topo("ring", 6), device="cpu", engine="eager", epochs=1, seed=5,
                                instant_detection=False, aggregation_timeout=60.0, node_timeout=20.0)
        fed.init_params()
        fed.set_local_data(*synthetic_shards(6, 32, 8, 16, 10, seed=1))
        perturb(fed)
        fed.schedule_fault(1, "kill", 3)
        fed.schedule_fault(2, "kill", 5, phase="mid")
        fed.schedule_fault(4, "revive", 3)
        return fed
    full = build()
    for _ in range(4):
        full.run_round(evaluate=False)
    resumed = build()
    resumed.fast_forward(4)
    strip = lambda ev: [(what, node) for _, what, node in ev]
    assert strip(resumed.membership.events) == strip(full.membership.events)
    assert ("evicted", 3) in strip(full.membership.events) and ("aggregation_timeout", 2) in strip(full.membership.events)
    assert (resumed.believed == full.believed).all() and (resumed.alive == full.alive).all()
    assert resumed.round == full.round == 4
    assert [resumed.train_set(i) for i in range(6)] == [full.train_set(i) for i in range(6)]
    # round 4 revives node 3 on both
    full.run_round(evaluate=False); resumed.run_round(evaluate=False)
    assert strip(resumed.membership.events) == strip(full.membership.events)


# ------------------------------------------------------------------ 2 ranks over gloo
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, kind, encrypt, out, block_sizes=None):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    fed = make_fed(kind, 6, rank=rank, world=world, pg=dist.group.WORLD, encrypt=encrypt, block_sizes=block_sizes)
    perturb(fed)
    fed.run_round(evaluate=False)
    fed.run_round(evaluate=False)
    out[rank] = fed.engine.params[:, :8].clone()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,encrypt", [("ring", False), ("random", True), ("fully", False), ("fully", True)])
def test_two_rank_exchange_matches_single_rank(kind, encrypt):
    ref = make_fed(kind, 6)
    perturb(ref)
    ref.run_round(evaluate=False)
    ref.run_round(evaluate=False)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), kind, encrypt, out), nprocs=2, join=True)
    got = torch.cat([out[0], out[1]], dim=0)
    assert torch.allclose(got, ref.engine.params[:, :8], atol=1e-6)


@pytest.mark.parametrize("kind", ["random", "fully"])
def test_uneven_placement_matches_single_rank(kind):
    """explicit placement (device_args.placement -> block sizes): 5 nodes on rank 0, 1 on rank 1"""
    ref = make_fed(kind, 6)
    perturb(ref)
    ref.run_round(evaluate=False)
    ref.run_round(evaluate=False)
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(2, _free_port(), kind, False, out, [5, 1]), nprocs=2, join=True)
    assert out[0].shape[0] == 5 and out[1].shape[0] == 1
    assert torch.allclose(torch.cat([out[0], out[1]], dim=0), ref.engine.params[:, :8], atol=1e-6)


# ------------------------------------------------------------------ churn with the reference's failure semantics
def make_churn_fed(kind, n, **kw):
    fed = BatchedFederation(SPEC, topo(kind, n), device="cpu", engine="eager", epochs=1, seed=5,
                            instant_detection=False, node_timeout=20.0, aggregation_timeout=60.0, **kw)
    fed.init_params()
    fed.set_local_data(*synthetic_shards(n, 32, 8, 16, 10, seed=1))
    perturb(fed)
    return fed


def test_killed_node_costs_one_aggregation_timeout_then_is_evicted():
    """heartbeater.py:88-101 + aggregator.py:47-76: a node that dies keeps its place in its neighbours' train sets
    until NODE_TIMEOUT passes; meanwhile they wait AGGREGATION_TIMEOUT and aggregate the subset that arrived."""
    fed = make_churn_fed("ring", 6)                       # ring 0-1-2-3-4-5-0
    base = fed.engine.params[0, 0].item()
    fed.schedule_fault(1, "kill", 2)
    r0 = fed.run_round(evaluate=False)
    assert r0.timeouts == 0 and fed.believed.all()
    after0 = fed.engine.params[:, 0].clone()
    r1 = fed.run_round(evaluate=False)
    # round 1: node 2 is down but still EXPECTED by 1 and 3 -> both time out and average who is left
    assert r1.timeouts == 2 and sorted(fed.timed_out) == [1, 3]
    ev = [(what, node) for _, what, node in fed.membership.events]
    assert ev == [("kill", 2), ("aggregation_timeout", 1), ("aggregation_timeout", 3), ("evicted", 2)]
    # the 60 s wait is on the virtual clock and is what expired NODE_TIMEOUT
    assert fed.membership.clock >= 60.0 and not fed.believed[2] and not fed.alive[2]
    trained = after0 + torch.tensor([(g + 1) / 100.0 for g in range(6)])
    assert abs(fed.engine.params[1, 0].item() - (trained[0] + trained[1]).item() / 2) < 1e-6      # {0, 1}, without 2
    assert abs(fed.engine.params[3, 0].item() - (trained[3] + trained[4]).item() / 2) < 1e-6      # {3, 4}, without 2
    assert abs(fed.engine.params[2, 0].item() - after0[2].item()) < 1e-7                          # dead: untouched
    # round 2: 2 is evicted, the train sets shrank, nobody waits any more
    r2 = fed.run_round(evaluate=False)
    assert r2.timeouts == 0 and fed.train_set(1) == [0, 1] and fed.train_set(3) == [3, 4]
    assert base == base


def test_mid_round_death_trains_but_never_delivers():
    fed = make_churn_fed("ring", 5)
    fed.schedule_fault(0, "kill", 4, phase="mid")
    before = fed.engine.params[:, 0].clone()
    rep = fed.run_round(evaluate=False)
    trained = before + torch.tensor([(g + 1) / 100.0 for g in range(5)])
    assert rep.train_samples == 5 * 32                   # node 4 did train this round ...
    assert rep.timeouts == 2 and sorted(fed.timed_out) == [0, 3]      # ... but its neighbours waited in vain
    assert abs(fed.engine.params[0, 0].item() - (trained[0] + trained[1]).item() / 2) < 1e-6
    assert abs(fed.engine.params[4, 0].item() - trained[4].item()) < 1e-6         # its own row: trained, not aggregated
    rep = fed.run_round(evaluate=False)
    assert rep.train_samples == 4 * 32 and rep.timeouts == 0         # evicted by the timeout round's clock


@pytest.mark.parametrize("kind,encrypt", [("random", True), ("ring", False), ("fully", True)])
def test_virtual_devices_take_the_inter_gpu_path_on_one_device(kind, encrypt):
    """virtual_devices=V: rows crossing from one virtual device to another go gather -> link cipher -> staging ->
    decrypt -> reduce; the result must equal the plain single-shard aggregation."""
    ref = make_fed(kind, 7)
    perturb(ref)
    fed = BatchedFederation(SPEC, topo(kind, 7), device="cpu", engine="eager", epochs=1, seed=5, encrypt=encrypt,
                            virtual_devices=3)
    fed.init_params()
    fed.set_local_data(*synthetic_shards(7, 32, 8, 16, 10, seed=1))
    perturb(fed)
    for _ in range(2):
        ref.run_round(evaluate=False)
        rep = fed.run_round(evaluate=False)
    assert torch.allclose(fed.engine.params, ref.engine.params, atol=1e-6)
    assert rep.exchanged_rows > 0
    assert (rep.encrypted_bytes == 2 * rep.exchanged_rows * fed.engine.layout.row_stride * 4) if encrypt \
        else rep.encrypted_bytes == 0


def test_out_of_range_labels_are_refused():
    """loss.hip indexes logits / the confusion matrix with the label: a foreign data file with a label outside
    [0, C) must raise like torch's CrossEntropyLoss, on every engine."""
    from p2pfl_amd.runtime.engine import EagerEngine
    eng = EagerEngine(SPEC, 2, "cpu")
    tx, ty, vx, vy, ex, ey = synthetic_shards(2, 16, 4, 8, 10, seed=1)
    bad = ty.clone()
    bad[1, 3] = 10
    with pytest.raises(ValueError, match="labels must lie in"):
        eng.set_data(tx, bad, vx, vy, ex, ey)
    bad[1, 3] = -1
    with pytest.raises(ValueError):
        eng.set_data(tx, bad, vx, vy, ex, ey)
    eng.set_data(tx, ty, vx, vy, ex, ey)


def test_waiting_node_adopts_the_aggregator_that_finishes_first():
    """node.py:526-531: a trainer takes the first aggregated model that reaches it. On the virtual clock that is the
    neighbouring aggregator with the fewest models to collect - and never one that is sitting out
    AGGREGATION_TIMEOUT for a dead member while another neighbour is already done."""
    # line 0 - 1 - 2 - 3 - 4 with a trainer in the middle: aggregator 1 collects {0, 1, 2}, aggregator 3 {2, 3, 4}
    adj = np.zeros((5, 5), dtype=np.int64)
    for a, b in ((0, 1), (1, 2), (2, 3), (3, 4)):
        adj[a, b] = adj[b, a] = 1
    roles = [Role.AGGREGATOR, Role.AGGREGATOR, Role.TRAINER, Role.AGGREGATOR, Role.AGGREGATOR]
    fed = BatchedFederation(SPEC, adj, roles=roles, device="cpu", engine="eager", epochs=1, seed=5,
                            instant_detection=False, node_timeout=20.0, aggregation_timeout=60.0)
    fed.init_params()
    fed.set_local_data(*synthetic_shards(5, 32, 8, 16, 10, seed=1))
    _, adopt = fed._round_structure()
    assert adopt == {2: 1}                                # equal sizes: the index breaks the tie
    fed.alive[0] = False                                  # dead but not yet evicted: 1 waits for a model that never comes
    _, adopt = fed._round_structure()
    assert fed.timed_out == [1] and adopt == {2: 3}


def test_eager_engine_per_step_series():
    """log_steps on the CPU engine: per-step loss and the confusion matrix of every 20th step's batch
    (reference: Train/* per step, every 20 steps - lightninglearner.py:188, mnist/models/mlp.py:33-106)."""
    from p2pfl_amd.models.specs import get_spec
    from p2pfl_amd.runtime.engine import EagerEngine
    spec = get_spec("mnist-mlp")
    K, n, B = 2, 96, 4
    g = torch.Generator().manual_seed(0)
    x, y = torch.randn(K, n, 784, generator=g), torch.randint(0, 10, (K, n), generator=g)
    eng = EagerEngine(spec, K, "cpu", batch_size=B, seed=3, log_steps=True)
    eng.init_params()
    eng.set_data(x, y, x[:, :8], y[:, :8], x[:, :8], y[:, :8])
    out = eng.fit(2)
    losses = eng.step_losses()
    assert losses.shape == (48, K) and bool((losses > 0).all())
    # the mean of the second epoch's per-step losses is the epoch loss the engine reports
    assert torch.allclose(losses[24:].mean(0), out["Train"]["loss"].float().cpu(), atol=1e-5)
    cms = eng.step_confusions()
    assert sorted(cms) == [20, 40] and all(c.sum(dim=(1, 2)).tolist() == [B, B] for c in cms.values())
    m = eng.step_metrics()[20]
    assert len(m) == K and set(m[0]) == {"Accuracy", "Precision", "Recall", "F1Score"}
