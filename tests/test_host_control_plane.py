"""Control plane of co-resident nodes (runtime/host.py): node façades, command bus, timer wheel, and the TCP gateway
through which a reference-protocol ``Node`` in the outside world peers with a virtual node."""
import time

import numpy as np
import pytest
import torch

from p2pfl_amd.config import Config
from p2pfl_amd.config.config import fast_timings
from p2pfl_amd.datasets.synthetic import synthetic_shards
from p2pfl_amd.learning.pytorch.mnist.mnist import MNISTDataModule
from p2pfl_amd.learning.pytorch.mnist.models.mlp import MNISTModelMLP
from p2pfl_amd.models.specs import get_spec
from p2pfl_amd.node import Node
from p2pfl_amd.role import Role
from p2pfl_amd.runtime.federation import BatchedFederation
from p2pfl_amd.runtime.host import CommandBus, FederationHost, TimerWheel
from p2pfl_amd.utils.topologymanager import TopologyManager

SPEC = get_spec("mnist-mlp")


def fast_participant(federation="DFL"):
    p = Config(participant={"scenario_args": {"federation": federation, "name": "", "controller": ""},
                            "tracking_args": {"local_tracking": "none", "log_dir": ""}}).participant
    fast_timings(p, 0.02)
    p["AGGREGATION_TIMEOUT"] = 20
    p["NODE_TIMEOUT"] = 20
    return p


def make_host(n=8, federation="CFL", kind="star", **kw):
    tm = TopologyManager(n_nodes=n, undirected_neighbor_num=n - 1 if kind == "fully" else 3, seed=0)
    {"star": tm.generate_server_topology, "ring": tm.generate_ring_topology, "fully": tm.generate_topology}[kind]()
    fed = BatchedFederation(SPEC, tm.topology, federation=federation, device="cpu", engine="eager", epochs=1, seed=5,
                            instant_detection=False)
    fed.init_params()
    fed.set_local_data(*synthetic_shards(n, 32, 8, 16, 10, seed=1))
    return FederationHost(fed, participant=fast_participant(federation), **kw)


def external_node(idx, role, federation):
    c = Config(participant={"scenario_args": {"federation": federation, "n_nodes": 9, "name": "", "controller": ""},
                            "device_args": {"role": role, "idx": idx, "accelerator": "cpu"},
                            "tracking_args": {"local_tracking": "none", "log_dir": ""}})
    fast_timings(c.participant, 0.02)
    c.participant["AGGREGATION_TIMEOUT"] = 20
    c.participant["NODE_TIMEOUT"] = 20
    node = Node(idx, "test", MNISTModelMLP(seed=100 + idx), MNISTDataModule(sub_id=idx, number_sub=200), config=c)
    node.start()
    return node


def wait_for(cond, timeout=90.0):
    t0 = time.time()
    while time.time() - t0 < timeout:
        if cond():
            return True
        time.sleep(0.05)
    return False


def test_timer_wheel_and_bus():
    wheel, fired = TimerWheel(tick=0.1, slots=8), []
    wheel.schedule(0.25, lambda: fired.append("once"))
    wheel.schedule(0.3, lambda: fired.append("tick"), period=0.3)
    wheel.advance(2)
    assert fired == []
    wheel.advance(1)
    assert sorted(fired) == ["once", "tick"]
    wheel.advance(30)                      # several turns of an 8-slot wheel
    assert fired.count("tick") == 11 and fired.count("once") == 1
    bus, seen = CommandBus(), []
    bus.subscribe(lambda s, m: seen.append((s, m)))
    bus.publish(3, b"BEAT 127.0.0.1:1 42\n")
    bus.publish(4, b"MODELS_READY 0\n")
    assert bus.pump() == 2 and seen[0] == (3, b"BEAT 127.0.0.1:1 42\n") and bus.counts["BEAT"] == 1


def test_virtual_nodes_have_the_node_surface_and_one_shared_heartbeat_thread():
    host = make_host(6, "DFL", "ring")
    seen = []
    host.bus.subscribe(lambda s, m: seen.append((s, m.split()[0])))
    n3 = host.nodes[3]
    assert n3.round is None and n3.totalrounds is None and n3.role == Role.AGGREGATOR
    assert n3.get_name() == "127.0.0.1:45003" and n3.get_neighbors() == ["127.0.0.1:45002", "127.0.0.1:45004"]
    assert len(n3.get_network_nodes()) == 6
    # a fake clock drives the ONE wheel: every live node beats once per period, ROLE every second beat
    period = host.heartbeat_period
    t = 1000.0
    host._last_tick = t
    host.fed.membership.schedule(0, "kill", 5)
    host.fed.membership.apply_round(0)
    for _ in range(int(4 * period / host.wheel.tick)):
        t += host.wheel.tick
        host.tick_once(now=t)
    beats = [s for s, h in seen if h == b"BEAT"]
    assert 5 not in beats and all(3 <= beats.count(i) <= 5 for i in range(5))
    assert any(h == b"ROLE" for _, h in seen)
    # the silent node is evicted NODE_TIMEOUT later - on the same clock
    for _ in range(int((host.node_timeout + 2 * period) / host.wheel.tick)):
        t += host.wheel.tick
        host.tick_once(now=t)
    assert "127.0.0.1:45005" not in n3.get_network_nodes() and len(n3.get_network_nodes()) == 5
    assert ("evicted", 5) in [(w, n) for _, w, n in host.fed.membership.events]


def test_virtual_start_node_runs_rounds_and_reports_protocol_messages():
    host = make_host(5, "DFL", "fully").start()
    try:
        host.nodes[2].set_start_learning(rounds=2, epochs=1)
        assert wait_for(lambda: host.round is not None, 5)
        assert wait_for(lambda: host.round is None, 60)
        assert len(host.reports) == 2 and all(n.round is None for n in host.nodes)
        p = host.fed.engine.params
        assert all(torch.equal(p[0], p[k]) for k in range(1, 5))
        time.sleep(0.2)
        c = host.bus.counts
        assert c["START_LEARNING"] == 1 and c["MODELS_AGGREGATED"] == 10 and c["MODELS_READY"] == 10 and c["BEAT"] > 0
    finally:
        host.stop()


@pytest.mark.parametrize("starter", ["external", "virtual"])
def test_external_tcp_node_trains_with_eight_virtual_nodes(starter):
    """CFL star: the server is virtual node 0 (gateway), 7 trainers are co-resident rows, the 8th trainer is a real
    TCP Node in the reference protocol. Two rounds; all nine nodes end with the identical model."""
    host = make_host(8, "CFL", "star").start()
    gw = host.gateway(0, port=0)
    ext = external_node(8, Role.TRAINER, "CFL")
    try:
        assert ext.connect_to(gw.host, gw.port) is not None
        assert wait_for(lambda: gw.get_neighbors_names() == [ext.get_name()], 5)
        # heartbeats of the co-resident nodes reach the external node through the gateway
        assert wait_for(lambda: len(ext.get_network_nodes()) == 9, 10), ext.get_network_nodes()
        assert ext.get_name() in host.nodes[0].get_neighbors() and ext.get_name() in host.nodes[3].get_network_nodes()
        if starter == "external":
            ext.set_start_learning(rounds=2, epochs=1)
        else:
            host.nodes[0].set_start_learning(rounds=2, epochs=1)
        assert wait_for(lambda: ext.round is not None and host.round is not None, 10)
        assert wait_for(lambda: ext.round is None and host.round is None and gw.round is None, 120)
        assert len(host.reports) == 2
        ref = ext.learner.get_parameters()
        for i in range(8):
            sd = host.nodes[i].get_parameters()
            for k in ref:
                assert torch.allclose(sd[k].cpu(), ref[k].cpu(), atol=1e-6), (i, k)
        # and it is a real average: the external node's 270 training samples took part in the weights
        assert int(ext.learner.get_num_samples()[0]) > 0
    finally:
        ext.stop()
        host.stop()
        time.sleep(0.3)


def test_stop_learning_from_the_tcp_node_ends_the_batched_run():
    host = make_host(8, "CFL", "star").start()
    gw = host.gateway(0, port=0)
    ext = external_node(8, Role.TRAINER, "CFL")
    try:
        assert ext.connect_to(gw.host, gw.port) is not None
        assert wait_for(lambda: len(ext.get_network_nodes()) == 9, 10)
        ext.set_start_learning(rounds=500, epochs=1)
        assert wait_for(lambda: len(host.reports) >= 1, 60)
        ext.set_stop_learning()
        assert wait_for(lambda: host.round is None and gw.round is None and ext.round is None, 60)
        assert 1 <= len(host.reports) < 500
        assert host.bus.counts["STOP_LEARNING"] == 1
    finally:
        ext.stop()
        host.stop()
        time.sleep(0.3)


def test_batched_launcher_serves_the_control_plane(tmp_path, monkeypatch):
    """`python -m p2pfl_amd.runtime.launch --control-plane --gateway 0:PORT --wait-start S`: the launcher's federation
    is started by START_LEARNING from a reference-protocol TCP node that peers with virtual node 0, runs the rounds
    that node asked for, and logs them per participant like the plain launcher."""
    import json
    import os
    import socket
    import threading
    from p2pfl_amd.controller import Controller
    from p2pfl_amd.runtime.launch import run_scenario
    from p2pfl_amd.learning.pytorch.datamodule import synthetic_classification
    tx, ty, ex, ey = synthetic_classification(8000, 1600, 784, 10, seed=1234)     # a small shared MNIST-shaped set
    monkeypatch.setattr(MNISTDataModule, "mnist_train", (tx, ty))
    monkeypatch.setattr(MNISTDataModule, "mnist_val", (ex, ey))
    cdir = os.path.join(str(tmp_path), "config", "scen")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    fast = fast_participant("DFL")
    over = {k: fast[k] for k in fast if k.isupper()}
    over["tracking_args"] = {"log_dir": str(tmp_path / "logs")}
    Controller.generate_participants(cdir, 4, "DFL", "MNIST", "MLP", rounds=5, epochs=1, accelerator="cpu",
                                     controller="", overrides=over)
    for i in range(4):
        path = os.path.join(cdir, f"participant_{i}.json")
        with open(path) as fh:
            p = json.load(fh)
        p["network_args"]["neighbors"] = " ".join(f"127.0.0.1:{45000 + j}" for j in ((i - 1) % 4, (i + 1) % 4))
        p["scenario_args"]["name"] = "scen"
        with open(path, "w") as fh:
            json.dump(p, fh)
    out = {}
    t = threading.Thread(target=lambda: out.update(run_scenario(cdir, device="cpu", gateways=[f"0:{port}"],
                                                                wait_start=60.0)), daemon=True)
    t.start()
    ext = external_node(39, Role.AGGREGATOR, "DFL")
    try:
        assert wait_for(lambda: _port_open(port), 60)
        ext.connect_to("127.0.0.1", port, full=False)
        assert wait_for(lambda: len(ext.get_neighbors()) == 1, 30)
        ext.set_start_learning(rounds=2, epochs=1)
        t.join(240)
        assert not t.is_alive()
    finally:
        ext.stop()
    assert out["rounds"] == 2 and out["nodes"] == 4
    log0 = (tmp_path / "logs" / "scen" / "participant_0.log").read_text()
    assert "TCP gateway of virtual node 0" in log0 and "round 1 role" in log0
    assert os.path.exists(tmp_path / "logs" / "scen" / "metrics" / "participant_2" / "metrics.csv")


def _port_open(port):
    import socket
    s = socket.socket()
    s.settimeout(0.2)
    try:
        s.connect(("127.0.0.1", port))
        return True
    except OSError:
        return False
    finally:
        s.close()
