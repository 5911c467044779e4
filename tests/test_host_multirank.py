"""Control plane that spans ranks (runtime/host.py + runtime/bridge.py): two gloo ranks host four virtual nodes each;
a reference-protocol TCP ``Node`` in THIS process peers with a virtual node hosted on rank 1, sees all eight names,
starts the run (its model is the initial one on both ranks), takes part in two rounds, and can stop both ranks.
Reference behaviour: fedstellar/heartbeater.py:66-139 (network-wide membership from flooded beats),
fedstellar/node.py:224-271 (START / STOP_LEARNING reach every node)."""
import os
import socket
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from p2pfl_amd.models.specs import get_spec
from p2pfl_amd.role import Role

SPEC = get_spec("mnist-mlp")
N, HUB = 8, 6            # ranks host nodes 0-3 / 4-7; the CFL server (and TCP gateway) is node 6 on rank 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _star(hub):
    adj = np.zeros((N, N), dtype=np.float32)
    adj[hub, :] = adj[:, hub] = 1
    adj[hub, hub] = 0
    return adj


def _host_worker(rank, world, port, gw_port, out, federation):
    from p2pfl_amd.datasets.synthetic import synthetic_shards
    from p2pfl_amd.runtime.federation import BatchedFederation
    from p2pfl_amd.runtime.host import FederationHost
    from tests.test_host_control_plane import fast_participant, wait_for
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    if federation == "CFL":
        adj, roles = _star(HUB), [Role.SERVER if i == HUB else Role.TRAINER for i in range(N)]
    else:        # DFL ring
        adj = np.zeros((N, N), dtype=np.float32)
        for i in range(N):
            adj[i, (i + 1) % N] = adj[(i + 1) % N, i] = 1
        roles = None
    fed = BatchedFederation(SPEC, adj, federation=federation, roles=roles, device="cpu", engine="eager", epochs=1,
                            seed=5, rank=rank, world_size=world, process_group=dist.group.WORLD,
                            instant_detection=False)
    fed.init_params()
    fed.set_local_data(*synthetic_shards(fed.n_local, 32, 8, 16, 10, seed=1, first_client=fed.local.start))
    host = FederationHost(fed, participant=fast_participant(federation)).start()
    seen = []
    host.bus.subscribe(lambda s, m: seen.append((s, m.split()[0])))
    try:
        if rank == 1 and gw_port:
            host.gateway(HUB, port=gw_port)
        with pytest.raises(ValueError):
            host.gateway(0 if rank == 1 else HUB)                   # a gateway opens on the rank that hosts the node
        if not gw_port and rank == 0:
            time.sleep(0.5)
            host.nodes[5].set_start_learning(rounds=2, epochs=1)    # a node of rank 1 starts: the call works anywhere
        assert host.wait_started(90)
        assert wait_for(lambda: host.round is None, 180)
        done = len(host.reports)          # (the last frames of the other rank are still on their way through the pumps)
        wait_for(lambda: sum(h == b"MODELS_READY" for _, h in list(seen)) >= N * done, 15)   # delivered, not just queued
        out[rank] = dict(params=fed.engine.params.clone(), reports=len(host.reports), names=list(host.names),
                         network=host.network_nodes(), counts=dict(host.bus.counts),
                         beats=sorted({s for s, h in seen if h == b"BEAT"}),
                         ready=sorted({s for s, h in seen if h == b"MODELS_READY"}),
                         local=list(fed.local))
    finally:
        host.stop()
        dist.destroy_process_group()


def _run_hosts(federation, gw_port):
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_host_worker, args=(r, 2, port, gw_port, out, federation), daemon=True) for r in range(2)]
    for p in procs:
        p.start()
    return procs, out


def _join(procs, timeout=240):
    for p in procs:
        p.join(timeout)
    alive = [p for p in procs if p.is_alive()]
    for p in alive:
        p.kill()
    assert not alive, "a rank host did not finish"
    assert [p.exitcode for p in procs] == [0, 0]


def test_two_rank_hosts_share_one_control_plane():
    """No TCP at all: a virtual node of rank 1 is started through rank 0's host; both ranks run the two rounds, every
    rank's bus carries the BEAT / MODELS_READY of all eight nodes, and the ring's models match the one-rank run."""
    procs, out = _run_hosts("DFL", 0)
    _join(procs)
    a, b = out[0], out[1]
    assert a["reports"] == b["reports"] == 2
    assert a["beats"] == b["beats"] == list(range(N)) and a["ready"] == b["ready"] == list(range(N))
    assert len(a["network"]) == len(b["network"]) == N
    assert a["counts"]["START_LEARNING"] == b["counts"]["START_LEARNING"] == 1
    # the same scenario on one rank
    from p2pfl_amd.datasets.synthetic import synthetic_shards
    from p2pfl_amd.runtime.federation import BatchedFederation
    adj = np.zeros((N, N), dtype=np.float32)
    for i in range(N):
        adj[i, (i + 1) % N] = adj[(i + 1) % N, i] = 1
    threads = torch.get_num_threads()
    torch.set_num_threads(1)              # like the rank workers: the very same summation order, so bit-equal models
    try:                                  # (Adam's first step turns rounding noise of tiny gradients into 1e-5 steps)
        ref = BatchedFederation(SPEC, adj, federation="DFL", device="cpu", engine="eager", epochs=1, seed=5)
        ref.init_params()
        ref.set_local_data(*synthetic_shards(N, 32, 8, 16, 10, seed=1))
        ref.run_round(); ref.run_round()
    finally:
        torch.set_num_threads(threads)
    got = torch.cat([a["params"], b["params"]], dim=0)
    assert torch.equal(got, ref.engine.params)


@pytest.mark.parametrize("what", ["two_rounds", "stop"])
def test_external_tcp_node_peers_with_a_virtual_node_on_rank_one(what):
    from tests.test_host_control_plane import _port_open, external_node, wait_for
    gw_port = _free_port()
    procs, out = _run_hosts("CFL", gw_port)
    ext = external_node(8, Role.TRAINER, "CFL")
    try:
        assert wait_for(lambda: _port_open(gw_port), 90)
        assert ext.connect_to("127.0.0.1", gw_port) is not None
        # the beats of ALL eight virtual nodes - four of them hosted by rank 0 - reach the TCP node through the gateway
        assert wait_for(lambda: len(ext.get_network_nodes()) == N + 1, 30), ext.get_network_nodes()
        assert {f"127.0.0.1:{45000 + i}" for i in range(N) if i != HUB} <= set(ext.get_network_nodes())
        if what == "two_rounds":
            ext.set_start_learning(rounds=2, epochs=1)
            assert wait_for(lambda: ext.round is not None, 20)
            assert wait_for(lambda: ext.round is None, 180)
            _join(procs)
            a, b = out[0], out[1]
            assert a["reports"] == b["reports"] == 2
            ref = ext.learner.get_parameters()
            from p2pfl_amd.runtime.slab import ParamLayout
            layout = ParamLayout(SPEC)
            for rank_out in (a, b):                       # every virtual node on BOTH ranks holds the external node's model
                assert ext.get_name() in rank_out["network"]
                for k in range(4):
                    sd = layout.row_to_state_dict(rank_out["params"][k])
                    for name in ref:
                        assert torch.allclose(sd[name], ref[name].cpu(), atol=1e-6), (rank_out["local"][k], name)
            assert a["names"][HUB] == b["names"][HUB] == f"127.0.0.1:{gw_port}"      # the gateway's name reached rank 0
        else:
            ext.set_start_learning(rounds=500, epochs=1)
            assert wait_for(lambda: ext.round is not None and ext.round >= 1, 120)
            ext.set_stop_learning()
            _join(procs)
            a, b = out[0], out[1]
            assert 1 <= a["reports"] == b["reports"] < 500           # both ranks left the loop at the same round
            assert wait_for(lambda: ext.round is None, 30)
    finally:
        ext.stop()
        for p in procs:
            if p.is_alive():
                p.kill()
        time.sleep(0.3)


def test_batched_launcher_control_plane_on_two_ranks(tmp_path):
    """``torch.distributed.run --nproc-per-node 2 -m p2pfl_amd.runtime.launch --control-plane --gateway 3:PORT``: the
    launcher no longer refuses (or silently shrinks to one rank); the gateway of participant 3 opens on the rank that
    hosts it, rank 0 starts the run, both ranks log their nodes' rounds."""
    import json
    import subprocess
    import sys
    from p2pfl_amd.controller import Controller
    from tests.test_host_control_plane import fast_participant
    cdir = os.path.join(str(tmp_path), "config", "scen")
    fast = fast_participant("DFL")
    over = {k: fast[k] for k in fast if k.isupper()}
    over["tracking_args"] = {"log_dir": str(tmp_path / "logs")}
    Controller.generate_participants(cdir, 4, "DFL", "MNIST", "MLP", rounds=2, epochs=1, accelerator="cpu",
                                     controller="", overrides=over)
    for i in range(4):
        path = os.path.join(cdir, f"participant_{i}.json")
        with open(path) as fh:
            p = json.load(fh)
        p["network_args"]["neighbors"] = " ".join(f"127.0.0.1:{45000 + j}" for j in ((i - 1) % 4, (i + 1) % 4))
        p["scenario_args"]["name"] = "scen"
        with open(path, "w") as fh:
            json.dump(p, fh)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gw_port = _free_port()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "-m", "p2pfl_amd.runtime.launch",
                        cdir, "--device", "cpu", "--control-plane", "--gateway", f"3:{gw_port}"], cwd=root,
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="2"))
    assert r.returncode == 0, r.stderr[-3000:]
    logs = tmp_path / "logs" / "scen"
    with open(logs / "summary.json") as f:
        s = json.load(f)
    assert s["nodes"] == 4 and s["world"] == 2 and s["local_nodes"] == 2 and s["rounds"] == 2
    for i in range(4):
        assert "round 1 role aggregator alive True" in (logs / f"participant_{i}.log").read_text()
    assert f"TCP gateway of virtual node 3 listening on 127.0.0.1:{gw_port}" in (logs / "participant_3.log").read_text()
    assert "TCP gateway" not in (logs / "participant_0.log").read_text()


def test_rank_bridge_refuses_connections_without_the_job_token(monkeypatch):
    """The mesh is for the ranks of ONE job: a process of the machine that dials a bridge port without the token the
    ranks exchanged over their process group is turned away, and the real rank still gets in."""
    import struct
    import threading

    from p2pfl_amd.runtime import bridge as B
    shared, cv = {}, threading.Condition()

    def fake_gather(out, obj, group=None):
        with cv:
            shared[group] = obj                      # group = the rank, in this test
            cv.notify_all()
            cv.wait_for(lambda: len(shared) == 2, timeout=10)
        out[0], out[1] = shared[0], shared[1]
        if group == 1:                               # before rank 1 dials: an intruder takes the first accept slot
            s = socket.create_connection(tuple(shared[0][:2]), timeout=5)
            s.sendall(struct.pack("!I", 1) + b"x" * 16)
            intruders.append(s)
    monkeypatch.setattr(B.dist, "all_gather_object", fake_gather)
    intruders, got, bridges = [], [], {}

    def make(rank):
        bridges[rank] = B.RankBridge(rank, 2, rank, lambda src, kind, sender, payload: got.append((rank, src, kind, sender, payload)),
                                     timeout=10)
    ts = [threading.Thread(target=make, args=(r,)) for r in (0, 1)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(20)
    assert sorted(bridges) == [0, 1]
    try:
        bridges[1].send_all(B.BUS, 5, b"BEAT 127.0.0.1:45005")
        deadline = time.time() + 5
        while not got and time.time() < deadline:
            time.sleep(0.01)
        assert got == [(0, 1, B.BUS, 5, b"BEAT 127.0.0.1:45005")]
        intruders[0].settimeout(2)
        assert intruders[0].recv(1) == b""           # closed by rank 0
        # an oversized frame ends the link instead of allocating what the header claims
        bridges[1]._peers[0].sendall(B.HEADER.pack(B.MAX_PAYLOAD + 1, B.BUS, 0))
        bridges[0]._threads[0].join(5)
        assert not bridges[0]._threads[0].is_alive()
    finally:
        for b in bridges.values():
            b.close()
        for s in intruders:
            s.close()
