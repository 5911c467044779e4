"""Controller: scenario generation, participant rewriting, both launchers (config #1 = 2-node
MNIST-MLP over TCP with one OS process per node; batched = one process hosting all nodes)."""
import argparse
import glob
import json
import os

import pytest

from p2pfl_amd.controller import Controller


def args_for(tmp_path, **kw):
    d = dict(federation="DFL", topology="fully", webserver=False, simulation=True, docker=False,
             config=str(tmp_path / "config"), logs=str(tmp_path / "logs"), env=None, python=None,
             scenario_name="scen", engine="auto")
    d.update(kw)
    os.makedirs(d["logs"], exist_ok=True)
    return argparse.Namespace(**d)


def test_participants_are_rewritten_like_the_reference(tmp_path):
    a = args_for(tmp_path, topology="ring", engine="batched")
    Controller.generate_participants(os.path.join(a.config, "scen"), 5, "DFL", "MNIST", "MLP", rounds=1, epochs=1,
                                     accelerator="cpu", controller="")
    c = Controller(a)
    c.start_nodes_batched = lambda: None           # only the preparation is under test
    c.load_configurations_and_start_nodes()
    parts = [json.load(open(f)) for f in sorted(glob.glob(os.path.join(a.config, "scen", "participant_*.json")))]
    assert len(parts) == 5 and sum(p["device_args"]["start"] for p in parts) == 1
    for i, p in enumerate(parts):
        assert p["device_args"]["idx"] == i and len(p["device_args"]["uid"]) == 40
        assert p["scenario_args"]["n_nodes"] == 5 and p["scenario_args"]["name"] == "scen"
        assert len(p["network_args"]["neighbors"].split()) >= 2          # ring (+10 % random links)
        assert p["tracking_args"]["log_dir"] == a.logs
        assert isinstance(p["geo_args"]["latitude"], float)
    assert os.path.exists(os.path.join(a.logs, "scen", "topology.png"))


def test_two_start_nodes_are_rejected(tmp_path):
    a = args_for(tmp_path)
    files = Controller.generate_participants(os.path.join(a.config, "scen"), 3, accelerator="cpu", controller="")
    p = json.load(open(files[1]))
    p["device_args"]["start"] = True
    json.dump(p, open(files[1], "w"))
    with pytest.raises(ValueError):
        Controller(a).load_configurations_and_start_nodes()


def test_cfl_star_and_unknown_topology(tmp_path):
    a = args_for(tmp_path, federation="CFL", topology="star", engine="batched")
    Controller.generate_participants(os.path.join(a.config, "scen"), 4, "CFL", accelerator="cpu", controller="")
    c = Controller(a)
    c.start_nodes_batched = lambda: None
    c.load_configurations_and_start_nodes()
    p0 = json.load(open(os.path.join(a.config, "scen", "participant_0.json")))
    assert p0["device_args"]["role"] == "server" and len(p0["network_args"]["neighbors"].split()) == 3
    with pytest.raises(ValueError):
        Controller(args_for(tmp_path, topology="star", federation="DFL")).load_configurations_and_start_nodes()


def test_docker_compose_generation(tmp_path):
    a = args_for(tmp_path, docker=True)
    Controller.generate_participants(os.path.join(a.config, "scen"), 3, accelerator="cpu", controller="")
    c = Controller(a)
    c.load_configurations_and_start_nodes()
    text = open(os.path.join(a.config, "scen", "docker-compose.yml")).read()
    assert text.count("participant") >= 3 and "/dev/kfd" in text and "ipv4_address: 192.168.50.2" in text


@pytest.mark.slow
def test_batched_launcher_runs_scenario(tmp_path, monkeypatch):
    monkeypatch.setenv("P2PFL_SYNTHETIC_MNIST_ROWS", "12000")     # (the launcher process trains on the CPU here)
    a = args_for(tmp_path, topology="ring", engine="batched")
    Controller.generate_participants(os.path.join(a.config, "scen"), 4, "DFL", "MNIST", "MLP", rounds=2, epochs=1,
                                     accelerator="cpu", controller="")
    c = Controller(a)
    c.load_configurations_and_start_nodes()
    assert c.wait(timeout=300) == [0]
    s = json.load(open(os.path.join(a.logs, "scen", "summary.json")))
    assert s["nodes"] == 4 and s["rounds"] == 2 and s["final_test_accuracy"] > 0.5
    assert os.path.exists(os.path.join(a.logs, "scen", "metrics", "participant_3", "metrics.csv"))


def test_batched_launcher_faults_checkpoints_and_resume(tmp_path):
    """faults.json drives the virtual membership; --save-models writes reference-format checkpoints; a second
    run resumes every node from them and continues the round numbering."""
    from p2pfl_amd.runtime.launch import run_scenario
    cdir = os.path.join(str(tmp_path), "config", "scen")
    Controller.generate_participants(cdir, 4, "DFL", "MNIST", "MLP", rounds=2, epochs=1, accelerator="cpu",
                                     controller="", overrides={"tracking_args": {"log_dir": str(tmp_path / "logs")}})
    # neighbours (ring) written by hand: the controller normally does it in load_configurations_and_start_nodes
    for i in range(4):
        path = os.path.join(cdir, f"participant_{i}.json")
        p = json.load(open(path))
        p["network_args"]["neighbors"] = " ".join(f"127.0.0.1:{45000 + j}" for j in ((i - 1) % 4, (i + 1) % 4))
        p["scenario_args"]["name"] = "scen"
        json.dump(p, open(path, "w"))
    json.dump([{"round": 1, "action": "kill", "node": 3}, {"round": 2, "action": "revive", "node": 3}],
              open(os.path.join(cdir, "faults.json"), "w"))
    s1 = run_scenario(cdir, device="cpu", save_models=True)
    assert s1["rounds"] == 2 and s1["first_round"] == 0 and [e[1] for e in s1["membership_events"]] == ["kill", "aggregation_timeout", "aggregation_timeout", "evicted"]
    log3 = open(tmp_path / "logs" / "scen" / "participant_3.log").read()
    assert "round 0 role aggregator alive True" in log3 and "round 1 role aggregator alive False" in log3
    rows = open(tmp_path / "logs" / "scen" / "metrics" / "participant_0" / "metrics.csv").read()
    assert "Federation/RoundSeconds" in rows and "Federation/Phase_train_seconds" in rows
    ck = tmp_path / "logs" / "scen" / "participant_0_final.ckpt"
    assert ck.exists() and json.load(open(str(ck) + ".json"))["round"] == 2
    # the resumed run does what is left of a 3-round experiment; the fault schedule (absolute rounds) continues:
    # the kill of round 1 is replayed without training, the revival of round 2 happens in the resumed round
    s2 = run_scenario(cdir, device="cpu", rounds=3, resume=str(tmp_path / "logs" / "scen"))
    assert s2["first_round"] == 2 and s2["rounds"] == 1
    assert [e[1:] for e in s2["membership_events"] if e[1] in ("kill", "revive")] == [["kill", 3], ["revive", 3]]
    assert "round 2 role aggregator alive True" in open(tmp_path / "logs" / "scen" / "participant_3.log").read()
    # resumed from trained weights: the very first evaluation is already good
    first_eval = [l for l in open(tmp_path / "logs" / "scen" / "participant_1.log").read().splitlines() if "round 2 " in l][0]
    assert float(first_eval.split("test_acc ")[1].split()[0]) > 0.5


@pytest.mark.slow
def test_config1_two_processes_tcp_mnist_mlp(tmp_path, monkeypatch):
    monkeypatch.setenv("P2PFL_SYNTHETIC_MNIST_ROWS", "12000")
    monkeypatch.setenv("P2PFL_START_DELAY", "1")
    monkeypatch.setenv("P2PFL_LAUNCH_GAP", "0.5")
    a = args_for(tmp_path, engine="process")
    fast = {"HEARTBEAT_PERIOD": 0.5, "WAIT_HEARTBEATS_CONVERGENCE": 1, "GOSSIP_MODELS_FREC": 20,
            "AGGREGATION_TIMEOUT": 30, "tracking_args": {"local_tracking": "csv"}}
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); base = s.getsockname()[1]; s.close()
    Controller.generate_participants(os.path.join(a.config, "scen"), 2, "DFL", "MNIST", "MLP", rounds=2, epochs=1,
                                     base_port=base, accelerator="cpu", controller="", overrides=fast)
    c = Controller(a)
    c.load_configurations_and_start_nodes()
    codes = c.wait(timeout=240)
    logs = [open(os.path.join(a.logs, "scen", f"participant_{i}.log")).read() for i in range(2)]
    assert codes == [0, 0], logs
    assert all("FL experiment finished" in l for l in logs)
    assert os.path.exists(os.path.join(a.logs, "scen", "metrics", "participant_0", "metrics.csv"))
