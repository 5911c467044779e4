"""scenario.json: one file for all N nodes (config/scenario.py); either form can be produced from the other."""
import json
import os

import pytest

from p2pfl_amd.config.scenario import (SCENARIO_FILE, emit_participants, expand_scenario, pack_scenario, read_scenario,
                                       write_scenario)
from p2pfl_amd.controller import Controller
from p2pfl_amd.runtime.launch import default_participant, load_participants


def _participants(n):
    out = []
    for i in range(n):
        p = default_participant()
        p["scenario_args"].update(name="one-file", n_nodes=n, rounds=2)
        p["device_args"].update(idx=i, uid=f"uid{i}", start=(i == 0), role="aggregator" if i % 3 else "trainer")
        p["network_args"].update(ip="127.0.0.1", port=45000 + i,
                                 neighbors=" ".join(f"127.0.0.1:{45000 + j}" for j in ((i - 1) % n, (i + 1) % n)))
        if i == 2:
            p["adversarial_args"] = {"attacks": "Label Flipping", "poisoned_ratio": 40}
        out.append(p)
    return out


def test_pack_expand_round_trips_and_is_compact():
    parts = _participants(1000)
    doc = pack_scenario(parts)
    assert doc["n_nodes"] == 1000 and len(doc["nodes"]) == 1000
    assert expand_scenario(json.loads(json.dumps(doc))) == parts
    # a node lists only what it changes: far smaller than 1000 full documents
    assert len(json.dumps(doc)) < 0.25 * sum(len(json.dumps(p)) for p in parts)
    assert set(doc["nodes"][5]) <= {"device_args", "network_args"}
    assert "adversarial_args" in doc["nodes"][2]


def test_removed_keys_survive_the_round_trip():
    a = {"x": {"k": 1, "gone": 2}, "device_args": {"idx": 0}}
    b = {"x": {"k": 1}, "device_args": {"idx": 1}, "extra": [1, 2]}
    assert expand_scenario(pack_scenario([a, b])) == [a, b]


def test_short_hand_written_scenario_fills_the_missing_nodes():
    d = default_participant()
    d["network_args"]["port"] = 47000
    parts = expand_scenario({"version": 1, "n_nodes": 4, "defaults": d,
                             "nodes": [{"device_args": {"idx": 0, "start": True}}]})
    assert [p["device_args"]["idx"] for p in parts] == [0, 1, 2, 3]
    assert [p["network_args"]["port"] for p in parts] == [47000, 47001, 47002, 47003]
    assert [bool(p["device_args"]["start"]) for p in parts] == [True, False, False, False]
    with pytest.raises(ValueError):
        expand_scenario({"version": 2, "defaults": d, "nodes": [{}]})
    with pytest.raises(ValueError):
        expand_scenario({"version": 1, "n_nodes": 1, "defaults": d, "nodes": [{}, {}]})


def test_launcher_reads_a_directory_that_only_holds_scenario_json(tmp_path):
    parts = _participants(6)
    write_scenario(str(tmp_path), parts)
    assert os.listdir(tmp_path) == [SCENARIO_FILE]
    loaded = load_participants(str(tmp_path))
    assert [p["device_args"]["idx"] for p in loaded] == list(range(6))
    assert loaded[2]["adversarial_args"]["attacks"] == "Label Flipping"
    assert loaded[3]["network_args"]["neighbors"] == "127.0.0.1:45002 127.0.0.1:45004"
    # and the other way round
    files = emit_participants(str(tmp_path))
    assert [os.path.basename(f) for f in files] == [f"participant_{i}.json" for i in range(6)]
    docs = []
    for f in files:
        with open(f) as fh:
            docs.append(json.load(fh))
    assert read_scenario(str(tmp_path)) == docs == parts


def test_controller_writes_and_ingests_scenario_json(tmp_path):
    from test_controller import args_for
    a = args_for(tmp_path, topology="ring", engine="batched")
    cdir = os.path.join(a.config, "scen")
    Controller.generate_participants(cdir, 5, "DFL", "MNIST", "MLP", rounds=1, epochs=1, accelerator="cpu", controller="")
    c = Controller(a)
    c.start_nodes_batched = lambda: None
    c.load_configurations_and_start_nodes()
    sc = os.path.join(cdir, SCENARIO_FILE)
    per_file = []
    for i in range(5):
        with open(os.path.join(cdir, f"participant_{i}.json")) as f:
            per_file.append(json.load(f))
    assert read_scenario(sc) == per_file
    # a second controller given only scenario.json gets the participant files back
    for i in range(5):
        os.remove(os.path.join(cdir, f"participant_{i}.json"))
    c2 = Controller(a)
    c2.start_nodes_batched = lambda: None
    c2.load_configurations_and_start_nodes()
    for i in range(5):
        with open(os.path.join(cdir, f"participant_{i}.json")) as f:
            p = json.load(f)
        # (the controller draws the topology again, so only the ring itself is the same)
        assert {f"127.0.0.1:{per_file[j]['network_args']['port']}" for j in ((i - 1) % 5, (i + 1) % 5)} \
            <= set(p["network_args"]["neighbors"].split())
        assert p["network_args"]["port"] == per_file[i]["network_args"]["port"]
        assert p["device_args"]["idx"] == i and p["device_args"]["role"] == per_file[i]["device_args"]["role"]


def test_two_rank_launch_honours_device_args_placement(tmp_path):
    """``device_args.placement`` = the rank that hosts a node: the launcher orders the nodes into per-rank blocks
    internally, everything that leaves the process (logs, metrics, checkpoints, fault schedule, events) keeps the
    participant's own index."""
    import socket
    import subprocess
    import sys
    cdir = os.path.join(str(tmp_path), "config", "scen")
    Controller.generate_participants(cdir, 5, "DFL", "MNIST", "MLP", rounds=2, epochs=1, accelerator="cpu",
                                     controller="", overrides={"tracking_args": {"log_dir": str(tmp_path / "logs")}})
    placement = [1, 0, 1, 0, 0]
    for i in range(5):
        path = os.path.join(cdir, f"participant_{i}.json")
        with open(path) as f:
            p = json.load(f)
        p["network_args"]["neighbors"] = " ".join(f"127.0.0.1:{45000 + j}" for j in ((i - 1) % 5, (i + 1) % 5))
        p["scenario_args"]["name"] = "scen"
        p["device_args"]["placement"] = placement[i]
        with open(path, "w") as f:
            json.dump(p, f)
    with open(os.path.join(cdir, "faults.json"), "w") as f:
        json.dump([{"round": 1, "action": "kill", "node": 2}], f)        # participant 2 lives on rank 1
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "p2pfl_amd.runtime.launch",
                        cdir, "--device", "cpu", "--save-models"], cwd=root, capture_output=True, text=True,
                       timeout=600, env=dict(os.environ, OMP_NUM_THREADS="2"))
    assert r.returncode == 0, r.stderr[-3000:]
    logs = tmp_path / "logs" / "scen"
    with open(logs / "summary.json") as f:
        s = json.load(f)
    assert s["nodes"] == 5 and s["world"] == 2 and s["local_nodes"] == 3 and s["rounds"] == 2
    assert s["placement"] == {"block_sizes": [3, 2], "order": [1, 3, 4, 0, 2]}
    assert ["kill", 2] in [e[1:] for e in s["membership_events"]]
    for i in range(5):
        assert (logs / f"participant_{i}.log").exists() and (logs / f"participant_{i}_final.ckpt").exists()
        assert (logs / "metrics" / f"participant_{i}" / "metrics.csv").exists()
    with open(logs / "participant_2.log") as f:
        log2 = f.read()
    assert "round 0 role aggregator alive True" in log2 and "round 1 role aggregator alive False" in log2
    with open(logs / "participant_0.log") as f:
        assert "round 1 role aggregator alive True" in f.read()


# ------------------------------------------------------------------ typed view (config/schema.py)
def test_typed_view_coerces_defaults_and_round_trips():
    from p2pfl_amd.config.config import Config
    from p2pfl_amd.config.schema import ParticipantView
    d = default_participant()
    d["device_args"].update(idx="3", start="true", placement="1", gpu="")
    d["network_args"].update(port="45003", neighbors="127.0.0.1:45002 127.0.0.1:45004")
    d["geo_args"].update(latitude="38.02", longitude=-1.17)
    d["adversarial_args"] = {"attacks": "No Attack"}
    d["data_args"]["root_dir"] = "/data"
    v = ParticipantView.from_dict(d)
    assert v.device.idx == 3 and v.device.start is True and v.device.placement == 1 and v.device.gpu is None
    assert v.network.port == 45003 and v.network.neighbor_list() == ["127.0.0.1:45002", "127.0.0.1:45004"]
    assert v.tunables.BLOCK_SIZE == 2048 and v.dataset == "MNIST" and v.epochs == 3
    back = v.to_dict()
    assert back["adversarial_args"] == {"attacks": "No Attack"} and back["data_args"] == {"dataset": "MNIST", "root_dir": "/data"}
    assert back["network_args"]["delay-distro"] == 0 and back["device_args"]["idx"] == 3
    assert ParticipantView.from_dict(back) == v
    assert Config(participant=d).view().device.idx == 3


@pytest.mark.parametrize("section,key,value", [
    ("scenario_args", "federation", "XFL"), ("device_args", "role", "boss"), ("device_args", "accelerator", "tpu"),
    ("network_args", "port", 70000), ("network_args", "port", "http"), ("network_args", "neighbors", "127.0.0.1"),
    ("geo_args", "latitude", 123.0), ("training_args", "epochs", 0), ("aggregator_args", "algorithm", "Krum"),
    (None, "NODE_TIMEOUT", 2), (None, "GOSSIP_MODELS_FREC", 0), ("device_args", "placement", -1)])
def test_typed_view_rejects_what_the_runtime_cannot_act_on(section, key, value):
    from p2pfl_amd.config.config import Config
    d = default_participant()
    if section is None:
        d[key] = value
    else:
        d[section][key] = value
    with pytest.raises(ValueError):
        Config.validate(d)


def test_scenario_wide_validation():
    from p2pfl_amd.config.schema import validate_scenario
    parts = _participants(4)
    assert [v.device.idx for v in validate_scenario(parts)] == [0, 1, 2, 3]
    two_starts = _participants(4)
    two_starts[2]["device_args"]["start"] = True
    same_port = _participants(4)
    same_port[1]["network_args"]["port"] = 45000
    self_nb = _participants(4)
    self_nb[1]["network_args"]["neighbors"] += " 127.0.0.1:45001"
    mixed = _participants(4)
    mixed[3]["scenario_args"]["federation"] = "CFL"
    cfl = _participants(4)
    for p in cfl:
        p["scenario_args"]["federation"] = "CFL"       # nobody is the server
    for bad in (two_starts, same_port, self_nb, mixed, cfl, []):
        with pytest.raises(ValueError):
            validate_scenario(bad)
