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
        assert p["network_args"]["neighbors"] == per_file[i]["network_args"]["neighbors"]
        assert p["device_args"]["role"] == per_file[i]["device_args"]["role"]
