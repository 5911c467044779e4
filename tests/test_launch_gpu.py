"""Whole scenarios through the batched launcher on the HIP engine: controller-made participant files ->
one BatchedFederation on the GPU -> logs / metrics / checkpoints / resume in the reference's layout."""
import json
import os

import pytest

pytestmark = pytest.mark.gpu


def _scenario(tmp_path, n, federation, dataset, model, rounds, topology="ring"):
    from p2pfl_amd.controller import Controller
    cdir = os.path.join(str(tmp_path), "config", "scen")
    Controller.generate_participants(cdir, n, federation, dataset, model, rounds=rounds, epochs=1, accelerator="gpu",
                                     controller="", overrides={"tracking_args": {"log_dir": str(tmp_path / "logs")}})
    for i in range(n):
        path = os.path.join(cdir, f"participant_{i}.json")
        p = json.load(open(path))
        if topology == "ring":
            nbrs = ((i - 1) % n, (i + 1) % n)
        else:      # star around node 0
            nbrs = tuple(range(1, n)) if i == 0 else (0,)
        p["network_args"]["neighbors"] = " ".join(f"127.0.0.1:{45000 + j}" for j in nbrs)
        p["scenario_args"]["name"] = "scen"
        json.dump(p, open(path, "w"))
    return cdir


@pytest.mark.parametrize("dataset,model,federation,topology", [("MNIST", "CNN", "DFL", "ring"),
                                                               ("MNIST", "MLP", "CFL", "star"),
                                                               ("FEMNIST", "CNN", "SDFL", "ring"),
                                                               ("WADI", "MLP", "DFL", "ring"),            # one logit: BCE kernel
                                                               ("SYSCALL", "Autoencoder", "DFL", "ring"),  # reconstruction loss
                                                               # no kernel-level spec: stock ops, state in the slab
                                                               ("SYSCALL", "MLP", "DFL", "ring"),
                                                               ("CIFAR10", "simplemobilenet", "DFL", "ring")])
def test_scenario_on_the_hip_engine(tmp_path, dataset, model, federation, topology):
    from p2pfl_amd.runtime.launch import run_scenario
    cdir = _scenario(tmp_path, 4, federation, dataset, model, rounds=2, topology=topology)
    json.dump([{"round": 1, "action": "kill", "node": 3}], open(os.path.join(cdir, "faults.json"), "w"))
    s = run_scenario(cdir, device="cuda", save_models=True)
    ev = [e[1] for e in s["membership_events"]]
    assert s["nodes"] == 4 and s["rounds"] == 2 and ev[0] == "kill"
    # whoever expected node 3's model waited AGGREGATION_TIMEOUT (virtual clock), which is what evicts it; in SDFL
    # the dead trainer may not be in the leader's train set at all - then nobody waits and it is not evicted yet
    assert ("evicted" in ev) == ("aggregation_timeout" in ev)
    logs = tmp_path / "logs" / "scen"
    assert "alive False" in open(logs / "participant_3.log").read()
    rows = open(logs / "metrics" / "participant_0" / "metrics.csv").read()
    assert "Test/Loss" in rows and "Federation/Phase_train_seconds" in rows
    assert (logs / "participant_0_final.ckpt").exists()
    # (the schedule stays: the resumed run replays the kill of round 1 without training and does round 2 of 3)
    s2 = run_scenario(cdir, device="cuda", rounds=3, resume=str(logs))
    assert s2["first_round"] == 2 and s2["rounds"] == 1 and [e[1:] for e in s2["membership_events"] if e[1] == "kill"] == [["kill", 3]]
    if dataset == "MNIST":       # synthetic MNIST-shaped shards are learnable: resumed weights evaluate well at once
        first_eval = [l for l in open(logs / "participant_1.log").read().splitlines() if "round 2 " in l][0]
        assert float(first_eval.split("test_acc ")[1].split()[0]) > 0.5


@pytest.mark.gpu
@pytest.mark.parametrize("ds,model,seg", [("SYSCALL", "MLP", None), ("SYSCALL", "SVM", None), ("WADI", "MLP", None),
                                          ("SYSCALL", "MLP", 2)])
def test_module_engine_graph_replay_matches_eager_batched_passes(ds, model, seg, monkeypatch):
    """ModuleEngine on the GPU: passes replayed as hipGraphs (captured at their second occurrence; the optimizer object
    kept and reset in place per round) against the same client-batched passes launched eagerly. ``seg``: a pass split
    into graph segments of that many steps (how long passes are captured: one graph of a 390-step CIFAR-10 pass
    crashes hipStreamEndCapture, tools/probe/capture_augment.py)."""
    if seg:
        monkeypatch.setenv("P2PFL_MODULE_GRAPH_STEPS", str(seg))
    import torch
    from p2pfl_amd.node_start import build_dataset_and_model, build_model
    from p2pfl_amd.runtime.module_engine import ModuleEngine, ModuleSpec
    monkeypatch.setattr(ModuleEngine, "LOG_EVERY", 2)
    monkeypatch.setenv("P2PFL_MODULE_VMAP_STRICT", "1")
    probe = build_dataset_and_model(ds, model, 0, 20)[0]
    spec = ModuleSpec(f"{ds}-{model}", lambda: build_model(ds, model), probe.num_classes, tuple(probe.input_shape))
    K, n = 4, 96
    t = probe.tensors()
    reps = -(-n // t[0].shape[0])
    data = [torch.stack([torch.cat([t[j]] * reps)[:n if j < 2 else 32] for _ in range(K)]) for j in range(6)]
    engines = {}
    for graph in ("0", "1"):
        monkeypatch.setenv("P2PFL_MODULE_GRAPH", graph)
        eng = ModuleEngine(spec, K, "cuda", seed=1, batched=True, batch_size=32, log_steps=True)
        for m in eng.modules:
            for mod in m.modules():
                if isinstance(mod, torch.nn.Dropout):
                    mod.p = 0.0
        eng.set_data(*data)
        eng.init_params(identical=False)
        engines[graph] = eng
    assert engines["1"].use_graph and not engines["0"].use_graph
    for rnd in range(2):
        res = {}
        for graph, eng in engines.items():
            ev = eng.evaluate()
            out = eng.fit(2, active=torch.tensor([True, True, False, True]))
            res[graph] = (ev["loss"], out["Train"]["loss"], out["Train"]["count"])
        assert torch.allclose(res["0"][0], res["1"][0], atol=2e-3, rtol=2e-3), rnd
        assert torch.allclose(res["0"][1], res["1"][1], atol=2e-3, rtol=2e-3), rnd
        assert torch.equal(res["0"][2], res["1"][2])
        # per-step series written from inside the replayed passes (epoch-local tables, collected after every epoch)
        l0, l1 = engines["0"].step_losses(), engines["1"].step_losses()
        assert l0.shape == l1.shape == (6, K) and torch.allclose(l0, l1, atol=2e-3, rtol=2e-3), rnd
    g = engines["1"]
    assert g.use_graph and len(g._graphs) >= 3          # test pass, training epoch, validation pass
    if seg:                                             # 96 samples = 3 steps: two segments per training pass
        assert max(len(v) for v in g._graphs.values()) == 2
    assert torch.equal(g.params[2], engines["0"].params[2])          # the idle node kept its (identical) model
    if engines["0"].classifier:
        c0, c1 = engines["0"].step_confusions(), g.step_confusions()
        assert sorted(c0) == sorted(c1) == [2, 4, 6]
        for t in c0:                                     # (the idle node trains too and is restored afterwards)
            assert c1[t].sum(dim=(1, 2)).tolist() == [32] * K
    g.set_data(*data)
    assert not g._graphs                                 # new data: the captured passes are dropped
    g.evaluate(); g.fit(1)
