"""Batched runtime for model families without a kernel-level spec (runtime/module_engine.py)."""
import json
import os

import numpy as np
import pytest
import torch

from p2pfl_amd.node_start import build_dataset_and_model, build_model
from p2pfl_amd.runtime.federation import BatchedFederation
from p2pfl_amd.runtime.module_engine import ModuleEngine, ModuleSpec, StateLayout
from p2pfl_amd.utils.topologymanager import TopologyManager


def _spec(ds="SYSCALL", model="MLP"):
    probe = build_dataset_and_model(ds, model, 0, 20)[0]
    return ModuleSpec(f"{ds}-{model}", lambda: build_model(ds, model), probe.num_classes, tuple(probe.input_shape))


def test_state_layout_covers_buffers_and_keeps_integer_entries():
    m = build_model("SYSCALL", "MLP")
    sd = m.state_dict()
    lay = StateLayout(sd)
    assert "batchnorm1.running_mean" in lay.by_name and lay.int_keys == ["batchnorm1.num_batches_tracked",
                                                                          "batchnorm2.num_batches_tracked"]
    assert lay.row_stride % 64 == 0 and all(s.offset % 64 == 0 for s in lay.slots)
    row = torch.zeros(lay.row_stride)
    ints = lay.arrays_to_row([v.numpy() for v in sd.values()], row)
    arrays = lay.full_arrays(row, ints)
    assert len(arrays) == len(sd)
    for a, (k, v) in zip(arrays, sd.items()):
        assert a.shape == tuple(v.shape) and np.allclose(a, v.numpy())


def test_module_state_aliases_the_slab_and_training_writes_through():
    eng = ModuleEngine(_spec(), 3, "cpu", seed=1)
    shards = [build_dataset_and_model("SYSCALL", "MLP", i, 20)[0] for i in range(3)]
    eng.set_data(*(torch.stack([s.tensors()[j][:40] for s in shards]) for j in range(6)))
    eng.init_params()
    assert torch.equal(eng.params[0], eng.params[2])
    w = eng.layout.view(eng.params, "l1.weight")
    assert eng.modules[1].l1.weight.data_ptr() == w[1].data_ptr()
    rm = eng.layout.view(eng.params, "batchnorm1.running_mean")
    before = eng.params.clone()
    l0 = eng.evaluate()["loss"]
    out = eng.fit(2)
    assert not torch.equal(before, eng.params) and not torch.equal(rm[0], before[0, :0].new_zeros(rm[0].shape))
    assert torch.equal(eng.modules[2].batchnorm1.running_mean, rm[2])          # BN statistics live in the row
    assert (eng.evaluate()["loss"] < l0).all() and out["Train"]["count"].tolist() == [40] * 3
    # an aggregation result written into the slab is what the modules compute with
    eng.params[1].copy_(eng.params[0])
    x = eng.test_x[0][:8]
    eng.modules[0].eval(); eng.modules[1].eval()
    assert torch.equal(eng.modules[0](x), eng.modules[1](x))
    # payload round trip incl. the integer buffers
    arrays = eng.get_arrays(0)
    assert len(arrays) == len(eng.modules[0].state_dict()) and arrays[-1].shape == eng.modules[0].l3.bias.shape
    eng.set_arrays(2, arrays)
    assert torch.equal(eng.params[2], eng.params[0])
    assert int(eng.modules[2].batchnorm1.num_batches_tracked) == int(eng.modules[0].batchnorm1.num_batches_tracked)


def test_federation_of_batchnorm_models_averages_rows_over_the_train_sets():
    tm = TopologyManager(n_nodes=4, b_symmetric=True)
    tm.generate_ring_topology()
    fed = BatchedFederation(_spec(), tm.topology, device="cpu", epochs=1, seed=3)
    shards = [build_dataset_and_model("SYSCALL", "MLP", i, 20)[0] for i in range(4)]
    fed.set_local_data(*(torch.stack([s.tensors()[j][:40] for s in shards]) for j in range(6)))
    fed.init_params()
    g = torch.Generator().manual_seed(0)
    fed.engine.params.copy_(torch.randn(fed.engine.params.shape, generator=g))
    before = fed.engine.params.clone()
    fed.aggregate()
    for i in range(4):
        ts = [(i - 1) % 4, i, (i + 1) % 4]
        assert torch.allclose(fed.engine.params[i], before[ts].mean(0), atol=1e-6)
    rep = fed.run_round()
    assert rep.train_samples == 4 * 40 and torch.isfinite(fed.engine.params).all()


@pytest.mark.parametrize("dataset,model", [("SYSCALL", "MLP"), ("SYSCALL", "Autoencoder"), ("WADI", "MLP")])
def test_launcher_runs_non_spec_families_with_faults_checkpoints_resume(tmp_path, dataset, model):
    from p2pfl_amd.controller import Controller
    from p2pfl_amd.runtime.launch import run_scenario
    cdir = os.path.join(str(tmp_path), "config", "scen")
    Controller.generate_participants(cdir, 3, "DFL", dataset, model, rounds=2, epochs=1, accelerator="cpu",
                                     controller="", overrides={"tracking_args": {"log_dir": str(tmp_path / "logs")}})
    for i in range(3):
        path = os.path.join(cdir, f"participant_{i}.json")
        p = json.load(open(path))
        p["network_args"]["neighbors"] = " ".join(f"127.0.0.1:{45000 + j}" for j in ((i - 1) % 3, (i + 1) % 3))
        p["scenario_args"]["name"] = "scen"
        json.dump(p, open(path, "w"))
    json.dump([{"round": 1, "action": "kill", "node": 2}], open(os.path.join(cdir, "faults.json"), "w"))
    s = run_scenario(cdir, device="cpu", save_models=True)
    assert s["nodes"] == 3 and s["rounds"] == 2 and np.isfinite(s["final_test_loss"])
    logs = tmp_path / "logs" / "scen"
    assert "alive False" in open(logs / "participant_2.log").read()
    # the per-step series of the reference's models (every 20 optimizer steps) for the stock-op families too
    rows = [r.split(",") for r in open(logs / "metrics" / "participant_0" / "metrics.csv").read().strip().splitlines()[1:]]
    at20 = {r[2] for r in rows if int(r[1]) == 20}
    assert "Train/Loss" in at20 and ("Train/F1Score" in at20) == (model != "Autoencoder")
    s2 = run_scenario(cdir, device="cpu", rounds=3, resume=str(logs))
    assert s2["first_round"] == 2 and s2["final_test_loss"] <= s["final_test_loss"] * 1.5 + 1e-3


def test_unknown_family_is_refused():
    from p2pfl_amd.controller import Controller
    from p2pfl_amd.runtime.launch import run_scenario
    import tempfile
    tmp = tempfile.mkdtemp()
    cdir = os.path.join(tmp, "config", "scen")
    Controller.generate_participants(cdir, 2, "DFL", "MNIST", "MLP", rounds=1, epochs=1, accelerator="cpu", controller="")
    for i in range(2):
        path = os.path.join(cdir, f"participant_{i}.json")
        with open(path) as f:
            p = json.load(f)
        p["model_args"]["model"] = "Transformer"
        with open(path, "w") as f:
            json.dump(p, f)
    with pytest.raises(ValueError):
        run_scenario(cdir, device="cpu")


# ------------------------------------------------------------------ 2 ranks over gloo
def _two_rank_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    tm = TopologyManager(n_nodes=6, b_symmetric=True)
    tm.generate_ring_topology()
    fed = BatchedFederation(_spec(), tm.topology, device="cpu", epochs=1, seed=3, rank=rank, world_size=world,
                            process_group=dist.group.WORLD)
    shards = [build_dataset_and_model("SYSCALL", "MLP", i, 20)[0] for i in fed.local]
    fed.set_local_data(*(torch.stack([s.tensors()[j][:40] for s in shards]) for j in range(6)))
    fed.init_params()
    rows = torch.randn(6, fed.engine.params.shape[1], generator=torch.Generator().manual_seed(0))
    fed.engine.params.copy_(rows[fed.local.start:fed.local.stop])
    fed.aggregate()
    got = fed.engine.params.clone()
    rep = fed.run_round()                       # and a whole round (training with dropout: finite, not pinned)
    out[rank] = (got, bool(torch.isfinite(fed.engine.params).all()), rep.exchanged_rows)
    dist.destroy_process_group()


def test_two_ranks_exchange_rows_of_module_state():
    """Rows of nn.Module state (BatchNorm statistics included) cross the rank boundary like spec-model rows."""
    import socket

    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = mp.Manager().dict()
    mp.spawn(_two_rank_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.cat([out[0][0], out[1][0]], dim=0)
    rows = torch.randn(6, got.shape[1], generator=torch.Generator().manual_seed(0))
    for i in range(6):
        assert torch.allclose(got[i], rows[[(i - 1) % 6, i, (i + 1) % 6]].mean(0), atol=1e-6)
    assert out[0][1] and out[1][1] and out[0][2] == 2 and out[1][2] == 2


def _pair(ds, model, K=3, n=24, B=8, **kw):
    probe = build_dataset_and_model(ds, model, 0, 20)[0]
    spec = ModuleSpec(f"{ds}-{model}", lambda: build_model(ds, model), probe.num_classes, tuple(probe.input_shape))
    shards = [build_dataset_and_model(ds, model, i, 20)[0] for i in range(K)]
    data = [torch.stack([s.tensors()[j][:n] for s in shards]) for j in range(6)]
    engines = []
    for batched in (False, True):
        eng = ModuleEngine(spec, K, "cpu", seed=1, batched=batched, batch_size=B, **kw)
        eng.set_data(*data)
        eng.init_params(identical=False)
        for m in eng.modules:
            for mod in m.modules():
                if isinstance(mod, torch.nn.Dropout):
                    mod.p = 0.0                          # the two paths draw their masks from different streams
        engines.append(eng)
    return engines


@pytest.mark.parametrize("ds,model", [("SYSCALL", "MLP"), ("SYSCALL", "SVM"), ("SYSCALL", "Autoencoder"), ("WADI", "MLP")])
def test_client_batched_step_matches_the_per_node_loop(ds, model, monkeypatch):
    """torch.func.vmap over all replicas (one stacked optimizer) == K Python-looped replicas with K optimizers:
    weights, BatchNorm running statistics (kept in the slab), integer buffers, metrics."""
    monkeypatch.setenv("P2PFL_MODULE_VMAP_STRICT", "1")
    loop, vm = _pair(ds, model)
    assert vm.batched and not loop.batched
    e0, e1 = loop.evaluate(), vm.evaluate()
    assert torch.allclose(e0["loss"], e1["loss"], atol=1e-5) and torch.equal(e0["count"], e1["count"])
    active = torch.tensor([True, False, True])
    before = vm.params[1].clone()
    o0, o1 = loop.fit(2, active=active), vm.fit(2, active=active)
    assert vm.batched, "the batched path must not have fallen back"
    assert torch.allclose(o0["Train"]["loss"][active], o1["Train"]["loss"][active], atol=1e-4)
    assert torch.equal(o0["Train"]["confusion"][active], o1["Train"]["confusion"][active])
    assert torch.equal(vm.params[1], before) and torch.equal(loop.params[1], before)      # sat the round out
    for s in loop.layout.slots:
        a, b = loop.layout.view(loop.params, s.name), vm.layout.view(vm.params, s.name)
        if "batchnorm1.weight" in loop.layout.by_name:
            # a bias in front of BatchNorm has a zero true gradient: Adam turns rounding noise into +-lr steps, and
            # the running mean behind it averages that wandering bias in. Neither changes what the model computes
            # in training mode; the evaluation losses below cover their net effect.
            if s.name in ("l1.bias", "l2.bias", "batchnorm1.running_mean", "batchnorm2.running_mean"):
                continue
        assert torch.allclose(a, b, atol=2e-4), s.name
    for k in range(3):
        for (n0, b0), (_, b1) in zip(loop.modules[k].named_buffers(), vm.modules[k].named_buffers()):
            if not torch.is_floating_point(b0):
                assert torch.equal(b0, b1), n0          # num_batches_tracked advanced alike (not for the idle node)
    t0, t1 = loop.evaluate(), vm.evaluate()
    assert torch.allclose(t0["loss"], t1["loss"], atol=1e-2)      # (eval mode sees the wandering pre-BatchNorm biases through the running means)
    # the replicas still see the slab: an aggregation result written into a row is what they compute with
    vm.params[1].copy_(vm.params[0])
    x = vm.test_x[0][:8]
    vm.modules[0].eval(); vm.modules[1].eval()
    assert torch.equal(vm.modules[0](x), vm.modules[1](x))


def test_client_batched_falls_back_when_vmap_cannot_express_the_step():
    """A family with data-dependent Python control flow in its loss trains through the per-node loop."""
    class Odd(torch.nn.Module):
        is_classifier = True

        def __init__(self):
            super().__init__()
            self.l = torch.nn.Linear(4, 3)

        def forward(self, x):
            return self.l(x)

        def training_loss(self, x, y):
            out = self(x)
            scale = 2.0 if float(out.sum()) > 0 else 1.0           # .item(): not traceable by vmap
            return torch.nn.functional.cross_entropy(out, y) * scale, out.detach().argmax(1)

        def configure_optimizers(self):
            return torch.optim.SGD(self.parameters(), lr=0.1)
    spec = ModuleSpec("odd", Odd, 3, (4,))
    eng = ModuleEngine(spec, 2, "cpu", batch_size=4)
    assert eng.batched
    g = torch.Generator().manual_seed(0)
    x, y = torch.randn(2, 8, 4, generator=g), torch.randint(0, 3, (2, 8), generator=g)
    eng.set_data(x, y, x, y, x, y)
    before = eng.params.clone()
    l0 = eng.evaluate()["loss"]
    assert not eng.batched                                          # evaluation already declined
    out = eng.fit(3)
    assert out["Train"]["count"].tolist() == [8, 8] and not torch.equal(before, eng.params)
    assert (eng.evaluate()["loss"] < l0 * 2.0 + 1e-6).all()


@pytest.mark.parametrize("ds,model", [("SYSCALL", "MLP"), ("SYSCALL", "Autoencoder")])
def test_per_step_series_of_module_families(ds, model, monkeypatch):
    """log_steps: loss of every optimizer step and (classifiers) the confusion matrix of every 20th step's batch, the
    same from the vmap path (epoch-local tables written inside the replayed pass) and from the per-node loop.
    Reference: Train/* logged per step, every 20 steps (lightninglearner.py:188, syscall/models/mlp.py)."""
    monkeypatch.setenv("P2PFL_MODULE_VMAP_STRICT", "1")
    monkeypatch.setattr(ModuleEngine, "LOG_EVERY", 4)
    loop, vm = _pair(ds, model, n=24, B=4, log_steps=True)                 # 6 steps per epoch
    loop.fit(2); vm.fit(2)
    assert vm.batched
    l0, l1 = loop.step_losses(), vm.step_losses()
    assert l0.shape == l1.shape == (12, 3)
    assert torch.allclose(l0, l1, atol=1e-4) and (l0 > 0).all()
    c0, c1 = loop.step_confusions(), vm.step_confusions()
    if loop.classifier:
        assert sorted(c0) == sorted(c1) == [4, 8, 12]
        for t in c0:
            assert torch.equal(c0[t], c1[t])
            assert c0[t].sum(dim=(1, 2)).tolist() == [4, 4, 4]            # one batch of four samples per client
        m = vm.step_metrics()[8][0]
        assert set(m) == {"Accuracy", "Precision", "Recall", "F1Score"}
    else:
        assert c0 == c1 == {} and vm.step_metrics() == {}
    # a second fit re-uses the tables (and, on a GPU, the captured passes that write them)
    vm.fit(2)
    assert vm.step_losses().shape == (12, 3)
