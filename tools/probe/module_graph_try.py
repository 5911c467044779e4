import sys, time, torch, os
sys.path.insert(0, "/root/repo")
os.environ["P2PFL_MODULE_VMAP_STRICT"] = "1"
from p2pfl_amd.node_start import build_dataset_and_model, build_model
from p2pfl_amd.runtime.module_engine import ModuleEngine, ModuleSpec
dev = "cuda"
for ds, model, K, n in (("SYSCALL", "MLP", 50, 320), ("SYSCALL", "SVM", 50, 320), ("SYSCALL", "Autoencoder", 50, 320), ("WADI", "MLP", 50, 320), ("CIFAR10", "ResNet9", 10, 128)):
    probe = build_dataset_and_model(ds, model, 0, 20)[0]
    spec = ModuleSpec(f"{ds}-{model}", lambda: build_model(ds, model), probe.num_classes, tuple(probe.input_shape))
    t = probe.tensors()
    reps = -(-n // t[0].shape[0])
    data = [torch.stack([torch.cat([t[j]] * reps)[:n if j < 2 else 32] for _ in range(K)]) for j in range(6)]
    res = {}
    for graph in ("0", "1"):
        os.environ["P2PFL_MODULE_GRAPH"] = graph
        eng = ModuleEngine(spec, K, dev, seed=1, batched=True, batch_size=32)
        for m in eng.modules:
            for mod in m.modules():
                if isinstance(mod, torch.nn.Dropout): mod.p = 0.0
        eng.set_data(*data); eng.init_params(identical=False)
        for r in range(3):
            eng.evaluate(); out = eng.fit(2)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ev = eng.evaluate(); out = eng.fit(2)
        l = float(out["Train"]["loss"].mean()); torch.cuda.synchronize()
        res[graph] = (time.perf_counter() - t0, eng.params.clone(), l, float(ev["loss"].mean()), eng.use_graph, len(eng._graphs))
    a, b = res["0"], res["1"]
    print(f"{ds}-{model}: eager {a[0]*1e3:.1f} ms  graph {b[0]*1e3:.1f} ms (graphs {b[5]}, on={b[4]})  param diff {float((a[1]-b[1]).abs().max()):.2e} "
          f"train loss {a[2]:.5f} / {b[2]:.5f}  eval {a[3]:.5f} / {b[3]:.5f}", flush=True)
