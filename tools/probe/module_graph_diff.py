import sys, torch, os
sys.path.insert(0, ".")
os.environ["P2PFL_MODULE_VMAP_STRICT"] = "1"
from p2pfl_amd.node_start import build_dataset_and_model, build_model
from p2pfl_amd.runtime.module_engine import ModuleEngine, ModuleSpec
dev = "cuda"
ds, model, K, n = sys.argv[1], sys.argv[2], 4, 96
probe = build_dataset_and_model(ds, model, 0, 20)[0]
spec = ModuleSpec(f"{ds}-{model}", lambda: build_model(ds, model), probe.num_classes, tuple(probe.input_shape))
t = probe.tensors()
reps = -(-n // t[0].shape[0])
data = [torch.stack([torch.cat([t[j]] * reps)[:n if j < 2 else 32] for _ in range(K)]) for j in range(6)]
engs = {}
for graph in ("0", "1"):
    os.environ["P2PFL_MODULE_GRAPH"] = graph
    eng = ModuleEngine(spec, K, dev, seed=1, batched=True, batch_size=32)
    for m in eng.modules:
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout): mod.p = 0.0
    eng.set_data(*data); eng.init_params(identical=False)
    engs[graph] = eng
for r in range(4):
    outs = {}
    for graph, eng in engs.items():
        ev = eng.evaluate(); out = eng.fit(2)
        outs[graph] = (float(ev["loss"].mean()), float(out["Train"]["loss"].mean()))
    d = (engs["0"].params - engs["1"].params).abs()
    worst = max(engs["0"].layout.slots, key=lambda s: float(d[:, s.offset:s.offset + s.numel].max()))
    print(f"round {r}: max param diff {float(d.max()):.3e} (in {worst.name}) eval {outs['0'][0]:.6f}/{outs['1'][0]:.6f} train {outs['0'][1]:.6f}/{outs['1'][1]:.6f} graphs {len(engs['1']._graphs)}", flush=True)
