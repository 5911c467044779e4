"""Model families without hand-written kernels (ModuleEngine): per-node Python loop vs the client-batched torch.func.vmap
step, one local-training round (evaluate + 1 epoch + validation) of K co-resident nodes on the GPU.

usage (GPU box): python tools/bench_module.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from p2pfl_amd.node_start import build_dataset_and_model, build_model  # noqa: E402
from p2pfl_amd.runtime.module_engine import ModuleEngine, ModuleSpec  # noqa: E402

CASES = [("SYSCALL", "MLP", 50, 320), ("SYSCALL", "SVM", 50, 320), ("SYSCALL", "Autoencoder", 50, 320), ("WADI", "MLP", 50, 320), ("CIFAR10", "simplemobilenet", 10, 128), ("CIFAR10", "ResNet9", 10, 128),
         ("CIFAR10", "ResNet18", 10, 64)]
dev = "cuda"
os.environ["P2PFL_MODULE_VMAP_STRICT"] = "1"
for ds, model, K, n in CASES:
    probe = build_dataset_and_model(ds, model, 0, 20)[0]
    spec = ModuleSpec(f"{ds}-{model}", lambda: build_model(ds, model), probe.num_classes, tuple(probe.input_shape))
    t = probe.tensors()
    reps = -(-n // t[0].shape[0])
    data = [torch.stack([torch.cat([t[j]] * reps)[:n if j < 2 else 32] for _ in range(K)]) for j in range(6)]
    times = {}
    for mode in ("loop", "vmap", "vmap+graph"):
        os.environ["P2PFL_MODULE_GRAPH"] = "1" if mode == "vmap+graph" else "0"
        try:
            eng = ModuleEngine(spec, K, dev, seed=1, batched=mode != "loop", batch_size=32)
            eng.set_data(*data)
            eng.init_params()
            for _ in range(3):                               # warm-up: MIOpen kernel search, allocator, graph capture
                eng.evaluate(); eng.fit(1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.evaluate(); out = eng.fit(1)
            float(out["Train"]["loss"].mean())
            torch.cuda.synchronize()
            times[mode] = time.perf_counter() - t0
            assert eng.batched == (mode != "loop") and (mode != "vmap+graph" or eng.use_graph)
        except Exception as exc:  # noqa: BLE001
            times[mode] = float("nan")
            print(f"  {ds}-{model} {mode} failed: {type(exc).__name__}: {str(exc)[:200]}")
        del eng
        torch.cuda.empty_cache()
    hip_ms = ""
    kspec = getattr(build_model(ds, model), "spec", None)
    if kspec is not None:                     # a family that ALSO has a kernel-level spec: the same round on the HIP engine
        from p2pfl_amd.runtime.engine import HipEngine
        eng = HipEngine(kspec, K, dev, seed=1, batch_size=32)
        eng.set_data(*data)
        eng.init_params()
        for _ in range(3):
            eng.evaluate(); eng.fit(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.evaluate(); out = eng.fit(1)
        float(out["Train"]["loss"].mean())
        torch.cuda.synchronize()
        hip_ms = f"   HIP kernels {(time.perf_counter() - t0)*1e3:8.1f} ms"
        del eng
        torch.cuda.empty_cache()
    print(f"{ds}-{model:16s} K={K:3d} n={n:4d}: per-node loop {times['loop']*1e3:8.1f} ms   client-batched {times['vmap']*1e3:8.1f} ms "
          f"(x{times['loop'] / times['vmap']:.1f})   + hipGraph replay {times['vmap+graph']*1e3:8.1f} ms (x{times['loop'] / times['vmap+graph']:.1f}){hip_ms}",
          flush=True)
