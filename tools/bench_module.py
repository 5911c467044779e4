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

CASES = [("SYSCALL", "MLP", 50, 320), ("WADI", "MLP", 50, 320), ("CIFAR10", "simplemobilenet", 10, 128), ("CIFAR10", "ResNet9", 10, 128),
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
    for batched in (False, True):
        try:
            eng = ModuleEngine(spec, K, dev, seed=1, batched=batched, batch_size=32)
            eng.set_data(*data)
            eng.init_params()
            eng.evaluate(); eng.fit(1)                       # warm-up (MIOpen kernel search, allocator)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.evaluate(); eng.fit(1)
            torch.cuda.synchronize()
            times[batched] = time.perf_counter() - t0
            assert eng.batched == batched
        except Exception as exc:  # noqa: BLE001
            times[batched] = float("nan")
            print(f"  {ds}-{model} batched={batched} failed: {type(exc).__name__}: {str(exc)[:200]}")
        del eng
        torch.cuda.empty_cache()
    print(f"{ds}-{model:16s} K={K:3d} n={n:4d}: per-node loop {times[False]*1e3:9.1f} ms   client-batched {times[True]*1e3:9.1f} ms   "
          f"x{times[False] / times[True]:.1f}", flush=True)
