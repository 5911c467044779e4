"""Which layer makes two client groups differ from one? One optimizer step (and then a few) for 84 MNIST-CNN clients under
P2PFL_GROUPS=1 and =2; per layout slot: how many clients differ and by how much.   (GPU box)"""
import gc
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from p2pfl_amd.datasets.synthetic import synthetic_shards  # noqa: E402
from p2pfl_amd.models.specs import get_spec  # noqa: E402
from p2pfl_amd.runtime.engine import HipEngine  # noqa: E402

spec = get_spec("mnist-cnn")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 84
for n in (32, 96):
    data = synthetic_shards(K, n, 16, 40, spec.num_classes, seed=5, device="cuda")
    res = {}
    for groups in ("1", "2"):
        os.environ["P2PFL_GROUPS"] = groups
        h = HipEngine(spec, K, "cuda", seed=7, use_graph=False)
        h.init_params(identical=False)
        p0 = h.params.clone()
        h.set_data(*data)
        h.fit(1)
        torch.cuda.synchronize()
        res[groups] = (h.params - p0).clone()
        slots = h.layout.slots
        h.close()
        del h
        gc.collect()
    a, b = res["1"], res["2"]
    print(f"--- {n // 32} step(s)")
    for s in slots:
        x, y = a[:, s.offset:s.offset + s.numel], b[:, s.offset:s.offset + s.numel]
        off = ((x != y).any(dim=1)).nonzero().flatten().tolist()
        print(f"{s.name:12s} clients that differ: {len(off):3d} {off[:12]}  max |diff| {(x - y).abs().max().item():.3g} "
              f"(update scale {x.abs().max().item():.3g})", flush=True)
