"""Client groups vs one group: which parameters of which clients differ? (round-3 diagnostic)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from p2pfl_amd.datasets.synthetic import synthetic_shards
from p2pfl_amd.models.specs import get_spec
from p2pfl_amd.runtime.engine import HipEngine

spec = get_spec("mnist-cnn")
K = int(os.environ.get("K", "84"))
data = synthetic_shards(K, 80, 16, 40, spec.num_classes, seed=5, device="cuda")
res = []
for groups in ("1", "2", "3"):
    os.environ["P2PFL_GROUPS"] = groups
    h = HipEngine(spec, K, "cuda", seed=7, use_graph=False)
    h.init_params(identical=False)
    h.set_data(*data)
    p0 = h.params.clone()
    for _ in range(int(os.environ.get("FITS", "1"))):
        h.fit(int(os.environ.get("EPOCHS", "1")))
    torch.cuda.synchronize()
    res.append((h.params.clone() - p0, h))
ref = res[0][0]
for name, (d, h) in zip(("2 groups", "3 groups"), res[1:]):
    print(name)
    for s in h.layout.slots:
        a, b = d[:, s.offset:s.offset + s.numel], ref[:, s.offset:s.offset + s.numel]
        pc = (a - b).norm(dim=1) / b.norm(dim=1).clamp(min=1e-20)
        bad = (pc > 1e-4).nonzero().flatten().tolist()
        print(f"  {s.name:14s} max rel {pc.max().item():.3g}  clients off: {bad[:12]}{'...' if len(bad) > 12 else ''} ({len(bad)})")
