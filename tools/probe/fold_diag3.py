"""graph vs eager after a prelude engine: first diverging step / parameter (round-3 diagnostic)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from p2pfl_amd.datasets.synthetic import synthetic_shards
from p2pfl_amd.models.specs import get_spec
from p2pfl_amd.runtime.engine import HipEngine

spec = get_spec("mnist-cnn")
K = int(os.environ.get("K", "84"))
N = int(os.environ.get("N", "80"))
h = HipEngine(spec, 3, "cuda", seed=7, use_graph=False)
h.init_params(identical=False)
h.set_data(*synthetic_shards(3, 80, 16, 40, spec.num_classes, seed=5, device="cuda"))
h.fit(1)
torch.cuda.synchronize()
del h
data = synthetic_shards(K, N, 16, 40, spec.num_classes, seed=5, device="cuda")
res = []
for graph in (True, False):
    h = HipEngine(spec, K, "cuda", seed=7, use_graph=graph, log_steps=True)
    h.init_params(identical=False)
    h.set_data(*data)
    p0 = h.params.clone()
    h.fit(1)
    torch.cuda.synchronize()
    res.append((h.params.clone() - p0, h.step_losses().clone(), h.m.clone(), h.v.clone(), h))
(dg, lg, mg, vg, h), (de, le, me, ve, _) = res
print("step losses differ at (step, #clients):", [(t, int((lg[t] != le[t]).sum())) for t in range(lg.shape[0])])
for s in h.layout.slots:
    sl = slice(s.offset, s.offset + s.numel)
    for nm, a, b in (("p", dg, de), ("m", mg, me), ("v", vg, ve)):
        bad = ((a[:, sl] != b[:, sl]).any(dim=1)).nonzero().flatten().tolist()
        print(f"  {s.name:14s} {nm}: clients off {bad[:10]}{'...' if len(bad) > 10 else ''} ({len(bad)})")
