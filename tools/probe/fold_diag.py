"""Which parameter ranges differ between two runs of the same schedule (graph vs eager)? (round-3 diagnostic)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from p2pfl_amd.datasets.synthetic import synthetic_shards
from p2pfl_amd.models.specs import get_spec
from p2pfl_amd.runtime.engine import HipEngine

spec = get_spec("mnist-cnn")
K = int(os.environ.get("K", "84"))
data = synthetic_shards(K, 80, 16, 40, spec.num_classes, seed=5, device="cuda")
if os.environ.get("GARBAGE"):      # poison the allocator's free blocks: anything read before it is written shows up
    junk = [torch.full((1 << 26,), float(os.environ["GARBAGE"]), device="cuda") for _ in range(24)]
    del junk
if os.environ.get("PRELUDE"):      # an earlier, smaller CNN engine in the same process (as in the test suite)
    h = HipEngine(spec, 3, "cuda", seed=7, use_graph=False)
    h.init_params(identical=False)
    h.set_data(*synthetic_shards(3, 80, 16, 40, spec.num_classes, seed=5, device="cuda"))
    h.fit(1)
    torch.cuda.synchronize()
    del h
res = []
for graph, overlap in ((True, True), (False, True), (False, False)):
    h = HipEngine(spec, K, "cuda", seed=7, use_graph=graph, overlap=overlap, log_steps=bool(os.environ.get("LOGSTEPS")))
    h.init_params(identical=False)
    h.set_data(*data)
    p0 = h.params.clone()
    for _ in range(int(os.environ.get("FITS", "1"))):
        h.fit(int(os.environ.get("EPOCHS", "1")))
    torch.cuda.synchronize()
    res.append((h.params.clone() - p0, h))
ref = res[2][0]
for name, (d, h) in zip(("graph+overlap", "eager+overlap"), res[:2]):
    print(name)
    for s in h.layout.slots:
        a, b = d[:, s.offset:s.offset + s.numel], ref[:, s.offset:s.offset + s.numel]
        pc = (a - b).norm(dim=1) / b.norm(dim=1).clamp(min=1e-20)
        bad = (pc > 1e-4).nonzero().flatten().tolist()
        print(f"  {s.name:14s} max rel {pc.max().item():.3g}  clients off: {bad[:12]}{'...' if len(bad) > 12 else ''} ({len(bad)})")
