"""state of the l2 moments right before / after the graph replay (round-3 diagnostic)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from p2pfl_amd.datasets.synthetic import synthetic_shards
from p2pfl_amd.models.specs import get_spec
from p2pfl_amd.runtime.engine import HipEngine

spec = get_spec("mnist-cnn")
K, N = 84, 32
h = HipEngine(spec, 3, "cuda", seed=7, use_graph=False)
h.init_params(identical=False)
h.set_data(*synthetic_shards(3, 80, 16, 40, spec.num_classes, seed=5, device="cuda"))
h.fit(1)
torch.cuda.synchronize()
del h
data = synthetic_shards(K, N, 16, 40, spec.num_classes, seed=5, device="cuda")
h = HipEngine(spec, K, "cuda", seed=7, use_graph=True)
h.init_params(identical=False)
h.set_data(*data)
s = h.layout.by_name["l2.weight"]
sl = slice(s.offset, s.offset + s.numel)
orig = torch.cuda.CUDAGraph.replay
def replay(self):
    torch.cuda.synchronize()
    print("before replay: |v| per client (nonzero clients):", (h.v[:, sl].abs().sum(dim=1) != 0).nonzero().flatten().tolist()[:20],
          " |m|:", (h.m[:, sl].abs().sum(dim=1) != 0).nonzero().flatten().tolist()[:20])
    orig(self)
    torch.cuda.synchronize()
torch.cuda.CUDAGraph.replay = replay
h.fit(1)
torch.cuda.synchronize()
v1 = h.v[:, sl].clone(); m1 = h.m[:, sl].clone()
g = m1 / (1 - 0.9)
vexp = (1 - 0.999) * g * g
print("v == (1-b2) g^2 per client:", [(k, float(((v1[k] - vexp[k]).abs().max() / vexp[k].abs().max()))) for k in (0, 10, 15, 20, 40, 83)])
print("v / vexp ratio sample client 40:", (v1[40] / vexp[40].clamp(min=1e-30)).flatten()[:8].tolist())
