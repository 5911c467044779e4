import sys, torch
sys.path.insert(0, ".")
from p2pfl_amd.ops import hip
h = hip.bound("fp32")
K, B, dev = 100, 32, "cuda"
a1 = torch.randn(K, B, 32, 196, device=dev).relu()
code = torch.randint(0, 5, (K, B, 64, 49), device=dev).to(torch.uint8)
g2 = torch.randn(K, B, 64, 49, device=dev)
dw2, db2 = torch.empty(K, 64, 800, device=dev), torch.empty(K, 64, device=dev)
scratch = h.conv2_wgrad_scratch(K, B, 5, dev)
f = lambda: h.conv2_wgrad(g2, code, a1, dw2, db2, 5, scratch=scratch)
for _ in range(3): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
print("ns", __import__("os").environ.get("P2PFL_WGRAD_NS"), "wgrad+reduce ms", e0.elapsed_time(e1) / 20)
