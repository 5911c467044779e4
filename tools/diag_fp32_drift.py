"""Where do the fp32 HIP engine and the eager engine part ways? Per layer, after 1 and 8 Adam steps."""
import sys
import torch
sys.path.insert(0, ".")
from tests.test_engine_gpu import _pair

spec, h, e = _pair("mnist-cnn", precision="fp32-mfma")
p0 = h.params.clone()
# one un-fused step: gradients
K, B = h.K, h.B
sidx = torch.stack([torch.randperm(64, device="cuda")[:B] for _ in range(K)]).int()
h._forward(h.train_x, sidx, h.train_y, h._nv(B), "Train", train=True)
h._backward(h.train_x, sidx)
torch.cuda.synchronize()
for k in range(1):
    m = e.modules[k]
    m.zero_grad()
    xb = e.train_x[k][sidx[k].long()]
    yb = e.train_y[k][sidx[k].long()].long()
    m.criterion(m(xb), yb).backward()
    for pname, p in m.named_parameters():
        g = h.grad_view(pname)[k]
        ref = p.grad
        d = (g - ref).abs()
        sign_flip = ((g * ref) < 0).float().mean().item()
        tiny = (ref.abs() < 1e-7).float().mean().item()
        zero = (ref == 0).float().mean().item()
        print(f"{pname:14s} |g| med {ref.abs().median():.2e} max {ref.abs().max():.2e} maxdiff {d.max():.2e} "
              f"relnorm {(g-ref).norm()/ref.norm():.2e} signflip {sign_flip:.2e} tiny<1e-7 {tiny:.3f} exact0 {zero:.3f}")
h.params.copy_(p0); e.params.copy_(p0)
for steps in (1, 4):
    h.fit(steps); e.fit(steps)   # epochs of 2 steps
    torch.cuda.synchronize()
    print(f"after fit({steps}):")
    for s in h.layout.specs if hasattr(h.layout, 'specs') else h.layout.by_name.values():
        a = h.layout.view(h.params, s.name); b = h.layout.view(e.params, s.name); z = h.layout.view(p0, s.name)
        dd = (a - b)
        print(f"  {s.name:14s} rel {dd.norm()/b.norm():.2e} delta-rel {dd.norm()/(b-z).norm():.2e} frac>1e-5 {(dd.abs()>1e-5).float().mean():.2e} max {dd.abs().max():.2e}")
