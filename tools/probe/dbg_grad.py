"""First-step gradients of the HIP engine against autograd, every parameter, and the pool decisions of both
(debug aid of tests/test_engine_gpu.py::test_first_step_gradients_match_autograd)."""
import sys, os, torch
import torch.nn.functional as F
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from test_engine_gpu import _pair
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32-mfma"
spec, h, e = _pair("mnist-cnn", fuse=False, precision=prec)
K, B = h.K, h.B
if len(sys.argv) > 2:
    torch.manual_seed(int(sys.argv[2]))
sidx = torch.stack([torch.randperm(64, device="cuda")[:B] for _ in range(K)]).int()
h.stats["Train"].zero_()
h._forward(h.train_x, sidx, h.train_y, h._nv(B), "Train", train=True)
h._backward(h.train_x, sidx)
torch.cuda.synchronize()
for k in range(K):
    m = e.modules[k]; m.zero_grad()
    xb = e.train_x[k][sidx[k].long()]; yb = e.train_y[k][sidx[k].long()].long()
    m.criterion(m(xb), yb).backward()
    errs = {pn: ((h.grad_view(pn)[k] - p.grad).norm() / p.grad.norm()).item() for pn, p in m.named_parameters()}
    print(prec, k, " ".join(f"{pn}={v:.1e}" for pn, v in errs.items()))
    # pool decisions of both stages against float64 pre-pool values
    P = dict(m.named_parameters())
    x = xb.view(B, 1, 28, 28).double()
    pre1 = F.conv2d(x, P["conv1.weight"].double(), P["conv1.bias"].double(), padding=2)
    a1 = F.max_pool2d(pre1.relu(), 2, 2)
    pre2 = F.conv2d(a1, P["conv2.weight"].double(), P["conv2.bias"].double(), padding=2)
    for name, pre, code in (("stage 1", pre1, h.code[0][k]), ("stage 2", pre2, h.code[1][k])):
        C, S = pre.shape[1], pre.shape[2] // 2
        win = pre.view(B, C, S, 2, S, 2).permute(0, 1, 2, 4, 3, 5).reshape(B, C, S * S, 4)
        best = win.argmax(-1)
        dead = win.max(-1).values <= 0
        ref = torch.where(dead, torch.full_like(best, 4), best)
        diff = (ref != code.view(B, C, S * S).long())
        top2 = win.topk(2, -1).values
        gap = (top2[..., 0] - top2[..., 1]).abs()
        print("   ", name, "decisions that differ from float64:", int(diff.sum()), "of", diff.numel(),
              "| their top-2 gaps / |max|:", [f"{(gap[tuple(d.tolist())] / win[tuple(d.tolist())].abs().max()).item():.1e}" for d in diff.nonzero()[:4].cpu()])
