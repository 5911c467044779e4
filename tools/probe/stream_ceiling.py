"""HBM ceilings for the access patterns of the update kernels (100 clients x 6.4 M weights)."""
import sys
import torch
sys.path.insert(0, ".")
from p2pfl_amd.ops import hip
K, P, dev = 100, 3136 * 2048 + 2048, "cuda"
p, g, m, v = (torch.randn(K, P, device=dev) * 0.02 for _ in range(4))
v.abs_()
GB = K * P * 4 / 1e9


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, fn, words in (
        ("torch copy_ (1r 1w)", lambda: g.copy_(p), 2),
        ("torch mul_ in place (1r 1w same address)", lambda: g.mul_(1.0001), 2),
        ("torch add out (2r 1w)", lambda: torch.add(p, m, out=g), 3),
        ("adam_step (4r 3w)", lambda: hip.adam_step(p, g, m, v, lr=1e-3, t=2), 7),
        ("sgdm_step (3r 2w)", lambda: hip.sgdm_step(p, g, m, lr=1e-3), 5),
        ("torch sum (1r)", lambda: p.sum(), 1),
        ("torch zero_ (1w)", lambda: g.zero_(), 1)):
    t = timed(fn)
    print(f"{name:45s} {t:.3f} ms  {words * GB / t:.2f} TB/s", flush=True)
