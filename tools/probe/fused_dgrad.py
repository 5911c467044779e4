"""Times linear_dgrad + linear_wgrad_adam against the fused linear_wgrad_adam_dgrad (100 clients, 3136 x 2048)."""
import sys
import torch
sys.path.insert(0, ".")
from p2pfl_amd.ops import hip
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
h = hip.bound(prec)
K, B, dev = 100, 32, "cuda"
P = 3136 * 2048 + 2048
slab = [torch.randn(K, P, device=dev) * 0.02 for _ in range(3)]
slab[2].abs_()
view = lambda s: (s[:, :3136 * 2048].unflatten(1, (2048, 3136)), s[:, 3136 * 2048:])
(w, b), (mw, mb), (vw, vb) = view(slab[0]), view(slab[1]), view(slab[2])
dy = torch.randn(K, B, 2048, device=dev) * 0.01
x = torch.randn(K, B, 3136, device=dev).relu()
dx = torch.empty(K, B, 3136, device=dev)


def timed(fn, reps=20):
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


def two():
    h.linear_dgrad(dy, w, None, dx)
    h.linear_wgrad_adam(dy, x, w, mw, vw, b, mb, vb, lr=1e-3, t=2)


print(f"[{prec}] update alone       {timed(lambda: h.linear_wgrad_adam(dy, x, w, mw, vw, b, mb, vb, lr=1e-3, t=2)):.3f} ms")
print(f"[{prec}] dgrad + update     {timed(two):.3f} ms")
print(f"[{prec}] fused update+dgrad {timed(lambda: h.linear_wgrad_adam_dgrad(dy, x, w, mw, vw, b, mb, vb, dx, lr=1e-3, t=2)):.3f} ms")
