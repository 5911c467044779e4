"""Does the 256 MB Infinity Cache (MALL) carry the fused wgrad+Adam working set (77 MB per client: w, m, v of the
3136x2048 layer) from one launch to the next? Times the update kernel per client for growing client counts, and
the dense forward / data-gradient kernels that read the same weights.

usage: python tools/mall_probe.py [fp32|bf16]
"""
import sys
import torch
sys.path.insert(0, ".")
from p2pfl_amd.ops import hip

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
h = hip.bound(prec)
dev = "cuda"
B = 32
P = 3136 * 2048 + 2048


def run(K, reps=30):
    slab = [torch.randn(K, P, device=dev) * 0.02 for _ in range(3)]
    view = lambda s: (s[:, :3136 * 2048].unflatten(1, (2048, 3136)), s[:, 3136 * 2048:])
    (w, b), (mw, mb), (vw, vb) = view(slab[0]), view(slab[1]), view(slab[2])
    slab[2].abs_()
    dy = torch.randn(K, B, 2048, device=dev) * 0.01
    x = torch.randn(K, B, 3136, device=dev).relu()
    y = torch.empty(K, B, 2048, device=dev)
    dx = torch.empty(K, B, 3136, device=dev)
    out = {}
    for name, fn in (("update", lambda: h.linear_wgrad_adam(dy, x, w, mw, vw, b, mb, vb, lr=1e-3, t=2)),
                     ("fwd", lambda: h.linear_fwd(x, w, b, y, True)),
                     ("dgrad", lambda: h.linear_dgrad(dy, w, x, dx)),
                     ("step", None)):
        if fn is None:
            def fn():
                h.linear_fwd(x, w, b, y, True)
                h.linear_dgrad(dy, w, x, dx)
                h.linear_wgrad_adam(dy, x, w, mw, vw, b, mb, vb, lr=1e-3, t=2)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[name] = e0.elapsed_time(e1) / reps
    return out


print(f"[{prec}] per-launch ms (ms per client) ; working set w+m+v = 77 MB per client")
for K in (1, 2, 3, 4, 6, 8, 12, 16, 32, 100):
    r = run(K)
    print(f"  K={K:3d} ({77 * K:5d} MB): " + "  ".join(f"{n} {v:7.3f} ({v / K * 1e3:6.1f} us/client)" for n, v in r.items()),
          flush=True)
