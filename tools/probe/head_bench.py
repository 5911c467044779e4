"""The class head alone (csrc/head.hip): one-kernel form against the two-kernel form over column slices.
usage (GPU box): python tools/probe/head_bench.py [K ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from p2pfl_amd.ops import hip  # noqa: E402


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    B, In, C, n, dev = 32, 2048, 10, 540, "cuda"
    for K in [int(a) for a in sys.argv[1:]] or [10, 50, 100]:
        P = C * In
        off_b = -(-P // 64) * 64
        p, m, v = (torch.randn(K, off_b + 64, device=dev) * 0.1 for _ in range(3))
        v.abs_()
        W = lambda s: s[:, :P].unflatten(1, (C, In))
        Bv = lambda s: s[:, off_b:off_b + C]
        x = torch.randn(K, B, In, device=dev).relu()
        labels = torch.randint(0, C, (K, n), device=dev).int()
        sidx = torch.stack([torch.randperm(n, device=dev)[:B] for _ in range(K)]).int()
        stats = (torch.zeros(K, device=dev), torch.zeros(K, dtype=torch.int32, device=dev),
                 torch.zeros(K, dtype=torch.int32, device=dev))
        conf = torch.zeros(K, C, C, dtype=torch.int32, device=dev)
        dx = torch.zeros(K, B, In, device=dev)
        opt = hip.fold_opt("adam", W(p), W(m), W(v), Bv(p), Bv(m), Bv(v), lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8,
                           weight_decay=0.0, momentum=0.0, t=2)
        scratch = hip.linear_head_scratch(K, B, In, C, dev)
        for name, sc in (("one kernel per client", None), ("column slices", scratch)):
            t = timeit(lambda: hip.linear_head(x, W(p), Bv(p), labels, sidx, *stats, conf, None, dx=dx, relu_mask=True,
                                               opt=opt, scratch=sc))
            print(f"K={K:4d}  {name:22s} {t:7.1f} us")


if __name__ == "__main__":
    main()
