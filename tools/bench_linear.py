"""Micro-benchmark of the big dense layer kernels (MNIST-CNN l1 shape, 100 clients): forward,
data-gradient and the fused wgrad+Adam. Prints TB/s of weight traffic.

usage (GPU box): python tools/bench_linear.py [K]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from p2pfl_amd.ops import hip  # noqa: E402


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    B, In, Out = 32, 3136, 2048
    dev = "cuda"
    torch.manual_seed(0)
    x = torch.randn(K, B, In, device=dev)
    w = torch.randn(K, Out, In, device=dev) * 0.02
    b = torch.randn(K, Out, device=dev)
    y = torch.empty(K, B, Out, device=dev)
    dy = torch.randn(K, B, Out, device=dev)
    dx = torch.empty(K, B, In, device=dev)
    wbytes = w.numel() * 4
    t = timeit(lambda: hip.linear_fwd(x, w, b, y, True))
    print(f"fwd   {t*1e6:8.1f} us  {wbytes/t/1e12:5.2f} TB/s")
    t = timeit(lambda: hip.linear_dgrad(dy, w, None, dx))
    print(f"dgrad {t*1e6:8.1f} us  {wbytes/t/1e12:5.2f} TB/s")
    ref = torch.relu(torch.einsum("kbi,koi->kbo", x[:2], w[:2]) + b[:2, None])
    print("fwd vs fp32 ref rel err", ((y[:2] - ref).norm() / ref.norm()).item())
    refd = torch.einsum("kbo,koi->kbi", dy[:2], w[:2])
    print("dgrad vs fp32 ref rel err", ((dx[:2] - refd).norm() / refd.norm()).item())
    m = torch.zeros_like(w)
    v = torch.zeros_like(w)
    mb, vb = torch.zeros_like(b), torch.zeros_like(b)
    t = timeit(lambda: hip.linear_wgrad_adam(dy, x, w, m, v, b, mb, vb, lr=1e-3, t=1), n=5)
    print(f"wgrad+adam    {t*1e6:8.1f} us  {6*wbytes/t/1e12:5.2f} TB/s (6 words per weight)")


if __name__ == "__main__":
    main()
