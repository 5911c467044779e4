"""Micro-benchmark of the convolution stages (client-batched): time, effective TFLOP/s and GB/s.

usage (GPU box): python tools/bench_conv.py [K] [ks1] [ks2]      (defaults: 1000 clients, FEMNIST 7x7 / 3x3)
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
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    ks1 = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    ks2 = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    B, n, dev = 32, 54, "cuda"
    data = torch.rand(K, n, 784, device=dev)
    sidx = torch.stack([torch.randperm(n, device=dev)[:B] for _ in range(K)]).int()
    w1 = torch.randn(K, 32, ks1 * ks1, device=dev) * 0.1
    b1 = torch.zeros(K, 32, device=dev)
    a1 = torch.empty(K, B, 32, 196, device=dev)
    c1 = torch.empty(K, B, 32, 196, dtype=torch.uint8, device=dev)
    imgs = K * B
    t = timeit(lambda: hip.conv1_fwd(data, sidx, w1, b1, a1, c1, ks1))
    fl = imgs * 784 * 32 * ks1 * ks1 * 2
    by = imgs * (784 * 4 + 32 * 196 * 5)
    print(f"conv1_fwd   {t*1e6:8.1f} us  {fl/t/1e12:6.1f} TFLOP/s  {by/t/1e9:7.0f} GB/s")
    g1 = torch.randn_like(a1)
    dw1, db1 = torch.zeros_like(w1), torch.zeros_like(b1)
    t = timeit(lambda: hip.conv1_wgrad(g1, c1, data, sidx, dw1, db1, ks1))
    by = imgs * (784 * 4 + 32 * 196 * 5)
    print(f"conv1_wgrad {t*1e6:8.1f} us  {fl/t/1e12:6.1f} TFLOP/s  {by/t/1e9:7.0f} GB/s")
    w2 = torch.randn(K, 64, 32 * ks2 * ks2, device=dev) * 0.05
    b2 = torch.zeros(K, 64, device=dev)
    a2 = torch.empty(K, B, 64, 49, device=dev)
    c2 = torch.empty(K, B, 64, 49, dtype=torch.uint8, device=dev)
    wp = hip.conv2_prep_alloc(K, ks2, dev)
    t = timeit(lambda: hip.conv2_prep(w2, wp, ks2))
    print(f"conv2_prep  {t*1e6:8.1f} us")
    fl2 = imgs * 196 * 64 * 32 * ks2 * ks2 * 2
    t = timeit(lambda: hip.conv2_fwd(a1, w2, b2, a2, c2, ks2, wprep=wp))
    by = imgs * (32 * 196 * 4 + 64 * 49 * 5)
    print(f"conv2_fwd   {t*1e6:8.1f} us  {fl2/t/1e12:6.1f} TFLOP/s  {by/t/1e9:7.0f} GB/s")
    g2 = torch.randn_like(a2)
    dw2, db2 = torch.zeros_like(w2), torch.zeros_like(b2)
    t = timeit(lambda: hip.conv2_wgrad(g2, c2, a1, dw2, db2, ks2))
    print(f"conv2_wgrad {t*1e6:8.1f} us  {fl2/t/1e12:6.1f} TFLOP/s  {by/t/1e9:7.0f} GB/s")
    da1 = torch.empty_like(a1)
    t = timeit(lambda: hip.conv2_dgrad(g2, c2, w2, da1, ks2, wprep=wp))
    print(f"conv2_dgrad {t*1e6:8.1f} us  {fl2/t/1e12:6.1f} TFLOP/s  {by/t/1e9:7.0f} GB/s")


if __name__ == "__main__":
    main()
