"""Throughput of the device AES-128-ECB kernel (csrc/aes.hip) on parameter rows: the per-link cipher of an encrypted
exchange (reference: fedstellar/encrypter.py AESCipher on every payload). Prints GB/s for encrypt and decrypt.

usage: python tools/aes_probe.py
"""
import sys
import torch
sys.path.insert(0, ".")
from p2pfl_amd.ops import hip
from p2pfl_amd.utils.aes import expand_key

dev = "cuda"
rk = torch.frombuffer(bytearray(expand_key(bytes(range(16)))), dtype=torch.uint8).to(dev)
for mb in (26, 256, 2600):
    n = mb * (1 << 20) // 16 * 16
    src = torch.randint(0, 256, (n,), dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    back = torch.empty_like(src)
    for name, a, b, dec in (("encrypt", src, dst, False), ("decrypt", dst, back, True)):
        for _ in range(2):
            hip.aes128_ecb(a, b, rk, dec)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        reps = 10
        e0.record()
        for _ in range(reps):
            hip.aes128_ecb(a, b, rk, dec)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"aes128_ecb {name} {mb:5d} MiB: {ms:8.3f} ms  {n / ms / 1e6:8.1f} GB/s (payload; read + write = 2x)", flush=True)
    assert torch.equal(src, back)
