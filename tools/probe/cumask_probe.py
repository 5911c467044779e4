"""CU-masked streams (hipExtStreamCreateWithCUMask): can the matrix-pipe-bound fp32 convolutions of a training step run
on a FIXED subset of the CUs while the HBM-bound fused update streams on the rest? (tools/overlap_probe.py showed
they cannot share SIMDs.)

1. where do the blocks of a masked stream land (XCC id / CU id per block)?
2. update kernel time against the number of CUs it may use
3. convolution chain (stage-2 wgrad + dgrad + fwd, stage-1 wgrad + fwd) against the number of CUs
4. both together: update on all CUs / on the complement, chain on the masked side stream
5. does a captured graph keep the mask of the stream a kernel was captured on?

usage: python tools/probe/cumask_probe.py [fp32|fp32-mfma]
"""
import ctypes
import os
import sys
from collections import Counter

import torch

sys.path.insert(0, ".")
from p2pfl_amd.ops import hip  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
K, B = 100, 32
dev = "cuda"
h = hip.bound(prec)
burn = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libburn.so"))
burn.masked_stream_create.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
burn.whereami_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
NCU = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(bits):
    """torch stream restricted to the CUs whose mask bits are set (iterable of bit indices)."""
    words = (NCU + 31) // 32
    arr = (ctypes.c_uint32 * words)()
    for b in bits:
        arr[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    rc = burn.masked_stream_create(ctypes.byref(s), words, arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def where(stream, blocks=2048):
    out = torch.zeros(blocks, 2, dtype=torch.int32, device=dev)
    with torch.cuda.stream(stream):
        burn.whereami_launch(blocks, 20, out.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    o = out.cpu()
    xcc = o[:, 0] & 0xF
    cu = (o[:, 1] >> 8) & 0xF
    sh = (o[:, 1] >> 12) & 1
    se = (o[:, 1] >> 13) & 7
    places = Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
    per_xcc = Counter(x for (x, _, _, _) in places)
    return len(places), dict(sorted(per_xcc.items()))


# ---------------------------------------------------------------- operands (headline shard)
P = 3136 * 2048 + 2048
slab = [torch.randn(K, P, device=dev) * 0.02 for _ in range(3)]
view = lambda s: (s[:, :3136 * 2048].unflatten(1, (2048, 3136)), s[:, 3136 * 2048:])
(w, b), (mw, mb), (vw, vb) = view(slab[0]), view(slab[1]), view(slab[2])
slab[2].abs_()
dy = torch.randn(K, B, 2048, device=dev) * 0.01
x = torch.randn(K, B, 3136, device=dev).relu()
dx = torch.empty(K, B, 3136, device=dev)
data = torch.randn(K, 540, 784, device=dev)
sidx = torch.randint(0, 540, (K, B), device=dev, dtype=torch.int32)
w1 = torch.randn(K, 32, 25, device=dev) * 0.1
b1 = torch.randn(K, 32, device=dev) * 0.1
dw1, db1 = torch.empty_like(w1), torch.empty_like(b1)
a1 = torch.randn(K, B, 32, 196, device=dev).relu()
code1 = torch.randint(0, 4, (K, B, 32, 196), device=dev).to(torch.uint8)
w2 = torch.randn(K, 64, 800, device=dev) * 0.05
b2 = torch.randn(K, 64, device=dev)
a2 = torch.empty(K, B, 64, 49, device=dev)
code = torch.randint(0, 4, (K, B, 64, 49), device=dev).to(torch.uint8)
g2 = torch.randn(K, B, 64, 49, device=dev)
da1 = torch.empty(K, B, 32, 196, device=dev)
dw2, db2 = torch.empty(K, 64, 800, device=dev), torch.empty(K, 64, device=dev)
wprep = h.conv2_prep_alloc(K, 5, dev)
h.conv2_prep(w2, wprep, 5)
scratch = h.conv2_wgrad_scratch(K, B, 5, dev)


def update_plain():
    h.linear_wgrad_adam(dy, x, w, mw, vw, b, mb, vb, lr=1e-3, t=2)


def update_dgrad():
    assert h.linear_wgrad_adam_dgrad(dy, x, w, mw, vw, b, mb, vb, dx, lr=1e-3, t=2)


def dgrad():
    h.linear_dgrad(dy, w, None, dx)


def chain(co):
    kw = dict(coresident=True) if co else {}
    h.conv2_wgrad(g2, code, a1, dw2, db2, 5, scratch=scratch, **kw)
    h.conv2_dgrad(g2, code, w2, da1, 5, wprep=wprep, **kw)
    h.conv1_wgrad(da1, code1, data, sidx, dw1, db1, 5)
    h.conv1_fwd(data, sidx, w1, b1, a1, code1, 5)
    h.conv2_prep(w2, wprep, 5)
    h.conv2_fwd(a1, w2, b2, a2, code, 5, wprep=wprep, **kw)


def timed(fn_main, fn_side, main, side, reps=5):
    """min over reps of max(main, side) elapsed, both started together"""
    best, last = 1e9, None
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1, s1 = torch.cuda.Event(True), torch.cuda.Event(True), torch.cuda.Event(True)
        with torch.cuda.stream(main):
            e0.record()
            if side is not None:
                side.wait_stream(main)
            if fn_main:
                fn_main()
            e1.record()
        if fn_side:
            with torch.cuda.stream(side):
                fn_side()
                s1.record()
        torch.cuda.synchronize()
        tm = e0.elapsed_time(e1) if fn_main else 0.0
        ts = e0.elapsed_time(s1) if fn_side else 0.0
        if max(tm, ts) < best:
            best, last = max(tm, ts), (tm, ts)
    return best, last


print(f"[{prec}] {NCU} CUs", flush=True)
full = torch.cuda.Stream()
for _ in range(2):
    update_plain(); update_dgrad(); dgrad(); chain(False); chain(True)
torch.cuda.synchronize()

# 1. placement
for n in (8, 64, 128):
    s = masked_stream(range(n))
    print(f"  mask = low {n:3d} bits: blocks seen on {where(s)} (distinct CUs, per XCC)", flush=True)
s = masked_stream(range(128, 256))
print(f"  mask = bits 128..255: {where(s)}", flush=True)
print(f"  unmasked stream: {where(full)}", flush=True)

# 2./3. time against the CU count
tu, _ = timed(update_plain, None, full, None)
tud, _ = timed(update_dgrad, None, full, None)
tdg, _ = timed(dgrad, None, full, None)
tc0, _ = timed(lambda: chain(False), None, full, None)
tc1, _ = timed(lambda: chain(True), None, full, None)
print(f"  all CUs: update {tu:.3f}  update+dgrad fused {tud:.3f}  dgrad {tdg:.3f}  chain big {tc0:.3f}  chain small {tc1:.3f}",
      flush=True)
streams = {}
for n in (224, 192, 160, 128, 96, 64):
    streams[n] = s = masked_stream(range(n))
    a, _ = timed(update_plain, None, s, None)
    a2_, _ = timed(update_dgrad, None, s, None)
    c0, _ = timed(lambda: chain(False), None, s, None)
    c1, _ = timed(lambda: chain(True), None, s, None)
    print(f"  {n:3d} CUs: update {a:.3f}  update+dgrad {a2_:.3f}  chain big {c0:.3f}  chain small {c1:.3f}", flush=True)

# 4. together
print(f"  serial: update + chain big {tu + tc0:.3f}; fused-dgrad step part {tud + tc0:.3f}; split step part {tdg + tu:.3f} + hidden chain",
      flush=True)
for n in (64, 96, 128, 160):
    side = streams[n]
    comp = masked_stream(range(n, NCU))
    for co in (False, True):
        t_all, (m1, s1_) = timed(update_plain, lambda: chain(co), full, side)
        t_cmp, (m2, s2_) = timed(update_plain, lambda: chain(co), comp, side)
        print(f"  chain {'small' if co else 'big  '} on {n:3d} CUs | update on all: {t_all:.3f} (update {m1:.3f}, chain {s1_:.3f})"
              f" | update on the other {NCU - n}: {t_cmp:.3f} (update {m2:.3f}, chain {s2_:.3f})", flush=True)

# 5. graph capture: does the mask survive?
side = streams[64]
g = torch.cuda.CUDAGraph()
cap = torch.cuda.Stream()
with torch.cuda.stream(cap):
    with torch.cuda.graph(g, stream=cap):
        side.wait_stream(cap)
        with torch.cuda.stream(side):
            chain(False)
        cap.wait_stream(side)
for _ in range(2):
    g.replay()
torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record(); g.replay(); e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
c64, _ = timed(lambda: chain(False), None, side, None)
print(f"  graph replay of a chain captured on the 64-CU stream: {best:.3f} ms (eager on it {c64:.3f}, on all CUs {tc0:.3f})",
      flush=True)
