"""Two half-federations beside each other: can the HBM-bound dense chain of clients 0..49 run on one set of CUs while the
matrix-pipe-bound convolution chain of clients 50..99 runs on the rest (and vice versa)? The halves are independent
(no client depends on another during local training), so - unlike update || convolutions of the SAME clients - the
fused data gradient stays, and the two chains only have to alternate.

Per step and half: dense stream: forward product -> fused update + data gradient; conv stream: convolution backward,
small optimizer, convolution forward of the next batch. Schedules, averaged over STEPS steps:
  serial      : the headline schedule (100 clients, one stream)
  halves      : two halves, dense / conv streams without CU masks (what a captured graph could replay)
  halves D/C  : the same on CU-masked streams (dense on D CUs, convolutions on the other C) - eager launches only

usage: python tools/probe/halves_probe.py [fp32|fp32-mfma]
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



y = torch.empty(K, B, 2048, device=dev)
STEPS = 8
H = K // 2
sl = [slice(0, H), slice(H, K)]


def dense(s):
    h.linear_fwd(x[s], w[s], b[s], y[s], True)
    assert h.linear_wgrad_adam_dgrad(dy[s], x[s], w[s], mw[s], vw[s], b[s], mb[s], vb[s], dx[s], lr=1e-3, t=2)


def convs(s):
    h.conv2_wgrad(g2[s], code[s], a1[s], dw2[s], db2[s], 5, scratch=scr[s])
    h.conv2_dgrad(g2[s], code[s], w2[s], da1[s], 5, wprep=wprep[s])
    h.conv1_wgrad(da1[s], code1[s], data[s], sidx[s], dw1[s], db1[s], 5)
    h.conv1_fwd(data[s], sidx[s], w1[s], b1[s], a1[s], code1[s], 5)
    h.conv2_prep(w2[s], wprep[s], 5)
    h.conv2_fwd(a1[s], w2[s], b2[s], a2[s], code[s], 5, wprep=wprep[s])


allk = slice(0, K)
scr = scratch
if scr is None or scr.shape[0] != K:
    class _S:                                   # scratch is not per-client sliceable: one per half
        def __init__(self):
            self.h = {}
        def __getitem__(self, s):
            n = s.stop - s.start
            if (s.start, n) not in self.h:
                self.h[(s.start, n)] = h.conv2_wgrad_scratch(n, B, 5, dev)
            return self.h[(s.start, n)]
    scr = _S()


def run_serial(stream):
    with torch.cuda.stream(stream):
        for _ in range(STEPS):
            dense(allk)
            convs(allk)


def run_halves(ds, cs):
    upd = [None, None]
    chain_done = [None, None]
    for _ in range(STEPS):
        for i in (0, 1):
            with torch.cuda.stream(ds):
                if chain_done[i] is not None:
                    ds.wait_event(chain_done[i])
                dense(sl[i])
                upd[i] = torch.cuda.Event()
                upd[i].record(ds)
            with torch.cuda.stream(cs):
                cs.wait_event(upd[i])
                convs(sl[i])
                chain_done[i] = torch.cuda.Event()
                chain_done[i].record(cs)


def measure(fn, streams, reps=4):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for s in streams:
            s.wait_stream(torch.cuda.current_stream())
        fn()
        for s in streams:
            torch.cuda.current_stream().wait_stream(s)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / STEPS)
    return best


print(f"[{prec}] {NCU} CUs, {STEPS} steps per measurement, ms per step", flush=True)
full, full2 = torch.cuda.Stream(), torch.cuda.Stream()
run_serial(full); run_halves(full, full2)
torch.cuda.synchronize()
print(f"  serial (100 clients, one stream)            : {measure(lambda: run_serial(full), [full]):.3f}", flush=True)
print(f"  halves on ONE stream (50 + 50, in order)    : {measure(lambda: run_halves(full, full), [full]):.3f}", flush=True)
print(f"  halves, dense / conv streams, no masks      : {measure(lambda: run_halves(full, full2), [full, full2]):.3f}", flush=True)
for d in (128, 144, 160, 176, 192):
    ds, cs = masked_stream(range(d)), masked_stream(range(d, NCU))
    run_halves(ds, cs)
    torch.cuda.synchronize()
    print(f"  halves, dense on {d:3d} CUs / convs on {NCU - d:3d} CUs : {measure(lambda: run_halves(ds, cs), [ds, cs]):.3f}", flush=True)
for d in (160, 192):
    ds, cs = masked_stream(range(d)), full2           # convolutions may go anywhere, the dense chain keeps d CUs
    run_halves(ds, cs)
    torch.cuda.synchronize()
    print(f"  halves, dense on {d:3d} CUs / convs unmasked    : {measure(lambda: run_halves(ds, cs), [ds, cs]):.3f}", flush=True)
