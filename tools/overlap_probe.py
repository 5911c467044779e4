"""How well does a convolution kernel hide under the fused wgrad+Adam kernel? Times each alone and both together
(two streams, the side one with priority -1) for the headline shard shape (100 clients, batch 32).

usage: python tools/overlap_probe.py [fp32|bf16]
"""
import sys
import torch
sys.path.insert(0, ".")
from p2pfl_amd.ops import hip

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32-mfma"
K, B = 100, 32
dev = "cuda"
h = hip.bound(prec)
P = 3136 * 2048 + 2048
slab = [torch.randn(K, P, device=dev) * 0.02 for _ in range(3)]
view = lambda s: (s[:, :3136 * 2048].unflatten(1, (2048, 3136)), s[:, 3136 * 2048:])
(w, b), (mw, mb), (vw, vb) = view(slab[0]), view(slab[1]), view(slab[2])
dy = torch.randn(K, B, 2048, device=dev) * 0.01
x = torch.randn(K, B, 3136, device=dev).relu()
a1 = torch.randn(K, B, 32, 196, device=dev).relu()
w2 = torch.randn(K, 64, 800, device=dev) * 0.05
b2 = torch.randn(K, 64, device=dev)
a2 = torch.empty(K, B, 64, 49, device=dev)
code = torch.randint(0, 5, (K, B, 64, 49), device=dev).to(torch.uint8)
g2 = torch.randn(K, B, 64, 49, device=dev)
da1 = torch.empty(K, B, 32, 196, device=dev)
dw2, db2 = torch.empty(K, 64, 800, device=dev), torch.empty(K, 64, device=dev)
wprep = h.conv2_prep_alloc(K, 5, dev)
h.conv2_prep(w2, wprep, 5)
scratch = h.conv2_wgrad_scratch(K, B, 5, dev)
co = dict(coresident=True) if prec == "fp32-mfma" else {}

def update():
    h.linear_wgrad_adam(dy, x, w, mw, vw, b, mb, vb, lr=1e-3, t=2)

convs = {
    "fwd": lambda kw: h.conv2_fwd(a1, w2, b2, a2, code, 5, wprep=wprep, **kw),
    "dgrad": lambda kw: h.conv2_dgrad(g2, code, w2, da1, 5, wprep=wprep, **kw),
    "wgrad": lambda kw: h.conv2_wgrad(g2, code, a1, dw2, db2, 5, scratch=scratch, **kw),
}
import os
side = torch.cuda.Stream(priority=int(os.environ.get("PROBE_SIDE_PRIO", "-1")))

def timed(fn_main, fn_side, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1, s1 = torch.cuda.Event(True), torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        side.wait_stream(torch.cuda.current_stream())
        if fn_main:
            fn_main()
        e1.record()
        if fn_side:
            with torch.cuda.stream(side):
                fn_side()
                s1.record()
            torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        tm = e0.elapsed_time(e1) if fn_main else 0.0
        ts = e0.elapsed_time(s1) if fn_side else 0.0
        best = min(best, max(tm, ts))
        last = (tm, ts)
    return best, last

for _ in range(2):
    update()
tu, _ = timed(update, None)
print(f"[{prec}] update alone {tu:.3f} ms")
for name, f in convs.items():
    for label, kw in (("big", {}), ("small", co)):
        if label == "small" and not co:
            continue
        f(kw)
        tc, _ = timed(None, lambda: f(kw))
        tb, (tm, ts) = timed(update, lambda: f(kw))
        print(f"  conv2_{name:5s} {label:5s}: alone {tc:.3f}  together {tb:.3f} (update {tm:.3f}, conv {ts:.3f})  "
              f"serial {tu + tc:.3f}  hidden {100 * (tu + tc - tb) / tc:.0f} %")

# ---- synthetic co-runners (tools/probe/burn.hip): what exactly does the update kernel lose time to?
import ctypes, os
lib_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "libburn.so")
if os.path.exists(lib_path) and prec == "fp32-mfma":
    burn = ctypes.CDLL(lib_path)
    burn.burn_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    out = torch.zeros(4, device=dev)
    for which, name, iters in ((0, "mfma_f32", 9000),):
        for blocks in (256,):
            fn = lambda: burn.burn_launch(which, blocks, iters * 256 // blocks if which != 3 else iters, out.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream)
            fn()
            tc, _ = timed(None, fn)
            tb, (tm, ts) = timed(update, fn)
            print(f"  burn {name:9s} x{blocks:5d} blocks: alone {tc:.3f}  together {tb:.3f} (update {tm:.3f}, burn {ts:.3f})  "
                  f"hidden {100 * (tu + tc - tb) / tc:.0f} %")
    # VALU-only work beside MFMA-only work: do they share anything?
    fv = lambda: burn.burn_launch(2, 1024, 30000 // 4, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    fm = lambda: burn.burn_launch(0, 256, 9000, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    fv(); fm()
    tv, _ = timed(fv, None)
    tm_, _ = timed(None, fm)
    tb, (a_, b_) = timed(fv, fm)
    print(f"  valu burn alone {tv:.3f}, mfma burn alone {tm_:.3f}, together {tb:.3f} (valu {a_:.3f}, mfma {b_:.3f})")

    # whole CUs taken away from the update kernel (LDS-hogging sleepers): how many can it spare?
    burn.burn_hog_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    for blocks in (32, 64, 96, 128, 160):
        fh = lambda: burn.burn_hog_launch(blocks, 4000, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        fh()
        th, _ = timed(None, fh)
        # hog first (side), then the update on main: the hogs are resident for the update's whole duration
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            with torch.cuda.stream(side):
                fh()
            time_sleep = __import__("time").sleep(0.0005)
            e0.record()
            update()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        print(f"  {blocks:3d} CUs hogged (sleep {th:.2f} ms): update {best:.3f} ms")
