"""Which op of the CIFAR-10 crop / flip augmentation breaks hipStreamEndCapture? Every case runs in a CHILD process
(faulthandler on, stderr kept) and the sweep stops at the first child that dies, so one call never continues on the
GPU after a crash.

    python tools/probe/capture_augment.py [out_dir]        (GPU box)

cases, cheapest first:
  pad        F.pad alone
  rng        device randint / rand alone (graph-safe Philox path)
  gather     the 4-index advanced-index gather with STATIC index tensors
  where      torch.where / arange index arithmetic with static offsets
  augment    CIFAR10DataModule.augment as shipped (RNG + index arithmetic + gather in one capture)
  predrawn   the same gather with offsets drawn OUTSIDE the capture into static buffers
  engine     ModuleEngine pass (FasterMobileNet, K=2, 64 samples) with the augmentation hook, graph forced on
"""
import os
import subprocess
import sys

CASES = ["pad", "rng", "gather", "where", "augment", "predrawn", "engine",
         # the launcher test that crashed: simplemobilenet, 4 nodes; growing passes (steps per captured graph)
         "engine:simplemobilenet:4:256", "engine:simplemobilenet:4:2048", "engine:simplemobilenet:4:12500"]


def child(case: str) -> None:
    import faulthandler
    faulthandler.enable()
    import torch
    import torch.nn.functional as F
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from p2pfl_amd.learning.pytorch.cifar10.cifar10 import CIFAR10DataModule
    dev = torch.device("cuda")
    b = 64
    x = torch.randn(b, 3, 32, 32, device=dev)
    ar = torch.arange(32, device=dev)
    ox = torch.randint(0, 9, (b,), device=dev)
    oy = torch.randint(0, 9, (b,), device=dev)
    flip = torch.rand(b, device=dev) < 0.5
    bi = torch.arange(b, device=dev).view(b, 1, 1)
    rows = (oy.view(b, 1) + ar.view(1, 32)).view(b, 32, 1)
    cols = (ox.view(b, 1) + ar.view(1, 32)).view(b, 1, 32)
    xp_static = F.pad(x, (4, 4, 4, 4))

    def body():
        if case == "pad":
            return F.pad(x, (4, 4, 4, 4))
        if case == "rng":
            return torch.randint(0, 9, (b,), device=dev).float() + torch.rand(b, device=dev)
        if case == "gather":
            return xp_static[bi, :, rows, cols].permute(0, 3, 1, 2).contiguous()
        if case == "where":
            c = ox.view(b, 1) + torch.where(flip.view(b, 1), 31 - ar.view(1, 32), ar.view(1, 32))
            return c.float()
        if case == "augment":
            return CIFAR10DataModule.augment(x)
        if case == "predrawn":
            xp = F.pad(x, (4, 4, 4, 4))
            r = oy.view(b, 1) + ar.view(1, 32)
            c = ox.view(b, 1) + torch.where(flip.view(b, 1), 31 - ar.view(1, 32), ar.view(1, 32))
            return xp[bi, :, r.view(b, 32, 1), c.view(b, 1, 32)].permute(0, 3, 1, 2).contiguous()
        raise SystemExit(f"unknown case {case}")

    if case.startswith("engine"):
        bits = case.split(":")
        model, K, n = (bits[1], int(bits[2]), int(bits[3])) if len(bits) == 4 else ("fastermobilenet", 2, 64)
        os.environ["P2PFL_MODULE_GRAPH_FORCE"] = "1"
        from p2pfl_amd.node_start import build_dataset_and_model, build_model
        from p2pfl_amd.runtime.module_engine import ModuleEngine, ModuleSpec
        probe = build_dataset_and_model("CIFAR10", model, 0, 4)[0]
        spec = ModuleSpec("CIFAR10-" + model, lambda: build_model("CIFAR10", model),
                          probe.num_classes, tuple(probe.input_shape), augment=getattr(probe, "augment", None))
        t = probe.tensors()
        data = [torch.stack([t[j][:n if j < 2 else 32] for _ in range(K)]) for j in range(6)]
        eng = ModuleEngine(spec, K, dev, seed=1, batched=True, batch_size=32)
        eng.set_data(*data)
        eng.init_params(identical=False)
        for r in range(3):
            eng.evaluate()
            out = eng.fit(1)
            print(f"  round {r}: graphs {len(eng._graphs)} use_graph {eng.use_graph} loss {float(out['Train']['loss'].mean()):.4f}", flush=True)
        torch.cuda.synchronize()
        print(f"case {case}: ok (graph on: {eng.use_graph}, graphs: {len(eng._graphs)})", flush=True)
        return
    body()                                   # eager warm-up
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        out = body()
    g.replay()
    torch.cuda.synchronize()
    print(f"case {case}: ok, output {tuple(out.shape)} finite {bool(torch.isfinite(out).all())}", flush=True)


def main() -> int:
    out_dir = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/capture_augment"
    os.makedirs(out_dir, exist_ok=True)
    only = sys.argv[2:] or CASES
    for case in only:
        with open(os.path.join(out_dir, f"{case.replace(':', '_')}.log"), "w") as log:
            rc = subprocess.run([sys.executable, "-u", os.path.abspath(__file__), "--child", case],
                                stdout=log, stderr=subprocess.STDOUT, timeout=240).returncode
        print(f"{case}: rc={rc}", flush=True)
        if rc != 0:
            print(f"stopping at the first failing case ({case}); see its log in {out_dir}", flush=True)
            return 1
    return 0


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        sys.exit(main())
