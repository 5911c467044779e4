"""Would two independent half-shards on two streams beat one full shard? Clients of a round are independent, so a
100-client local-training round can run as 2 x 50 (or 4 x 25) clients, each group on its own stream with its own
captured graph: whenever one group sits in its HBM-bound update the other may be in its matrix-core-bound
convolutions. Times `fit(3 epochs)` of the headline shard shape for 1, 2 and 4 groups.

usage: python tools/probe/two_shards.py [fp32|fp32-mfma|bf16]
"""
import sys
import time
import torch
sys.path.insert(0, ".")
from p2pfl_amd.datasets.synthetic import synthetic_shards
from p2pfl_amd.models.specs import get_spec
from p2pfl_amd.runtime.engine import HipEngine

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
spec = get_spec("mnist-cnn")
K = 100
for groups in (1, 2, 4):
    per = K // groups
    engines, streams = [], []
    for g in range(groups):
        e = HipEngine(spec, per, "cuda", seed=11 + g, use_graph=True, precision=prec)
        e.init_params(identical=False)
        e.set_data(*synthetic_shards(per, 540, 60, 100, spec.num_classes, seed=3 + g, device="cuda"))
        engines.append(e)
        streams.append(torch.cuda.Stream())
    cur = torch.cuda.current_stream()

    def round_():
        for e, s in zip(engines, streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                e.fit(3)
        for s in streams:
            cur.wait_stream(s)

    for _ in range(2):
        round_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        round_()
    torch.cuda.synchronize()
    print(f"[{prec}] {groups} group(s) of {per} clients: fit(3 epochs) {1e3 * (time.perf_counter() - t0) / reps:.1f} ms", flush=True)
    for e in engines:
        e.close()
    del engines
    torch.cuda.empty_cache()
