"""HIP engine (default precision policy, or P2PFL_PRECISION) against the fp32 eager engine over whole federations: per-round test loss / accuracy of both.

usage (GPU box): python tools/convergence_check.py [rounds] [nodes] [model ...] > profiles/<name>.json
Same topology, data shards, initial weights and shuffles for both engines; differences come from the summation
order (fp32 policies) or from bf16 operand rounding (bf16 policy: fp32 accumulation, master weights, optimizer state).
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from p2pfl_amd.datasets.synthetic import synthetic_shards  # noqa: E402
from p2pfl_amd.models.specs import get_spec  # noqa: E402
from p2pfl_amd.runtime.federation import BatchedFederation  # noqa: E402
from p2pfl_amd.utils.topologymanager import TopologyManager  # noqa: E402


def run(model, impl, rounds, nodes):
    spec = get_spec(model)
    tm = TopologyManager(n_nodes=nodes, b_symmetric=True)
    tm.generate_ring_topology()
    fed = BatchedFederation(spec, tm.topology, device="cuda", engine=impl, epochs=2, seed=5)
    fed.init_params()
    fed.set_local_data(*synthetic_shards(nodes, 256, 32, 128, spec.num_classes, seed=5, device="cuda"))
    hist = []
    for _ in range(rounds):
        rep = fed.run_round()
        hist.append({"test_loss": float(rep.test_loss.mean()), "test_acc": float(rep.test_accuracy.mean()),
                     "train_loss": float(rep.train_loss.mean())})
    final = fed.engine.evaluate()
    return hist, {"test_loss": float(final["loss"].mean()), "test_acc": float(final["accuracy"].mean())}


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    nodes = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    out = {"rounds": rounds, "nodes": nodes, "topology": "ring", "epochs": 2, "models": {}}
    for model in (sys.argv[3:] or ("mnist-mlp", "mnist-cnn", "femnist-cnn")):
        res = {}
        for impl in ("hip", "eager"):
            hist, final = run(model, impl, rounds, nodes)
            res[impl] = {"per_round": hist, "final": final}
        res["max_abs_test_loss_gap"] = max(abs(a["test_loss"] - b["test_loss"])
                                           for a, b in zip(res["hip"]["per_round"], res["eager"]["per_round"]))
        res["final_acc_gap"] = res["hip"]["final"]["test_acc"] - res["eager"]["final"]["test_acc"]
        out["models"][model] = res
        print(f"{model}: final acc hip {res['hip']['final']['test_acc']:.4f} eager {res['eager']['final']['test_acc']:.4f} "
              f"max per-round test-loss gap {res['max_abs_test_loss_gap']:.4f}", file=sys.stderr)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
