"""The client-batched runtime as a library: all nodes of a federation in ONE process per GPU.

    python examples/batched_federation.py                       # 16 MNIST-CNN nodes, ring, hand-written HIP kernels on a GPU
    python examples/batched_federation.py --dataset SYSCALL --model MLP --nodes 32 --topology random
    python examples/batched_federation.py --federation CFL --topology star

Spec models (MNIST MLP / CNN, FEMNIST CNN) train on the client-batched gfx950 kernels (eager PyTorch on a CPU-only
host); every other family of the reference's factory keeps its ``nn.Module`` and runs one ``torch.func.vmap`` step for
all nodes. Either way the models live as rows of one ``[nodes, P]`` slab and a round is
evaluate -> local epochs -> exchange -> FedAvg over each node's neighbourhood (reference node.py:402-456).
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from p2pfl_amd.models.specs import get_spec  # noqa: E402
from p2pfl_amd.node_start import build_dataset_and_model, build_model  # noqa: E402
from p2pfl_amd.runtime.federation import BatchedFederation  # noqa: E402
from p2pfl_amd.runtime.module_engine import ModuleSpec  # noqa: E402
from p2pfl_amd.utils.topologymanager import TopologyManager  # noqa: E402

SPECS = {("MNIST", "MLP"): "mnist-mlp", ("MNIST", "CNN"): "mnist-cnn", ("FEMNIST", "CNN"): "femnist-cnn"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", default="MNIST")
    ap.add_argument("--model", default="CNN")
    ap.add_argument("--nodes", type=int, default=16)
    ap.add_argument("--topology", default="ring", choices=["ring", "fully", "random", "star"])
    ap.add_argument("--federation", default="DFL", choices=["DFL", "SDFL", "CFL"])
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--samples", type=int, default=256, help="training samples per node (taken from the node's shard)")
    a = ap.parse_args()
    n = a.nodes
    dev = torch.device("cuda" if torch.cuda.is_available() else "cpu")

    tm = TopologyManager(n_nodes=n, b_symmetric=True, undirected_neighbor_num=3, seed=0)
    {"ring": tm.generate_ring_topology, "fully": tm.generate_topology, "star": tm.generate_server_topology,
     "random": lambda: tm.generate_topology(extra_random_links=2.0)}[a.topology]()

    shards = [build_dataset_and_model(a.dataset, a.model, i, n)[0] for i in range(n)]
    key = (a.dataset.upper(), a.model.upper() if a.model.upper() in ("MLP", "CNN") else a.model)
    if key in SPECS:
        spec = get_spec(SPECS[key])
    else:
        spec = ModuleSpec(f"{a.dataset}-{a.model}", lambda: build_model(a.dataset, a.model), int(shards[0].num_classes),
                          tuple(shards[0].input_shape), augment=getattr(shards[0], "augment", None))
    roles = None
    if a.federation == "CFL":
        roles = ["server"] + ["trainer"] * (n - 1)
    elif a.federation == "SDFL":
        roles = ["aggregator"] + ["trainer"] * (n - 1)
    fed = BatchedFederation(spec, tm.topology, federation=a.federation, roles=roles, device=dev, epochs=a.epochs)

    def stack(j, limit):
        m = min(min(s.tensors()[j].shape[0] for s in shards), limit)
        return torch.stack([s.tensors()[j][:m] for s in shards])
    fed.init_params()
    fed.set_local_data(stack(0, a.samples), stack(1, a.samples), stack(2, 64), stack(3, 64), stack(4, 128), stack(5, 128))
    print(f"{n} x {a.dataset}/{a.model} on {dev} ({type(fed.engine).__name__}), {a.federation} / {a.topology}")
    for rep in fed.run(a.rounds):
        acc = "" if rep.test_accuracy is None else f" test acc {float(rep.test_accuracy.mean()):.3f}"
        print(f"round {rep.round}: {rep.seconds * 1e3:8.1f} ms  test loss {float(rep.test_loss.mean()):.4f}{acc}  "
              f"train loss {float(rep.train_loss.mean()):.4f}")
    final = fed.final_test
    print(f"final: test loss {float(final['loss'].mean()):.4f}  accuracy {float(final['accuracy'].mean()):.3f}  "
          f"(spread over nodes {float(final['accuracy'].std()):.3f})")


if __name__ == "__main__":
    main()
