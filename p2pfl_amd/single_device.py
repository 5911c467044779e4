"""Two reference-style nodes in one process (reference: fedstellar/single_device.py:12-41, a dead stub there).

    python -m p2pfl_amd.single_device [--rounds 2] [--epochs 1]        (examples/single_device.py runs the same)

Starts two ``Node`` objects on localhost ports, connects them, runs a short DFL session on synthetic
MNIST-shaped shards and prints each node's final test accuracy.
"""
import argparse
import time

from .config.config import Config, fast_timings
from .learning.pytorch.mnist.mnist import MNISTDataModule
from .learning.pytorch.mnist.models.mlp import MNISTModelMLP
from .node import Node


def make_node(idx, port, n):
    cfg = Config(participant={"scenario_args": {"federation": "DFL", "n_nodes": n, "name": "single_device", "controller": ""},
                              "device_args": {"role": "aggregator", "idx": idx, "accelerator": "auto"},
                              "tracking_args": {"local_tracking": "none", "log_dir": ""}})
    fast_timings(cfg.participant, 0.02)
    return Node(idx, "single_device", MNISTModelMLP(), MNISTDataModule(sub_id=idx, number_sub=100), host="127.0.0.1",
                port=port, config=cfg)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--base-port", type=int, default=45500)
    a = ap.parse_args(argv)
    nodes = [make_node(i, a.base_port + i, 2) for i in range(2)]
    for n in nodes:
        n.start()
    time.sleep(0.5)
    nodes[0].connect_to("127.0.0.1", a.base_port + 1)
    time.sleep(1.0)
    nodes[0].set_start_learning(rounds=a.rounds, epochs=a.epochs)
    while any(n.round is not None for n in nodes):
        time.sleep(0.5)
    results = {}
    for n in nodes:
        results[n.get_name()] = n.learner.evaluate()
        print(n.get_name(), "test (loss, accuracy):", results[n.get_name()])
        n.stop()
    return results


if __name__ == "__main__":
    main()
