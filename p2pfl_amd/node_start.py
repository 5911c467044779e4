"""Per-process node entry point: ``python -m p2pfl_amd.node_start <participant.json>``
(reference: fedstellar/node_start.py:28-125).

Loads the participant config, builds the dataset x model pair, starts the node, dials the
configured neighbours and — on the start node — kicks off learning. The reference hard-codes a
30 s settle sleep and 5 s between neighbour dials (node_start.py:106,112); both are configurable
here (``P2PFL_START_DELAY`` / ``P2PFL_CONNECT_DELAY`` seconds, defaults 2 / 0.2): co-resident
processes on one MI355X box do not need WAN-sized pauses.
"""
from __future__ import annotations

import logging
import os
import sys
import time

from .config import Config
from .learning.pytorch.lightninglearner import LightningLearner
from .node import Node


def build_dataset_and_model(dataset: str, model_name: str, idx: int, n_nodes: int, root_dir=None):
    """The dataset x model factory of the reference (node_start.py:48-85) + the WADI branch it lacks."""
    if dataset == "MNIST":
        from .learning.pytorch.mnist.mnist import MNISTDataModule
        data = MNISTDataModule(sub_id=idx, number_sub=n_nodes, iid=True, root_dir=root_dir)
        if model_name == "MLP":
            from .learning.pytorch.mnist.models.mlp import MNISTModelMLP as M
        elif model_name == "CNN":
            from .learning.pytorch.mnist.models.cnn import MNISTModelCNN as M
        else:
            raise ValueError(f"Model {model_name} not supported for MNIST")
        return data, M()
    if dataset == "FEMNIST":
        from .learning.pytorch.femnist.femnist import FEMNISTDataModule
        data = FEMNISTDataModule(sub_id=idx, number_sub=n_nodes, root_dir=root_dir)
        if model_name == "CNN":
            from .learning.pytorch.femnist.models.cnn import FEMNISTModelCNN as M
        else:
            raise ValueError(f"Model {model_name} not supported for FEMNIST")
        return data, M()
    if dataset == "SYSCALL":
        from .learning.pytorch.syscall.syscall import SYSCALLDataModule
        data = SYSCALLDataModule(sub_id=idx, number_sub=n_nodes, root_dir=root_dir)
        if model_name == "MLP":
            from .learning.pytorch.syscall.models.mlp import SyscallModelMLP as M
        elif model_name == "SVM":
            from .learning.pytorch.syscall.models.svm import SyscallModelSGDOneClassSVM as M
        elif model_name == "Autoencoder":
            from .learning.pytorch.syscall.models.autoencoder import SyscallModelAutoencoder as M
        else:
            raise ValueError(f"Model {model_name} not supported for SYSCALL")
        return data, M()
    if dataset == "CIFAR10":
        from .learning.pytorch.cifar10.cifar10 import CIFAR10DataModule
        data = CIFAR10DataModule(sub_id=idx, number_sub=n_nodes, root_dir=root_dir)
        if model_name == "ResNet9":
            from .learning.pytorch.cifar10.models.resnet import CIFAR10ModelResNet
            return data, CIFAR10ModelResNet(classifier="resnet9")
        if model_name == "ResNet18":
            from .learning.pytorch.cifar10.models.resnet import CIFAR10ModelResNet
            return data, CIFAR10ModelResNet(implementation="own", classifier="resnet18")
        if model_name == "fastermobilenet":
            from .learning.pytorch.cifar10.models.fastermobilenet import FasterMobileNet as M
        elif model_name == "simplemobilenet":
            from .learning.pytorch.cifar10.models.simplemobilenet import SimpleMobileNetV1 as M
        else:
            raise ValueError(f"Model {model_name} not supported for CIFAR10")
        return data, M()
    if dataset == "WADI":
        from .learning.pytorch.wadi.models.mlp import WADIModelMLP
        from .learning.pytorch.wadi.wadi import WADIDataModule
        return WADIDataModule(sub_id=idx, number_sub=n_nodes, root_dir=root_dir), WADIModelMLP()
    raise ValueError(f"Dataset {dataset} not supported")


def build_model(dataset: str, model_name: str):
    """Only the model of :func:`build_dataset_and_model` (a tiny throw-away split keeps the data part cheap)."""
    return build_dataset_and_model(dataset, model_name, 0, 1000)[1]


def main():
    config_path = str(sys.argv[1])
    config = Config(entity="participant", participant_config_file=config_path)
    p = config.participant
    n_nodes = p["scenario_args"]["n_nodes"]
    experiment_name = p["scenario_args"]["name"]
    host, port = p["network_args"]["ip"], p["network_args"]["port"]
    neighbors = p["network_args"]["neighbors"].split()
    idx = p["device_args"]["idx"]
    rounds, epochs = p["scenario_args"]["rounds"], p["training_args"]["epochs"]
    aggregation_algorithm = p["aggregator_args"]["algorithm"]
    if aggregation_algorithm != "FedAvg":
        raise ValueError(f"Aggregation algorithm {aggregation_algorithm} not supported")

    dataset, model = build_dataset_and_model(p["data_args"]["dataset"], p["model_args"]["model"], int(idx), int(n_nodes),
                                             p["data_args"].get("root_dir"))
    encrypt = bool(p["scenario_args"].get("encrypt", False))
    node = Node(idx=idx, experiment_name=experiment_name, model=model, data=dataset, hostdemo=p["network_args"].get("ipdemo"),
                host=host, port=port, config=config, learner=LightningLearner, encrypt=encrypt)
    node.start()
    print("Node started, grace time for network start-up", flush=True)
    time.sleep(float(os.environ.get("P2PFL_START_DELAY", "2")))
    for nb in neighbors:
        addr, nport = nb.split(":")[0], int(nb.split(":")[1])
        node.connect_to(addr, nport, full=False)
        time.sleep(float(os.environ.get("P2PFL_CONNECT_DELAY", "0.2")))
    logging.info("Neighbors: %s", node.get_neighbors_names())
    logging.info("Network nodes: %s", node.get_network_nodes())
    if p["device_args"]["start"]:
        time.sleep(float(os.environ.get("P2PFL_START_DELAY", "2")))
        node.set_start_learning(rounds=rounds, epochs=epochs)
    # keep the process alive until the experiment is over (node threads are daemons)
    try:
        started = False
        while node.is_alive():
            if node.round is not None:
                started = True
            elif started:
                time.sleep(float(p["HEARTBEAT_PERIOD"]) * 2)   # let the last MODELS_READY / gossip drain
                break
            time.sleep(0.5)
    except KeyboardInterrupt:
        pass
    node.stop()


if __name__ == "__main__":
    os.system("clear") if os.environ.get("P2PFL_CLEAR") else None
    main()
