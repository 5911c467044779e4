"""Model descriptors for the client-batched engine.

A :class:`ModelSpec` is the static description the slab engine executes: the ordered parameter
list (== ``state_dict`` order of the reference Lightning modules, which is also the array order
of the serialized payload), the layer sequence, and the optimizer policy.

Reference architectures:
* MNIST MLP      – fedstellar/learning/pytorch/mnist/models/mlp.py:144-167
* MNIST CNN      – fedstellar/learning/pytorch/mnist/models/cnn.py:134-164 (its ``forward`` returns
  the input instead of the logits, cnn.py:159; this build returns logits)
* FEMNIST CNN    – fedstellar/learning/pytorch/femnist/models/cnn.py:136-157
* SYSCALL auto-encoder – fedstellar/learning/pytorch/syscall/models/autoencoder.py (six linear layers, MSE against the input)
* WADI MLP       – fedstellar/learning/pytorch/wadi/models/mlp.py (nine linear layers, ReLU between, ONE logit through a
  sigmoid + BCELoss = the two-class problem the loss kernel solves for C = 1)
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple


@dataclass(frozen=True)
class ParamSpec:
    name: str
    shape: Tuple[int, ...]

    @property
    def numel(self) -> int:
        n = 1
        for s in self.shape:
            n *= s
        return n


@dataclass(frozen=True)
class LayerSpec:
    kind: str                 # "conv1pool" | "conv2pool" | "linear"
    weight: str               # parameter names
    bias: str
    in_features: int = 0      # linear
    out_features: int = 0
    relu: bool = False
    cin: int = 1              # conv
    cout: int = 0
    ksize: int = 0


@dataclass(frozen=True)
class OptimSpec:
    kind: str                 # "adam" | "sgd"
    lr: float
    momentum: float = 0.0
    beta1: float = 0.9
    beta2: float = 0.999
    eps: float = 1e-8
    weight_decay: float = 0.0


@dataclass(frozen=True)
class ModelSpec:
    name: str
    params: Tuple[ParamSpec, ...]
    layers: Tuple[LayerSpec, ...]
    optim: OptimSpec
    num_classes: int
    input_shape: Tuple[int, int, int] = (1, 28, 28)
    loss: str = "ce"          # "ce": softmax cross-entropy over the last layer's outputs (ONE output: sigmoid + BCE);
    #                           "mse_input": reconstruction error against the input batch (auto-encoders: no class metrics)

    @property
    def num_params(self) -> int:
        return sum(p.numel for p in self.params)

    def param(self, name: str) -> ParamSpec:
        for p in self.params:
            if p.name == name:
                return p
        raise KeyError(name)

    def flops_per_sample_fwd(self) -> int:
        """Multiply-add FLOPs (2/MAC) of one forward pass, for throughput accounting."""
        total = 0
        for l in self.layers:
            if l.kind == "linear":
                total += 2 * l.in_features * l.out_features
            elif l.kind == "conv1pool":
                total += 2 * 28 * 28 * l.cout * l.ksize * l.ksize
            elif l.kind == "conv2pool":
                total += 2 * 14 * 14 * l.cout * l.cin * l.ksize * l.ksize
        return total


def mnist_mlp(num_classes: int = 10, lr: float = 1e-3) -> ModelSpec:
    return ModelSpec(
        name="mnist-mlp",
        params=(
            ParamSpec("l1.weight", (256, 784)), ParamSpec("l1.bias", (256,)),
            ParamSpec("l2.weight", (128, 256)), ParamSpec("l2.bias", (128,)),
            ParamSpec("l3.weight", (num_classes, 128)), ParamSpec("l3.bias", (num_classes,)),
        ),
        layers=(
            LayerSpec("linear", "l1.weight", "l1.bias", 784, 256, relu=True),
            LayerSpec("linear", "l2.weight", "l2.bias", 256, 128, relu=True),
            LayerSpec("linear", "l3.weight", "l3.bias", 128, num_classes),
        ),
        optim=OptimSpec("adam", lr),
        num_classes=num_classes,
    )


def mnist_cnn(num_classes: int = 10, lr: float = 1e-3) -> ModelSpec:
    return ModelSpec(
        name="mnist-cnn",
        params=(
            ParamSpec("conv1.weight", (32, 1, 5, 5)), ParamSpec("conv1.bias", (32,)),
            ParamSpec("conv2.weight", (64, 32, 5, 5)), ParamSpec("conv2.bias", (64,)),
            ParamSpec("l1.weight", (2048, 3136)), ParamSpec("l1.bias", (2048,)),
            ParamSpec("l2.weight", (num_classes, 2048)), ParamSpec("l2.bias", (num_classes,)),
        ),
        layers=(
            LayerSpec("conv1pool", "conv1.weight", "conv1.bias", cin=1, cout=32, ksize=5),
            LayerSpec("conv2pool", "conv2.weight", "conv2.bias", cin=32, cout=64, ksize=5),
            LayerSpec("linear", "l1.weight", "l1.bias", 3136, 2048, relu=True),
            LayerSpec("linear", "l2.weight", "l2.bias", 2048, num_classes),
        ),
        optim=OptimSpec("adam", lr),
        num_classes=num_classes,
    )


def femnist_cnn(num_classes: int = 62, lr: float = 1e-3) -> ModelSpec:
    return ModelSpec(
        name="femnist-cnn",
        params=(
            ParamSpec("conv1.weight", (32, 1, 7, 7)), ParamSpec("conv1.bias", (32,)),
            ParamSpec("conv2.weight", (64, 32, 3, 3)), ParamSpec("conv2.bias", (64,)),
            ParamSpec("out.weight", (num_classes, 3136)), ParamSpec("out.bias", (num_classes,)),
        ),
        layers=(
            LayerSpec("conv1pool", "conv1.weight", "conv1.bias", cin=1, cout=32, ksize=7),
            LayerSpec("conv2pool", "conv2.weight", "conv2.bias", cin=32, cout=64, ksize=3),
            LayerSpec("linear", "out.weight", "out.bias", 3136, num_classes),
        ),
        optim=OptimSpec("sgd", lr, momentum=0.9),
        num_classes=num_classes,
    )


WADI_DIMS = (123, 1024, 512, 256, 128, 64, 32, 16, 8)


def wadi_mlp(num_classes: int = 1, lr: float = 1e-3) -> ModelSpec:
    """``num_classes`` = outputs of the last layer as the model's constructor names them (``out_channels`` = 1: one
    logit); the labels are binary, so the engine's counters and confusion matrix have TWO classes."""
    if num_classes != 1:
        raise ValueError("the WADI MLP has one output (sigmoid + BCE)")
    dims = WADI_DIMS + (1,)
    params, layers = [], []
    for i in range(9):
        params += [ParamSpec(f"l{i + 1}.weight", (dims[i + 1], dims[i])), ParamSpec(f"l{i + 1}.bias", (dims[i + 1],))]
        layers.append(LayerSpec("linear", f"l{i + 1}.weight", f"l{i + 1}.bias", dims[i], dims[i + 1], relu=i < 8))
    return ModelSpec(name="wadi-mlp", params=tuple(params), layers=tuple(layers), optim=OptimSpec("adam", lr),
                     num_classes=2, input_shape=(1, 1, 123))


def syscall_autoencoder(num_classes: int = 9, lr: float = 1e-2) -> ModelSpec:
    """17 -> 64 -> 16 -> 8 -> 16 -> 64 -> 17, ReLU after all but the last layer, MSE against the input, Adam
    (reference syscall/models/autoencoder.py). ``num_classes`` = classes of the data set's labels (unused by the loss)."""
    dims = (17, 64, 16, 8, 16, 64, 17)
    params, layers = [], []
    for i in range(6):
        params += [ParamSpec(f"fc{i + 1}.weight", (dims[i + 1], dims[i])), ParamSpec(f"fc{i + 1}.bias", (dims[i + 1],))]
        layers.append(LayerSpec("linear", f"fc{i + 1}.weight", f"fc{i + 1}.bias", dims[i], dims[i + 1], relu=i < 5))
    return ModelSpec(name="syscall-autoencoder", params=tuple(params), layers=tuple(layers), optim=OptimSpec("adam", lr),
                     num_classes=max(int(num_classes), 2), input_shape=(1, 1, 17), loss="mse_input")


SPECS = {
    "syscall-autoencoder": syscall_autoencoder,
    "wadi-mlp": wadi_mlp,
    "mnist-mlp": mnist_mlp,
    "mnist-cnn": mnist_cnn,
    "femnist-cnn": femnist_cnn,
}


def get_spec(name: str, **kw) -> ModelSpec:
    key = name.lower().replace("_", "-")
    if key not in SPECS:
        raise KeyError(f"unknown model spec {name!r}; known: {sorted(SPECS)}")
    return SPECS[key](**kw)
