"""MNIST MLP (reference: fedstellar/learning/pytorch/mnist/models/mlp.py:28-222)."""
from .....models.torch_models import MNISTModelMLP  # noqa: F401
