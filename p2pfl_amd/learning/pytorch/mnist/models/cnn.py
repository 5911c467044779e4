"""MNIST CNN (reference: fedstellar/learning/pytorch/mnist/models/cnn.py:18-219; forward fixed to
return the logits)."""
from .....models.torch_models import MNISTModelCNN  # noqa: F401
