"""FEMNIST CNN (reference: fedstellar/learning/pytorch/femnist/models/cnn.py:18-212)."""
from .....models.torch_models import FEMNISTModelCNN  # noqa: F401
