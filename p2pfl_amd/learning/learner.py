"""NodeLearner: the template every learner implements (reference: fedstellar/learning/learner.py:24-177)."""
from __future__ import annotations


class NodeLearner:
    """
    Template to implement learning processes, including metric monitoring during training.

    Args:
        model: model of the learner.
        data: data of the learner.
        config: node configuration.
        logger: metric logger.
    """

    def __init__(self, model, data, config=None, logger=None):
        pass

    def set_model(self, model):
        """Set the model (the architecture, not only its weights)."""
        pass

    def set_data(self, data):
        """Set the data module of the learner."""
        pass

    def encode_parameters(self, params=None, contributors=None, weight=None):
        """Serialize parameters (``params`` None: the current model) into the wire payload."""
        pass

    def decode_parameters(self, data):
        """Inverse of :meth:`encode_parameters`: ``(state_dict, contributors, weight)``.

        Raises:
            DecodingParamsError: if the payload is malformed."""
        pass

    def check_parameters(self, params):
        """True if ``params`` fits the local model (same keys, same shapes)."""
        pass

    def set_parameters(self, params):
        """Load ``params`` (a state_dict) into the model.

        Raises:
            ModelNotMatchingError: if they do not fit."""
        pass

    def get_parameters(self):
        """The model's state_dict."""
        pass

    def set_epochs(self, epochs):
        pass

    def fit(self):
        """Run local training for the configured number of epochs."""
        pass

    def interrupt_fit(self):
        """Ask a running :meth:`fit` to stop as soon as possible."""
        pass

    def evaluate(self):
        """Evaluate on the test shard; returns ``(loss, metric)`` or None."""
        pass

    def log_validation_metrics(self, loss, metric, round=None, name=None):
        pass

    def get_num_samples(self):
        """``(n_train, n_test)``; the first is the node's FedAvg weight."""
        pass

    def init(self):
        """Called once when learning starts."""
        pass

    def close(self):
        """Release loggers / resources."""
        pass

    def finalize_round(self):
        """Round bookkeeping (folds the logger's local step into its global step)."""
        pass

    def create_trainer(self):
        """(Re)create the per-round trainer / optimizer state."""
        pass
