"""Learning exceptions (reference: fedstellar/learning/exceptions.py:7-20)."""


class DecodingParamsError(Exception):
    """The received payload could not be decoded into model parameters."""


class ModelNotMatchingError(Exception):
    """The received parameters do not match the local model (keys or shapes)."""
