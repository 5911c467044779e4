from .config import Config, default_participant  # noqa: F401
