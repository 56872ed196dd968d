from .registry import registers  # noqa: F401
from .dummy import DummyLoader  # noqa: F401
from .checkpoint import load_matched_weights, save_checkpoint  # noqa: F401
