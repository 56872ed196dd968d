from .registry import registers  # noqa: F401
from .dummy import DummyLoader  # noqa: F401
