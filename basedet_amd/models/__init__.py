from .retinanet import RetinaNet  # noqa: F401
