from .retinanet import RetinaNet  # noqa: F401
from .free_anchor import FreeAnchor  # noqa: F401
from .fcos import ATSS, FCOS, OTA  # noqa: F401
from .faster_rcnn import FasterRCNN  # noqa: F401
