"""basedet_amd: MI355X-native training hot path behind BaseDet's operator / model / solver surface."""
import os

# One hardware queue per stream the step uses (main, weight-gradient, top-block, communication + RCCL's own): with HIP's default of
# four, the weight-gradient side stream can land on the main stream's queue once a process group exists, and the two then run
# back to back (bench.py, scripts/queue_map.py).  Read by the HIP runtime when it initialises, i.e. at the first device call.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
