"""Synthetic batch generator restating basedet/utils/dummy.py:8-63 (the reference's benchmark input,
tools/benchmark.py:173).  The annotation pattern is data (also pinned in tests/golden/dummy_loader.npz)."""
import numpy as np

__all__ = ["DummyLoader"]

_ANNO = np.array([
    [[0., 0., 800., 800., 61.], [148.33984, 488.73206, 667.7124, 602.64056, 52.],
     [170.45752, 422.78433, 572.1176, 552.15686, 52.], [228.24835, 486.88892, 600.71893, 589.39874, 52.],
     [71.803894, 54.444447, 110.20911, 78.19608, 43.], [237.46405, 0., 418.64053, 32.03922, 41.],
     [315.08798, 101.472, 464.52798, 797.696, 80.], [280.448, 118.096, 370.336, 786.864, 70.],
     [228.31999, 104.71999, 307.40802, 791.456, 40.], [145.61601, 94.736, 246.288, 786.86395, 20.]],
    [[315.08798, 101.472, 464.52798, 797.696, 30.], [280.448, 118.096, 370.336, 786.864, 20.],
     [228.31999, 104.71999, 307.40802, 791.456, 10.], [145.61601, 94.736, 246.288, 786.86395, 60.],
     [68.32, 101.12, 244.496, 787.872, 70.], [0., 0., 0., 0., 0.], [0., 0., 0., 0., 0.], [0., 0., 0., 0., 0.],
     [0., 0., 0., 0., 0.], [0., 0., 0., 0., 0.]],
], dtype="float32")


class DummyLoader:
    def __init__(self, batch_size=2, output_size=(800, 1344), seed=None):
        self.batch_size = batch_size
        self.output_size = output_size
        self.anno = _ANNO.copy()
        self.anno *= min(output_size[0] / 800, output_size[1] / 800)          # dummy.py:40-41
        self.im_info = np.array([[*output_size, 612., 612., 10.], [*output_size, 500., 375., 5.]], dtype="float32")
        # the reference draws unseeded float64 noise (dummy.py:60); a seed makes parity runs reproducible
        self._rng = np.random.default_rng(seed)

    def __iter__(self):
        return self

    def _tile(self, x):
        repeat = self.batch_size // len(self.anno)                           # dummy.py:51-57
        remain = self.batch_size % len(self.anno)
        return np.concatenate([np.repeat(x, repeat, axis=0), x[:remain, ...]], axis=0)

    def __next__(self):
        return {
            "data": self._rng.random(size=(self.batch_size, 3, *self.output_size)),
            "gt_boxes": self._tile(self.anno),
            "im_info": self._tile(self.im_info),
        }
