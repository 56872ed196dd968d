"""Aspect-ratio grouped batch sampling (behaviour of basedet's GroupedRandomSampler / AspectRatioGroupSampler, group_sampler.py):
every batch holds images of ONE aspect-ratio group, so padding to the per-batch maximum wastes little and the device sees few
distinct (H, W) plans.

Behaviour kept: one seeded permutation of the indices per pass; with world_size > 1 the permutation is padded (by wrapping around)
to a multiple of world_size and rank r takes every world_size-th entry; an index joins its group's queue and a batch is emitted the
moment a queue holds batch_size entries; queues that did not fill up carry over into the next pass; the length is undefined.
(The reference inherits permutation and scatter from megengine.data.RandomSampler, which is not vendored.)"""
import numpy as np

__all__ = ["GroupedRandomSampler", "AspectRatioGroupSampler"]


class GroupedRandomSampler:
    def __init__(self, dataset, batch_size, group_ids, indices=None, world_size=None, rank=None, seed=None):
        n = len(dataset)
        self.group_ids = np.asarray(group_ids)
        assert self.group_ids.shape[0] == n
        self.batch_size = int(batch_size)
        self.indices = np.arange(n) if indices is None else np.asarray(list(indices))
        self.world_size = int(world_size or 1)
        self.rank = int(rank or 0)
        self.rng = np.random.RandomState(seed or 0)
        self._pending = {}              # group id -> indices waiting for their batch to fill (survives across passes)

    def sample(self):
        return self.rng.permutation(self.indices)

    def scatter(self, order):
        order = np.asarray(order)
        short = -len(order) % self.world_size
        if short:
            order = np.concatenate([order, order[:short]])
        return order[self.rank::self.world_size]

    def batch(self):
        order = self.sample()
        if self.world_size > 1:
            order = self.scatter(order)
        batches = []
        for idx, gid in zip(order.tolist(), self.group_ids[order].tolist()):
            queue = self._pending.setdefault(gid, [])
            queue.append(idx)
            if len(queue) >= self.batch_size:
                batches.append(queue)
                self._pending[gid] = []
        return iter(batches)

    __iter__ = batch

    def __len__(self):
        raise NotImplementedError("length of GroupedRandomSampler is not well-defined.")


class AspectRatioGroupSampler(GroupedRandomSampler):
    """Groups by height / width against the sorted ``aspect_grouping`` thresholds: a ratio equal to a threshold falls into the
    upper group (bisect-right rule)."""

    def __init__(self, dataset, batch_size, aspect_grouping=(1,), *args, **kwargs):
        infos = (dataset.get_img_info(i) for i in range(len(dataset)))
        ratios = np.fromiter((info["height"] / info["width"] for info in infos), dtype=np.float64, count=len(dataset))
        group_ids = np.searchsorted(np.sort(np.asarray(aspect_grouping, np.float64)), ratios, side="right")
        super().__init__(dataset, batch_size, group_ids, *args, **kwargs)
