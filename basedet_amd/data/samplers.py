"""GroupedRandomSampler / AspectRatioGroupSampler (basedet/data/samplers/group_sampler.py:8-93): every batch holds samples of one
aspect-ratio group, so that padding to the per-batch maximum wastes little and the device sees few distinct (H, W) plans.

The reference inherits the permutation and the rank scatter from ``megengine.data.RandomSampler`` (un-vendored); restated here as:
a seeded permutation of the indices per epoch, padded to a multiple of world_size, rank r taking every world_size-th index."""
import bisect

import numpy as np

__all__ = ["GroupedRandomSampler", "AspectRatioGroupSampler"]


class GroupedRandomSampler:
    def __init__(self, dataset, batch_size, group_ids, indices=None, world_size=None, rank=None, seed=None):
        self.batch_size = batch_size
        self.indices = list(range(len(dataset))) if indices is None else list(indices)
        self.world_size = 1 if world_size is None else int(world_size)
        self.rank = 0 if rank is None else int(rank)
        self.rng = np.random.RandomState(0 if seed is None else seed)
        self.group_ids = group_ids
        assert len(group_ids) == len(dataset)
        # buffer the indices of each group until batch size is reached (group_sampler.py:36-39)
        self.buffer_per_group = {k: [] for k in np.unique(self.group_ids).tolist()}

    def sample(self):
        return self.rng.permutation(self.indices).tolist()

    def scatter(self, indices):
        total = (len(indices) + self.world_size - 1) // self.world_size * self.world_size
        indices = indices + indices[: total - len(indices)]
        return indices[self.rank: total: self.world_size]

    def batch(self):
        """group_sampler.py:41-55."""
        indices = list(self.sample())
        if self.world_size > 1:
            indices = self.scatter(indices)
        batch_index = []
        for ind in indices:
            group_id = self.group_ids[ind]
            group_buffer = self.buffer_per_group[group_id]
            group_buffer.append(ind)
            if len(group_buffer) == self.batch_size:
                batch_index.append(group_buffer)
                self.buffer_per_group[group_id] = []
        return iter(batch_index)

    def __iter__(self):
        return self.batch()

    def __len__(self):
        raise NotImplementedError("length of GroupedRandomSampler is not well-defined.")


class AspectRatioGroupSampler(GroupedRandomSampler):
    """group_sampler.py:60-93: group id = bisect_right(sorted(aspect_grouping), height / width)."""

    def __init__(self, dataset, batch_size, aspect_grouping=(1,), *args, **kwargs):
        aspect_ratios = []
        for i in range(len(dataset)):
            info = dataset.get_img_info(i)
            aspect_ratios.append(info["height"] / info["width"])
        bins = sorted(aspect_grouping)
        group_ids = [bisect.bisect_right(bins, r) for r in aspect_ratios]
        super().__init__(dataset, batch_size, group_ids, *args, **kwargs)
