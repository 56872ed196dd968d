"""Checkpoint loading with the reference's name-matching rules (basedet/utils/checkpoint.py:13-140).

The reference reads checkpoints with ``megengine.load`` (a pickle of numpy arrays for the published backbone / model-zoo
files).  MegEngine is not a dependency here: ``.pkl`` files are read with a restricted unpickler that only reconstructs
numpy arrays and plain containers, ``.npz`` files with numpy.  Nothing on this path touches the GPU kernels except the
final re-bind of the model (``FPNDetector._bind_params``)."""
import io
import pickle

import numpy as np

__all__ = ["load_matched_weights", "unwarp_ckpt", "full_match", "load_checkpoint_file", "save_checkpoint"]


class _NumpyOnlyUnpickler(pickle.Unpickler):
    _ALLOWED = {
        ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
        ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
        ("numpy", "ndarray"), ("numpy", "dtype"), ("collections", "OrderedDict"),
        ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer"),
    }

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"checkpoint contains a {module}.{name} object; only numpy arrays and containers are accepted")


def load_checkpoint_file(path):
    if str(path).endswith(".npz"):
        with np.load(path) as z:
            return {k: z[k] for k in z.files}
    with open(path, "rb") as f:
        return _NumpyOnlyUnpickler(io.BytesIO(f.read())).load()


def save_checkpoint(path, state_dict, **extra):
    """{"model": {"state_dict": ...}, **extra} like the reference's CheckpointHook payload, numpy arrays only."""
    payload = dict(extra, model={"state_dict": {k: np.asarray(v) for k, v in state_dict.items()}})
    with open(path, "wb") as f:
        pickle.dump(payload, f, protocol=4)


def unwarp_ckpt(weights, model_key="model"):
    """checkpoint.py:32-38."""
    if model_key in weights:
        weights = weights[model_key]
    if "state_dict" in weights:
        weights = weights["state_dict"]
    return weights


def _name_matched(key, keys):
    return {k for k in keys if key == k or k.endswith(key)}


def full_match(weights, self_key_shape):
    """checkpoint.py:41-90: exact name first, then unique suffix match, then the element count breaks ties.
    Returns ({model key: checkpoint key}, unused checkpoint keys)."""
    self_key_shape = dict(self_key_shape)
    mapping, pending, unused = {}, {}, []
    for w_key, w_val in weights.items():
        cands = _name_matched(w_key, self_key_shape.keys())
        if not cands:
            unused.append(w_key)
            continue
        if w_key in cands:
            hit = w_key
        elif len(cands) == 1:
            hit = next(iter(cands))
        else:
            same = {k for k in cands if int(np.prod(self_key_shape[k])) == int(np.prod(np.shape(w_val)))}
            if len(same) == 1:
                hit = next(iter(same))
            else:
                pending[w_key] = same
                continue
        for v in pending.values():
            v.discard(hit)
        self_key_shape.pop(hit)
        mapping[hit] = w_key
    for w_key, cands in pending.items():
        assert len(cands) == 1, f"{w_key} matched more than 1 keys: {sorted(cands)}"
        mapping[cands.pop()] = w_key
    return mapping, unused


def load_matched_weights(model, weights, strict=False):
    """checkpoint.py:96-140.  Same-size / different-shape tensors are reshaped (MegEngine dumps BatchNorm vectors as
    (1, C, 1, 1)); other shape mismatches are skipped unless ``strict``."""
    if weights is None:
        return model
    if not isinstance(weights, dict):
        weights = load_checkpoint_file(weights)
    weights = unwarp_ckpt(weights)
    state = model.state_dict()
    mapping, _ = full_match(weights, {k: v.shape for k, v in state.items()})
    for k, src in mapping.items():
        val = np.asarray(weights[src], np.float32)
        if val.shape != state[k].shape:
            if val.size == state[k].size:
                val = val.reshape(state[k].shape)
            elif strict:
                raise ValueError(f"param `{k}` size mismatch, get {state[k].shape} vs {val.shape}")
            else:
                continue
        state[k] = val
    model._bind_params(state)
    return model
