"""load_model: mirror of utils/model_utils.py:6-15 (state_dict ``.pth`` -> MobilePoserNet on the GPU)."""
import numpy as np

from .manifest import state_dict_manifest


def state_dict_to_blob(sd):
    """Flatten a state dict (torch tensors or ndarrays) into the fp32 blob mp_create expects."""
    parts = []
    man = state_dict_manifest()
    missing = [k for k in man if k not in sd]
    if missing:
        raise KeyError("state dict lacks %d keys, e.g. %s" % (len(missing), missing[:3]))
    for key, shape in man.items():
        v = sd[key]
        if hasattr(v, "detach"):
            v = v.detach().cpu().numpy()
        v = np.asarray(v, dtype=np.float32)
        if tuple(v.shape) != tuple(shape):
            raise ValueError("%s has shape %s, expected %s" % (key, tuple(v.shape), tuple(shape)))
        parts.append(v.reshape(-1))
    return np.ascontiguousarray(np.concatenate(parts))


def blob_to_state_dict(blob):
    out, off = {}, 0
    for key, shape in state_dict_manifest().items():
        n = int(np.prod(shape))
        out[key] = np.asarray(blob[off:off + n], dtype=np.float32).reshape(shape).copy()
        off += n
    return out


def normalise_state_dict(obj):
    """What ``load_model`` accepts (utils/model_utils.py:6-15): the plain state dict of the 72 tensors written by
    combine_weights.py:53-56, or a Lightning checkpoint (``MobilePoserNet.load_from_checkpoint``: the weights sit under
    'state_dict', possibly behind a common wrapper prefix such as 'model.' / 'module.').  Extra entries are ignored
    the way a non-strict load would; a missing weight raises."""
    sd = obj
    if isinstance(sd, dict) and "state_dict" in sd and not hasattr(sd["state_dict"], "shape"):
        sd = sd["state_dict"]
    if not isinstance(sd, dict):
        raise TypeError("expected a state dict or a checkpoint dict, got %s" % type(obj).__name__)
    man = state_dict_manifest()
    first = next(iter(man))
    if first not in sd:
        hits = [k for k in sd if isinstance(k, str) and k.endswith(first)]
        if len(hits) == 1:
            prefix = hits[0][:-len(first)]
            sd = {k[len(prefix):]: v for k, v in sd.items() if isinstance(k, str) and k.startswith(prefix)}
    return {k: sd[k] for k in man if k in sd} if all(k in sd for k in man) else sd


class UntrustedFileError(RuntimeError):
    """A file needs full unpickling and the caller has not declared it trusted.  A class of its own so that loaders which
    tolerate broken files (PoseDataset: "a bad file is reported, not fatal", data.py:50-54) do not swallow it."""


def safe_torch_load(path, trusted=False):
    """``torch.load`` restricted to tensors and plain containers (``weights_only=True``).  A file the safe loader refuses
    -- e.g. a Lightning checkpoint whose hyper-parameters pickle arbitrary classes -- is unpickled in full only when the
    caller says the file is trusted (``trusted=True`` or env ``MP_TRUSTED_CHECKPOINTS=1``): a silent fallback would make
    the safe attempt pointless."""
    import os
    import pickle
    import torch
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except (pickle.UnpicklingError, AttributeError, ImportError, ModuleNotFoundError, TypeError) as e:
        # (everything the restricted unpickler raises on content it does not allow: UnpicklingError for unknown globals,
        #  the others when a pickled class or its module cannot be resolved / rebuilt)
        if not (trusted or os.environ.get("MP_TRUSTED_CHECKPOINTS", "") not in ("", "0")):
            raise UntrustedFileError("%s holds pickled objects beyond tensors and plain containers (%s); pass trusted=True (or set "
                               "MP_TRUSTED_CHECKPOINTS=1) to unpickle it in full -- only for files you trust"
                               % (path, str(e).splitlines()[0])) from e
        return torch.load(path, map_location="cpu", weights_only=False)


def load_model(model_path, smpl_file=None, device="cuda:0", smpl=None, trusted=False):
    """utils/model_utils.py:6-15: ``load_model(path) -> MobilePoserNet`` with the weights of ``path`` (a ``torch.save``d
    state dict, or a Lightning checkpoint as the reference's fallback branch reads).  ``smpl_file`` defaults to
    ``config.paths.smpl_file`` when that file exists (the reference always reads it, net.py:37), else the synthetic body."""
    import os
    from .config import paths
    from .net import MobilePoserNet
    obj = safe_torch_load(model_path, trusted)
    if smpl is None and smpl_file is None and os.path.exists(str(paths.smpl_file)):
        smpl_file = str(paths.smpl_file)
    model = MobilePoserNet(smpl_file=smpl_file, smpl=smpl, device=device)
    model.load_state_dict(normalise_state_dict(obj))
    return model
