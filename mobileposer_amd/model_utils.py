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


def load_model(model_path, smpl_file=None, device="cuda:0"):
    """utils/model_utils.py:6-15.  ``model_path``: a ``torch.save``d state dict of the 72 tensors
    (combine_weights.py:53-56) or a Lightning checkpoint holding it under 'state_dict'."""
    import torch
    from .net import MobilePoserNet
    sd = torch.load(model_path, map_location="cpu")
    if isinstance(sd, dict) and "state_dict" in sd:
        sd = sd["state_dict"]
    model = MobilePoserNet(smpl_file=smpl_file, device=device)
    model.load_state_dict(sd)
    return model
