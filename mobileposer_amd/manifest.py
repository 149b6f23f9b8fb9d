"""State-dict manifest of MobilePoserNet: the 72 tensors a ``weights.pth`` holds.

Reference: the key names are what ``MobilePoserNet().state_dict()`` yields for the modules built at
models/net.py:40-43 from models/rnn.py:13-18 (``rnn`` = nn.LSTM, ``linear1``, ``linear2``), written by
combine_weights.py:53-56.  LSTM gate order inside every [4H, .] matrix is PyTorch's i, f, g, o.
"""
from collections import OrderedDict

# (facade attribute, state-dict prefix, n_in, n_out, hidden, bidirectional)
MODULES = (
    ("pose",         "pose.pose.",                 132, 96, 256, True),    # models/poser.py:32
    ("joints",       "joints.joints.",              60, 72, 256, True),    # models/joints.py:29
    ("foot_contact", "foot_contact.footcontact.",  132,  2,  64, True),    # models/footcontact.py:28
    ("velocity",     "velocity.vel.",              132, 72, 256, False),   # models/velocity.py:29
)
MODULE_INDEX = {"joints": 0, "pose": 1, "foot_contact": 2, "velocity": 3}   # order used by the C-ABI


def rnn_keys(n_in, n_out, hidden, bidirectional, n_layers=2):
    """Ordered (suffix, shape) list of one RNN block (nn.LSTM registers its params first, rnn.py:15-17)."""
    dirs = 2 if bidirectional else 1
    out = []
    for layer in range(n_layers):
        in_l = hidden if layer == 0 else hidden * dirs
        for sfx in ([""] if not bidirectional else ["", "_reverse"]):
            out.append((f"rnn.weight_ih_l{layer}{sfx}", (4 * hidden, in_l)))
            out.append((f"rnn.weight_hh_l{layer}{sfx}", (4 * hidden, hidden)))
            out.append((f"rnn.bias_ih_l{layer}{sfx}", (4 * hidden,)))
            out.append((f"rnn.bias_hh_l{layer}{sfx}", (4 * hidden,)))
    out.append(("linear1.weight", (hidden, n_in)))
    out.append(("linear1.bias", (hidden,)))
    out.append(("linear2.weight", (n_out, hidden * dirs)))
    out.append(("linear2.bias", (n_out,)))
    return out


def state_dict_manifest():
    """OrderedDict key -> shape for the full net, in the reference's registration order."""
    m = OrderedDict()
    for _, prefix, n_in, n_out, hidden, bi in MODULES:
        for sfx, shape in rnn_keys(n_in, n_out, hidden, bi):
            m[prefix + sfx] = shape
    return m


def n_params():
    n = 0
    for shape in state_dict_manifest().values():
        k = 1
        for s in shape:
            k *= s
        n += k
    return n
