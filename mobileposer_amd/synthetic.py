"""Deterministic synthetic weights, SMPL constants and IMU streams.

The reference ships neither pretrained weights nor the SMPL model file (.MISSING_LARGE_BLOBS:1,
config.py:30-31) and there is no network, so tests and the benchmark run on recipes that both
this container (where the reference can be imported to make golden vectors) and the GPU box can
regenerate bit-identically from a seed with numpy's PCG64 generator -- no torch RNG involved.
"""
import numpy as np

from .config import SMPL_PARENT, amass
from .manifest import state_dict_manifest


def _random_rotations(rng, n):
    """n random SO(3) matrices (QR of a Gaussian, det forced to +1), float64."""
    a = rng.standard_normal((n, 3, 3))
    q, r = np.linalg.qr(a)
    q = q * np.sign(np.diagonal(r, axis1=1, axis2=2))[:, None, :]
    det = np.linalg.det(q)
    q[:, :, 2] *= det[:, None]
    return q


def make_weights(seed=0, profile="init"):
    """Seeded state dict (key -> float32 ndarray) with PyTorch-like fan-in uniform init.

    ``profile="trained"`` (round 4) rescales the SAME draw into the regime a trained net works in -- what the reference's
    evaluate.py feeds its 12 sensor combos through (combine_weights.py:42-57): LSTM ``weight_ih`` / ``weight_hh`` x 3
    (recurrent gain > 1, gates that saturate, hidden activations near +-1), forget-gate biases + 1.0 (bias_ih slice
    [H:2H]: long memory, rounding differences are carried over many steps instead of being forgotten), ``linear1``
    weights x 2.  The output conditioning tweaks below are the same for both profiles.

    Tweaks that make the random net behave like a trained one where the path is sensitive to it:
      * pose linear2.bias  = 6D of random rotations, so r6d outputs are well conditioned
        (trained posers emit near-orthonormal 6D; a near-zero 6D blows up Gram-Schmidt),
      * foot-contact linear2 is scaled up and biased so logits change sign / arg-max over time and
        the contact weight (net.py:90-91,144,197) leaves its clamp interval on both sides.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for key, shape in state_dict_manifest().items():
        if ".rnn." in key:
            hidden = shape[0] // 4
            k = 1.0 / np.sqrt(hidden)
        elif key.endswith("weight"):
            k = 1.0 / np.sqrt(shape[1])
        else:  # linear bias: fan-in of its weight
            wshape = state_dict_manifest()[key[:-4] + "weight"]
            k = 1.0 / np.sqrt(wshape[1])
        sd[key] = rng.uniform(-k, k, size=shape).astype(np.float32)
    if profile == "trained":
        for key in sd:
            if ".rnn.weight_" in key:
                sd[key] = (sd[key] * np.float32(3.0)).astype(np.float32)
            elif ".rnn.bias_ih" in key:
                hidden = sd[key].shape[0] // 4
                sd[key][hidden:2 * hidden] += np.float32(1.0)
            elif key.endswith("linear1.weight"):
                sd[key] = (sd[key] * np.float32(2.0)).astype(np.float32)
    elif profile != "init":
        raise ValueError("profile must be 'init' or 'trained'")
    rot = _random_rotations(rng, 16)
    # 6D = first two columns of R (articulate/math/angular.py:180,192)
    sd["pose.pose.linear2.bias"] = np.ascontiguousarray(
        rot[:, :, :2].transpose(0, 2, 1).reshape(96)).astype(np.float32)
    sd["foot_contact.footcontact.linear2.weight"] = (sd["foot_contact.footcontact.linear2.weight"] * 6.0).astype(np.float32)
    sd["foot_contact.footcontact.linear2.bias"] = np.array([0.80, 0.72], dtype=np.float32)
    return sd


# Plausible SMPL rest-pose joint positions (metres, y up); only parent[] and J[10:12] matter on the
# hot path (net.py:47-49,132), J as a whole for forward kinematics (articulate/model.py:228-231).
_J = np.array([
    [0.00, 0.00, 0.00], [0.07, -0.09, 0.00], [-0.07, -0.09, 0.00], [0.00, 0.11, -0.02],
    [0.10, -0.47, 0.00], [-0.10, -0.47, 0.00], [0.00, 0.24, 0.00], [0.09, -0.87, -0.03],
    [-0.09, -0.87, -0.03], [0.00, 0.30, 0.02], [0.11, -0.93, 0.09], [-0.11, -0.92, 0.09],
    [0.00, 0.51, -0.01], [0.08, 0.42, 0.00], [-0.08, 0.42, 0.00], [0.00, 0.60, 0.03],
    [0.17, 0.45, -0.01], [-0.17, 0.45, -0.01], [0.43, 0.44, -0.03], [-0.43, 0.44, -0.03],
    [0.68, 0.45, -0.03], [-0.68, 0.45, -0.03], [0.76, 0.44, -0.04], [-0.76, 0.44, -0.04],
], dtype=np.float32)


def synthetic_smpl(n_vertex=96, seed=7):
    """Dict with the keys articulate/model.py:28-38 reads, on a toy mesh of ``n_vertex`` vertices.

    The pelvis is offset from the origin so that ``J - J[0]`` (model.py:87) is exercised.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    J = (_J + np.array([0.01, -0.24, 0.03], dtype=np.float32)).astype(np.float32)
    owner = np.arange(n_vertex) % 24
    v_template = (J[owner] + 0.05 * rng.standard_normal((n_vertex, 3))).astype(np.float32)
    w = rng.uniform(0.0, 1.0, size=(n_vertex, 24)).astype(np.float32) ** 6
    w[np.arange(n_vertex), owner] += 1.0
    w /= w.sum(axis=1, keepdims=True)
    jreg = np.zeros((24, n_vertex), dtype=np.float32)
    for j in range(24):
        jreg[j, owner == j] = 1.0 / max(1, int((owner == j).sum()))
    faces = np.stack([np.arange(n_vertex - 2), np.arange(1, n_vertex - 1), np.arange(2, n_vertex)], axis=1)
    kintree = np.stack([np.array([2 ** 32 - 1] + SMPL_PARENT[1:], dtype=np.int64), np.arange(24, dtype=np.int64)])
    return {
        "J_regressor": jreg, "weights": w.astype(np.float32),
        "posedirs": (0.001 * rng.standard_normal((n_vertex, 3, 207))).astype(np.float32),
        "shapedirs": (0.01 * rng.standard_normal((n_vertex, 3, 10))).astype(np.float32),
        "v_template": v_template, "J": J, "f": faces.astype(np.int64), "kintree_table": kintree,
    }


def make_imu(batch, frames, seed=1, combo="lw_rp", smooth=0.95):
    """Synthetic IMU windows [batch, frames, 60] float32 shaped like data.py:69-76 produces.

    acc block [.., :15]  ~ N(0, 0.3^2) (what a/30 looks like), AR(1)-smoothed in time;
    ori block [.., 15:60] = 5 rotation matrices row-major, a random walk on SO(3);
    devices not in ``combo`` are zeroed (data.py:72-76).  ``combo`` is one name of config.py:60-73 for the whole batch or a
    list of ``batch`` names (row k keeps the devices of combo[k]) -- evaluate.py:56 feeds all 12.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    acc = np.empty((batch, frames, 5, 3), dtype=np.float64)
    e = rng.standard_normal((batch, frames, 5, 3)) * 0.3
    acc[:, 0] = e[:, 0]
    s = np.sqrt(1.0 - smooth * smooth)
    for t in range(1, frames):
        acc[:, t] = smooth * acc[:, t - 1] + s * e[:, t]
    ori = np.empty((batch, frames, 5, 3, 3), dtype=np.float64)
    ori[:, 0] = _random_rotations(rng, batch * 5).reshape(batch, 5, 3, 3)
    for t in range(1, frames):
        w = rng.standard_normal((batch, 5, 3)) * 0.06          # small axis-angle step
        th = np.linalg.norm(w, axis=-1, keepdims=True) + 1e-12
        k = w / th
        K = np.zeros((batch, 5, 3, 3))
        K[..., 0, 1], K[..., 0, 2] = -k[..., 2], k[..., 1]
        K[..., 1, 0], K[..., 1, 2] = k[..., 2], -k[..., 0]
        K[..., 2, 0], K[..., 2, 1] = -k[..., 1], k[..., 0]
        th = th[..., None]
        dR = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)
        ori[:, t] = ori[:, t - 1] @ dR
    names = [combo] * batch if isinstance(combo, str) else list(combo)
    if len(names) != batch:
        raise ValueError("need one combo name or %d of them" % batch)
    for b, name in enumerate(names):
        keep = np.zeros(5, dtype=bool)
        keep[amass.combos[name]] = True
        acc[b][:, ~keep] = 0.0
        ori[b][:, ~keep] = 0.0
    imu = np.concatenate([acc.reshape(batch, frames, 15), ori.reshape(batch, frames, 45)], axis=-1)
    return np.ascontiguousarray(imu.astype(np.float32))
