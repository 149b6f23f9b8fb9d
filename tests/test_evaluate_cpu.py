"""The translation-window statistics of evaluate_pose (evaluate.py:66-91,106-107) against the reference's own output
(golden G11: its evaluate_pose driven with canned predictions).  CPU only -- the error table of the same golden needs
the GPU forward kinematics and is checked in tests/test_gpu_parity.py."""
import numpy as np
import torch

from conftest import load_golden
from mobileposer_amd.evaluate import distance_window_pairs, translation_window_errors


def test_g11_translation_window_errors():
    g = load_golden("g11_evaluate.npz")
    per_window = {w: [] for w in range(1, 8)}
    for k in range(int(g["n_seq"])):
        for w, v in translation_window_errors(g[f"s{k}_tran_p"], g[f"s{k}_tran_t"]).items():
            per_window[w].append(v)
    got = [0.0] + [float(torch.tensor(v).mean()) if v else float("nan") for v in per_window.values()]
    np.testing.assert_allclose(got, g["tran_errors"], rtol=1e-5, atol=1e-7)
    assert sum(len(v) for v in per_window.values()) > 7          # several sequences contribute


def test_distance_window_pairs_matches_the_two_pointer_walk():
    """Same pairs as the reference's while-loop on adversarial inputs: plateaus, exact hits, windows never reached."""
    def walk(d, w):                                               # independent re-statement used only as a cross-check
        pairs, s, e = [], 0, 1
        while e < len(d):
            if np.float32(d[e] - d[s]) < np.float32(w):
                e += 1
            else:
                if not pairs or pairs[-1][1] != e:
                    pairs.append((s, e))
                s += 1
        return pairs
    rng = np.random.default_rng(3)
    for trial in range(20):
        steps = rng.choice([0.0, 0.25, 0.5, 1.0, 0.37], size=rng.integers(1, 60)).astype(np.float32)
        d = np.concatenate(([0.0], np.cumsum(steps))).astype(np.float32)
        for w in (1, 2, 7):
            assert distance_window_pairs(d, w) == walk(d, w), (trial, w)
    assert distance_window_pairs(np.zeros(1, np.float32), 1) == [] and distance_window_pairs(np.zeros(9, np.float32), 1) == []
