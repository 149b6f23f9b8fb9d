"""Constants of the MobilePoser inference path.

Restated (not imported) from the reference's class-attribute config so the package is
self-contained on the GPU box.  Reference: mobileposer/config.py:40-54 (model_config),
:57-83 (amass), :86-126 (datasets), :129-142 (joint_set).
"""


class paths:
    """config.py:26-38: locations relative to the working directory, exactly as the reference resolves them
    (``Path().absolute()`` at import time).  Only the entries the inference / evaluation path reads."""
    from pathlib import Path as _P
    root_dir = _P().absolute()
    checkpoint = root_dir / "checkpoints"
    smpl_file = root_dir / "smpl/basicmodel_m.pkl"
    weights_file = root_dir / "checkpoints/weights.pth"
    eval_dir = root_dir / "data/processed_datasets/eval"
    processed_datasets = root_dir / "data/processed_datasets"


class model_config:
    n_joints = 5                 # IMU locations (lw, rw, lp, rp, head)      config.py:46
    n_imu = 12 * n_joints        # 5 x (3 acc + 9 ori) = 60                  config.py:47
    n_output_joints = 24         # SMPL joints                               config.py:48
    past_frames = 40             # online window: past                       config.py:52
    future_frames = 5            # online window: future                     config.py:53
    total_frames = past_frames + future_frames


class amass:
    # device-location combinations (indices into the 5 IMU slots)           config.py:60-73
    combos = {
        'lw_rp_h': [0, 3, 4], 'rw_rp_h': [1, 3, 4], 'lw_lp_h': [0, 2, 4], 'rw_lp_h': [1, 2, 4],
        'lw_lp': [0, 2], 'lw_rp': [0, 3], 'rw_lp': [1, 2], 'rw_rp': [1, 3],
        'lp_h': [2, 4], 'rp_h': [3, 4], 'lp': [2], 'rp': [3],
    }
    acc_scale = 30               # config.py:74
    vel_scale = 2                # config.py:75


class datasets:
    fps = 30                     # config.py:89
    window_length = 125          # config.py:126
    test_datasets = {'dip': 'dip_test.pt', 'totalcapture': 'totalcapture.pt', 'imuposer': 'imuposer_test.pt'}


class joint_set:
    gravity_velocity = -0.018    # config.py:131
    full = list(range(24))
    reduced = [0, 1, 2, 3, 4, 5, 6, 9, 12, 13, 14, 15, 16, 17, 18, 19]      # config.py:134
    ignored = [0, 7, 8, 10, 11, 20, 21, 22, 23]                              # config.py:135
    n_full, n_ignored, n_reduced = 24, 9, 16


class train_hypers:
    batch_size = 256             # config.py:8  (the "256 x 125" benchmark shape)


# SMPL kinematic tree (kintree_table[0] of basicmodel_m.pkl, parent[0] -> -1);
# reference: articulate/model.py:36-37.
SMPL_PARENT = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]

# foot-contact weight thresholds, net.py:53
PROB_THRESHOLD = (0.5, 0.9)
# frames-per-second / vel_scale: the divisor applied to the network root velocity, net.py:141,196
VEL_DIVISOR = datasets.fps / amass.vel_scale      # 15.0
LFOOT, RFOOT = 10, 11
