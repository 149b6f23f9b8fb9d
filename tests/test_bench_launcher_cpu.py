"""bench.py's own multi-rank launcher, end to end on CPU: `python bench.py --gpus 2 --dry-run` must start two ranks by itself
(torch.distributed.run, rendezvous on 127.0.0.1), run the broadcast / shard / barrier / max-over-ranks / gather path over gloo
and print ONE JSON line whose n_gpus and n_ranks_seen say 2.  (--dry-run replaces only the step: no GPU here.)"""
import json
import os
import subprocess
import sys

from conftest import REPO


def _run(argv, env_extra=None, timeout=300):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + argv, cwd=REPO, env=env,
                          capture_output=True, text=True, timeout=timeout)


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_bench_gpus_2_launches_two_ranks_by_itself():
    r = _run(["--gpus", "2", "--steps", "5", "--warmup", "1", "--dry-run"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["dry_run"] is True
    assert out["weights_broadcast_ok"] is True
    assert out["steps"] == 5 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 512 and out["config"]["batch_per_gpu"] == 256
    assert [p["rank"] for p in out["per_rank"]] == [0, 1]
    assert all(p["frames"] == 256 * 125 * 5 for p in out["per_rank"])
    # value = whole-job frames / max-over-ranks seconds
    assert abs(out["value"] - 512 * 125 * 5 / (out["ms_per_step"] * 5e-3)) < 1e-3 * out["value"]
    assert out["ms_per_step"] * 5e-3 >= max(p["seconds"] for p in out["per_rank"]) - 1e-4


def test_bench_strong_scaling_shards_the_global_batch():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "0", "--dry-run", "--scaling", "strong", "--global-batch", "1023"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = _json_line(r.stdout)
    assert out["n_ranks_seen"] == 2 and out["scaling"] == "strong" and out["config"]["global_batch"] == 1023
    assert [p["frames"] for p in out["per_rank"]] == [512 * 125 * 3, 511 * 125 * 3]
    assert "global batch" in out["metric"]


def test_bench_single_rank_line_has_the_same_fields():
    r = _run(["--steps", "3", "--warmup", "0", "--dry-run"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 1 and out["n_ranks_seen"] == 1 and "per_rank" not in out


def test_bench_refuses_a_world_that_is_not_gpus():
    r = _run(["--gpus", "2", "--dry-run"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
    r = _run(["--gpus", "1", "--dry-run"], env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0",
                                                     "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_bench_verify_rows_accepts_the_oracle_and_rejects_a_wrong_output():
    """bench.py compares 16 rows of the step it timed with the oracle (verify_rows, round 5).  The check itself, on CPU: the
    oracle's own outputs pass with zero error; a 2e-4 offset on one velocity row or a 2 mm offset on one translation row fails
    the run; rows outside the sample do not matter."""
    import numpy as np
    import pytest
    import torch
    sys.path.insert(0, REPO)
    import bench
    from mobileposer_amd import synthetic
    from oracle import mp_oracle as O
    B, T = 8, 12
    imu = synthetic.make_imu(B, T, seed=3)
    ref = O.OracleNet(synthetic.make_weights(0), synthetic.synthetic_smpl()["J"])
    pose, joints, vel, contact = ref.forward(imu, [T] * B)
    vel = vel.reshape(B, T, 72)
    rg, jg = O.forward_kinematics(pose, ref.J)
    tran = np.stack([O.translate_offline(joints[b].reshape(T, 24, 3), vel[b], contact[b], ref.floor_y) for b in range(B)])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    outs = {"joints": t(joints), "vel": t(vel), "contact": t(contact), "tran": t(tran), "pose": t(pose), "rglob": t(rg), "jglob": t(jg)}
    ok = bench.verify_rows(outs, t(imu), B, T, rows=4, seed=1)
    assert ok["rows"] == 4 and ok["of"] == B and max(ok["max_err"].values()) < 1e-5   # (the oracle on 4 rows vs on 8: BLAS blocking)
    pick = np.sort(np.random.Generator(np.random.PCG64(1)).choice(B, 4, replace=False))
    other = [b for b in range(B) if b not in pick][0]
    bad = dict(outs, vel=outs["vel"].clone())
    bad["vel"][other] += 1.0                                   # not sampled: passes
    bench.verify_rows(bad, t(imu), B, T, rows=4, seed=1)
    bad["vel"][int(pick[0]), 3, 5] += 2e-4
    with pytest.raises(RuntimeError, match="differ from the oracle"):
        bench.verify_rows(bad, t(imu), B, T, rows=4, seed=1)
    bad = dict(outs, tran=outs["tran"].clone())
    bad["tran"][int(pick[1]), 0, 1] += 2e-3
    with pytest.raises(RuntimeError, match="tran_m"):
        bench.verify_rows(bad, t(imu), B, T, rows=4, seed=1)


def test_bench_gpus_8_dry_run_weak_and_strong():
    """The shape of the first real 8-GPU run (BASELINE configs[3] / the scaling curve), on CPU: 8 gloo ranks started by bench.py
    itself, weak (256 sequences per rank) and strong (1024 split into 128 per rank) -- n_ranks_seen == 8, one line."""
    r = _run(["--gpus", "8", "--steps", "3", "--warmup", "1", "--dry-run"], timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 8 and out["n_ranks_seen"] == 8 and out["weights_broadcast_ok"] is True
    assert out["config"]["global_batch"] == 2048 and [p["rank"] for p in out["per_rank"]] == list(range(8))
    r = _run(["--gpus", "8", "--steps", "3", "--warmup", "1", "--dry-run", "--scaling", "strong"], timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = _json_line(r.stdout)
    assert out["n_ranks_seen"] == 8 and out["scaling"] == "strong" and out["config"]["global_batch"] == 1024
    assert [p["frames"] for p in out["per_rank"]] == [128 * 125 * 3] * 8 and out["config"]["batch_per_gpu"] == 128


def test_bench_prints_one_error_line_when_a_rank_dies_or_hangs():
    """A rank that dies (exit in front of the timed region) or hangs (sleeps for ever) must not take the result line with it
    or hang the job: ONE JSON line with value null and an error text, non-zero exit status, within --timeout."""
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "0", "--dry-run"], env_extra={"MP_BENCH_TEST_KILL_RANK": "1"}, timeout=300)
    assert r.returncode != 0
    out = _json_line(r.stdout)
    assert out["value"] is None and out["n_gpus"] == 2 and "error" in out
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "0", "--dry-run", "--timeout", "20"],
             env_extra={"MP_BENCH_TEST_HANG_RANK": "1"}, timeout=300)
    assert r.returncode != 0
    out = _json_line(r.stdout)
    assert out["value"] is None and ("--timeout" in out["error"] or "terminated" in out["error"])   # (whichever rank's timer fires first)
