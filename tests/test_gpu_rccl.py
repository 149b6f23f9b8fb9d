"""The RCCL branch of bench.py on the hardware there is: ONE rank under torch.distributed.run on the leased MI355X
(no 8-GPU node was available to any round so far).  What this exercises on a real GPU, that the gloo dry
run on CPU cannot: init_process_group("nccl") with a device id, the gloo side group for the barriers, the broadcast of the
weight blob + SMPL constants as CUDA tensors, a model built from the blob in HBM (mp_create_from_device) under a rank, the
MAX all-reduce over ranks, the all-gathers behind n_ranks_seen / per_rank -- and that none of it costs step time or changes
a single output bit against the plain single-process run.  The JSON lines are kept under gpurun_out/ (copied to profiles/)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _bench(argv, launcher):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable]
    if launcher:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port())]
    cmd += [os.path.join(REPO, "bench.py"), "--gpus", "1", "--no-cpu-baseline"] + argv
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    return json.loads(lines[0])


def _keep(name, line):
    out = os.path.join(REPO, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, name), "w") as f:
            json.dump(line, f, indent=1)
    except OSError:
        pass


def test_one_rank_rccl_launch_matches_the_plain_run():
    args = ["--steps", "50", "--warmup", "5"]
    plain = _bench(args, launcher=False)
    ranked = _bench(args, launcher=True)
    _keep("r06_rccl_1rank_weak.json", ranked)
    _keep("r06_plain_1gpu.json", plain)
    assert plain["n_ranks_seen"] == 1 and [p["rank"] for p in plain["per_rank"]] == [0]
    assert plain["verified"]["rows"] == 16 and ranked["verified"]["rows"] == 16        # (bench.py checks what it timed)
    assert plain["build_id"] == ranked["build_id"] == ranked["per_rank"][0]["build_id"]
    assert ranked["per_rank"][0]["device"] == 0 and ranked["per_rank"][0]["xcd_round_robin"] is True
    assert ranked["n_gpus"] == 1 and ranked["n_ranks_seen"] == 1
    assert [p["rank"] for p in ranked["per_rank"]] == [0] and ranked["per_rank"][0]["frames"] == 256 * 125 * 50
    assert "RCCL" in ranked["config"]["launcher"] and ranked["config"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # the model built from the broadcast blob in HBM computes the same bits as the one built from host weights
    assert ranked["output_sha1"] == plain["output_sha1"]
    # and the process group costs no step time (host-side barriers; same box, back to back): `value` within 1 % (round 6;
    # measured 0.9997, 1.0039 -- and once 1.022: two processes, a 0.18 s timed region each, and a step whose launches wait for the
    # host after every synchronised call.  A pair that disagrees is measured again, up to three times, and the FASTEST run of each kind
    # is compared: a cost of the process group shows in every ranked run, a slow process start in one)
    t_plain, t_ranked = [plain["ms_per_step"]], [ranked["ms_per_step"]]
    for _ in range(3):
        if 0.99 < min(t_ranked) / min(t_plain) < 1.01:
            break
        t_plain.append(_bench(args, launcher=False)["ms_per_step"])
        t_ranked.append(_bench(args, launcher=True)["ms_per_step"])
    ratio = min(t_ranked) / min(t_plain)
    print("1-rank RCCL launch: %s ms/step, plain: %s ms/step, ratio of the fastest %.4f" % (t_ranked, t_plain, ratio))
    assert 0.99 < ratio < 1.01, (t_ranked, t_plain)
    assert "numa_node" in ranked["per_rank"][0] and "cpus_pinned" in ranked["per_rank"][0]
    for leg in ("configs0_single_sequence", "configs1_joints_only", "configs3_strong", "configs4_stream"):        # every BASELINE config in the line
        assert leg in plain, leg
    assert ranked["config"]["recoveries_during_run"] == 0


def test_one_rank_rccl_launch_strong_scaling_and_streams():
    strong = _bench(["--steps", "10", "--warmup", "2", "--scaling", "strong"], launcher=True)
    _keep("r06_rccl_1rank_strong.json", strong)
    assert strong["n_ranks_seen"] == 1 and strong["scaling"] == "strong" and strong["config"]["global_batch"] == 1024
    assert strong["per_rank"][0]["frames"] == 1024 * 125 * 10
    stream = _bench(["--steps", "10", "--warmup", "2", "--workload", "stream", "--streams", "512"], launcher=True)
    _keep("r06_rccl_1rank_stream.json", stream)
    assert stream["n_ranks_seen"] == 1 and stream["config"]["streams_per_gpu"] == 512 and stream["config"]["meets_60hz"]
