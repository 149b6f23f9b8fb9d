"""bench.py's multi-rank bookkeeping on CPU: two gloo ranks run the strong-scaling shard logic (configs[3]: a global batch
split with shard_range), the one-broadcast start-up (weights + SMPL constants) and the per-rank gather the JSON line
reports -- with a stand-in "forward" (the HIP path needs a GPU; what is tested here is the rank logic)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mobileposer_amd import synthetic
from mobileposer_amd.dist import broadcast_model, gather_counts, shard_range
from mobileposer_amd.model_utils import state_dict_to_blob


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sd = synthetic.make_weights(0) if rank == 0 else None
        smpl = synthetic.synthetic_smpl() if rank == 0 else None
        blob, smpl_r = broadcast_model(sd, smpl, "cpu", src=0)
        ref_smpl = synthetic.synthetic_smpl()
        ok_w = bool(np.array_equal(blob.numpy(), state_dict_to_blob(synthetic.make_weights(0))))
        ok_s = bool(np.array_equal(smpl_r["J"], ref_smpl["J"])) and \
            [-1] + list(np.asarray(smpl_r["kintree_table"])[0][1:]) == [-1] + list(np.asarray(ref_smpl["kintree_table"])[0][1:])
        G, T, steps = 10, 4, 3
        lo, hi = shard_range(G, rank, world)
        imu = synthetic.make_imu(G, T, seed=1)[lo:hi]              # every rank draws the same global batch, keeps its rows
        checksum = float(np.abs(imu).sum())
        counts = gather_counts((hi - lo) * T * steps, 0.5 + rank, "cpu")
        tt = torch.tensor([0.5 + rank], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        value = G * T * steps / float(tt.item())                     # whole-job frames / max-over-ranks seconds
        q.put((rank, ok_w, ok_s, (lo, hi), checksum, counts.tolist(), value))
    finally:
        dist.destroy_process_group()


def test_strong_scaling_rank_logic_two_gloo_ranks():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] and r[2] for r in res)
    assert [r[3] for r in res] == [(0, 5), (5, 10)]
    whole = float(np.abs(synthetic.make_imu(10, 4, seed=1)).sum())
    assert abs(res[0][4] + res[1][4] - whole) < 1e-3 * whole         # the shards are exactly the global batch
    assert res[0][5] == res[1][5] == [[60.0, 0.5], [60.0, 1.5]]
    assert res[0][6] == res[1][6] == 10 * 4 * 3 / 1.5
