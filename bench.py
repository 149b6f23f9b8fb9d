#!/usr/bin/env python3
"""bench.py -- IMU frames/s of the MobilePoser inference path on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one synthetic batch of 256 IMU windows x 125 frames that is
already resident in HBM:  mp_forward_offline = forward (4 stacked-LSTM modules + 6D->SO(3) + global->local)
 + SMPL forward kinematics of the predicted pose + foot-contact/velocity translation solver, one captured graph.  With N > 1 every rank (one process per GPU, launched by torch.distributed.run) runs
the same per-GPU workload on its own seeded batch (weak scaling); the only collective is the RCCL broadcast
of the weight blob from rank 0 at start-up.

Prints ONE JSON line on rank 0 (see the field list in README / DESIGN.md section "Measurement").
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

B_PER_GPU, T_WIN = 256, 125
FLOP_PER_FRAME = 2 * 6651392            # 13.30 MFLOP / frame for the 4 modules
BYTES_PER_FRAME = 1688                  # compulsory HBM bytes / frame (240 in + 1448 out)
PEAK_FP32_MFMA_TFLOPS = 157.3           # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0          # MI355X_MICROARCH.md: dense bf16 MFMA peak (no sparsity)


X3_NAMES = {         # the same timing classes when the H = 256 blocks run on split-bf16 operands (--lstm-mode x3)
    0: "mp_gemm_x3 (linear1 / linear2 of the H=256 blocks) + mp_gemm_f32 (foot-contact block)",
    1: "mp_lstm_x3<8,256> bidirectional layer 0 (joints, pose)",
    4: "mp_lstm_x3w<512> bidirectional layer 1 (joints, pose)",
    5: "mp_lstm_x3<8,256> unidirectional layers (velocity; 128 workgroups)",
}

KERNEL_CLASSES = {   # timing classes of the C ABI (include/mobileposer_hip.h, mp_timing_read)
    0: "mp_gemm_f32 (linear1 / linear2)",
    1: "mp_lstm_fused<256,8,256,2> bidirectional layer 0 (joints, pose)",
    4: "mp_lstm_fused<256,8,512,2> bidirectional layer 1 (joints, pose)",
    5: "mp_lstm_fused<256,16,256,1> unidirectional layers (velocity)",
    6: "mp_lstm_fused<64,4,*,1> (foot contact)",
    7: "mp_lstm_step (per-step fallback)",
    2: "mp_r6d_ik",
}


def cpu_baseline(seconds=10.0):
    """The numpy oracle (a port of the reference's CPU path) on a bounded sample of the same workload."""
    from mobileposer_amd import synthetic
    from oracle import mp_oracle as O
    sd = synthetic.make_weights(0)
    smpl = synthetic.synthetic_smpl()
    Bs = 32
    imu = synthetic.make_imu(Bs, T_WIN, seed=1)[:Bs]
    ref = O.OracleNet(sd, smpl["J"])

    def one():
        ref.velocity_rnn_state = None
        pose, joints, vel, contact = ref.forward(imu, [T_WIN] * Bs)
        O.forward_kinematics(pose, smpl["J"])
        for b in range(Bs):
            O.translate_offline(joints[b].reshape(T_WIN, 24, 3), vel[b], contact[b], ref.floor_y)

    one()                                   # warm-up
    reps, t0 = 0, time.perf_counter()
    while True:
        one()
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or reps >= 50:
            break
    out = {"value": round(reps * Bs * T_WIN / dt, 1), "unit": "frames/s", "cores": os.cpu_count(), "kind": "port",
           "sample": "%d reps of %d sequences x %d frames (numpy oracle: forward + FK + translation), %.1f s"
                     % (reps, Bs, T_WIN, dt)}
    # the same workload through torch's own CPU nn.LSTM / nn.Linear (what the reference calls): the full 256 x 125
    # batch, a thread count ATen scales to (more threads than that are slower on these small GEMMs)
    try:
        from oracle.torch_ref import TorchNet
        nthr = max(1, min(32, os.cpu_count() or 1))
        torch.set_num_threads(nthr)
        tnet = TorchNet(sd, smpl["J"])
        imu_full = synthetic.make_imu(B_PER_GPU, T_WIN, seed=1)

        def one_t():
            tnet.vel_state = None
            pose, joints, vel, contact, _ = tnet.forward(imu_full, [T_WIN] * B_PER_GPU)
            O.forward_kinematics(pose, smpl["J"])
            for b in range(B_PER_GPU):
                O.translate_offline(joints[b].reshape(T_WIN, 24, 3), vel[b], contact[b], tnet.floor_y)

        one_t()
        reps_t, t0 = 0, time.perf_counter()
        while True:
            one_t()
            reps_t += 1
            dt_t = time.perf_counter() - t0
            if dt_t >= seconds or reps_t >= 50:
                break
        out["torch_cpu"] = {"value": round(reps_t * B_PER_GPU * T_WIN / dt_t, 1), "unit": "frames/s", "threads": nthr,
                            "sample": "%d reps of %d sequences x %d frames through torch CPU nn.LSTM "
                                      "(oracle/torch_ref.py) + numpy FK / translation, %.1f s"
                                      % (reps_t, B_PER_GPU, T_WIN, dt_t)}
    except Exception as e:
        out["torch_cpu"] = {"error": str(e)}
    # report the stronger CPU figure as the baseline proper, keep the other one beside it
    t = out.get("torch_cpu", {})
    if "value" in t and t["value"] > out["value"]:
        numpy_leg = {k: out[k] for k in ("value", "unit", "cores", "sample")}
        out = {"value": t["value"], "unit": "frames/s", "cores": t["threads"], "kind": "port", "sample": t["sample"],
               "numpy_oracle": numpy_leg}
    return out


def bench_stream(args, net, dev, dist, rank, world):
    """BASELINE configs[4]: S concurrent streams per GPU, one new 60-d frame per stream per tick; every tick runs
    the reference's forward_online semantics (45-frame window re-evaluated, net.py:173-219) from one captured graph."""
    from mobileposer_amd import synthetic
    S = args.streams
    frames = torch.from_numpy(synthetic.make_imu(S, args.steps + args.warmup + 1, seed=7 + rank)).to(dev)
    net.stream_create(S)
    f32 = torch.float32
    io = {"pose": torch.empty(S, 24, 9, device=dev, dtype=f32), "joints": torch.empty(S, 45, 72, device=dev, dtype=f32),
          "root": torch.empty(S, 3, device=dev, dtype=f32), "contact": torch.empty(S, 2, device=dev, dtype=f32)}

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for k in range(args.warmup):
        net.stream_step_into(frames[:, k].contiguous(), io["pose"], io["joints"], io["root"], io["contact"])
    cur = [frames[:, args.warmup + k].contiguous() for k in range(args.steps)]
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        net.stream_step_into(cur[k], io["pose"], io["joints"], io["root"], io["contact"])
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank == 0:
        ticks = args.steps / elapsed
        print(json.dumps({
            "metric": "streaming output frames/s (forward_online semantics: 45-frame window per output frame)",
            "value": round(world * S * ticks, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 / ticks, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[4]: %d concurrent streams per GPU, hipGraph-captured tick" % S,
                       "streams_per_gpu": S, "ticks_per_s": round(ticks, 2), "meets_60hz": ticks >= 60.0,
                       "window_frames_per_s": round(world * S * 45 * ticks, 1)}}))
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--workload", choices=["offline", "stream"], default="offline",
                    help="offline (default, the BASELINE metric) or stream: config 5, S concurrent 45-frame windows per GPU")
    ap.add_argument("--streams", type=int, default=512)
    ap.add_argument("--lstm-mode", choices=["fp32", "x3"], default="x3",
                    help="MFMA operands of the H=256 LSTM layers for the headline value: x3 (the library default: every fp32 "
                         "product as 3 bf16 products on v_mfma_f32_16x16x32_bf16, fp32 accumulate) or fp32 (exact "
                         "v_mfma_f32_16x16x4_f32); the other mode is timed as well and reported beside it")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or "RANK" in os.environ:           # launched by torch.distributed.run: one process per GPU over RCCL
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
        local_rank = 0
    dev = torch.device("cuda", local_rank)

    import __graft_entry__
    if not os.path.exists(__graft_entry__.LIB):
        if rank == 0:
            __graft_entry__.build()
        if dist is not None:
            dist.barrier()

    from mobileposer_amd import synthetic
    from mobileposer_amd.dist import broadcast_weights
    from mobileposer_amd.net import MobilePoserNet

    smpl = synthetic.synthetic_smpl()
    if dist is not None:
        blob = broadcast_weights(synthetic.make_weights(0) if rank == 0 else None, dev, src=0)
        net = MobilePoserNet.from_device_blob(blob, smpl, device=dev)
    else:
        net = MobilePoserNet.from_numpy(synthetic.make_weights(0), smpl, device=dev)
    if args.no_graph:
        net.set_graph_mode(False)

    if args.workload == "stream":
        return bench_stream(args, net, dev, dist, rank, world)

    B, T = B_PER_GPU, T_WIN
    imu = torch.from_numpy(synthetic.make_imu(B, T, seed=1 + rank)).to(dev)
    f32 = torch.float32
    pose = torch.empty(B * T, 24, 3, 3, device=dev, dtype=f32)
    joints = torch.empty(B, T, 72, device=dev, dtype=f32)
    vel = torch.empty(B, T, 72, device=dev, dtype=f32)
    contact = torch.empty(B, T, 2, device=dev, dtype=f32)
    tran = torch.empty(B, T, 3, device=dev, dtype=f32)
    rglob = torch.empty(B * T, 24, 3, 3, device=dev, dtype=f32)
    jglob = torch.empty(B * T, 24, 3, device=dev, dtype=f32)
    lens = (C.c_int32 * B)(*([T] * B))
    lib, h = net._lib, net._h
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    vp = lambda t: C.c_void_p(t.data_ptr())

    def step():
        lib.mp_reset_state(h, 1)            # every step is a fresh, independent batch
        rc = lib.mp_forward_offline(h, vp(imu), lens, B, T, vp(pose), vp(joints), vp(vel), vp(contact), vp(tran),
                                    vp(rglob), vp(jglob), stream)       # forward + FK + translation solver, one graph
        if rc:
            raise RuntimeError(lib.mp_last_error(h).decode())

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    MODE_ID = {"fp32": 1, "x3": 3}

    def timed(mode):
        """W warm-up steps, then exactly K steps between barrier + synchronize; max over ranks."""
        net.set_lstm_mode(MODE_ID[mode])
        for _ in range(args.warmup):
            step()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        dt = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    other = "x3" if args.lstm_mode == "fp32" else "fp32"
    elapsed_other = timed(other)
    outs_other = [t.clone() for t in (joints, vel, contact, tran, pose)]
    elapsed = timed(args.lstm_mode)                       # the headline measurement (its outputs stay in the buffers)
    mode_dev = max(float((a - b).abs().max()) for a, b in zip(outs_other, (joints, vel, contact, tran, pose)))

    # ---- per-kernel-class timing: HIP events around every launch, on the library stream that launches it ----
    kern, dominant = {}, None
    if rank == 0:
        net.timing_enable(True)
        acc = {c: [0, 0.0, 0.0] for c in list(KERNEL_CLASSES) + [3]}
        reps = 3
        for _ in range(reps):
            lib.mp_reset_state(h, 1)
            lib.mp_forward(h, vp(imu), lens, B, T, vp(pose), vp(joints), vp(vel), vp(contact), None, stream)
            torch.cuda.synchronize(dev)
            for cls in acc:
                n, ms, gf = net.timing_read(cls)
                acc[cls][0] += n
                acc[cls][1] += ms
                acc[cls][2] += gf
        net.timing_enable(False)
        names = dict(KERNEL_CLASSES)
        if args.lstm_mode == "x3":
            names.update(X3_NAMES)
        for cls, name in names.items():
            n, ms, gf = acc[cls]
            if n == 0:
                continue
            kern[name] = {"launches_per_forward": n // reps, "avg_launch_ms": round(ms / n, 4),
                          "ms_per_forward": round(ms / reps, 4), "gflop_per_launch": round(gf / n, 3),
                          "tflops": round(gf / ms, 2) if ms > 0 else None}
        kern["forward_eager_ms"] = round(acc[3][1] / reps, 4)
        # the kernel class with the largest share of the forward is the one the roofline line describes
        dominant = max((c for c in KERNEL_CLASSES if acc[c][0]), key=lambda c: acc[c][1])
    if dist is not None:
        dist.barrier()

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    frames = world * B * T * args.steps
    value = frames / elapsed
    value_other = frames / elapsed_other
    dn, dms, dgf = acc[dominant]
    achieved = dgf / dms if dms > 0 else 0.0              # GFLOP / ms = TFLOP/s
    x3_dom = args.lstm_mode == "x3" and dominant in X3_NAMES
    if x3_dom:
        # every algorithmic fp32 multiply-add is executed as 3 bf16 multiply-adds: price the EXECUTED MFMA flops
        # against the dense bf16 peak (the algorithmic rate is reported beside it)
        roof_peak, roof_achieved = PEAK_BF16_MFMA_TFLOPS, 3.0 * achieved
        roof_note = ("3 x algorithmic FLOPs of the launch (each fp32 product = hi*hi + hi*lo + lo*hi on "
                     "v_mfma_f32_16x16x32_bf16) / HIP-event duration, dense bf16 MFMA peak; the kernel is bound by the "
                     "per-step hidden-state exchange latency, not by the matrix pipe (DESIGN.md)")
    else:
        roof_peak, roof_achieved = PEAK_FP32_MFMA_TFLOPS, achieved
        roof_note = ("algorithmic FLOPs of the launch (input projection + recurrence, SURVEY.md 8(d)) / "
                     "HIP-event duration of the launch, v_mfma_f32_16x16x4_f32 dense peak")
    # HBM bytes per launch of that kernel from the separate rocprofv3 --pmc passes (profiles/r01_pmc_summary.json):
    # (2*FETCH_SIZE + WRITE_SIZE)*1024, corrected as MI355X_MICROARCH.md prescribes; null when no profile is present
    traffic = None
    try:
        pmc = json.load(open(os.path.join(REPO, "profiles", "r01_pmc_summary.json")))["kernels"]
        key = {1: "mp_lstm_fused<256, 8, 256, 2, false>", 4: "mp_lstm_fused<256, 8, 512, 2, false>",
               5: "mp_lstm_fused<256, 16, 256, 1, false>", 0: "mp_gemm_f32<2, 2, 2, 2>"}.get(dominant)
        if args.lstm_mode == "x3":
            key = {1: "mp_lstm_x3<8, 256, false>", 4: "mp_lstm_x3w<512, false>", 5: "mp_lstm_x3<8, 256, false>",
                   0: "mp_gemm_x3<128, 64>"}.get(dominant)
        traffic = pmc[key]["hbm_bytes_per_launch_corrected"] if key in pmc else None
    except Exception:
        traffic = None
    out = {
        "metric": "imu_frames_per_sec (MobilePoserNet fwd + FK + translation solver, batch 256 x window 125 per GPU)",
        "value": round(value, 1), "unit": "frames/s", "per_gpu": round(value / world, 1),
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.lstm_mode == "fp32" else "f32 (LSTM and linear-layer matrix products as 3-term split-bf16 MFMA, fp32 accumulate)",
        "data": "synthetic",
        "config": {"workload": "configs[2]+solver: full MobilePoserNet (4 LSTM modules) + r6d/IK + SMPL FK + offline "
                               "translation solver, B=256 x T=125 per GPU, seeded synthetic IMU (lw_rp combo), "
                               "seeded random weights, synthetic SMPL constants",
                   "batch_per_gpu": B, "window": T, "global_batch": world * B,
                   "parallelism": "independent sequences sharded, dp%d" % world,
                   "graph": not args.no_graph, "lstm_mode": args.lstm_mode},
        "modes": {
            args.lstm_mode: {"value": round(value, 1), "ms_per_step": round(1e3 * elapsed / args.steps, 4), "headline": True},
            other: {"value": round(value_other, 1), "ms_per_step": round(1e3 * elapsed_other / args.steps, 4),
                    "headline": False},
            "max_abs_output_difference_between_modes": mode_dev,
            "note": "fp32 = exact v_mfma_f32_16x16x4_f32 operands; x3 = each fp32 product as hi*hi+hi*lo+lo*hi of bf16 "
                    "parts on v_mfma_f32_16x16x32_bf16 with fp32 accumulate and fp32 state (mp_set_lstm_mode(h, 3)); both "
                    "pass the same parity tests at 1e-4 / 1 mm (tests/test_gpu_parity.py runs every test in both modes)"},
        "end_to_end": {"tflops": round(value * FLOP_PER_FRAME / 1e12, 3),      # algorithmic fp32 FLOPs of the 4 modules
                       "frac_of_fp32_mfma_peak": round(value / world * FLOP_PER_FRAME / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                       "frac_of_bf16_mfma_peak_executed": (round(3.0 * value / world * FLOP_PER_FRAME / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4)
                                                           if args.lstm_mode != "fp32" else None),
                       "hbm_gbps_compulsory": round(value / world * BYTES_PER_FRAME / 1e9, 2)},
        "roofline": {"kernel": names[dominant], "bound": "mfma",
                     "achieved": round(roof_achieved, 2), "peak": roof_peak, "unit": "TFLOP/s",
                     "frac": round(roof_achieved / roof_peak, 4), "traffic": traffic,
                     "flop_per_launch": round(dgf / dn * 1e9), "algorithmic_tflops": round(achieved, 2),
                     "avg_launch_ms": round(dms / dn, 4), "note": roof_note},
        "kernels": kern,
    }
    if not args.no_cpu_baseline and world == 1:          # the CPU leg is timed on rank 0 of the single-GPU run only
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
