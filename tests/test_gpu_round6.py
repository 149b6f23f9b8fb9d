"""Round-6 GPU parity tests.

  * G17 -- the reference's OWN call shape on trained-regime weights (evaluate.py:54-58, models/net.py:121-155): ONE sequence of
    2000 / 2500 / 3000 frames x 3 input seeds, recorded from the reference together with the distances of an ensemble of fp32
    evaluations from the float64 result (tests/golden/make_golden.py, oracle/ensemble.py).  The one-sequence kernels (mp_lstm_v1 /
    mp_lstm_v1s, the default at B = 1) and the MFMA path (MP_VARIANT=vec=0) are both held to the band that ensemble occupies.
  * configs[1] of BASELINE.json through its OWN entry point: mp_rnn_forward("joints") at 256 x 125 (models/joints.py:48-52),
    ragged lengths and a carried state, against the oracle's rnn_forward.

Why a band and not "1e-4 of the golden": at this length the trained-regime net is chaotic at fp32 resolution.  Permuting the
summation order of the numpy oracle alone moves its maximum distance from the float64 result by up to 10 x, and float64 dot
products with fp32 state do not lower it (profiles/r06_b1_precision_emulation.txt): the reference itself is 2e-5 ... 9e-4 from the
float64 result on these nine cases.  The ensemble has 16 members (the reference, the oracle, 14 permuted orders); its first version
had 5, and the maximum over 5 draws of so heavy-tailed a distance was no envelope: kernels whose LEVEL was 1.0-1.3 fell outside
twice of it in 2 of 90 checks (profiles/r06_accuracy_g17_5members.txt).  The rule (NOISE_FACTOR_B1): per case and output, max |x - f64| <= 2 x the largest maximum of
the ensemble (or the north-star bound, 1e-4 / 1 mm, where that is larger) AND mean |x - f64| <= 2 x the largest mean of the
ensemble; over the nine cases, the geometric mean of (mean distance / median member's mean distance) <= 1.5 -- a kernel whose
noise LEVEL is above fp32's shows up there even when every single draw is inside the band.
"""
import os

import numpy as np
import pytest

from conftest import cu, geodesic, load_golden, npy

pytestmark = pytest.mark.gpu

TOL = {"r6d": 1e-4, "joints": 1e-4, "vel": 1e-4, "contact": 1e-4, "tran": 1e-3}
NOISE_FACTOR_B1 = 2.0        # per case: x the ensemble's envelope
LEVEL_FACTOR_B1 = 1.5        # over all cases: geometric mean of (mean distance / the median member's)
OUTPUTS = ("r6d", "joints", "vel", "contact", "tran")


@pytest.fixture(scope="module")
def g17():
    return load_golden("g17_single_sequence.npz")


@pytest.fixture(scope="module")
def weights_trained():
    from mobileposer_amd.synthetic import make_weights
    return make_weights(0, profile="trained")


@pytest.fixture(scope="module")
def g17_truth(g17, weights_trained, smpl):
    """The float64 result of every case (the oracle's arithmetic carried out in float64; ~4 s per case on the host)."""
    from mobileposer_amd import synthetic
    from oracle import ensemble as ENS
    truth = {}
    for T in g17["lengths"].tolist():
        for k, seed in enumerate(g17["seeds"].tolist()):
            imu = synthetic.make_imu(1, T, seed=seed, combo=str(g17["combos"][k]))
            assert abs(float(imu.astype(np.float64).sum()) - float(g17["T%d_s%d_imu_sum" % (T, seed)])) < 1e-9      # the recorded input
            truth[(T, seed)] = (imu, ENS.offline_outputs(weights_trained, smpl["J"], imu, T, dtype=np.float64))
    return truth


def _gpu_outputs(torch_mod, net, imu, T):
    import ctypes as C
    net.reset_all()
    pose, joints, vel, contact, r6d = net.forward(cu(torch_mod, imu), [T], return_r6d=True)
    tran = torch_mod.empty(1, T, 3, device="cuda")
    net.translate_offline_into(joints, vel.reshape(1, T, 72), contact, (C.c_int32 * 1)(T), tran)
    # (forward_offline itself -- the call evaluate.py makes -- must give the same bits)
    net.reset_all()
    pose2, joints2, tran2, contact2 = net.forward_offline(cu(torch_mod, imu), [T])
    assert torch_mod.equal(tran2, tran[0]) and torch_mod.equal(joints2, joints) and torch_mod.equal(contact2, contact[0])
    return {"r6d": npy(r6d).reshape(T, 96), "joints": npy(joints).reshape(T, 72), "vel": npy(vel).reshape(T, 72),
            "contact": npy(contact).reshape(T, 2), "tran": npy(tran).reshape(T, 3)}


@pytest.mark.parametrize("variant", ["", "vec=0"], ids=["one-sequence-kernels", "mfma-path"])
def test_g17_single_sequence_trained_regime(torch_mod, g17, g17_truth, weights_trained, smpl, variant, monkeypatch):
    from mobileposer_amd.net import MobilePoserNet
    from oracle import ensemble as ENS
    if variant:
        monkeypatch.setenv("MP_VARIANT", variant)
    members = [str(m) for m in g17["members"]]
    stride = int(g17["stride"])
    ratios = {k: [] for k in OUTPUTS}
    report = {}
    failures = []
    with MobilePoserNet.from_numpy(weights_trained, smpl) as net:
        for (T, seed), (imu, truth) in g17_truth.items():
            tag = "T%d_s%d" % (T, seed)
            got = _gpu_outputs(torch_mod, net, imu, T)
            assert net.device_error() == 0 and net.recovery_count == 0
            d = ENS.distance(got, truth)
            dist = g17[tag + "_dist"]                                   # [member][output][max, mean]
            env_max, env_mean = dist[:, :, 0].max(axis=0), dist[:, :, 1].max(axis=0)
            med_mean = np.median(dist[:, :, 1], axis=0)
            report[tag] = {k: "%.1e/%.1e (band %.1e/%.1e)" % (d[k][0], d[k][1], env_max[i], env_mean[i]) for i, k in enumerate(OUTPUTS)}
            for i, k in enumerate(OUTPUTS):                 # (collected, reported in full, asserted at the end)
                if d[k][0] > max(TOL[k], NOISE_FACTOR_B1 * env_max[i]):
                    failures.append((tag, k, "max", d[k][0], env_max[i]))
                if d[k][1] > max(0.01 * TOL[k], NOISE_FACTOR_B1 * env_mean[i]):
                    failures.append((tag, k, "mean", d[k][1], env_mean[i]))
                ratios[k].append(max(d[k][1], 1e-12) / max(med_mean[i], 1e-12))
            # against the reference's own outputs (every `stride`-th frame; contact and translation in full): by the triangle
            # inequality no farther than the two distances from the float64 result -- a check of the golden's layout and of
            # this test's indexing rather than of the kernels
            ref_d = dist[members.index("reference")]
            sub = slice((T - 1) % stride, None, stride)
            for i, k in enumerate(OUTPUTS):
                ref = g17[tag + "_" + k]
                mine = got[k] if k in ("contact", "tran") else got[k][sub]
                assert np.abs(mine - ref).max() <= 1.001 * (d[k][0] + ref_d[i, 0]) + 1e-7, (tag, k)
    level = {k: float(np.exp(np.mean(np.log(v)))) for k, v in ratios.items()}
    print("G17 (%s): noise level relative to the ensemble's median member (geometric mean over %d cases): %s"
          % (variant or "default", len(g17_truth), {k: "%.2f" % v for k, v in level.items()}))
    for tag, r in report.items():
        print("   %s max/mean |x - f64|: %s" % (tag, r))
    import json                                         # keep the numbers (profiles/r06_accuracy_g17*.json are copies of these files)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump({"level": level, "cases": report, "outside_band": [list(map(str, f)) for f in failures],
               "rule": {"per_case": NOISE_FACTOR_B1, "level": LEVEL_FACTOR_B1}},
              open(os.path.join("gpurun_out", "r06_accuracy_g17%s.json" % ("_" + variant.replace("=", "") if variant else "")), "w"), indent=1)
    assert not failures, failures
    for k in OUTPUTS:
        assert level[k] <= LEVEL_FACTOR_B1, (k, level)


@pytest.mark.parametrize("profile", ["init", "trained"])
def test_config1_joints_module_through_its_own_entry(torch_mod, smpl, profile):
    """BASELINE configs[1]: the joints module alone, 256 x 125, through mp_rnn_forward (models/joints.py:48-52 -> rnn.py:20-33) --
    the tail of this entry (joints.linear2 by itself) is not the one the full forward runs (mp_gemm_l2l1).  Full lengths, ragged
    lengths, and a second call on the state the first one returned; outputs AND final states against the oracle."""
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    from oracle import mp_oracle as O
    B, T = 256, 125
    sd = synthetic.make_weights(0, profile=profile)
    x = synthetic.make_imu(B, T, seed=91)
    ragged = [T - (7 * b) % 60 for b in range(B)]
    ragged[17] = T

    def exact(lengths, state=None):              # the oracle's arithmetic in float64 (the yardstick on trained-regime weights)
        O.F32 = np.float64
        try:
            return O.rnn_forward(sd, O.PREFIX["joints"], x, lengths, state)
        finally:
            O.F32 = np.float32

    def close(tag, got, ref, truth, scale=1.0):
        """1e-4 against the fp32 oracle -- or, on the trained-regime net (250 steps of amplified rounding by the second call: two
        fp32 evaluations are more than 1e-4 apart somewhere in 256 x 256 states), no farther from the float64 result than
        NOISE_FACTOR x the fp32 oracle is (the rule of tests/test_gpu_round4.py at this size)."""
        e = float(np.abs(np.asarray(got, np.float64) - ref).max())
        if e < 1e-4 * scale:
            return
        assert truth is not None, (tag, e)
        e64 = float(np.abs(np.asarray(got, np.float64) - truth).max())
        n64 = float(np.abs(np.asarray(ref, np.float64) - truth).max())
        print("   %s %s: %.2e from the fp32 oracle; from float64: library %.2e, fp32 oracle %.2e" % (profile, tag, e, e64, n64))
        assert e64 < NOISE_FACTOR_B1 * n64, (tag, e, e64, n64)

    with MobilePoserNet.from_numpy(sd, smpl) as net:
        for lengths in ([T] * B, ragged):
            y, (h, c) = net.rnn_forward("joints", cu(torch_mod, x), lengths)
            ry, (rh, rc) = O.rnn_forward(sd, O.PREFIX["joints"], x, lengths)
            ty, (th, tc) = exact(lengths) if profile == "trained" else (None, (None, None))
            # (rows past a sequence's length: linear2(0) = bias on both sides)
            close("y", npy(y), ry, ty)
            close("h", npy(h), rh, th)
            close("c", npy(c), rc, tc, max(1.0, float(np.abs(rc).max())))
            # carried state: the same input again, starting from (h, c)
            y2, (h2, c2) = net.rnn_forward("joints", cu(torch_mod, x), lengths, (h, c))
            ry2, (rh2, rc2) = O.rnn_forward(sd, O.PREFIX["joints"], x, lengths, (rh, rc))
            ty2, (th2, tc2) = exact(lengths, (th, tc)) if profile == "trained" else (None, (None, None))
            close("y2", npy(y2), ry2, ty2)
            close("h2", npy(h2), rh2, th2)
            close("c2", npy(c2), rc2, tc2, max(1.0, float(np.abs(rc2).max())))
        assert net.device_error() == 0 and net.recovery_count == 0


def test_three_stream_schedule_at_full_batch_vs_oracle(torch_mod, weights, smpl, monkeypatch):
    """MP_VARIANT=one_stream=0 (the round-3 three-stream schedule, the cross-check of tools/debug/fuzz_modes.py) at
    B > 128 runs the velocity block as the two-layer wavefront, which carries foot-contact layer 1 only -- layer 0 has to ride in
    pose layer 0 there too (it never did before round 6: stale foot-contact layer-0 output).  160 x 40 ragged against the oracle,
    and bitwise against the default schedule."""
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    from oracle import mp_oracle as O
    B, T = 160, 40
    imu = synthetic.make_imu(B, T, seed=93)
    lengths = [T - (b % 7) for b in range(B)]
    rp, rj, rv, rc = O.OracleNet(weights, smpl["J"]).forward(imu, lengths)
    outs = {}
    for variant in ("", "one_stream=0"):
        if variant:
            monkeypatch.setenv("MP_VARIANT", variant)
        with MobilePoserNet.from_numpy(weights, smpl) as net:
            pose, joints, vel, contact = net.forward(cu(torch_mod, imu), lengths)
            outs[variant] = [npy(t) for t in (joints, vel, contact, pose)]
            for b, n in enumerate(lengths):
                assert np.abs(outs[variant][0][b, :n] - rj[b, :n]).max() < 1e-4, (variant, b)
                assert np.abs(outs[variant][1][b, :n] - rv.reshape(B, T, 72)[b, :n]).max() < 1e-4, (variant, b)
                assert np.abs(outs[variant][2][b, :n] - rc[b, :n]).max() < 1e-4, (variant, b)
            assert net.device_error() == 0 and net.recovery_count == 0
    for a, b in zip(outs[""], outs["one_stream=0"]):          # same kernels, another launch order
        assert np.array_equal(a, b)


def _plan_stats(net):
    import ctypes as C
    n, a, r = C.c_int(0), C.c_int(0), C.c_longlong(0)
    assert net._lib.mp_debug_plan_stats(net._h, C.byref(n), C.byref(a), C.byref(r)) == 0
    return n.value, a.value, r.value


def test_many_shapes_reuse_plans(torch_mod, weights, smpl):
    """Workspaces by capacity class (csrc/mp_plans.hip get_plan): 200 sequence lengths between 45 and 3000 at B = 1 -- the way
    evaluate.py walks through a dataset -- allocate at most 4 plans; a length never seen before costs what a seen one costs; and
    results on a shared plan are the bits a fresh handle computes (workspaces carry nothing from shape to shape)."""
    import time
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    rng = np.random.Generator(np.random.PCG64(5))
    Ts = [int(t) for t in rng.integers(45, 3001, size=200)]
    imu = cu(torch_mod, synthetic.make_imu(1, 3000, seed=94))
    with MobilePoserNet.from_numpy(weights, smpl) as net:
        dt = {}
        for k, T in enumerate(Ts):
            net.reset_all()
            torch_mod.cuda.synchronize()
            t0 = time.perf_counter()
            pose, joints, tran, contact = net.forward_offline(imu[:, :T].contiguous(), [T])
            torch_mod.cuda.synchronize()
            dt[k] = (T, time.perf_counter() - t0)
        n, allocs, rows = _plan_stats(net)
        print("200 lengths at B = 1: %d plans alive, %d allocated, %d rows of capacity" % (n, allocs, rows))
        assert allocs <= 4 and rows <= 4 * 4096
        # a length never seen before (the plan exists: no allocation) against the same length seen again, both warm
        unseen = [T for T in range(1500, 1600) if T not in Ts][0]

        def timed(T):
            net.reset_all()
            x = imu[:, :T].contiguous()
            torch_mod.cuda.synchronize()
            t0 = time.perf_counter()
            net.forward_offline(x, [T])
            torch_mod.cuda.synchronize()
            return time.perf_counter() - t0

        first = timed(unseen)
        again = min(timed(unseen) for _ in range(3))
        print("T = %d: first call %.2f ms, seen %.2f ms" % (unseen, 1e3 * first, 1e3 * again))
        assert first < 1.10 * again + 2e-4
        assert _plan_stats(net)[1] == allocs
        # bits: three lengths on the shared plan against a fresh handle each
        for T in (45, 777, 2999):
            net.reset_all()
            got = [npy(t) for t in net.forward_offline(imu[:, :T].contiguous(), [T])]
            with MobilePoserNet.from_numpy(weights, smpl) as fresh:
                want = [npy(t) for t in fresh.forward_offline(imu[:, :T].contiguous(), [T])]
            for a, b in zip(got, want):
                assert np.array_equal(a, b), T
        assert net.device_error() == 0 and net.recovery_count == 0


def test_batch_classes_share_a_plan_and_stay_bitwise(torch_mod, weights, smpl):
    """Batches above 64 sequences share plans by class ({2^k, 1.5 * 2^k}): 100, 128, 97 and 128 again on ONE plan (the exchange
    areas are zeroed when the batch size changes), each bitwise what a fresh handle computes; the streaming plan is never
    evicted or handed out with another shape's lengths in it."""
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    T = 30
    imu = cu(torch_mod, synthetic.make_imu(128, T, seed=95))
    with MobilePoserNet.from_numpy(weights, smpl) as net:
        for B in (100, 128, 97, 128):
            net.reset_all()
            got = [npy(t) for t in net.forward(imu[:B].contiguous(), [T] * B)]
            with MobilePoserNet.from_numpy(weights, smpl) as fresh:
                want = [npy(t) for t in fresh.forward(imu[:B].contiguous(), [T] * B)]
            for a, b in zip(got, want):
                assert np.array_equal(a, b), B
        assert _plan_stats(net)[1] == 1
        assert net.device_error() == 0 and net.recovery_count == 0


def test_a_plan_that_does_not_fit_in_memory_never_enters_the_cache(torch_mod, weights, smpl):
    """csrc/mp_plans.hip get_plan: a shape whose workspaces cannot be allocated (2^30 rows: 2 TB for the first buffer alone) is
    MP_ERR_HIP with hipMalloc's message -- and nothing of the half-built plan stays behind: a plan with null buffers in the cache
    would be handed to the next call of its batch class.  The handle keeps working, bitwise what a fresh one computes."""
    import ctypes as C
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    B, T = 3, 20
    imu = cu(torch_mod, synthetic.make_imu(B, T, seed=96))
    with MobilePoserNet.from_numpy(weights, smpl) as net:
        before = [npy(t) for t in net.forward_offline(imu, [T] * B)]
        allocs0 = _plan_stats(net)[1]
        big_b, big_t = 1 << 20, 1 << 10
        lengths = (C.c_int32 * big_b)(*([big_t] * big_b))
        dummy = torch_mod.zeros(16, device="cuda")
        vp = C.c_void_p(dummy.data_ptr())
        rc = net._lib.mp_translate_offline(net._h, vp, vp, vp, lengths, big_b, big_t, vp, None)
        assert rc != 0
        msg = net._lib.mp_last_error(net._h).decode()
        assert "hipMalloc" in msg, msg
        n, allocs, rows = _plan_stats(net)
        assert allocs == allocs0 and rows < (1 << 30), (n, allocs, rows)       # nothing of the failed plan is in the cache
        net.reset_all()
        after = [npy(t) for t in net.forward_offline(imu, [T] * B)]
        for a, b in zip(before, after):
            assert np.array_equal(a, b)
        assert net.device_error() == 0 and net.recovery_count == 0


def test_mode3_smoke(torch_mod, weights, smpl):
    """What is left of the opt-in split-fp16 mode in the default suite (conftest.lstm_test_modes): the full forward against golden
    G2 (equal and ragged lengths, carried velocity state) and 20 online frames against golden G5, at the bound of the exact mode
    (init-scale weights: both modes are within 5e-7 there)."""
    from conftest import geodesic
    from mobileposer_amd.net import MobilePoserNet
    g = load_golden("g2_forward.npz")
    g5 = load_golden("g5_online.npz")
    with MobilePoserNet.from_numpy(weights, smpl) as net:
        net.set_lstm_mode(3)
        for tag in ("eq", "rag"):
            net.reset_all()
            pose, joints, vel, contact, r6d = net.forward(cu(torch_mod, g["imu"]), g[f"{tag}_lengths"].tolist(), return_r6d=True)
            for got, key in ((joints, "joints"), (vel, "vel"), (contact, "contact"), (r6d, "r6d")):
                assert np.abs(npy(got) - g[f"{tag}_{key}"]).max() < 1e-4, (tag, key)
            assert geodesic(npy(pose), g[f"{tag}_pose"]).max() < 1e-4
            h, c = net.velocity.rnn_state
            assert np.abs(npy(h) - g[f"{tag}_vel_h"]).max() < 1e-4 and np.abs(npy(c) - g[f"{tag}_vel_c"]).max() < 1e-4
        net.reset_all()
        net.reset()
        for k, f in enumerate(g5["imu"][:20]):
            pose, joints, tran, contact = net.forward_online(cu(torch_mod, f))
            assert geodesic(npy(pose).reshape(24, 3, 3), g5["pose"][k].reshape(24, 3, 3)).max() < 1e-4, k
            assert np.abs(npy(joints)[40] - g5["joints40"][k]).max() < 1e-4
            assert np.abs(npy(contact) - g5["contact"][k]).max() < 1e-4
            assert np.abs(npy(tran) - g5["tran"][k]).max() < 1e-3, k
        assert net.device_error() == 0 and net.recovery_count == 0


@pytest.mark.parametrize("B", [2, 3, 4, 5])
def test_few_sequences_run_as_sequence_clusters(torch_mod, weights, smpl, B, monkeypatch):
    """Batches of 2 ... 4 sequences on the one-sequence kernels (mp_lstm_v1 / mp_lstm_v1s: a cluster per (direction, SEQUENCE),
    mp_schedule.hip seq_clusters; B = 2 runs pose | velocity | foot contact side by side, B = 3, 4 the serial schedule; B = 5 is the
    first batch on the 32-slice MFMA kernels and runs the same checks): ragged lengths against the oracle, twice (carried velocity
    state); sequences BITWISE what they give alone at B = 1 (B <= 4: same kernels, same order of summation -- a batch is its
    sequences); against the 32-slice MFMA kernels (MP_VARIANT=vec=0) to fp32 rounding; 30 ticks of B streams bitwise what
    one-stream handles give (B <= 4)."""
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    from oracle import mp_oracle as O
    T = 90
    imu = synthetic.make_imu(B, T, seed=96)
    lengths = ([T, 1, 37, 64, 90, 2, 45, 89, 17, 90, 33, 5, 71, 60, 9, 88])[:B]
    sample = list(range(B)) if B <= 4 else [0, 1, B // 2, B - 1]         # sequences checked alone / as one-stream handles
    ref = O.OracleNet(weights, smpl["J"])
    want = []
    for call in range(2):                                # (the velocity state carries from call to call, velocity.py:45-48)
        rpose, rjoints, rvel, rcontact = ref.forward(imu, lengths)
        rvel = np.asarray(rvel).reshape(B, T, 72)
        rtran = [O.translate_offline(rjoints[b, :n].reshape(n, 24, 3), rvel[b, :n], rcontact[b, :n], ref.floor_y) for b, n in enumerate(lengths)]
        want.append((rjoints, rcontact, rtran))
    outs = {}
    for variant in ("", "vec=0"):
        if variant:
            monkeypatch.setenv("MP_VARIANT", variant)
        with MobilePoserNet.from_numpy(weights, smpl) as net:
            got = [[npy(t) for t in net.forward_offline(cu(torch_mod, imu), lengths)] for _ in range(2)]
            outs[variant] = got
            for call in range(2):
                rjoints, rcontact, rtran = want[call]
                pose, joints, tran, contact = got[call]
                for b, n in enumerate(lengths):
                    assert np.abs(joints[b, :n] - rjoints[b, :n]).max() < 1e-4, (variant, call, b)
                    assert np.abs(contact[b, :n] - rcontact[b, :n]).max() < 1e-4, (variant, call, b)
                    assert np.abs(tran[b, :n] - rtran[b]).max() < 1e-3, (variant, call, b)
            assert net.device_error() == 0 and net.recovery_count == 0
            if not variant and B <= 4:
                # each sequence alone (B = 1, the same two calls): the same bits
                for b in sample:
                    n = lengths[b]
                    with MobilePoserNet.from_numpy(weights, smpl) as one:
                        for call in range(2):
                            p1, j1, t1, c1 = [npy(t) for t in one.forward_offline(cu(torch_mod, imu[b:b + 1, :n]), [n])]
                            assert np.array_equal(j1[0, :n], got[call][1][b, :n]), (b, call)
                            assert np.array_equal(c1.reshape(-1, 2)[:n], got[call][3][b, :n]), (b, call)
                            assert np.array_equal(t1.reshape(-1, 3)[:n], got[call][2][b, :n]), (b, call)
                # streams: B streams on one handle against B one-stream handles
                frames = synthetic.make_imu(B, 30, seed=97)
                net.reset_all()
                net.stream_create(B)
                ticks = []
                for k in range(30):
                    ticks.append([npy(t) for t in net.stream_step(cu(torch_mod, frames[:, k]))])
                for b in sample:
                    with MobilePoserNet.from_numpy(weights, smpl) as one:
                        one.stream_create(1)
                        for k in range(30):
                            o1 = [npy(t) for t in one.stream_step(cu(torch_mod, frames[b:b + 1, k]))]
                            for a, bb in zip(o1, ticks[k]):
                                assert np.array_equal(a[0], bb[b]), (b, k)
                assert net.device_error() == 0 and net.recovery_count == 0
    for call in range(2):
        for a, b in zip(outs[""][call], outs["vec=0"][call]):
            assert np.abs(a - b).max() < 2e-5


def test_g18_lengths_none_is_time_major(torch_mod, weights, smpl):
    """forward(batch, None) / forward_offline(imu, None) as the reference computes them (golden G18; SURVEY Q3: without lengths the
    reference's nn.LSTM reads dim 0 as time): T sequences of B steps, a carried velocity state of batch T, forward_offline of
    one sequence = T one-step sequences + the solver.  Rounds 1-5 raised here."""
    from conftest import geodesic
    from mobileposer_amd.net import MobilePoserNet
    g = load_golden("g18_lengths_none.npz")
    with MobilePoserNet.from_numpy(weights, smpl) as net:
        for call in (0, 1):
            pose, joints, vel, contact = net.forward(cu(torch_mod, g["imu"]), None)
            assert tuple(pose.shape) == g[f"c{call}_pose"].shape and tuple(joints.shape) == g[f"c{call}_joints"].shape
            assert tuple(vel.shape) == g[f"c{call}_vel"].shape and tuple(contact.shape) == g[f"c{call}_contact"].shape
            assert np.abs(npy(joints) - g[f"c{call}_joints"]).max() < 1e-4, call
            assert np.abs(npy(vel) - g[f"c{call}_vel"]).max() < 1e-4, call
            assert np.abs(npy(contact) - g[f"c{call}_contact"]).max() < 1e-4, call
            assert geodesic(npy(pose), g[f"c{call}_pose"]).max() < 1e-4, call
        h, c = net.velocity.rnn_state
        assert tuple(h.shape) == g["vel_h"].shape
        assert np.abs(npy(h) - g["vel_h"]).max() < 1e-4 and np.abs(npy(c) - g["vel_c"]).max() < 1e-4
        net.reset_all()
        net.reset()
        pose, joints, tran, contact = net.forward_offline(cu(torch_mod, g["imu1"]), None)
        assert tuple(tran.shape) == g["off_tran"].shape and tuple(contact.shape) == g["off_contact"].shape
        assert geodesic(npy(pose), g["off_pose"]).max() < 1e-4
        assert np.abs(npy(joints) - g["off_joints"]).max() < 1e-4 and np.abs(npy(contact) - g["off_contact"]).max() < 1e-4
        assert np.abs(npy(tran) - g["off_tran"]).max() < 1e-3
        assert net.device_error() == 0 and net.recovery_count == 0


def test_g19_submodule_views_golden(torch_mod, weights, smpl):
    """Golden G19: the sub-modules as the reference's own code calls them (net.py:103-117) -- net.joints(x, l), net.pose(x, l),
    net.foot_contact(x, l), net.velocity(x, l), net.velocity.forward_online(x, l) twice on the carried state (the ONE state
    MobilePoserNet.forward runs on) -- and each with input_lengths=None (dim 0 is time, rnn.py:15,25)."""
    from mobileposer_amd.net import MobilePoserNet
    g = load_golden("g19_submodules.npz")
    lengths = g["lengths"].tolist()
    with MobilePoserNet.from_numpy(weights, smpl) as net:
        for name, mod in (("joints", net.joints), ("pose", net.pose), ("foot_contact", net.foot_contact), ("velocity", net.velocity)):
            x = cu(torch_mod, g[f"{name}_x"])
            y = mod(x, lengths)
            assert tuple(y.shape) == g[f"{name}_y"].shape and np.abs(npy(y) - g[f"{name}_y"]).max() < 1e-4, name
            yn = mod.forward(x)
            assert tuple(yn.shape) == g[f"{name}_y_none"].shape and np.abs(npy(yn) - g[f"{name}_y_none"]).max() < 1e-4, name
            assert net.velocity.rnn_state is None                       # none of these touches the carried state
        xv = cu(torch_mod, g["velocity_x"])
        for tag, lens in (("online", lengths), ("online_none", None)):
            net.velocity.rnn_state = None
            for call in (0, 1):
                y = net.velocity.forward_online(xv, lens)
                assert np.abs(npy(y) - g[f"{tag}{call}"]).max() < 1e-4, (tag, call)
            h, c = net.velocity.rnn_state
            assert tuple(h.shape) == g[f"{tag}_h"].shape
            assert np.abs(npy(h) - g[f"{tag}_h"]).max() < 1e-4 and np.abs(npy(c) - g[f"{tag}_c"]).max() < 1e-4
        # a carried state of another batch size is refused (nn.LSTM raises there too) -- not read as if it fitted
        with pytest.raises(RuntimeError):
            net.velocity.forward_online(xv[:2].contiguous(), lengths[:2])
        with pytest.raises(RuntimeError):
            net.rnn_forward("joints", cu(torch_mod, g["joints_x"]), lengths, (torch_mod.zeros(4, 2, 256, device="cuda"), torch_mod.zeros(4, 2, 256, device="cuda")))
        with pytest.raises(RuntimeError):                         # a sub-module fed the other one's input width
            net.pose(cu(torch_mod, g["joints_x"]), lengths)
        with pytest.raises(RuntimeError):
            net.velocity.rnn_state = (torch_mod.zeros(2, 3, 128, device="cuda"), torch_mod.zeros(2, 3, 128, device="cuda"))
        # poser.py:52-58 is net.py:93-99
        g3 = load_golden("g3_r6d_ik.npz")
        assert np.abs(npy(net.pose._reduced_global_to_full(cu(torch_mod, g3["r6d"]))) - g3["pose"]).max() < 1e-5
        assert net.device_error() == 0 and net.recovery_count == 0


def test_g20_rotation_kinematics_golden_and_edges(torch_mod, weights, smpl):
    """Golden G20: ParametricModel.forward_kinematics_R / inverse_kinematics_R (articulate/model.py:126-164) and
    MobilePoserNet.global_to_local_pose (net.py:38) through mp_fk / mp_inverse_kinematics_r -- bound to a net and on a body-only
    handle; every frame count around the 8-frame workgroup; a buffer that is not 16-byte aligned (the scalar kernel) bitwise what
    the staged kernel gives; an in-place call is refused."""
    import ctypes as C
    from mobileposer_amd.body_model import ParametricModel
    from mobileposer_amd.net import MobilePoserNet
    from oracle import mp_oracle as O
    g = load_golden("g20_rotation_kinematics.npz")
    R = cu(torch_mod, g["R"])
    with MobilePoserNet.from_numpy(weights, smpl) as net:
        bm = net.bodymodel
        assert np.abs(npy(bm.forward_kinematics_R(R)) - g["fk_R"]).max() < 1e-5
        assert np.abs(npy(bm.forward_kinematics_R(R.reshape(37, -1))) - g["fk_R"]).max() < 1e-5      # "[batch_size, *]"
        loc = npy(net.global_to_local_pose(R))
        assert loc.shape == g["ik_R"].shape and np.abs(loc - g["ik_R"]).max() < 1e-5
        assert np.abs(npy(bm.inverse_kinematics_R(cu(torch_mod, g["fk_R"]))) - g["ik_of_fk"]).max() < 1e-5
        assert float(net.gravity_velocity[1]) == pytest.approx(-0.018) and tuple(net.last_joints.shape) == (24, 3)
        # frame counts around a workgroup's 8 frames, against the oracle; unaligned buffers bitwise the same
        big = cu(torch_mod, np.concatenate([g["fk_R"]] * 3))
        pad = torch_mod.empty(big.numel() + 1, device="cuda")
        pad[1:] = big.reshape(-1)
        for n in (1, 7, 8, 9, 16, 17, 100):
            got = bm.inverse_kinematics_R(big[:n])
            assert np.abs(npy(got) - O.inverse_kinematics_R(npy(big[:n]))).max() < 1e-6, n
            out = torch_mod.empty(n * 216 + 1, device="cuda")
            rc = net._lib.mp_inverse_kinematics_r(net._h, C.c_void_p(pad.data_ptr() + 4), n, C.c_void_p(out.data_ptr() + 4), None)
            assert rc == 0
            torch_mod.cuda.synchronize()
            assert np.array_equal(npy(out[1:]).reshape(n, 24, 3, 3), npy(got)), n
        rc = net._lib.mp_inverse_kinematics_r(net._h, C.c_void_p(big.data_ptr()), 4, C.c_void_p(big.data_ptr()), None)
        assert rc != 0 and "in-place" in net._lib.mp_last_error(net._h).decode()
        assert net.device_error() == 0
    body = ParametricModel(data=smpl)                       # a body-only handle (mp_create_body), as data.py:24 builds one
    try:
        assert np.abs(npy(body.inverse_kinematics_R(R)) - g["ik_R"]).max() < 1e-5
        assert np.abs(npy(body.forward_kinematics_R(R)) - g["fk_R"]).max() < 1e-5
    finally:
        body.close()


@pytest.mark.parametrize("B", [4, 5, 16, 17, 32, 33, 64, 65, 96, 97, 128, 129])
def test_schedule_boundaries_vs_oracle(torch_mod, weights, smpl, B):
    """Every batch size at which the library changes the kernels or the schedule of a forward (csrc/mp_schedule.hip: per-sequence
    clusters up to 4 sequences, the 32-slice kernels up to 32, three blocks side by side up to 64, the half-chip schedules up to 96 /
    128, the one-stream schedule above) and the size right behind it: a ragged forward_offline against the oracle, twice -- the
    second call runs on the velocity state the first one left."""
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    from oracle import mp_oracle as O
    T = 11
    rng = np.random.Generator(np.random.PCG64(600 + B))
    lengths = rng.integers(1, T + 1, size=B).tolist()
    lengths[int(rng.integers(0, B))] = T
    imu = synthetic.make_imu(B, T, seed=600 + B)
    ref = O.OracleNet(weights, smpl["J"])
    with MobilePoserNet.from_numpy(weights, smpl) as net:
        x = cu(torch_mod, imu)
        for call in (0, 1):
            pose, joints, vel, contact = net.forward(x, lengths)
            rpose, rjoints, rvel, rcontact = ref.forward(imu, lengths)
            assert np.abs(npy(joints) - rjoints).max() < 1e-4, (B, call)
            assert np.abs(npy(vel).reshape(rvel.shape) - rvel).max() < 1e-4, (B, call)
            assert np.abs(npy(contact) - rcontact).max() < 1e-4, (B, call)
            assert geodesic(npy(pose), rpose).max() < 1e-4, (B, call)
        h, c = net.velocity.rnn_state
        assert np.abs(npy(h) - ref.velocity_rnn_state[0]).max() < 1e-4 and np.abs(npy(c) - ref.velocity_rnn_state[1]).max() < 1e-4
        assert net.device_error() == 0 and net.recovery_count == 0
