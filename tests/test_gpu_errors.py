"""Error behaviour of the fused (persistent) LSTM launches -- include/mobileposer_hip.h, "error behaviour":
a starved grid gives up its waits after a TIME bound, leaves NaN (never plausible numbers) and an error word behind;
with recovery on (default) the call repairs itself with per-step kernels and returns valid results; with recovery off
the error surfaces at the next entry / mp_finish / close().  Starvation is produced by a test hook that makes one workgroup of
a layer grid exit at once -- what its cluster sees when that workgroup never becomes resident on a shared GPU."""
import ctypes as C
import time
import warnings

import numpy as np
import pytest

from conftest import cu, npy

pytestmark = pytest.mark.gpu


def test_poked_error_is_reported_once_by_the_next_entry(torch_mod, weights, smpl):
    """The error word of an EARLIER call (poked the way a kernel stores it: system-scope store into the pinned host word)
    makes the next API entry return MP_ERR_DEVICE once; the handle stays usable."""
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    x = cu(torch_mod, synthetic.make_imu(2, 8, seed=1))
    with MobilePoserNet.from_numpy(weights, smpl) as m:
        m.forward(x, [8, 8])
        assert m.device_error() == 0
        assert m._lib.mp_debug_poke_error(m._h, 43) == 0
        with pytest.raises(RuntimeError, match="gave up a wait"):
            m.forward(x, [8, 8])
        m.forward(x, [8, 8])                          # reported once
        assert m.device_error() == 0
        assert m._lib.mp_debug_poke_error(m._h, 44) == 0
        with pytest.raises(RuntimeError, match="code 44"):
            m.finish()
        m.finish()


def _starve(m, skip=0, launches=1):
    """After `skip` fused-LSTM layer launches, workgroup 8 (slice 1 of a cluster) of the next one(s) never shows up."""
    assert m._lib.mp_debug_drop_workgroup(m._h, 8, skip, launches) == 0


@pytest.mark.parametrize("mode", [1, 3])
def test_starved_call_repairs_itself(torch_mod, weights, smpl, monkeypatch, mode):
    """Recovery on (default): the starved forward_offline returns the same values as an undisturbed one (to the
    difference between the fused and the per-step kernels), warns, and counts one recovery; the carried velocity state
    it started from is put back before the re-run (second call of a pair, quirk Q1)."""
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    monkeypatch.setenv("MP_WAIT_MS", "15")                        # the 0.25 s bound, shortened for the test
    B, T = 256, 24
    x = cu(torch_mod, synthetic.make_imu(B, T, seed=77))
    with MobilePoserNet.from_numpy(weights, smpl) as m:
        m.set_lstm_mode(mode)
        m.set_recovery(True)
        first = [t.clone() for t in m.forward_offline(x, [T] * B)]
        want = [t.clone() for t in m.forward_offline(x, [T] * B)]      # second call: velocity state carried
        assert m.recovery_count == 0
        m.reset_all()
        got1 = [t.clone() for t in m.forward_offline(x, [T] * B)]
        for a, b in zip(first, got1):
            assert torch_mod.equal(a, b)
        _starve(m)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            got = m.forward_offline(x, [T] * B)
        assert m.recovery_count == 1 and any("starved" in str(i.message) for i in w), [str(i.message) for i in w]
        for a, b in zip(want, got):
            assert bool(torch_mod.isfinite(b).all())
            assert float((a - b).abs().max()) < 2e-5
        m.reset_all()
        again = m.forward_offline(x, [T] * B)                     # back on the fused kernels, undisturbed
        for a, b in zip(first, again):
            assert float((a - b).abs().max()) < 2e-5              # (XCD tables are off now: 16-slice / serial schedule may differ)
        assert m.recovery_count == 1 and m.device_error() == 0


def test_starved_single_sequence_launches_repair_themselves(torch_mod, weights, smpl, monkeypatch):
    """B = 1 (round 5: mp_lstm_v1 -- granule hand-off, every wave polls for itself; the velocity block as a two-layer wavefront on
    one XCD): a workgroup that never shows up in the k-th layer launch ends every wait of its cluster -- and of the other layer of a
    wavefront -- in the time bound; the call reports it, is re-run on the per-step kernels and returns the undisturbed values.
    (mp_lstm_v1s, the H = 64 block in one workgroup, has no waits: the hook does nothing there.)"""
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    monkeypatch.setenv("MP_WAIT_MS", "15")
    monkeypatch.setenv("MP_VARIANT", "")
    T = 60
    x = cu(torch_mod, synthetic.make_imu(1, T, seed=79))
    repaired = 0
    with MobilePoserNet.from_numpy(weights, smpl) as m:
        m.set_lstm_mode(1)
        m.set_recovery(True)
        want = [t.clone() for t in m.forward_offline(x, [T])]
        for skip in range(7):
            m.reset_all()
            before = m.recovery_count
            _starve(m, skip=skip)
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                got = m.forward_offline(x, [T])
            assert m.recovery_count - before in (0, 1)
            if m.recovery_count > before:
                repaired += 1
                assert any("starved" in str(i.message) for i in w), [str(i.message) for i in w]
            for a, b in zip(want, got):
                assert bool(torch_mod.isfinite(b).all())
                assert float((a - b).abs().max()) < 2e-5, (skip, float((a - b).abs().max()))
            assert m._lib.mp_debug_drop_workgroup(m._h, 0, 0, 0) == 0       # disarm (a launch without the hook left it armed)
            assert m.device_error() == 0
        m.reset_all()
        again = m.forward_offline(x, [T])
        for a, b in zip(want, again):
            assert float((a - b).abs().max()) < 2e-5
    assert repaired >= 5, repaired          # joints L0 / L1, pose L0 / L1, the velocity wavefront


@pytest.mark.parametrize("S", [1, 64])
def test_starved_streaming_tick_repairs_itself(torch_mod, weights, smpl, monkeypatch, S):
    """A streaming tick (forward_online semantics) that loses a workgroup, recovery on: the state the tick started from --
    velocity h / c, last foot positions, root height, root position, snapshotted by ONE launch (mp_copy_words, round 5) -- is put
    back and network + solver are re-run on the per-step kernels (the frame is not pushed a second time).  The starved tick and
    every later one equal those of an undisturbed model fed the same frames."""
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    monkeypatch.setenv("MP_WAIT_MS", "15")
    monkeypatch.setenv("MP_VARIANT", "")
    frames = cu(torch_mod, synthetic.make_imu(S, 9, seed=83))
    with MobilePoserNet.from_numpy(weights, smpl) as m, MobilePoserNet.from_numpy(weights, smpl) as fresh:
        for n in (m, fresh):
            n.set_lstm_mode(1)
            n.set_recovery(True)
            n.stream_create(S)
        for k in range(9):
            if k in (3, 6):
                _starve(m, skip=(0 if k == 3 else 2))                    # joints layer 0 | a later layer launch of the tick
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                got = [t.clone() for t in m.stream_step(frames[:, k])]
            want = fresh.stream_step(frames[:, k])
            for a, b in zip(want, got):
                assert bool(torch_mod.isfinite(b).all()), k
                assert float((a - b).abs().max()) < 2e-5, (k, float((a - b).abs().max()))
            assert m._lib.mp_debug_drop_workgroup(m._h, 0, 0, 0) == 0
        assert m.recovery_count == 2 and fresh.recovery_count == 0
        hm, cm = m.velocity.rnn_state
        hf, cf = fresh.velocity.rnn_state
        assert float((hm - hf).abs().max()) < 2e-5 and float((cm - cf).abs().max()) < 2e-5
        for name in ("last_root_pos", "last_lfoot_pos", "last_rfoot_pos"):
            assert float((m.stream_state(0)[name] - fresh.stream_state(0)[name]).abs().max()) < 2e-5, name
        assert m.device_error() == 0


def test_starved_call_without_recovery_is_loud(torch_mod, weights, smpl, monkeypatch):
    """Recovery off: the starved call returns at once (asynchronous); its outputs are NaN, never plausible numbers;
    finish() raises MP_ERR_DEVICE; afterwards the handle works again."""
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    monkeypatch.setenv("MP_WAIT_MS", "15")
    B, T = 256, 24
    x = cu(torch_mod, synthetic.make_imu(B, T, seed=78))
    with MobilePoserNet.from_numpy(weights, smpl) as m:
        m.set_lstm_mode(1)
        m.set_recovery(False)
        want = [t.clone() for t in m.forward_offline(x, [T] * B)]
        m.reset_all()
        _starve(m)
        got = [t.clone() for t in m.forward_offline(x, [T] * B)]
        with pytest.raises(RuntimeError, match="gave up a wait"):
            m.finish()
        assert any(bool(torch_mod.isnan(t).any()) for t in got), "a starved grid must not leave plausible numbers"
        joints = got[1]
        bad = torch_mod.isnan(joints).flatten(1).any(dim=1)          # rows (sequences) whose slab was poisoned
        ok_rows = (~bad).nonzero().flatten().tolist()
        # rows that are not NaN are right (a slab either finished undisturbed or is poisoned as a whole)
        if ok_rows:
            assert float((joints[ok_rows] - want[1][ok_rows]).abs().max()) < 1e-4
        m.reset_all()
        again = m.forward_offline(x, [T] * B)
        m.finish()
        for a, b in zip(want, again):
            assert float((a - b).abs().max()) < 2e-5


def test_starved_velocity_launch_poisons_its_rider_too(torch_mod, weights, smpl, monkeypatch):
    """B = 256, exact-fp32: foot-contact layer 1 rides in the velocity wavefront launch (round 5: the fifth fused launch of a
    forward, both velocity layers in one; rounds 3-4: the fifth and sixth, one per layer).  A workgroup missing there -- block 8
    = slice 1 of layer 0 of slab 0 -- starves that cluster AND the layer-1 cluster behind the link, and leaves NaN -- not
    plausible numbers -- in the slab's velocity AND contact rows, and therefore in its translation; joints and pose (earlier
    launches) are untouched.  With recovery the call is repaired."""
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    monkeypatch.setenv("MP_WAIT_MS", "15")
    monkeypatch.setenv("MP_VARIANT", "")                                   # (this is about the default schedule)
    B, T = 256, 20
    x = cu(torch_mod, synthetic.make_imu(B, T, seed=79))
    with MobilePoserNet.from_numpy(weights, smpl) as m:
        m.set_lstm_mode(1)
        m.set_recovery(False)
        want = [t.clone() for t in m.forward_offline(x, [T] * B)]          # pose, joints, tran, contact
        m.reset_all()
        _starve(m, skip=4)                                                  # joints L0/L1, pose L0/L1, then the velocity wavefront
        got = [t.clone() for t in m.forward_offline(x, [T] * B)]
        with pytest.raises(RuntimeError, match="gave up a wait"):
            m.finish()
        assert torch_mod.equal(got[0], want[0]) and torch_mod.equal(got[1], want[1])
        bad_c = torch_mod.isnan(got[3]).flatten(1).any(dim=1)
        bad_t = torch_mod.isnan(got[2]).flatten(1).any(dim=1)
        assert bool(bad_c.any()) and bool(bad_t.any()), "the starved slab's contact and translation rows must be NaN"
        ok = (~(bad_c | bad_t)).nonzero().flatten().tolist()
        if ok:
            assert float((got[3][ok] - want[3][ok]).abs().max()) < 1e-4 and float((got[2][ok] - want[2][ok]).abs().max()) < 1e-3
        m.reset_all()
        m.set_recovery(True)
        _starve(m, skip=4)
        with warnings.catch_warnings(record=True):
            warnings.simplefilter("always")
            rep = m.forward_offline(x, [T] * B)
        assert m.recovery_count == 1
        for a, b in zip(want, rep):
            assert float((a - b).abs().max()) < 2e-5


def test_close_reports_an_unreported_error(torch_mod, weights, smpl):
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    m = MobilePoserNet.from_numpy(weights, smpl)
    m.set_recovery(False)
    m.forward(cu(torch_mod, synthetic.make_imu(2, 8, seed=1)), [8, 8])
    assert m._lib.mp_debug_poke_error(m._h, 7) == 0
    with pytest.raises(RuntimeError, match="code 7"):
        m.close()
    assert m._h is None


def test_state_attributes_can_be_assigned(torch_mod, net):
    """The reference's per-stream variables are plain attributes (net.py:59-64): reads AND writes work, and repeated
    reads between two ticks share one device round trip."""
    from mobileposer_amd import synthetic
    frames = cu(torch_mod, synthetic.make_imu(1, 3, seed=5)[0])
    net.reset_all()
    net.forward_online(frames[0])
    a = net.last_root_pos
    assert net._state_cache[0][0] == net._tick                      # one round trip per tick: the second read is served from it
    b = net.last_root_pos
    assert b is not a and torch_mod.equal(a, b)                     # ... as a COPY: an in-place edit of a value read earlier
    a[1] = 123.0                                                    # does not masquerade as device state (only assignment
    assert float(net.last_root_pos[1]) != 123.0                     # writes through)
    net.last_root_pos = torch_mod.tensor([1.0, 2.0, 3.0])
    assert npy(net.last_root_pos).tolist() == [1.0, 2.0, 3.0]
    net.current_root_y = 0.25
    assert abs(net.current_root_y - 0.25) < 1e-12
    net.last_lfoot_pos = torch_mod.tensor([0.1, 0.2, 0.3])
    assert np.allclose(npy(net.last_lfoot_pos), [0.1, 0.2, 0.3]) and np.allclose(npy(net.last_rfoot_pos), npy(net.feet_pos[1]), atol=1.0)
    _, _, root, _ = net.forward_online(frames[1])
    assert abs(float(root[0]) - 1.0) < 0.5                           # the tick continued from the assigned position
    net.imu = None                                                  # what reset() does to the window
    assert net.imu is None
    net.forward_online(frames[2])
    win = net.imu
    assert win is not None and float((win - frames[2]).abs().max()) == 0.0    # fresh window = the frame repeated 45 times
    assert net.device_error() == 0


@pytest.mark.parametrize("B", [256, 128, 1])
def test_two_handles_two_threads_overlapping_full_chip_calls(torch_mod, weights, smpl, monkeypatch, B):
    """Real contention: two handles driven from two host threads, each issuing full-chip (256-workgroup) fused-LSTM launches
    at the same time (ctypes releases the GIL inside a call).  Every call plans for a GPU it has to itself, so grids of the
    two handles can starve each other; with recovery on every call still returns the undisturbed result -- repaired calls
    are counted, none raises, nothing hangs (all waits are time-bounded).  B = 128: each handle itself runs two grids side by
    side on host-chosen XCDs (schedule 4) until its first repaired call switches the placement tables off.  B = 1 (round 5): the
    one-sequence kernels of both handles want the same XCDs -- whole XCDs for a cluster, two workgroups per CU for the velocity
    wavefront -- so their clusters may be resident only in part."""
    import threading
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    monkeypatch.setenv("MP_WAIT_MS", "40")
    T, reps = (16, 6) if B > 1 else (400, 12)
    xs = [cu(torch_mod, synthetic.make_imu(B, T, seed=90 + k)) for k in range(2)]
    nets = [MobilePoserNet.from_numpy(weights, smpl) for _ in range(2)]
    try:
        want = []
        for n, x in zip(nets, xs):                       # undisturbed references, one handle at a time
            n.reset_all()
            want.append([t.clone() for t in n.forward_offline(x, [T] * B)])
        torch_mod.cuda.synchronize()
        errors, worst = [], [0.0, 0.0]

        def work(k):
            try:
                stream = torch_mod.cuda.Stream()
                with torch_mod.cuda.stream(stream), warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    for _ in range(reps):
                        nets[k].reset_all()
                        got = nets[k].forward_offline(xs[k], [T] * B)
                        stream.synchronize()
                        for a, b in zip(want[k], got):
                            assert bool(torch_mod.isfinite(b).all())
                            worst[k] = max(worst[k], float((a - b).abs().max()))
            except Exception as e:                          # noqa: BLE001 -- reported by the main thread
                errors.append((k, repr(e)))

        threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
        for th in threads:
            th.start()
        for th in threads:
            th.join(timeout=300)
        assert not any(th.is_alive() for th in threads), "a call hung"
        assert not errors, errors
        assert max(worst) < 2e-5, worst
        print("recoveries:", [n.recovery_count for n in nets], "worst diff:", worst)
        for n in nets:
            assert n.device_error() == 0
    finally:
        for n in nets:
            n.close()


@pytest.mark.parametrize("skip,what", [(3, "velocity layer 0 + rider"), (4, "pose layer 1")])
def test_starved_launch_in_the_half_chip_schedule(torch_mod, weights, smpl, monkeypatch, skip, what):
    """B = 128, exact-fp32 (schedule 4 of profiles/NOTES_r01-r03.md section 4): fused launches are issued in the order joints L0, joints L1,
    pose L0, velocity L0 (with the foot-contact rider), pose L1, velocity L1 -- pose L1 and the velocity layers run side by side
    on disjoint halves of the chip.  A workgroup missing from one of the two concurrent grids: without recovery NaN in what
    that grid feeds (and only there), finish() raises; with recovery the call is repaired; the other grid is not disturbed."""
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    monkeypatch.setenv("MP_WAIT_MS", "15")
    monkeypatch.setenv("MP_VARIANT", "")                                   # (this is about the default schedule)
    B, T = 128, 20
    x = cu(torch_mod, synthetic.make_imu(B, T, seed=81))
    with MobilePoserNet.from_numpy(weights, smpl) as m:
        m.set_lstm_mode(1)
        m.set_recovery(False)
        want = [t.clone() for t in m.forward_offline(x, [T] * B)]          # pose, joints, tran, contact
        m.reset_all()
        _starve(m, skip=skip)
        got = [t.clone() for t in m.forward_offline(x, [T] * B)]
        with pytest.raises(RuntimeError, match="gave up a wait"):
            m.finish()
        assert torch_mod.equal(got[1], want[1]), "joints ran before the starved launch"
        nan = [bool(torch_mod.isnan(t).any()) for t in got]
        if skip == 3:                                                       # velocity + rider: translation and contact
            assert nan[2] and nan[3] and not nan[0], nan
            assert torch_mod.equal(got[0], want[0]), "pose layer 1 ran beside the starved grid and must not notice"
        else:                                                               # pose layer 1: poses (and nothing else)
            # the poisoned r6d rows reach _reduced_global_to_full, whose own rule turns NaN into 0 (net.py:110, quirk Q8):
            # the starved slab's poses are all-zero matrices -- no rotation, never a plausible pose
            assert not nan[2] and not nan[3], nan
            assert torch_mod.equal(got[3], want[3]) and torch_mod.equal(got[2], want[2])
            p_got, p_want = got[0].reshape(-1, 24, 3, 3), want[0].reshape(-1, 24, 3, 3)
            bad = (p_got != p_want).flatten(1).any(dim=1)                    # frames of the starved slab
            assert bool(bad.any()) and int(bad.sum()) <= 16 * T, int(bad.sum())
            assert float(p_got[bad][:, 0].abs().max()) == 0.0               # (the root of every such frame: the zero matrix)
        m.reset_all()
        m.set_recovery(True)
        _starve(m, skip=skip)
        with warnings.catch_warnings(record=True):
            warnings.simplefilter("always")
            rep = m.forward_offline(x, [T] * B)
        assert m.recovery_count == 1, what
        for a, b in zip(want, rep):
            assert bool(torch_mod.isfinite(b).all()) and float((a - b).abs().max()) < 2e-5
