"""What does a GPU that sat idle cost, and which clock is it?  (verdict r5 item 4: "root-cause the post-idle slowdown")

The 512-stream tick costs 2.75 ms back to back and 3.1 - 3.3 ms at a 30 / 60 Hz cadence; the first tick after an idle stretch is
the slow one and the next six to eight get faster one by one.  mp_debug_clock_probe says the SHADER clock is where it always is
(2.39 - 2.42 GHz) in front of and behind such a tick.  This tool separates the other suspects, each as "warm up, sit idle for
idle_s, then 12 items back to back, each timed by itself":

  ticks      the streaming tick (S streams)                         -- the thing that is slow
  copy256    a 256 MB device-to-device copy (HIP events)            -- HBM / fabric bandwidth: memory-side clocks
  copy4      a 4 MB copy (stays in L2 / the Infinity Cache)         -- on-chip bandwidth
  probe      mp_debug_clock_probe: shader MHz and the duration of a fixed dependent FMA chain on one wave (compute only)
  loaded     mp_debug_clock_probe_loaded: both clocks and the duration of a fixed MFMA stream on EVERY wave slot of the chip
  launch     an empty 1-element kernel, host-timed submit -> done   -- wake-up / launch latency

and reads the DPM levels the driver publishes in sysfs (pp_dpm_sclk / mclk / fclk / socclk) when the container shows them.
Then the same ticks with a keep-warm thread that issues a 1 MB copy every `period` ms on a side stream during the idle stretch.

  python tools/debug/idle_probe.py [S] [idle_s ...]
"""
import ctypes as C
import glob
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mobileposer_amd import synthetic                       # noqa: E402
from mobileposer_amd.net import MobilePoserNet              # noqa: E402


def dpm_levels():
    out = {}
    for f in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_*")):
        try:
            cur = [l.strip() for l in open(f).read().splitlines() if l.strip().endswith("*")]
            out[f.split("/")[4] + ":" + os.path.basename(f)[7:]] = cur[0] if cur else "?"
        except OSError:
            pass
    return out


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    idles = [float(a) for a in sys.argv[2:]] or [0.016, 0.05, 1.0]
    dev = torch.device("cuda", 0)
    net = MobilePoserNet.from_numpy(synthetic.make_weights(0), synthetic.synthetic_smpl(), device=dev)
    lib, h = net._lib, net._h
    frames = torch.from_numpy(synthetic.make_imu(S, 400, seed=7)).to(dev)
    net.stream_create(S)
    f32 = torch.float32
    io = [torch.empty(S, 24, 9, device=dev, dtype=f32), torch.empty(S, 45, 72, device=dev, dtype=f32),
          torch.empty(S, 3, device=dev, dtype=f32), torch.empty(S, 2, device=dev, dtype=f32)]
    xin = torch.empty(S, 60, device=dev, dtype=f32)
    cur = [0]
    big_a, big_b = torch.empty(64 << 20, device=dev, dtype=f32), torch.empty(64 << 20, device=dev, dtype=f32)
    sm_a, sm_b = torch.empty(1 << 20, device=dev, dtype=f32), torch.empty(1 << 20, device=dev, dtype=f32)
    one = torch.zeros(1, device=dev)

    def tick():
        xin.copy_(frames[:, cur[0] % 400]); cur[0] += 1
        t0 = time.perf_counter()
        net.stream_step_into(xin, *io)
        torch.cuda.synchronize(dev)
        return 1e3 * (time.perf_counter() - t0)

    def copy(a, b):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); b.copy_(a); e1.record(); e1.synchronize()
        return e0.elapsed_time(e1)

    def probe():
        mhz, us = C.c_double(0), C.c_double(0)
        lib.mp_debug_clock_probe(h, C.byref(mhz), C.byref(us))
        return mhz.value, us.value

    def loaded(iters=2048):           # 8 192 MFMAs per wave on every SIMD of the chip: ~220 us at 2.4 GHz
        m, mn, us, usx = C.c_double(0), C.c_double(0), C.c_double(0), C.c_double(0)
        lib.mp_debug_clock_probe_loaded(h, iters, C.byref(m), C.byref(mn), C.byref(us), C.byref(usx))
        return round(m.value, 0), round(mn.value, 0), round(us.value, 1), round(usx.value, 1)

    def launch():
        t0 = time.perf_counter(); one.add_(1.0); torch.cuda.synchronize(dev)
        return 1e3 * (time.perf_counter() - t0)

    def warm():
        for _ in range(60):
            tick()

    items = {
        "ticks (ms)": lambda: round(tick(), 3),
        "copy256 (GB/s, read + write)": lambda: round(2 * 256 / 1024 / (copy(big_a, big_b) * 1e-3), 0),
        "copy4 (GB/s)": lambda: round(2 * 4 / 1024 / (copy(sm_a, sm_b) * 1e-3), 0),
        "probe (MHz, us)": lambda: tuple(round(v, 1) for v in probe()),
        "loaded (MHz mean, min, us mean, max)": loaded,
        "launch (ms)": lambda: round(launch(), 3),
    }
    print("sysfs DPM levels, busy:", end=" ")
    warm(); print(dpm_levels() or "(no pp_dpm_* files visible in this container)")
    time.sleep(1.0)
    print("sysfs DPM levels after 1 s idle:", dpm_levels() or "(none)")
    warm()
    back = [round(tick(), 3) for _ in range(12)]
    print("S = %d  ticks back to back (ms): %s" % (S, back))
    for idle in idles:
        print("---- idle %.0f ms" % (idle * 1e3))
        for name, fn in items.items():
            warm()
            time.sleep(idle)
            if name.startswith("loaded"):     # (a 20 ms ramp is ~90 of these probes: every 8th of 104)
                vals = [fn() for _ in range(104)]
                print("  %-30s %s" % (name + " every 8th", vals[::8]))
            else:
                print("  %-30s %s" % (name, [fn() for _ in range(12)]))
    # ---- keep-warm: does traffic during the idle stretch keep the first tick fast?
    side = torch.cuda.Stream(device=dev)
    for period_ms, what in ((0.5, "1 MB copy"), (2.0, "1 MB copy"), (0.5, "64 MB copy")):
        src, dst = (sm_a[: 1 << 18], sm_b[: 1 << 18]) if what == "1 MB copy" else (big_a[: 16 << 20], big_b[: 16 << 20])
        stop = threading.Event()

        def keeper():
            with torch.cuda.stream(side):
                while not stop.is_set():
                    dst.copy_(src)
                    time.sleep(period_ms * 1e-3)

        for idle in idles:
            warm()
            stop.clear()
            th = threading.Thread(target=keeper); th.start()
            time.sleep(idle)
            stop.set(); th.join(); side.synchronize()
            print("keep-warm (%s every %.1f ms) idle %.0f ms: ticks %s" % (what, period_ms, idle * 1e3, [round(tick(), 3) for _ in range(8)]))
    net.close()


if __name__ == "__main__":
    main()
