"""Timeline of ONE forward_offline: start offset / duration / queue of every kernel (rocprofv3 --kernel-trace).

    cd /tmp && python $GRAFT_REPO_ROOT/tools/debug/timeline.py [B] [T]      (on the GPU box; B = 0: one forward_online tick of one stream)

Phase 1 (no argument `--child`): runs itself under rocprofv3; phase 2 parses the kernel trace and prints the last forward.
"""
import csv, glob, os, re, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child_tick():
    """one stream, forward_online calls (live_demo.py:238-241): the last tick is printed"""
    import torch
    sys.path.insert(0, ROOT)
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    net = MobilePoserNet.from_numpy(synthetic.make_weights(0), synthetic.synthetic_smpl())
    net.set_lstm_mode(1)
    frames = torch.from_numpy(synthetic.make_imu(1, 80, seed=1)[0]).cuda()
    for f in frames:
        net.forward_online(f)
        torch.cuda.synchronize()
    for f in frames[:6]:
        time.sleep(0.05)
        net.forward_online(f)
        torch.cuda.synchronize()


def child(B, T):
    if B == 0:
        return child_tick()
    import torch
    sys.path.insert(0, ROOT)
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    net = MobilePoserNet.from_numpy(synthetic.make_weights(0), synthetic.synthetic_smpl())
    net.set_lstm_mode(int(os.environ.get("MP_TL_MODE", "1")))        # 1 = exact fp32 (default), 3 = split-fp16
    import ctypes as C
    x = torch.from_numpy(synthetic.make_imu(B, T, seed=1)).cuda()
    mk = lambda *sh: torch.empty(*sh, device="cuda", dtype=torch.float32)
    o = [mk(B * T, 24, 3, 3), mk(B, T, 72), mk(B, T, 72), mk(B, T, 2), mk(B, T, 3), mk(B * T, 24, 3, 3), mk(B * T, 24, 3)]
    lens = (C.c_int32 * B)(*([T] * B))
    for _ in range(6):                 # the call bench.py times: forward + FK + solver
        net.reset_all()
        net.forward_offline_into(x, lens, *o)
        torch.cuda.synchronize()
        time.sleep(0.05)


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*", "", n)
    return n.replace("void ", "")[:44]


def main():
    if "--child" in sys.argv:
        return child(int(sys.argv[2]), int(sys.argv[3]))
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 125
    out = "/tmp/tl_out"
    subprocess.run("rm -rf %s; rocprofv3 --kernel-trace --output-format csv -d %s -- %s %s --child %d %d > /tmp/tl.log 2>&1"
                   % (out, out, sys.executable, os.path.abspath(__file__), B, T), shell=True, env=dict(os.environ, TMPDIR="/tmp"))
    f = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)
    if not f:
        print(open("/tmp/tl.log").read()[-2000:])
        return
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last forward = everything after the last idle gap > 20 ms (the child sleeps 50 ms between forwards)
    cut, hi = 0, 0
    for i in range(len(rows)):
        if i and int(rows[i]["Start_Timestamp"]) - hi > 20000000:
            cut = i
        hi = max(hi, int(rows[i]["End_Timestamp"]))
    rows = rows[cut:]
    t0 = int(rows[0]["Start_Timestamp"])
    end = max(int(r["End_Timestamp"]) for r in rows)
    print("%s: %d kernels, %.1f us from first start to last end" % ("forward_offline %d x %d" % (B, T) if B else "forward_online, one stream (S = 1 tick)", len(rows), (end - t0) / 1e3))
    print("%9s %9s %9s  %-6s %s" % ("start us", "dur us", "end us", "queue", "kernel  [grid x block]"))
    for r in rows:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        grid = r.get("Grid_Size_X", r.get("Grid_Size", "?"))
        wg = r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))
        print("%9.1f %9.1f %9.1f  %-6s %s  [%s x %s]" % (s / 1e3, (e - s) / 1e3, e / 1e3, r.get("Queue_Id", "?"), short(r["Kernel_Name"]), grid, wg))


main()
