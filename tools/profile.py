"""rocprofv3 evidence for bench.py, run ON THE GPU BOX:  python tools/profile.py <tag> [bench args...]

  1. rocprofv3 --kernel-trace --stats  of  python bench.py --steps 20 --warmup 5 --no-cpu-baseline <bench args>
  2. four rocprofv3 --pmc passes (one counter group per pass, with --kernel-trace only -- never combined with
     sys/hip/hsa traces) of the same command with --steps 3 --warmup 1
  3. gpurun_out/<tag>_kernel_stats.{csv,md} and gpurun_out/<tag>_pmc_summary.json, which get copied into profiles/.

Counter arithmetic follows /opt/skills/guides/MI355X_MICROARCH.md: GRBM_GUI_ACTIVE is summed over the 8 XCDs;
HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (FETCH_SIZE reads half of wide coalesced streaming reads on gfx950).
"""
import csv
import glob
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "gpurun_out")
sys.path.insert(0, REPO)
PMC_GROUPS = [["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"], ["FETCH_SIZE"], ["WRITE_SIZE"],
              ["SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_VALU_MFMA_MOPS_F32"]]


def short(name):
    name = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    for tail in ("(LstmPersistArgs)", "(LstmStepArgs)", "(GemmArgs, int, int)"):  # noqa
        name = name.replace(tail, "")
    return name.split("(")[0] if name.startswith("mp_") or name.startswith("__amd") else name[:70]


def run(cmd):
    env = dict(os.environ, TMPDIR="/tmp")
    print("+", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    return r.returncode, r.stdout


def main():
    tag = sys.argv[1]
    extra = sys.argv[2:]
    # --legs headline: the legs of the other BASELINE configs (ticks of 512 streams at T = 45, one 3000-frame sequence, the joints
    # module alone) launch the SAME kernels at other shapes; in the default command they mix into the per-kernel averages and
    # the per-launch counters below (round 6's first profile: mp_lstm_fused<256,8,512> "380 us" over 1 169 launches against
    # 793 us by HIP events).  The profiled command is bench.py's default one without those legs; `<tag>_kernel_stats_all_legs.md`
    # keeps the summary of the default command beside it.
    bench = [sys.executable, os.path.join(REPO, "bench.py"), "--no-cpu-baseline", "--legs", "headline"] + extra
    bench_all = [sys.executable, os.path.join(REPO, "bench.py"), "--no-cpu-baseline"] + extra
    os.makedirs(OUT, exist_ok=True)
    # ---- 1. kernel trace + stats
    d = os.path.join(OUT, "prof_" + tag)
    rc, log = run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", "bench", "--"]
                  + bench + ["--steps", "20", "--warmup", "5"])
    line = [l for l in log.splitlines() if l.startswith("{")]
    bench_line = json.loads(line[-1]) if line else None
    stats = glob.glob(os.path.join(d, "**", "bench_kernel_stats.csv"), recursive=True)
    rows = list(csv.DictReader(open(stats[0]))) if stats else []
    with open(os.path.join(OUT, tag + "_kernel_stats.csv"), "w") as f:
        f.write(open(stats[0]).read() if stats else "")
    with open(os.path.join(OUT, tag + "_kernel_stats.md"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats (%s)\n\n" % tag)
        f.write("Command (MI355X, via gpurun): `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py "
                "--no-cpu-baseline --legs headline %s --steps 20 --warmup 5`\n" % " ".join(extra))
        f.write("(bench.py in one process: 25 forwards of the headline mode (exact-fp32 operands, eager launches on the library's "
                "streams), 23 of the opt-in split-fp16 mode, the configs[3] leg (B = 1024: chunks of 256 sequences, the same launches), "
                "5 event-timed forwards of the headline mode; B=256 x T=125; `--legs headline` leaves out the legs of the other "
                "BASELINE configs, which launch the same kernels at other shapes -- the default command's summary is "
                "%s_kernel_stats_all_legs.md; raw CSV next to this file)\n\n" % tag)
        if bench_line:
            f.write("bench line of this (profiled) run: %.4f ms/step = %.0f frames/s headline (%s); modes: %s\n\n"
                    % (bench_line["ms_per_step"], bench_line["value"], bench_line["config"].get("lstm_mode"),
                       json.dumps({k: v for k, v in bench_line["modes"].items() if isinstance(v, dict)})))
            f.write("roofline kernel (HIP events, live): %s: avg %.4f ms/launch\n\n"
                    % (bench_line["roofline"]["kernel"], bench_line["roofline"]["avg_launch_ms"]))
        trace = glob.glob(os.path.join(d, "**", "bench_kernel_trace.csv"), recursive=True)
        if bench_line and trace:
            want = bench_line["roofline"]["kernel"].split(" ")[0].replace(",", ", ").replace(">", ", false")
            dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(trace[0]))
                   if want in r["Kernel_Name"]]
            if dur:
                f.write("rocprof durations of `%s`: %d launches, avg %.1f us (HIP events inside bench.py: %.1f us)\n\n"
                        % (want, len(dur), sum(dur) / len(dur), 1e3 * bench_line["roofline"]["avg_launch_ms"]))
        f.write("| kernel | calls | total ms | % | avg us | min us | max us |\n|---|---|---|---|---|---|---|\n")
        for r in rows[:24]:
            f.write("| %s | %s | %.3f | %s | %.1f | %.1f | %.1f |\n" % (
                short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6, r["Percentage"],
                float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
    # ---- 1b. the default command (all legs), stats only
    d_all = os.path.join(OUT, "prof_" + tag + "_all")
    rc, log = run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d_all, "-o", "bench", "--"]
                  + bench_all + ["--steps", "20", "--warmup", "5"])
    stats_all = glob.glob(os.path.join(d_all, "**", "bench_kernel_stats.csv"), recursive=True)
    if stats_all:
        rows_all = list(csv.DictReader(open(stats_all[0])))
        with open(os.path.join(OUT, tag + "_kernel_stats_all_legs.md"), "w") as f:
            f.write("# rocprofv3 --kernel-trace --stats of bench.py's DEFAULT command (%s): every leg\n\n" % tag)
            f.write("`rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline %s --steps 20 --warmup 5`.  "
                    "The layer kernels appear with launches of three shapes here (256 x 125 forwards, 512 x 45 ticks, 1 x 3000): "
                    "their averages describe no single launch -- the per-launch figures of the roofline are in %s_kernel_stats.md.\n\n"
                    % (" ".join(extra), tag))
            f.write("| kernel | calls | total ms | % | avg us | min us | max us |\n|---|---|---|---|---|---|---|\n")
            for r in rows_all[:30]:
                f.write("| %s | %s | %.3f | %s | %.1f | %.1f | %.1f |\n" % (
                    short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6, r["Percentage"],
                    float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
    # ---- 2. PMC passes
    acc = {}
    for gi, group in enumerate(PMC_GROUPS):
        d = os.path.join(OUT, "pmc_%s_%d" % (tag, gi))
        rc, log = run(["rocprofv3", "--kernel-trace", "--pmc"] + group + ["--output-format", "csv", "-d", d, "-o", "pmc", "--"]
                      + bench + ["--steps", "3", "--warmup", "1"])
        files = glob.glob(os.path.join(d, "**", "pmc_counter_collection.csv"), recursive=True)
        if not files:
            print("no counter file for", group, "\n", log[-2000:])
            continue
        for r in csv.DictReader(open(files[0])):
            k = short(r["Kernel_Name"])
            if not k.startswith("mp_"):
                continue
            e = acc.setdefault(k, {}).setdefault(r["Counter_Name"], [0.0, 0])
            e[0] += float(r["Counter_Value"])
            e[1] += 1
    import hashlib
    lib = os.path.join(REPO, "mobileposer_amd", "libmobileposer_hip.so")
    summary = {"note": __doc__.split("Counter arithmetic")[1].strip(),
               # bench.py quotes roofline.traffic from this file only while the library it runs is this very binary
               "lib_md5": hashlib.md5(open(lib, "rb").read()).hexdigest(), "src_md5": __import__("__graft_entry__").source_md5(),
               "kernels": {}}
    for k, c in acc.items():
        e = {n: v[0] / v[1] for n, v in c.items()}
        e["launches_sampled"] = max(v[1] for v in c.values())
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("GRBM_GUI_ACTIVE"):
            e["mfma_util_pct"] = round(100.0 * e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 1)
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_bytes_per_launch_corrected"] = int((2.0 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024.0)
        summary["kernels"][k] = e
    json.dump(summary, open(os.path.join(OUT, tag + "_pmc_summary.json"), "w"), indent=1)
    print(json.dumps({k: {n: v for n, v in e.items() if n in ("mfma_util_pct", "hbm_bytes_per_launch_corrected")}
                      for k, e in summary["kernels"].items()}, indent=1))


if __name__ == "__main__":
    main()
