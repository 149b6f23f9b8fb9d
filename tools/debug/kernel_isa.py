"""Per-kernel instruction text of the gfx950 code in a .so / .o (addresses and encodings stripped), for diffing two builds:
  python tools/debug/kernel_isa.py <lib-or-object> <out-dir>      -> one file per kernel (demangled name, sanitised)
  python tools/debug/kernel_isa.py --diff <dirA> <dirB> [substr]  -> kernels whose instruction streams differ"""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from mobileposer_amd import _devcode  # noqa: E402


def kernels(path):
    text = _devcode.disassemble(path)
    out, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if cur is None or not line.startswith("\t"):
            continue
        ins = line.strip().split("//")[0].strip()
        if ins:
            out[cur].append(re.sub(r"\s+", " ", ins))
    return out


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return dict(zip(names, r.stdout.splitlines()))


if sys.argv[1] == "--diff":
    a, b = sys.argv[2], sys.argv[3]
    sub = sys.argv[4] if len(sys.argv) > 4 else ""
    for f in sorted(set(os.listdir(a)) | set(os.listdir(b))):
        if sub not in f:
            continue
        pa, pb = os.path.join(a, f), os.path.join(b, f)
        if not os.path.exists(pa) or not os.path.exists(pb):
            print("%-90s only in %s" % (f[:90], a if os.path.exists(pa) else b))
            continue
        ta, tb = open(pa).read().splitlines(), open(pb).read().splitlines()
        print("%-90s %s (%d vs %d instructions)" % (f[:90], "IDENTICAL" if ta == tb else "DIFFERENT", len(ta), len(tb)))
else:
    ks = kernels(sys.argv[1])
    dm = demangle(list(ks))
    os.makedirs(sys.argv[2], exist_ok=True)
    for k, ins in ks.items():
        name = re.sub(r"[^A-Za-z0-9_<>,]", "_", dm.get(k, k).replace("(anonymous namespace)::", "").replace(" ", ""))[:150]
        open(os.path.join(sys.argv[2], name + ".txt"), "w").write("\n".join(ins) + "\n")
    print(len(ks), "kernels ->", sys.argv[2])
