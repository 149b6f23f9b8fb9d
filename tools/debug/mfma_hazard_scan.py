"""Scan per-kernel ISA dumps (tools/debug/kernel_isa.py) for hazards the assembler's recogniser cannot see because the MFMAs are
inline asm: a VALU (or v_accvgpr_*) write of a register within `dist` instructions in front of a v_mfma that reads it as SrcA / SrcB,
and any v_accvgpr_* traffic at all in the kernels whose 256 AccVGPRs hold weights.   python tools/debug/mfma_hazard_scan.py <dir> [substr]"""
import os
import re
import sys


def regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"^([va])\[(\d+):(\d+)\]$", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r"^([va])(\d+)$", tok)
    return {(m.group(1), int(m.group(2)))} if m else set()


def scan(path, dist=2):
    L = open(path).read().splitlines()
    bad = []
    for i, l in enumerate(L):
        if not l.startswith("v_mfma"):
            continue
        ops = l.split(None, 1)[1].split(",")
        src = regs(ops[1]) | regs(ops[2])
        for k in range(1, dist + 1):
            if i - k < 0:
                break
            p = L[i - k]
            if p.startswith("v_mfma") or not p.startswith("v_"):
                continue
            dst = regs(p.split(None, 1)[1].split(",")[0])
            if dst & src:
                bad.append((i, k, p, l))
    return bad, sum("accvgpr" in l for l in L)


d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "mp_lstm_fused<256,8"
rc = 0
for f in sorted(os.listdir(d)):
    if sub not in f:
        continue
    bad, nacc = scan(os.path.join(d, f))
    print("%-70s VALU-write -> MFMA-read within 2: %d   v_accvgpr ops: %d" % (f[:70], len(bad), nacc))
    for b in bad[:5]:
        print("    line %d (-%d): %s  ->  %s" % b)
    rc |= bool(bad) or nacc > 0
sys.exit(rc)
