"""Pull the gfx950 code objects out of the built shared library and disassemble them (build / test tooling only).

Used by tests/test_cabi_cpu.py to check that no kernel contains packed-fp32 VALU instructions (insurance kept from
round 2, not a claimed hardware erratum: csrc/mp_common.h, profiles/r02_coexec_glitch.md) and by tools that want per-kernel ISA."""
import os
import struct
import subprocess
import tempfile

LLVM_BIN = "/opt/rocm/lib/llvm/bin"
EM_AMDGPU = 224


def device_elves(lib_path):
    """Every AMDGPU ELF image embedded in `lib_path` (one per translation unit), as bytes."""
    blob = open(lib_path, "rb").read()
    out, pos = [], 0
    while True:
        i = blob.find(b"\x7fELF", pos)
        if i < 0:
            break
        pos = i + 4
        if i == 0 or len(blob) < i + 64 or blob[i + 4] != 2:           # the host ELF itself / not ELF64
            continue
        e_machine = struct.unpack_from("<H", blob, i + 18)[0]
        if e_machine != EM_AMDGPU:
            continue
        e_shoff = struct.unpack_from("<Q", blob, i + 40)[0]
        e_shentsize, e_shnum = struct.unpack_from("<HH", blob, i + 58)
        size = e_shoff + e_shentsize * e_shnum
        out.append(blob[i:i + size])
        pos = i + size
    return out


def disassemble(lib_path):
    """Concatenated `llvm-objdump -d` text of all device code objects in the library."""
    objdump = os.path.join(LLVM_BIN, "llvm-objdump")
    text = []
    for elf in device_elves(lib_path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(elf)
            f.flush()
            r = subprocess.run([objdump, "-d", f.name], capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("llvm-objdump failed: " + r.stderr[:200])
            text.append(r.stdout)
    return "\n".join(text)
