#!/usr/bin/env python
"""Build-time guard for the split `load ... wait` inline-asm idiom of the persistent kernels (lstm.hip).

The hand-off loads are written as one asm statement that ISSUES `global_load_dwordx4 vX, ..., off sc1` and a later one that
WAITS (`s_waitcnt vmcnt(0)` with the value as a "+v" operand).  Between the two the destination registers are being written
asynchronously, which the compiler does not know: it may legally copy or spill the value there.  This script
compiles the given sources to gfx950 assembly and proves, per kernel, that no instruction reads or writes the destination
registers of an `sc1` load between the load and the first following `s_waitcnt vmcnt(0)`.

    python tools/check_async_loads.py [file.hip ...]      (default: lstm.hip)

Exit status 0 = the invariant holds in every kernel; used by tests/test_sanitizer_cpu.py.
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "unified_audio_amd", "csrc")

_REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def _regs(text: str) -> set[int]:
    out: set[int] = set()
    for m in _REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def assembly(src: str) -> str:
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        hipcc = "/opt/rocm/bin/hipcc"
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
                        "-S", "--cuda-device-only", "-o", out, src], check=True, stderr=subprocess.DEVNULL)
        return open(out).read()


def check(asm: str) -> tuple[int, list[str]]:
    """-> (number of sc1 loads seen, violations)"""
    kernel = "?"
    flight: list[tuple[set[int], str]] = []  # vector-memory operations in issue order: (destination registers, text)
    n_loads = 0
    bad: list[str] = []
    vm_ops = ("global_", "buffer_", "flat_", "scratch_")
    branches = ("s_endpgm", "s_branch", "s_cbranch_scc0", "s_cbranch_scc1", "s_cbranch_vccz", "s_cbranch_vccnz", "s_cbranch_execz",
                "s_cbranch_execnz", "s_setpc_b64")
    for ln, raw in enumerate(asm.splitlines(), 1):
        line = raw.split(";")[0].strip()
        if not line:
            continue
        if line.endswith(":") and not line.startswith("."):
            kernel = line[:-1]
            flight.clear()
            continue
        if line.startswith(".") or line.startswith("//"):
            continue
        op = line.split()[0]
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", line)
            if m:  # gfx9: loads and stores share vmcnt and retire in issue order: at most n stay outstanding
                n = int(m.group(1))
                flight = flight[len(flight) - n:] if n else []
            continue
        pending = {r: t for regs, t in flight for r in regs}
        touched = _regs(line[len(op):])
        hit = touched & pending.keys()
        if hit:
            r = min(hit)
            bad.append(f"{kernel}: line {ln}: `{line}` touches v{r} while `{pending[r]}` is in flight")
        if op.startswith(vm_ops):
            dst: set[int] = set()
            if "_load" in op and " sc1" in line:
                n_loads += 1
                dst = _regs(line[len(op):].split(",")[0])
            flight.append((dst, line))
        if op in branches and any(regs for regs, _ in flight):
            # the idiom never spans a branch in the sources; if hipcc put one there the linear scan is no proof
            bad.append(f"{kernel}: line {ln}: `{line}` while registers of an sc1 load are in flight")
            flight = [(set(), t) for _, t in flight]
    return n_loads, bad


def main(argv):
    files = argv or ["lstm.hip"]
    rc = 0
    for f in files:
        path = f if os.path.exists(f) else os.path.join(CSRC, f)
        n, bad = check(assembly(path))
        print(f"{os.path.basename(path)}: {n} sc1 loads, {len(bad)} violations")
        for b in bad[:20]:
            print("  " + b)
        rc |= bool(bad)
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
