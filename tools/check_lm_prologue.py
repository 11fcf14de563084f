#!/usr/bin/env python
"""Build-time guard for the decode step's launch prologues.

The GEMV / fused-MLP kernels of csrc/lm_decode.hip take the scalars their FIRST loads need as leading kernel arguments, which gfx950 preloads
into SGPRs (-amdgpu-kernarg-preload-count), and issue their weight batch before they touch the argument struct.  Both only pay if the ISA that
ships really has no scalar-memory round trip in front of a working wave's first global_load - round 6 found such a wait (`s_waitcnt lgkmcnt(0)`
behind the struct's s_loads) that the source order had hidden for a whole round.  This script compiles lm_decode.hip to gfx950 assembly with the
product's flags and checks, for every instance of the three kernels, the instruction stream from the kernel's real entry (behind the preload
header) to the first global_load of the working waves: no s_load, no s_waitcnt on lgkmcnt.

    python tools/check_lm_prologue.py          exit status 0 = the invariant holds; used by tests/test_isa_guard_cpu.py
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KERNELS = ("lm_gemv_kernel", "lm_gemv4_kernel", "lm_mlp_kernel")


def assembly() -> str:
    from unified_audio_amd import build as B

    src = "lm_decode.hip"
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "lm_decode.s")
        flags = [f for f in B.FLAGS if f != "-fPIC"]
        subprocess.run([B._hipcc(), *flags, *B.EXTRA_FLAGS.get(src, []), "--cuda-device-only", "-S", "-x", "hip", os.path.join(B.CSRC, src), "-o", out],
                       check=True, stderr=subprocess.DEVNULL)
        return open(out).read()


def check(asm: str):
    """-> (kernels checked, [violations])"""
    bad, n = [], 0
    for m in re.finditer(r"\n(_ZN2qa\d+(" + "|".join(KERNELS) + r")\w+):", asm):
        name = m.group(1)
        body = asm[m.end():asm.index(".Lfunc_end", m.end())]
        lines = [ln.strip() for ln in body.split("\n")]
        # the real entry: behind the kernarg-preload header (s_load of the preloaded arguments for firmware without preload, then `.p2align 8`)
        try:
            start = next(i for i, ln in enumerate(lines) if ln.startswith(".p2align"))
        except StopIteration:
            bad.append(f"{name}: no kernarg-preload header (.p2align) - was the preload flag dropped?")
            continue
        n += 1
        # follow the fall-through path of the WORKING waves: the prefetch plane (blockIdx.z != 0) branches away
        for ln in lines[start + 1:]:
            if ln.startswith("global_load"):
                break
            if ln.startswith("s_load") or (ln.startswith("s_waitcnt") and "lgkmcnt" in ln):
                bad.append(f"{name}: `{ln}` in front of the first global_load")
                break
    return n, bad


if __name__ == "__main__":
    n, bad = check(assembly())
    print(f"{n} kernel instances checked, {len(bad)} violations")
    for b in bad[:20]:
        print(" ", b)
    sys.exit(1 if bad or n == 0 else 0)
