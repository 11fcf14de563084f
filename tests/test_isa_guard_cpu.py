"""Build-time guard: the persistent recurrences issue their hand-off loads in one inline-asm statement and wait for them in a
later one; in between the destination registers are written asynchronously and the compiler does not know.  tools/check_async_loads.py
compiles lstm.hip to gfx950 assembly and proves that no instruction touches such a register between the `global_load ... sc1` and the
`s_waitcnt vmcnt` that retires it - for the ISA that actually ships, on the CPU."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_split_load_wait_idiom_is_safe_in_the_shipped_isa():
    import check_async_loads as G

    n, bad = G.check(G.assembly(os.path.join(G.CSRC, "lstm.hip")))
    assert n >= 100, f"only {n} sc1 loads found: the scanner no longer sees the kernels"
    assert not bad, "\n".join(bad[:10])


def test_scanner_flags_a_register_touched_in_flight():
    import check_async_loads as G

    asm = """
_Zkernel:
\tglobal_load_dwordx4 v[4:7], v[0:1], off sc1
\tv_mov_b32_e32 v9, v5
\ts_waitcnt vmcnt(0)
\ts_endpgm
"""
    n, bad = G.check(asm)
    assert n == 1 and len(bad) == 1 and "v5" in bad[0]
    ok = asm.replace("\tv_mov_b32_e32 v9, v5\n\ts_waitcnt vmcnt(0)\n", "\ts_waitcnt vmcnt(0)\n\tv_mov_b32_e32 v9, v5\n")
    assert G.check(ok) == (1, [])
    # vmcnt(n) retires all but the n youngest operations
    two = "_Zk:\n\tglobal_load_dword v1, v[2:3], off sc1\n\tglobal_load_dword v4, v[2:3], off sc1\n\ts_waitcnt vmcnt(1)\n\tv_add_f32_e32 v5, v1, v1\n\ts_waitcnt vmcnt(0)\n\ts_endpgm\n"
    assert G.check(two) == (2, [])
    assert len(G.check(two.replace("v_add_f32_e32 v5, v1, v1", "v_add_f32_e32 v5, v4, v4"))[1]) == 1


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_decode_kernels_reach_their_first_load_without_a_scalar_memory_round_trip():
    """r06: the decode step's GEMV / fused-MLP kernels take their hot scalars as preloaded kernel arguments and issue
    their weight batch before they read the argument struct.  The shipped ISA must show it: from a kernel's real entry to the first
    global_load of its working waves no s_load and no wait on lgkmcnt - in every instance.  (The source order alone had hidden such a wait
    for most of round 6: 2 % of a 16-segment generate.)"""
    import check_lm_prologue as G

    n, bad = G.check(G.assembly())
    assert n >= 40, f"only {n} kernel instances found: the scanner no longer sees the kernels"
    assert not bad, "\n".join(bad[:10])


def test_prologue_scanner_flags_a_scalar_load_in_front_of_the_first_global_load():
    import check_lm_prologue as G

    asm = """
_ZN2qa14lm_gemv_kernelILi1EEEvPKf:
\ts_load_dwordx2 s[2:3], s[0:1], 0x0
\ts_waitcnt lgkmcnt(0)
\ts_branch .LBB0_0
\t.p2align\t8
.LBB0_0:
\ts_load_dwordx4 s[24:27], s[0:1], 0x38
\ts_waitcnt lgkmcnt(0)
\tglobal_load_dwordx4 v[0:3], v[4:5], off
\ts_endpgm
.Lfunc_end0:
"""
    n, bad = G.check(asm)
    assert n == 1 and len(bad) == 1 and "s_load_dwordx4" in bad[0]
    ok = asm.replace("\ts_load_dwordx4 s[24:27], s[0:1], 0x38\n\ts_waitcnt lgkmcnt(0)\n\tglobal_load_dwordx4 v[0:3], v[4:5], off\n",
                     "\tglobal_load_dwordx4 v[0:3], v[4:5], off\n\ts_load_dwordx4 s[24:27], s[0:1], 0x38\n\ts_waitcnt lgkmcnt(0)\n")
    assert G.check(ok) == (1, [])
    assert len(G.check(asm.replace("\t.p2align\t8\n", ""))[1]) == 1  # no preload header: the flag was dropped
