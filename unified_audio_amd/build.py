"""Build recipe for libquarkaudio_hip.so (hipcc, gfx950 only).  Invoked by __graft_entry__.build() and usable as
`python -m unified_audio_amd.build`.  Objects go to unified_audio_amd/_build/, the library next to this file so that
it travels with the repo snapshot to the GPU box."""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
BUILD = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libquarkaudio_hip.so")
SOURCES = ["api.cpp", "conv_gemm.hip", "ew.hip", "attention.hip", "lstm.hip", "rvq.hip", "lm_kernels.hip", "lm_decode.hip", "ssl_kernels.hip", "bicodec_kernels.hip", "seanet_front.hip", "hcodec.cpp", "lm.cpp", "ssl.cpp", "bicodec.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-I", INCLUDE, "-I", CSRC]
# -amdgpu-kernarg-preload-count: gfx950 delivers the first 14 dwords of a kernel's leading SCALAR arguments in SGPRs at wave start, so a wave can form
# its first addresses without an s_load round trip to the kernel-argument segment.  Only for the latency-bound UniSE decode step, whose kernels pass
# their hot scalars in front of the argument struct (r06: 103.4 -> 101.0 ms per 16-segment generate); applied to EVERY source the same generate took
# 102.4 ms (profiles/r06_lm_prefetch_ab.txt), so the flag stays with the one file it was measured on.
EXTRA_FLAGS = {"lm_decode.hip": ["-mllvm", "-amdgpu-kernarg-preload-count=14"]}


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libquarkaudio_hip.so cannot be built on this machine")
    return exe


def _newest_dep() -> float:
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(INCLUDE, "quarkaudio.h"), __file__]
    return max(os.path.getmtime(d) for d in deps)


def _compile(src: str) -> str:
    obj = os.path.join(BUILD, src.rsplit(".", 1)[0] + ".o")
    if os.path.exists(obj) and os.path.getmtime(obj) >= _newest_dep():
        return obj
    cmd = [_hipcc(), *FLAGS, *EXTRA_FLAGS.get(src, []), "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-4000:]}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build_library(force: bool = False) -> str:
    os.makedirs(BUILD, exist_ok=True)
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _newest_dep():
        return LIB
    with cf.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(_compile, SOURCES))
    r = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
