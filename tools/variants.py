#!/usr/bin/env python
"""Build alternative versions of libquarkaudio_hip.so with extra -D flags (kernel-tuning experiments).
usage: python tools/variants.py name "-DQA_SB1=0x8f -DQA_SB2=8" [name2 "flags2" ...]
Each lands in tools/_variants/<name>/libquarkaudio_hip.so; select one with QA_LIBRARY=<path>."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unified_audio_amd import build as B  # noqa: E402


def main():
    args = sys.argv[1:]
    B.build_library()
    for name, flags in zip(args[::2], args[1::2]):
        out = os.path.join(ROOT, "tools", "_variants", name)
        os.makedirs(out, exist_ok=True)
        objs = []
        for src in B.SOURCES:
            obj = os.path.join(B.BUILD, src.rsplit(".", 1)[0] + ".o")
            if src == "conv_gemm.hip" or src in os.environ.get("QA_VARIANT_SOURCES", "").split(","):
                obj = os.path.join(out, src.rsplit(".", 1)[0] + ".o")
                subprocess.run([B._hipcc(), *B.FLAGS, *flags.split(), "-x", "hip", "-c", os.path.join(B.CSRC, src), "-o", obj], check=True,
                               stderr=subprocess.DEVNULL)
            objs.append(obj)
        lib = os.path.join(out, "libquarkaudio_hip.so")
        subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs], check=True)
        print(lib)


if __name__ == "__main__":
    main()
