#!/usr/bin/env python
"""Build alternative versions of libquarkaudio_hip.so with extra -D flags (kernel-tuning experiments).
usage: python tools/variants.py [--transform tools/experiments/X.py] name "-DQA_SB1=0x8f -DQA_SB2=8" [name2 "flags2" ...]
Each lands in tools/_variants/<name>/libquarkaudio_hip.so; select one with QA_LIBRARY=<path>."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unified_audio_amd import build as B  # noqa: E402


def _transform(path):
    """--transform FILE.py[:function]: a module with transform(source) -> source, applied to conv_gemm.hip (the product file stays untouched)."""
    path, _, fn = path.partition(":")
    fn = fn or "transform"
    import importlib.util

    spec = importlib.util.spec_from_file_location("qa_variant_transform", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return getattr(mod, fn)


def main():
    args = sys.argv[1:]
    transform = None
    if args[:1] == ["--transform"]:
        transform = _transform(args[1])
        args = args[2:]
    B.build_library()
    for name, flags in zip(args[::2], args[1::2]):
        out = os.path.join(ROOT, "tools", "_variants", name)
        os.makedirs(out, exist_ok=True)
        objs = []
        for src in B.SOURCES:
            obj = os.path.join(B.BUILD, src.rsplit(".", 1)[0] + ".o")
            if src == "conv_gemm.hip" or src in os.environ.get("QA_VARIANT_SOURCES", "").split(","):
                obj = os.path.join(out, src.rsplit(".", 1)[0] + ".o")
                path = os.path.join(B.CSRC, src)
                if transform and src == "conv_gemm.hip":
                    path = os.path.join(out, src)
                    with open(path, "w") as f:
                        f.write(transform(open(os.path.join(B.CSRC, src)).read()))
                subprocess.run([B._hipcc(), *B.FLAGS, *B.EXTRA_FLAGS.get(src, []), *flags.split(), "-x", "hip", "-c", path, "-o", obj], check=True,
                               stderr=subprocess.DEVNULL)
            objs.append(obj)
        lib = os.path.join(out, "libquarkaudio_hip.so")
        subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs], check=True)
        print(lib)


if __name__ == "__main__":
    main()
