#!/usr/bin/env python
"""Re-print the knob table of INTEGRATION.md section 6 from the library's own table (csrc/knobs.h via qa_knob_info).
usage: python tools/knob_table.py [--write]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unified_audio_amd import _lib  # noqa: E402

lines = ["| Knob (environment variable = initial value; `qa_set_knob(name, v)` at run time) | Default | Effect |", "|---|---|---|"]
for k, (_, d, doc) in _lib.knobs().items():
    lines.append(f"| `{k}` | {d} | {doc} |")
table = "\n".join(lines)
if "--write" in sys.argv:
    p = os.path.join(ROOT, "INTEGRATION.md")
    s = open(p).read()
    a = s.index("| Knob (environment variable = initial value")
    b = s.index("\n\nOther measurement hooks:")
    open(p, "w").write(s[:a] + table + s[b:])
else:
    print(table)
