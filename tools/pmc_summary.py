#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes (csv output, one counter group per pass) into the per-kernel table under profiles/.
usage: python tools/pmc_summary.py OUT_PREFIX PASS_DIR [PASS_DIR ...]
Each PASS_DIR holds *_counter_collection.csv of one `rocprofv3 --kernel-trace --pmc ... --output-format csv` run of the same
command.  FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled (gfx950 correction, MI355X_MICROARCH.md HBM section).
Kernels of one tile configuration (BM, BN) are also aggregated over their BK / prologue variants, which is the granularity
of bench.py's live profiler."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def short(name):
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "").strip()


def main(out_prefix, *dirs):
    agg = defaultdict(lambda: defaultdict(float))   # kernel -> counter -> sum
    cnt = defaultdict(lambda: defaultdict(int))     # kernel -> counter -> dispatches
    dur = defaultdict(list)
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            seen = set()
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if not k.startswith("qa::"):
                    continue
                c = r["Counter_Name"]
                agg[k][c] += float(r["Counter_Value"])
                cnt[k][c] += 1
                key = (r["Dispatch_Id"], f)
                if key not in seen:
                    seen.add(key)
                    dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    rows = {}
    for k in agg:
        per = {c: agg[k][c] / max(cnt[k][c], 1) for c in agg[k]}
        n = max(cnt[k].values())
        fetch, write = per.get("FETCH_SIZE"), per.get("WRITE_SIZE")
        busy = None
        if "SQ_VALU_MFMA_BUSY_CYCLES" in per and per.get("GRBM_GUI_ACTIVE"):
            busy = per["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * per["GRBM_GUI_ACTIVE"] / 8.0)
        rows[k] = {"launches": n, "avg_us": sum(dur[k]) / len(dur[k]), "fetch_kib": fetch, "write_kib": write,
                   "hbm_bytes_per_launch": (2 * fetch + write) * 1024 if fetch is not None and write is not None else None,
                   "mfma_busy_frac": busy}
    # aggregate conv_gemm variants per (BM, BN, WM, WN)
    groups = defaultdict(list)
    for k, v in rows.items():
        m = re.match(r"qa::conv_gemm_kernel<(\d+), (\d+), (\d+), (\d+),", k)
        if m:
            groups["qa::conv_gemm_kernel<%s, %s, %s, %s>" % m.groups()].append(v)
    for g, vs in groups.items():
        n = sum(v["launches"] for v in vs)
        w = lambda key: (sum(v[key] * v["launches"] for v in vs if v[key] is not None) / n) if all(v[key] is not None for v in vs) else None
        rows[g + " (all BK / prologue variants)"] = {"launches": n, "avg_us": w("avg_us"), "fetch_kib": w("fetch_kib"), "write_kib": w("write_kib"),
                                                     "hbm_bytes_per_launch": w("hbm_bytes_per_launch"), "mfma_busy_frac": w("mfma_busy_frac")}
    order = sorted(rows, key=lambda k: -(rows[k]["avg_us"] or 0) * rows[k]["launches"])
    f = lambda v, fmt: "-" if v is None else fmt % v
    lines = ["| kernel | launches | avg us | FETCH KiB/launch | WRITE KiB/launch | HBM MB/launch (2F+W) | HBM GB/s (traffic / duration) | MFMA busy frac |",
             "|---|---|---|---|---|---|---|---|"]
    for k in order[:40]:
        v = rows[k]
        gbps = v["hbm_bytes_per_launch"] / (v["avg_us"] * 1e-6) / 1e9 if v["hbm_bytes_per_launch"] is not None and v["avg_us"] else None
        v["hbm_gbps"] = gbps
        lines.append("| `%s` | %d | %s | %s | %s | %s | %s | %s |" % (k[:90], v["launches"], f(v["avg_us"], "%.1f"), f(v["fetch_kib"], "%.0f"), f(v["write_kib"], "%.0f"),
                                                                    f(v["hbm_bytes_per_launch"] and v["hbm_bytes_per_launch"] / 1e6, "%.1f"), f(gbps, "%.0f"),
                                                                    f(v["mfma_busy_frac"], "%.3f")))
    open(out_prefix + ".md", "w").write("\n".join(lines) + "\n")
    json.dump({k: rows[k] for k in order}, open(out_prefix + ".json", "w"), indent=1)
    print("\n".join(lines[:12]))


if __name__ == "__main__":
    main(*sys.argv[1:])
