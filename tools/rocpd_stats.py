#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME` writes NAME_results.db on
ROCm 7.2) as the per-kernel table `--stats` would print: calls, total, average, min, max, share."""
import sqlite3
import sys


def main(path, out=None, top=40):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                            "from kernels group by name order by sum(duration) desc"))
    tot = sum(r[2] for r in rows)
    lines = [f"# kernel stats from {path}", f"# total kernel time {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches", "",
             "| % | calls | total ms | avg us | min us | max us | kernel |", "|---|---|---|---|---|---|---|"]
    for r in rows[:top]:
        lines.append(f"| {100 * r[2] / tot:.2f} | {r[1]} | {r[2] / 1e6:.3f} | {r[3] / 1e3:.2f} | {r[4] / 1e3:.2f} | {r[5] / 1e3:.2f} | `{r[0][:110]}` |")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(*sys.argv[1:3])
