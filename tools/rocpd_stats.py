#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME` writes NAME_results.db on
ROCm 7.2) as the per-kernel table `--stats` would print: calls, total, average, min, max, share."""
import sqlite3
import sys


def timeline(cur):
    """Device timeline of the trace: how much of the span between the first and the last kernel has at least one kernel running
    (union of the [start, end] intervals), and which kernels the longest idle gaps follow - launch-bound stretches show up here."""
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    if "start" not in cols or "end" not in cols:
        return ["", f"(no start / end columns in the kernels view: {cols})"]
    ev = list(cur.execute('select start, "end", name from kernels order by start'))
    if not ev:
        return []
    busy, gaps, cur_end, prev_name = 0, {}, ev[0][0], ev[0][2]
    span0 = ev[0][0]
    for st, en, name in ev:
        if st > cur_end:  # nothing was running between cur_end and st: charge the gap to the kernel that ended last
            g = gaps.setdefault(prev_name[:70], [0, 0, []])
            g[0] += st - cur_end
            g[1] += 1
            g[2].append(st - cur_end)
            cur_end = st
        if en > cur_end:
            busy += en - cur_end
            cur_end, prev_name = en, name
    span = cur_end - span0
    out = ["", "## device timeline", f"span first kernel -> last kernel {span / 1e6:.3f} ms; at least one kernel running {busy / 1e6:.3f} ms "
           f"({100.0 * busy / span:.1f} %); idle {(span - busy) / 1e6:.3f} ms (includes the host-side setup between warm-up and timed steps)", "",
           "| idle after kernel | gaps | total idle ms | median us | max us |", "|---|---|---|---|---|"]
    for name, (tot, n, each) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:14]:
        each.sort()
        out.append(f"| `{name}` | {n} | {tot / 1e6:.3f} | {each[len(each) // 2] / 1e3:.2f} | {each[-1] / 1e3:.2f} |")
    return out


def main(path, out=None, top=40):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                            "from kernels group by name order by sum(duration) desc"))
    tot = sum(r[2] for r in rows)
    lines = [f"# kernel stats from {path}", f"# total kernel time {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches", "",
             "| % | calls | total ms | avg us | min us | max us | kernel |", "|---|---|---|---|---|---|---|"]
    for r in rows[:top]:
        lines.append(f"| {100 * r[2] / tot:.2f} | {r[1]} | {r[2] / 1e6:.3f} | {r[3] / 1e3:.2f} | {r[4] / 1e3:.2f} | {r[5] / 1e3:.2f} | `{r[0][:110]}` |")
    lines += timeline(cur)
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(*sys.argv[1:3])
