#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel stats table that is committed
under profiles/ (name, calls, total/avg/min/max duration, share of GPU busy time)."""
import sqlite3
import sys


def short(name: str) -> str:
    name = name.replace("mh::(anonymous namespace)::", "").replace("mh::dec::", "dec::").replace("void ", "")
    return name[:110]


def main(db, out=None):
    con = sqlite3.connect(db)
    rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                       "max(vgpr_count), max(accum_vgpr_count), max(lds_size), max(grid_x) "
                       "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    lines = [f"# rocprofv3 --kernel-trace summary of {db}", f"# total kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches",
             "kernel | calls | total_ms | avg_us | min_us | max_us | pct | vgpr | agpr | lds | grid_x"]
    for n, c, t, a, mn, mx, vg, ag, lds, gx in rows:
        lines.append(f"{short(n)} | {c} | {t / 1e6:.3f} | {a / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * t / total:.1f} | {vg} | {ag} | {lds} | {gx}")
    # the dominant kernel split by stream: bench.py's roofline probe launches it back to back on the engine's stream,
    # the decode step launches it from the chain streams
    if rows:
        top = rows[0][0]
        lines.append(f"# {short(top)} by stream (stream_id | calls | avg_us | min_us | max_us)")
        for sid, c, a, mn, mx in con.execute("select stream_id, count(*), avg(duration), min(duration), max(duration) from kernels "
                                             "where name = ? group by stream_id order by stream_id", (top,)):
            lines.append(f"#   stream {sid} | {c} | {a / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f}")
    txt = "\n".join(lines)
    if out:
        open(out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
