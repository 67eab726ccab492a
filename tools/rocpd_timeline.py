#!/usr/bin/env python
"""Per-stream timeline statistics of a rocprofv3 kernel trace (rocpd sqlite): for the busiest streams (the decode
chains replay one captured token step each) the median duration of every kernel AND the median idle gap in front of it,
so that launch-boundary cost and kernel body cost can be told apart; plus how much of the wall time of the decode phase
each stream keeps the GPU busy.

  rocpd_timeline.py DB [OUT.txt] [N_STREAMS=3]
"""
import sqlite3
import statistics as st
import sys


def short(name: str) -> str:
    name = name.replace("mh::(anonymous namespace)::", "").replace("mh::dec::", "dec::").replace("void ", "")
    return name[:96]


def main(db, out=None, n_streams=3):
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info('kernels')")]
    c_start = "start" if "start" in cols else [c for c in cols if "start" in c][0]
    c_end = "end" if "end" in cols else [c for c in cols if "end" in c][0]
    lines = [f"# timeline statistics of {db} (columns {c_start}/{c_end}; times in us)"]
    streams = con.execute("select stream_id, count(*) from kernels group by stream_id order by count(*) desc").fetchall()
    for sid, cnt in streams[:n_streams]:
        rows = con.execute(f"select name, {c_start}, {c_end} from kernels where stream_id = ? order by {c_start}", (sid,)).fetchall()
        per = {}
        busy = 0
        prev_end = None
        for name, s, e in rows:
            d = per.setdefault(name, {"dur": [], "gap": []})
            d["dur"].append((e - s) / 1e3)
            if prev_end is not None:
                d["gap"].append((s - prev_end) / 1e3)
            prev_end = e
            busy += e - s
        span = rows[-1][2] - rows[0][1]
        lines.append(f"## stream {sid}: {cnt} dispatches, span {span / 1e6:.2f} ms, kernels busy {busy / 1e6:.2f} ms "
                     f"({100.0 * busy / span:.1f} % of the span)")
        lines.append("kernel | calls | median dur | p10 dur | median gap before | p90 gap before | sum(dur+gap) ms")
        for name, d in sorted(per.items(), key=lambda kv: -sum(kv[1]["dur"])):
            if len(d["dur"]) < 8:
                continue
            g = sorted(x for x in d["gap"] if x < 200.0) or [0.0]     # drop the rare host stalls
            du = sorted(d["dur"])
            lines.append(f"{short(name)} | {len(du)} | {st.median(du):.2f} | {du[len(du) // 10]:.2f} | {st.median(g):.2f} | "
                         f"{g[(9 * len(g)) // 10]:.2f} | {(sum(du) + sum(g)) / 1e3:.2f}")
    txt = "\n".join(lines)
    if out:
        open(out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None, int(sys.argv[3]) if len(sys.argv) > 3 else 3)
