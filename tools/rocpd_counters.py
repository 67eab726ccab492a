#!/usr/bin/env python
"""Per-kernel averages of every counter in a rocprofv3 --pmc run (rocpd sqlite) + the kernel's average duration.

  rocpd_counters.py DB [OUT.txt] [NAME_SUBSTRING ...]
"""
import sqlite3
import sys


def short(name: str) -> str:
    return name.replace("mh::(anonymous namespace)::", "").replace("mh::dec::", "dec::").replace("void ", "")[:96]


def main(db, out=None, filters=()):
    con = sqlite3.connect(db)
    rows = con.execute("select name, counter_name, avg(counter_value), count(*) from pmc_events group by name, counter_name").fetchall()
    dur = {n: (a, c) for n, a, c in con.execute("select name, avg(duration), count(*) from kernels group by name")}
    per = {}
    for n, cn, v, c in rows:
        per.setdefault(n, {})[cn] = v
    names = sorted({cn for d in per.values() for cn in d})
    lines = [f"# rocprofv3 --pmc {' '.join(names)} (averages per launch) of {db}",
             "kernel | launches | avg_us | " + " | ".join(names)]
    for n, d in sorted(per.items(), key=lambda kv: -(dur.get(kv[0], (0, 0))[0] * dur.get(kv[0], (0, 0))[1])):
        if filters and not any(f in n for f in filters):
            continue
        a, c = dur.get(n, (0.0, 0))
        lines.append(f"{short(n)} | {c} | {a / 1e3:.2f} | " + " | ".join(f"{d.get(cn, float('nan')):.4g}" for cn in names))
    txt = "\n".join(lines)
    if out:
        open(out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None, sys.argv[3:])
