#!/usr/bin/env python
"""Median / p10 / p90 duration per kernel of a rocprofv3 kernel trace (rocpd sqlite), kernels matching a substring.
  rocpd_medians.py DB [SUBSTR ...]"""
import sqlite3, statistics as st, sys
con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info('kernels')")]
c_start = "start" if "start" in cols else [c for c in cols if "start" in c][0]
c_end = "end" if "end" in cols else [c for c in cols if "end" in c][0]
pats = sys.argv[2:] or [""]
per = {}
for name, s, e in con.execute(f"select name, {c_start}, {c_end} from kernels"):
    if any(p in name for p in pats):
        per.setdefault(name, []).append((e - s) / 1e3)
for name, d in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    d.sort()
    print(f"{name[:70]:70s} n {len(d):5d}  median {st.median(d):8.2f}  p10 {d[len(d) // 10]:8.2f}  p90 {d[9 * len(d) // 10]:8.2f} us")
