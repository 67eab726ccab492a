#!/usr/bin/env python
"""Token-step anatomy from a rocprofv3 kernel trace (rocpd sqlite): for the two busiest streams (the decode chains) the period of
a token step (dec_sample_kernel to dec_sample_kernel), the kernel time and the idle time inside it, where the idle time sits
(gap in front of which kernel), how much of the time both chains have a kernel in flight, and one step printed kernel by kernel.

  rocpd_step_trace.py DB [OUT.txt]
"""
import sqlite3
import statistics as st
import sys


def short(name):
    return name.replace("mh::(anonymous namespace)::", "").replace("mh::dec::", "dec::").replace("void ", "")[:60]


def main(db, out=None):
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info('kernels')")]
    c_start = "start" if "start" in cols else [c for c in cols if "start" in c][0]
    c_end = "end" if "end" in cols else [c for c in cols if "end" in c][0]
    streams = con.execute("select stream_id, count(*) from kernels where name like '%dec_sample_kernel%' group by stream_id order by count(*) desc").fetchall()
    lines = [f"# token-step anatomy of {db}"]
    per_stream = {}
    for sid, _ in streams[:2]:
        rows = con.execute(f"select name, {c_start}, {c_end} from kernels where stream_id = ? order by {c_start}", (sid,)).fetchall()
        idx = [i for i, r in enumerate(rows) if "dec_sample_kernel" in r[0]]
        mid = idx[len(idx) // 4: 3 * len(idx) // 4]
        periods, busy, gaps, nk = [], [], [], []
        gap_by = {}
        for a, b in zip(mid[:-1], mid[1:]):
            seg = rows[a + 1: b + 1]                      # the kernels of one step, ending with its sampler
            if not seg or len(seg) > 200:
                continue
            periods.append((rows[b][1] - rows[a][1]) / 1e3)
            busy.append(sum(e - s for _, s, e in seg) / 1e3)
            prev = rows[a][2]
            gsum = 0.0
            for k, (n, s, e) in enumerate(seg):
                g = (s - prev) / 1e3
                gsum += g
                gap_by.setdefault((k, short(n)), []).append(g)
                prev = e
            gaps.append(gsum)
            nk.append(len(seg))
        per_stream[sid] = rows
        lines.append(f"## stream {sid}: {len(periods)} steps, kernels per step {st.median(nk):.0f}; step period median {st.median(periods):.1f} us "
                     f"(p10 {sorted(periods)[len(periods) // 10]:.1f}, p90 {sorted(periods)[9 * len(periods) // 10]:.1f}); kernel time per step {st.median(busy):.1f} us; "
                     f"idle per step {st.median(gaps):.1f} us")
        big = sorted(gap_by.items(), key=lambda kv: -st.mean(kv[1]))[:8]
        lines.append("   largest mean gaps (position in step, kernel behind the gap, mean gap us, median, p90):")
        for (k, n), v in big:
            v2 = sorted(v)
            lines.append(f"     #{k:3d} {n:60s} {st.mean(v):7.2f} {st.median(v):7.2f} {v2[9 * len(v2) // 10]:7.2f}")
        a, b = mid[len(mid) // 2], mid[len(mid) // 2 + 1]
        lines.append("   one step, kernel by kernel (offset us from the previous sampler's start | gap before | duration):")
        t0, prev = rows[a][1], rows[a][2]
        for n, s, e in rows[a + 1: b + 1]:
            lines.append(f"     {(s - t0) / 1e3:8.2f} | {(s - prev) / 1e3:6.2f} | {(e - s) / 1e3:6.2f} | {short(n)}")
            prev = e
    if len(per_stream) == 2:
        (s1, r1), (s2, r2) = per_stream.items()
        lo = max(r1[len(r1) // 4][1], r2[len(r2) // 4][1])
        hi = min(r1[3 * len(r1) // 4][2], r2[3 * len(r2) // 4][2])
        ev = []
        for rows in (r1, r2):
            for _, s, e in rows:
                if e > lo and s < hi:
                    ev.append((max(s, lo), 1)); ev.append((min(e, hi), -1))
        ev.sort()
        t_prev, depth, acc = lo, 0, {0: 0, 1: 0, 2: 0}
        for t, d in ev:
            acc[min(depth, 2)] += t - t_prev
            t_prev, depth = t, depth + d
        tot = hi - lo
        lines.append(f"## both chains over the middle half of the run ({tot / 1e6:.1f} ms): no kernel in flight {100 * acc[0] / tot:.1f} %, "
                     f"one chain {100 * acc[1] / tot:.1f} %, both {100 * acc[2] / tot:.1f} %")
    txt = "\n".join(lines)
    if out:
        open(out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
