#!/usr/bin/env python
"""HBM traffic per kernel from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected separately, with
--kernel-trace only, as MI355X_MICROARCH.md prescribes).  Counter values are KiB; gfx950 correction for wide coalesced
streams: FETCH_SIZE tallies 64 B per 128-B request -> corrected fetch bytes = 2 * FETCH_SIZE * 1024.  WRITE_SIZE is
uncalibrated and reported raw.

  rocpd_pmc.py FETCH.db WRITE.db OUT.txt [KERNEL_SUBSTRING OUT.json WORKLOAD_NOTE [N_CHAINS ALG_STEP_BYTES]]

With N_CHAINS the JSON also carries the step-level figure: corrected fetch + write bytes of ALL decode-step kernels (dec::*,
dec_sample_kernel) per token step (token steps = dec_sample_kernel launches / N_CHAINS), and its ratio to ALG_STEP_BYTES.
"""
import json
import sqlite3
import sys


def short(name: str) -> str:
    return name.replace("mh::(anonymous namespace)::", "").replace("mh::dec::", "dec::").replace("void ", "")[:100]


def per_kernel(db, counter):
    con = sqlite3.connect(db)
    rows = con.execute("select name, count(*), avg(counter_value) from pmc_events where counter_name = ? group by name",
                       (counter,)).fetchall()
    return {n: (c, v) for n, c, v in rows}


def main(argv):
    fetch, write, out = per_kernel(argv[1], "FETCH_SIZE"), per_kernel(argv[2], "WRITE_SIZE"), argv[3]
    lines = ["# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only)",
             "# units: counter values are KiB; gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 64 B per 128-B",
             "# request on wide coalesced streams -> corrected_fetch_bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE raw * 1024.",
             "kernel | launches | FETCH_SIZE avg (KiB) | corrected fetch MB | WRITE_SIZE avg (KiB) | write MB"]
    for n, (c, f) in sorted(fetch.items(), key=lambda kv: -kv[1][1] * kv[1][0]):
        w = write.get(n, (0, 0.0))[1]
        lines.append(f"{short(n)} | {c} | {f:.1f} | {2 * f * 1024 / 1e6:.2f} | {w:.1f} | {w * 1024 / 1e6:.3f}")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:14]))
    if len(argv) > 5:
        key = [n for n in fetch if argv[4] in n]
        assert len(key) == 1, key
        c, f = fetch[key[0]]
        w = write.get(key[0], (0, 0.0))[1]
        js = {"kernel": argv[4], "fetch_size_kib_avg": f, "write_size_kib_avg": w,
              "hbm_bytes_per_launch": int(round(2 * f * 1024 + w * 1024)), "launches": c,
              "correction": "2 x FETCH_SIZE x 1024 (gfx950 wide-stream half-count) + WRITE_SIZE x 1024",
              "workload": argv[6] if len(argv) > 6 else ""}
        if len(argv) > 7:
            n_chains = int(argv[7])
            dec = [n for n in fetch if "dec::" in n or "dec_sample_kernel" in n]
            samp = [n for n in dec if "dec_sample_kernel" in n]
            steps = sum(fetch[n][0] for n in samp) / n_chains
            tot = sum(fetch[n][0] * (2 * fetch[n][1] * 1024 + write.get(n, (0, 0.0))[1] * 1024) for n in dec)
            js["step"] = {"decode_chains": n_chains, "token_steps": steps, "decode_kernels": len(dec),
                          "launches_per_token_step": sum(fetch[n][0] for n in dec) / steps,
                          "hbm_bytes_per_token_step": int(round(tot / steps))}
            if len(argv) > 8:
                js["step"]["alg_bytes_per_token_step"] = int(argv[8])
                js["step"]["traffic_ratio"] = round(tot / steps / int(argv[8]), 4)
        json.dump(js, open(argv[5], "w"), indent=1)
        print(js)


if __name__ == "__main__":
    main(sys.argv)
