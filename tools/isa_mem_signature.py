#!/usr/bin/env python
"""Memory-event signature of the decode kernels' gfx950 ISA: for every kernel whose mangled name contains one of the
given substrings, the sequence of vector loads (L), counted waits (Wn = s_waitcnt vmcnt(n)), stores (S), scalar loads
(s) and barriers (|) in TEXTUAL order, runs compressed (L8 = eight loads back to back).  A decode kernel is a handful of
memory round trips; hipcc sometimes sinks loads next to their uses (`L W0 L W0 ...` = one round trip per load), which
costs ~1 us each beside the other chain's K/V stream -- this tool is how such schedules are found after an edit.
(Loops are rotated by the compiler: textual order is not always execution order; read the .s next to it.)

  python tools/isa_mem_signature.py [-DMACRO=..] [--file t5.hip] substr [substr ...]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    defs = [a for a in sys.argv[1:] if a.startswith("-D")]
    args = [a for a in sys.argv[1:] if not a.startswith("-D")]
    src = "t5.hip"
    if "--file" in args:
        i = args.index("--file")
        src = args[i + 1]
        del args[i:i + 2]
    keys = args or ["gemv_kernelItLi1E", "cross_attn_q_kernelItLi6ELi1ELb0", "self_attn_qkv_kernelItLi6", "dec_sample_kernelIt"]
    with tempfile.TemporaryDirectory() as td:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-pass-failed",
               "-mllvm", "-amdgpu-kernarg-preload-count=14", *defs, "-save-temps", "-c",
               os.path.join(ROOT, "mapperatorinator_amd", "csrc", src), "-o", os.path.join(td, "x.o")]
        subprocess.run(cmd, cwd=td, check=True, stderr=subprocess.DEVNULL)
        asm = [f for f in os.listdir(td) if f.endswith("gfx950.s")][0]
        s = open(os.path.join(td, asm)).read()
        keep = os.environ.get("ISA_KEEP")
        if keep:
            open(keep, "w").write(s)
    for f in re.split(r"\n(?=_Z\w+:)", s):
        name = f.split(":", 1)[0]
        if not any(k in name for k in keys):
            continue
        ev = []
        for l in f.split("\n"):
            l = l.strip()
            if not l or l.startswith(";"):
                continue
            if l.startswith(("global_load", "buffer_load")):
                ev.append("L")
            elif l.startswith("s_waitcnt") and "vmcnt" in l:
                ev.append("W" + re.search(r"vmcnt\((\d+)\)", l).group(1))
            elif l.startswith(("global_store", "global_atomic", "buffer_store")):
                ev.append("S")
            elif l.startswith("s_barrier"):
                ev.append("|")
            elif l.startswith("s_load"):
                ev.append("s")
            elif l.startswith("s_endpgm"):
                break
        out, prev, cnt = [], None, 0
        for e in ev + [None]:
            if e == prev:
                cnt += 1
            else:
                if prev:
                    out.append(prev + (str(cnt) if cnt > 1 else ""))
                prev, cnt = e, 1
        m = re.search(r"\.vgpr_count:\s+(\d+)", s[s.find(".name:           " + name):][:3000]) if (".name:           " + name) in s else None
        print(name[:96] + (f"   [vgpr {m.group(1)}]" if m else ""))
        print("    " + " ".join(out))


if __name__ == "__main__":
    main()
