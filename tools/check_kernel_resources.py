#!/usr/bin/env python
"""Reads the gfx950 code objects out of libmapperhip.so (every translation unit's offload bundle inside .hip_fatbin), lists each
kernel's registers / LDS / scratch / spill counts from the AMDGPU metadata note, and FAILS when a kernel that issues its LDS or
global loads from inline asm with hand-counted `s_waitcnt`s has scratch or spills.

Why (DESIGN 4 `gemm_s3g_kernel`): hipcc believes an inline-asm output register valid the moment the asm statement
has issued; if it spills or copies such a register before the hand-placed wait, the kernel reads bytes that have not arrived.
That produced wrong results once (the 256-row bf16x3 form).  The kernels named in ASM_LOAD_KERNELS must therefore compile to
ZERO scratch and ZERO spilled registers -- checked at build time (`__graft_entry__.build()`) and in the CPU test suite.

One refinement: scratch traffic that sits entirely BEHIND the kernel's last MFMA (checked in the disassembly) is an epilogue
spill -- every asm-loaded fragment has been consumed by then, nothing asm-issued is in flight -- and is reported but allowed
(the 256 x 256 bf16 tile keeps 128 accumulators in AGPRs and has 128 VGPRs for everything else: some of its epilogues park an
address pair in scratch).

usage: check_kernel_resources.py [lib.so] [--all]        (exit status 1 on a violation)"""
from __future__ import annotations

import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
# demangled-name fragments of the kernels whose loads are issued by inline asm and covered by counted waits
ASM_LOAD_KERNELS = ("gemm_glds3_kernel", "gemm_s3g_kernel", "gemm_mx8_kernel", "gemm_glds4_kernel", "gemm_glds2s_kernel")


def code_objects(lib_path: str):
    """-> list of gfx950 ELF images (bytes) found in the library's .hip_fatbin section"""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib_path, fat], check=True)
        blob = open(fat, "rb").read()
    out, pos = [], 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            break
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, idlen = struct.unpack_from("<QQQ", blob, q)
            ident = blob[q + 24: q + 24 + idlen].decode()
            q += 24 + idlen
            if "amdgcn" in ident and size:
                out.append(blob[pos + off: pos + off + size])
        pos += len(MAGIC)
    return out


def kernels_of(elf: bytes):
    """-> the `amdhsa.kernels` records (dicts) of one code object"""
    import yaml
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(elf)
        f.flush()
        txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name], check=True, capture_output=True, text=True).stdout
    out = []
    for doc in re.findall(r"^\s*---\s*$(.*?)^\s*\.\.\.\s*$", txt, flags=re.S | re.M):
        meta = yaml.safe_load(doc) or {}
        out += meta.get("amdhsa.kernels", [])
    return out


def scratch_only_behind_last_mfma(elf: bytes, symbol: str) -> bool:
    """True iff every scratch_* instruction of kernel `symbol` comes after its last v_mfma instruction (program order)"""
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(elf)
        f.flush()
        txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", f"--disassemble-symbols={symbol}", f.name],
                             check=True, capture_output=True, text=True).stdout
    ins = [ln.split("//")[0].strip() for ln in txt.splitlines() if "\t" in ln]
    mf = [i for i, ln in enumerate(ins) if ln.startswith("v_mfma")]
    sc = [i for i, ln in enumerate(ins) if ln.startswith("scratch_")]
    return bool(mf) and bool(sc) and min(sc) > max(mf)


def demangle(names):
    """c++filt when the box has one; mangled names contain the kernel's plain name anyway, which is all the policy matches on"""
    import shutil
    tool = shutil.which("c++filt") or shutil.which(os.path.join(LLVM, "llvm-cxxfilt"))
    if not tool:
        return list(names)
    p = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True, check=True)
    return p.stdout.splitlines()


def main(argv):
    show_all = "--all" in argv
    paths = [a for a in argv if not a.startswith("--")]
    lib = paths[0] if paths else os.path.join(ROOT, "mapperatorinator_amd", "lib", "libmapperhip.so")
    rows, owner = [], []
    for elf in code_objects(lib):
        ks = kernels_of(elf)
        rows += ks
        owner += [elf] * len(ks)
    if not rows:
        print(f"{lib}: no gfx950 kernels found", file=sys.stderr)
        return 2
    names = demangle([r[".symbol"].removesuffix(".kd") for r in rows])
    bad, n_checked = [], 0
    n_epi = 0
    for r, nm, elf in zip(rows, names, owner):
        scratch = int(r.get(".private_segment_fixed_size", 0))
        spills = int(r.get(".vgpr_spill_count", 0)) + int(r.get(".sgpr_spill_count", 0))
        watched = any(w in nm for w in ASM_LOAD_KERNELS)
        n_checked += watched
        if watched and (scratch or spills) and scratch_only_behind_last_mfma(elf, r[".symbol"].removesuffix(".kd")):
            n_epi += 1
            if show_all:
                print(f"ep vgpr {r.get('.vgpr_count', '?'):>3} agpr {r.get('.agpr_count', '?'):>3} scratch {scratch:>5} spills {spills:>3}  (behind the last MFMA) {nm[:120]}")
            continue
        if show_all or (watched and (scratch or spills)):
            print(f"{'!!' if watched and (scratch or spills) else '  '} vgpr {r.get('.vgpr_count', '?'):>3} agpr {r.get('.agpr_count', '?'):>3} "
                  f"sgpr {r.get('.sgpr_count', '?'):>3} lds {r.get('.group_segment_fixed_size', '?'):>6} scratch {scratch:>5} spills {spills:>3}  {nm[:150]}")
        if watched and (scratch or spills):
            bad.append(nm)
    print(f"{len(rows)} kernels, {n_checked} with asm-issued loads checked for scratch / spills: {'FAIL ' + str(len(bad)) if bad else 'ok'}"
          + (f" ({n_epi} with epilogue-only spills behind their last MFMA)" if n_epi else ""))
    return 1 if bad else 0


if __name__ == "__main__":
    # exit status: 0 ok, 1 policy violation (a watched kernel spills / uses scratch), 2 no gfx950 kernels in the library,
    # 3 the tooling itself failed (PyYAML, llvm-objcopy / -readelf / -objdump missing or erroring): never to be read as "a kernel spills"
    try:
        rc = main(sys.argv[1:])
    except Exception:
        import traceback
        traceback.print_exc()
        rc = 3
    sys.exit(rc)
