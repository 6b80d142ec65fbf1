#!/usr/bin/env python3
"""Static check of a kernel's vector-memory waits: compile a .hip source for gfx950, pick one kernel by a substring of
its mangled name and print, for every stretch between two s_barrier, the `s_waitcnt vmcnt(N)` the compiler emitted
together with how many MFMAs / loads / stores had been issued since the barrier.

Why: `vmcnt` counts loads AND stores of a wave in issue order, and hipcc's N in front of the consumer of a load is an
upper bound computed over merged control flow -- often far below the number of YOUNGER operations, so the wait also
covers stores issued after the load.  In score_ws_kernel's helper waves that made every iteration sit out its own
sixteen score stores (`vmcnt(5..0)` in front of the staging ds_writes); moving the stores behind
those ds_writes was worth 3 %.  The bf16 kernel's unrolled loop gets exact counts (`vmcnt(20) .. vmcnt(14)`).

    python tools/isa_waits.py r-tucker_amd/csrc/rtk_score_ws.hip score_ws_kernelILi13ELi2E [-DNAME=VALUE ...]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src, pick, defs = sys.argv[1], sys.argv[2], sys.argv[3:]
    with tempfile.TemporaryDirectory() as d:
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include", f"-I{ROOT}/r-tucker_amd/csrc",
               *defs, "-c", os.path.abspath(src), "-o", os.path.join(d, "x.o"), "-save-temps=obj"]
        subprocess.run(cmd, check=True, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        asm = [f for f in os.listdir(d) if f.endswith(".s") and "amdgcn" in f][0]
        s = open(os.path.join(d, asm)).read()
    names = [n for n in re.findall(r"\.amdhsa_kernel (\S+)", s) if pick in n]
    if not names:
        sys.exit(f"no kernel matches {pick!r}")
    name = names[0]
    blk = s[s.index(".amdhsa_kernel " + name):]
    blk = blk[:blk.index(".end_amdhsa_kernel")]
    st = s.index("\n" + name + ":")
    body = s[st:s.index(".Lfunc_end", st)]
    vgpr = re.search(r"next_free_vgpr (\d+)", blk).group(1)
    scratch = re.search(r"private_segment_fixed_size (\d+)", blk).group(1)
    print(f"{name}\n  vgpr {vgpr}  scratch {scratch} B  {len(body.splitlines())} lines")
    m = ld = stc = 0
    waits = []
    for line in body.split("\n"):
        t = line.split(";")[0].strip()
        if not t:
            continue
        if t.startswith("s_barrier"):
            print(f"  barrier after {m} mfma, {ld} loads, {stc} stores; vmcnt waits since the previous one: {' '.join(waits) or '-'}")
            m = ld = stc = 0
            waits = []
        elif t.startswith("v_mfma"):
            m += 1
        elif re.match(r"(global|buffer)_load", t):
            ld += 1
        elif re.match(r"(global|buffer)_store", t):
            stc += 1
        elif t.startswith("s_waitcnt") and "vmcnt" in t:
            n = re.search(r"vmcnt\(\d+\)", t).group(0)
            waits.append(f"{n}@{m}m/{ld}l/{stc}s")


if __name__ == "__main__":
    main()
