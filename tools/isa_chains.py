#!/usr/bin/env python
"""Find serialized loads in the compiled kernels: for every kernel of every translation unit under prmf_amd/csrc, the longest
chain of `global_load (x1..3) ; s_waitcnt vmcnt(0)` repeated back to back in the gfx950 assembly.  Such a chain is what the
compiler makes of loads that sit behind per-element branches (every load gets an exec-masked block of its own and a full
wait after it): N dependent memory round trips where the source reads like N loads in flight.  Round 3 found the Gram
reduction job of the sweep prologue (13 at C3), the multi-rank assemble kernel (35..128) and a staging loop this way.
    python tools/isa_chains.py [MIN_CHAIN=4]       (needs hipcc; no GPU)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src_dir = os.path.join(ROOT, "prmf_amd", "csrc")
min_chain = int(sys.argv[1]) if len(sys.argv) > 1 else 4
res = []
with tempfile.TemporaryDirectory() as tmp:
    for f in sorted(os.listdir(src_dir)):
        if not f.endswith(".hip"):
            continue
        asm = os.path.join(tmp, f + ".s")
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I/opt/rocm/include",
                            "-S", "--cuda-device-only", os.path.join(src_dir, f), "-o", asm], capture_output=True, text=True)
        if r.returncode != 0:
            print("hipcc failed for", f, r.stderr[-300:])
            continue
        s = open(asm).read()
        for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\.Lfunc_end", s, flags=re.S | re.M):
            ev = []
            for line in m.group(2).split("\n"):
                line = line.strip()
                t = line.split(" ")[0]
                if t.startswith("global_load") or t.startswith("buffer_load"):
                    ev.append("L")
                elif t == "s_waitcnt" and "vmcnt(0)" in line:
                    ev.append("W")
            st = "".join(ev)
            longest = max([x.group(0).count("W") for x in re.finditer(r"(?:L{1,3}W){3,}", st)], default=0)
            if longest >= min_chain:
                res.append((longest, f, m.group(1)))
res.sort(reverse=True)
for n, f, name in res:
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    print("%4d  %-14s %s" % (n, f, dn[:120]))
if not res:
    print("no chain of %d or more" % min_chain)
