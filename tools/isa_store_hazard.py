#!/usr/bin/env python3
"""Scan the gfx950 ISA of every kernel source for the hazard that corrupted the stem's dy2 in round 2/3:

    buffer_store_dwordx3/x4 with a SCALAR-REGISTER offset, followed within 2 wait states by a VALU write of one of its data registers.

hipcc pads this write-after-read only when the store has no SGPR offset (it applies the rule of the older GCN parts), so such a
sequence overwrites the data before the store has read it.  Prints every occurrence; exit status 1 if any.

    python tools/isa_store_hazard.py            # compiles csrc/*.hip to ISA (about a minute)
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "panoswintransformerobjectdetection_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "--cuda-device-only", "-S"]
STORE = re.compile(r"^\s*buffer_store_dwordx[34]\s+v\[(\d+):(\d+)\],\s*(?:v\d+|off),\s*s\[\d+:\d+\],\s*(s\d+|\d+|0x[0-9a-f]+)")
VDST = re.compile(r"^\s*(v_\w+)\s+v(?:\[(\d+):(\d+)\]|(\d+))")


def scan(path):
    bad = 0
    lines = open(path).read().split("\n")
    kern = "?"
    for i, ln in enumerate(lines):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            kern = m.group(1)
        st = STORE.match(ln)
        if not st or not st.group(3).startswith("s"):
            continue
        lo, hi = int(st.group(1)), int(st.group(2))
        states, j = 0, i + 1
        while j < len(lines) and states < 2:
            t = lines[j].strip()
            j += 1
            if not t or t.startswith((";", ".")) or t.endswith(":"):
                continue
            if t.startswith("s_nop"):
                states += int(t.split()[1]) + 1
                continue
            w = VDST.match(t)
            if w and not t.startswith(("v_cmp", "v_mfma")):
                a, b = (int(w.group(2)), int(w.group(3))) if w.group(2) else (int(w.group(4)), int(w.group(4)))
                if a <= hi and b >= lo:
                    print(f"{os.path.basename(path)}: {kern[:70]} line {i + 1}: `{ln.strip()}` then `{t}` after {states} wait state(s)")
                    bad += 1
                    break
            states += 1
    return bad


def main():
    bad = 0
    with tempfile.TemporaryDirectory() as d:
        procs = []
        for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
            out = os.path.join(d, os.path.basename(src) + ".s")
            procs.append((out, subprocess.Popen(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", out, src], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)))
        for out, p in procs:
            p.wait()
            if os.path.exists(out):
                bad += scan(out)
    print("occurrences:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
