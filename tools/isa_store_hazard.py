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
sys.path.insert(0, ROOT)
from panoswintransformerobjectdetection_amd.build import FLAGS as BUILD_FLAGS, SOURCES  # noqa: E402  (the library's own flags and source list)

FLAGS = [f for f in BUILD_FLAGS if f != "-fPIC"] + ["--cuda-device-only", "-S"]
ANY_STORE = re.compile(r"^\s*buffer_store_dwordx[34]\s")
# kernel sources known to issue 128-bit buffer stores: a scan that sees none of them in one of these did not scan that file
EXPECT_STORES = ("pswin_stem.hip", "pswin_qkvattn.hip", "pswin_fused.hip", "pswin_attn.hip")
STORE = re.compile(r"^\s*buffer_store_dwordx[34]\s+v\[(\d+):(\d+)\],\s*(?:v\d+|off),\s*s\[\d+:\d+\],\s*(s\d+|\d+|0x[0-9a-f]+)")
VDST = re.compile(r"^\s*(v_\w+)\s+v(?:\[(\d+):(\d+)\]|(\d+))")


def scan(path, stats=None):
    bad = 0
    lines = open(path).read().split("\n")
    kern = "?"
    if stats is not None:
        stats[os.path.basename(path)[:-2]] = sum(1 for ln in lines if ANY_STORE.match(ln))
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
    """0: every source of the library compiled and scanned, no occurrence; 1: occurrences; 2: the scan itself is not trustworthy (a
    source did not compile, an ISA file is missing, or a file expected to contain 128-bit buffer stores shows none)."""
    bad, broken, stats = 0, 0, {}
    with tempfile.TemporaryDirectory() as d:
        procs = []
        srcs = [os.path.join(CSRC, f) for f in SOURCES]
        extra = sorted(set(glob.glob(os.path.join(CSRC, "*.hip"))) - set(srcs))
        for src in srcs + extra:
            out = os.path.join(d, os.path.basename(src) + ".s")
            procs.append((src, out, subprocess.Popen(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", out, src], stdout=subprocess.PIPE,
                                                     stderr=subprocess.STDOUT, text=True)))
        for src, out, p in procs:
            log, _ = p.communicate()
            if p.returncode != 0 or not os.path.exists(out):
                print(f"{os.path.basename(src)}: hipcc exit {p.returncode}, ISA {'missing' if not os.path.exists(out) else 'present'}\n{log}")
                broken += 1
                continue
            bad += scan(out, stats)
    for name in EXPECT_STORES:
        if name in stats and stats[name] == 0:
            print(f"{name}: no buffer_store_dwordx3/x4 seen -- the pattern no longer matches this compiler's output")
            broken += 1
        elif name not in stats:
            print(f"{name}: not scanned")
            broken += 1
    print("scanned:", {k: v for k, v in sorted(stats.items())})
    print("occurrences:", bad)
    return 2 if broken else (1 if bad else 0)


if __name__ == "__main__":
    sys.exit(main())
