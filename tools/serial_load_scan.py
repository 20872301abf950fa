"""Dev tool: compile every kernel file to gfx950 assembly and list the loops that wait for a single (or two) global / buffer loads per
iteration - the signature of a latency chain (load -> s_waitcnt vmcnt(0) -> add) the compiler would not pipeline because the trip
count is a run-time value. Remainder loops show up too; read the source before acting on a line.

    python tools/serial_load_scan.py            # run from the repo root
"""
import os
import re
import subprocess
import sys
import tempfile

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "openset-imagenet_amd", "csrc")


def scan(path):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", out, path],
                       check=True, stderr=subprocess.DEVNULL, cwd=CSRC)
        txt = open(out).read()
    for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)s_endpgm", txt, re.S | re.M):
        toks = []
        for line in m.group(2).split("\n"):
            t = line.strip()
            if t.startswith(".LBB"):
                toks.append(t.split(":")[0])
            elif "global_load" in t or "buffer_load" in t:
                toks.append("L")
            elif t.startswith("s_waitcnt") and "vmcnt" in t:
                toks.append("W" + re.search(r"vmcnt\((\d+)\)", t).group(1))
            elif t.startswith("s_cbranch"):
                toks.append("B:" + t.split()[-1])
        for lm in re.finditer(r"(\.LBB\d+_\d+) ((?:(?!\.LBB)[^ ]+ )*?)B:\1", " ".join(toks)):
            inner = lm.group(2).split()
            if 0 < inner.count("L") <= 2 and "W0" in inner:
                name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
                print(f"{os.path.basename(path):18s} {name[:100]:100s} loop: {' '.join(inner)[:50]}")


if __name__ == "__main__":
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".hip") and (len(sys.argv) < 2 or sys.argv[1] in f):
            scan(os.path.join(CSRC, f))
