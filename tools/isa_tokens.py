"""Dev tool: token-stream view of a kernel's gfx950 assembly, one line per basic block:
    BL / GL = buffer / global load, ST = store, Wn = s_waitcnt vmcnt(n), d*k = k LDS instructions, M*k = k MFMAs, BAR = barrier,
    B:label = branch.
This is how the MFMA kernels were checked for loads that wait on each other (a `BL W0 BL W0` chain, or a block per load): see
DESIGN.md "Straight-line loaders and epilogues".

    python tools/isa_tokens.py conv_igemm.hip 'k_conv_dgrad<1, 1, 1, 3, false, false>'
"""
import os
import re
import subprocess
import sys
import tempfile

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "openset-imagenet_amd", "csrc")


def main():
    src, pat = sys.argv[1], sys.argv[2]
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", out, src],
                       check=True, stderr=subprocess.DEVNULL, cwd=CSRC)
        txt = open(out).read()
    for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)s_endpgm", txt, re.S | re.M):
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        if pat not in name:
            continue
        toks = []
        for line in m.group(2).split("\n"):
            t = line.strip()
            if t.startswith(".LBB"):
                toks.append("\n" + t.split(":")[0])
            elif "global_load" in t:
                toks.append("GL")
            elif "buffer_load" in t:
                toks.append("BL")
            elif "global_store" in t or "buffer_store" in t or "scratch_store" in t:
                toks.append("ST")
            elif t.startswith("s_waitcnt") and "vmcnt" in t:
                toks.append("W" + re.search(r"vmcnt\((\d+)\)", t).group(1))
            elif t.startswith("s_cbranch") or t.startswith("s_branch"):
                toks.append("B:" + t.split()[-1])
            elif "v_mfma" in t:
                toks.append("M")
            elif t.startswith("s_barrier"):
                toks.append("BAR")
            elif t.startswith("ds_"):
                toks.append("d")
        s = " ".join(toks)
        s = re.sub(r"(M )+", lambda x: "M*%d " % (len(x.group(0)) // 2), s)
        s = re.sub(r"(d )+", lambda x: "d*%d " % (len(x.group(0)) // 2), s)
        print("=====", name)
        print("\n".join(l for l in s.split("\n") if len(l.split()) > 1))


if __name__ == "__main__":
    main()
