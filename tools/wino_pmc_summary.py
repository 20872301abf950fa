#!/usr/bin/env python3
"""Dev: per-layer table of the counter passes of tools/wino_pmc.sh. Dispatches of a kernel family are matched to the labels the driver
wrote (k-th dispatch of the family = k-th label), warm-ups dropped, counters averaged over the repetitions of a label.

Derived columns (per launch, MI355X: 256 CUs x 4 SIMDs):
  us            kernel duration under the counter pass (End - Start of the dispatch)
  clk           GRBM_GUI_ACTIVE / 8 / duration (GHz): the clock the chip held
  mfma_busy     SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GUI cycles / 8): share of the launch the matrix pipes were busy
  coexec        SQ_VALU_MFMA_COEXEC_CYCLES / SQ_VALU_MFMA_BUSY_CYCLES: share of the matrix-busy cycles in which a vector instruction issued too
  valu_active   4 x SQ_ACTIVE_INST_VALU / (1024 x GUI cycles / 8): share of the launch a SIMD's vector issue was active (quad-cycles -> cycles;
                with ONE wave per SIMD, as in the Winograd kernels, the wave's own share; MFMA issue counts as VALU issue here)
  wait / stall / issue   SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES: parked at a waitcnt / barrier, stalled at
                issue (pipe busy, dependency), issuing
  valu/mfma     (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA: ordinary vector instructions per matrix instruction
  serial_us     the kernel's SERIAL-PIPE floor: an fp32 matrix instruction (v_mfma_f32_32x32x2_f32: 64 pipe cycles) and an ordinary vector
                instruction never execute together on a SIMD (co-execution counter exactly 0; tools/probes/mfma_valu_coexec.hip), so a SIMD
                needs at least 64 cycles per matrix instruction PLUS the issue time of its vector instructions — 4 cycles each with one
                wave per SIMD (the Winograd kernels), 2 cycles each once several waves alternate (the direct kernels) — at the clock held
  of_floor      serial_us / us
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

src = sys.argv[1]
fam_of = lambda n: next((f for f in ("k_wino_wgrad<", "k_wino<", "k_conv_fwd<", "k_conv_dgrad<") if f in n), None)
short = lambda n: re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))
acc = defaultdict(lambda: defaultdict(list))     # label -> counter -> values
dur = defaultdict(list)
kname = {}
for i in range(1, 16):
    lp = os.path.join(src, f"labels_p{i}.json")
    if not os.path.exists(lp):
        continue
    labels = json.load(open(lp))["labels"]
    files = glob.glob(os.path.join(src, f"p{i}", "**", "p_counter_collection.csv"), recursive=True)
    rows = [r for f in files for r in csv.DictReader(open(f))]
    per_disp = defaultdict(dict)
    meta = {}
    for r in rows:
        fam = fam_of(r["Kernel_Name"])
        if fam is None or "fixup" in r["Kernel_Name"] or "tail" in r["Kernel_Name"]:
            continue
        did = int(r["Dispatch_Id"])
        per_disp[did][r["Counter_Name"]] = per_disp[did].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        meta[did] = (fam, short(r["Kernel_Name"]), float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    seq = defaultdict(list)
    for did in sorted(per_disp):
        seq[meta[did][0]].append(did)
    for fam, dids in seq.items():
        lab = labels.get(fam, [])
        if len(dids) != len(lab):
            print(f"# pass {i}: {fam} has {len(dids)} dispatches but {len(lab)} labels — skipped", file=sys.stderr)
            continue
        for did, l in zip(dids, lab):
            if l.endswith("(warm-up)"):
                continue
            kname[l] = meta[did][1]
            dur[l].append(meta[did][2])
            for c, v in per_disp[did].items():
                acc[l][c].append(v)
print("Per-kernel SQ counters of the Winograd kernels and their direct twins, B = 128 (tools/wino_pmc.sh: four rocprofv3 --kernel-trace --pmc passes over tools/wino_pmc_run.py)")
print(__doc__.strip().split("\n\n")[1])
print()
hdr = f"{'layer':22s} {'kernel':44s} {'us':>7s} {'clk':>5s} {'mfma_busy':>9s} {'coexec':>7s} {'valu_act':>8s} {'wait':>6s} {'stall':>6s} {'issue':>6s} {'valu/mfma':>9s} {'serial_us':>9s} {'of_floor':>8s}"
print(hdr)
mean = lambda v: sum(v) / len(v) if v else None
order = [f"{k} {C}@{H}" for k in ("fwd", "fwd-direct", "dgrad", "dgrad-direct", "wgrad") for C, H in ((64, 56), (128, 28), (256, 14), (512, 7))]
out = {}
for l in order:
    if l not in acc:
        continue
    c = {k: mean(v) for k, v in acc[l].items()}
    us = mean(dur[l]) / 1e3
    gui = c.get("GRBM_GUI_ACTIVE")
    cyc = gui / 8 if gui else None                       # GUI_ACTIVE is summed over the 8 XCDs
    f = lambda x, fmt="{:.3f}": "-" if x is None else fmt.format(x)
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES")
    wc = c.get("SQ_WAVE_CYCLES")
    row = {"us": us, "clock_GHz": cyc / (us * 1e3) if cyc else None,
           "mfma_busy": busy / (1024 * cyc) if busy is not None and cyc else None,
           "coexec_over_busy": c["SQ_VALU_MFMA_COEXEC_CYCLES"] / busy if busy and "SQ_VALU_MFMA_COEXEC_CYCLES" in c else None,
           "valu_active": 4 * c["SQ_ACTIVE_INST_VALU"] / (1024 * cyc) if cyc and "SQ_ACTIVE_INST_VALU" in c else None,
           "wait": c["SQ_WAIT_ANY"] / wc if wc and "SQ_WAIT_ANY" in c else None,
           "stall": c["SQ_WAIT_INST_ANY"] / wc if wc and "SQ_WAIT_INST_ANY" in c else None,
           "issue": c["SQ_ACTIVE_INST_ANY"] / wc if wc and "SQ_ACTIVE_INST_ANY" in c else None,
           "valu_per_mfma": (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"] if c.get("SQ_INSTS_MFMA") and "SQ_INSTS_VALU" in c else None,
           "raw_per_launch": {k: round(v, 1) for k, v in sorted(c.items())}}
    lone = kname[l].startswith("k_wino")          # one wave per SIMD: a vector instruction costs the wave's full 4-cycle issue
    if c.get("SQ_INSTS_MFMA") and "SQ_INSTS_VALU" in c and row["clock_GHz"]:
        cyc_simd = (64.0 * c["SQ_INSTS_MFMA"] + (4.0 if lone else 2.0) * (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"])) / 1024.0
        row["serial_floor_us"] = cyc_simd / (row["clock_GHz"] * 1e3)
        row["of_serial_floor"] = row["serial_floor_us"] / us
    else:
        row["serial_floor_us"] = row["of_serial_floor"] = None
    out[l] = dict(row, kernel=kname[l])
    print(f"{l:22s} {kname[l][:44]:44s} {us:7.1f} {f(row['clock_GHz'], '{:.2f}'):>5s} {f(row['mfma_busy']):>9s} {f(row['coexec_over_busy']):>7s} "
          f"{f(row['valu_active']):>8s} {f(row['wait']):>6s} {f(row['stall']):>6s} {f(row['issue']):>6s} {f(row['valu_per_mfma'], '{:.2f}'):>9s} "
          f"{f(row['serial_floor_us'], '{:.1f}'):>9s} {f(row['of_serial_floor']):>8s}")
json.dump(out, open(os.path.join(src, "wino_pmc.json"), "w"), indent=1)
