"""Dev tool: aggregate the PMC passes of tools/stall_breakdown.sh per kernel instantiation -> <dir>/stall_breakdown.json.

Every counter is summed over the launches of an instantiation (all passes ran the same command, so launch counts agree) and
reported per launch plus as the ratios that answer "where do the wave-cycles go":
  wait_any / wait_inst / active   share of SQ_WAVE_CYCLES a wave is parked at s_waitcnt or a barrier / stalled at issue / issuing
  mfma_busy                       SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x duration x clock): the matrix pipe's busy share
  per_mfma                        VALU / LDS / VMEM / SALU instructions issued per MFMA instruction (the loop's overhead mix)
  waves_per_simd                  SQ_LEVEL_WAVES / SQ_BUSY_CYCLES / 4 (mean resident waves while the SQ is busy)
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)


def main():
    src = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(int))
    dur = defaultdict(lambda: [0.0, 0])
    for f in sorted(glob.glob(os.path.join(src, "p*", "**", "p_counter_collection.csv"), recursive=True)):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if not k.startswith("k_"):
                continue
            c = r["Counter_Name"]
            acc[k][c] += float(r["Counter_Value"])
            calls[k][c] += 1
            key = (r["Dispatch_Id"], k)
            if key not in seen:
                seen.add(key)
                d = dur[k]
                d[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); d[1] += 1
    out = {}
    for k, cs in acc.items():
        n = max(calls[k].values())
        per = {c: v / calls[k][c] for c, v in cs.items()}
        avg_ns = dur[k][0] / max(1, dur[k][1])
        e = {"launches_seen": n, "avg_us_under_pmc": round(avg_ns / 1e3, 2), "per_launch": {c: round(v, 1) for c, v in sorted(per.items())}}
        wc = per.get("SQ_WAVE_CYCLES")
        if wc:
            e["share_of_wave_cycles"] = {name: round(per[c] / wc, 4) for name, c in
                                         (("wait_any", "SQ_WAIT_ANY"), ("wait_inst_any", "SQ_WAIT_INST_ANY"), ("active_inst_any", "SQ_ACTIVE_INST_ANY"),
                                          ("wait_inst_lds", "SQ_WAIT_INST_LDS"), ("active_valu", "SQ_ACTIVE_INST_VALU"), ("active_lds", "SQ_ACTIVE_INST_LDS"),
                                          ("active_vmem", "SQ_ACTIVE_INST_VMEM"), ("active_scalar", "SQ_ACTIVE_INST_SCA"), ("active_misc", "SQ_ACTIVE_INST_MISC"))
                                         if c in per}
        m = per.get("SQ_INSTS_MFMA")
        if m:
            e["instructions_per_mfma"] = {name: round(per[c] / m, 3) for name, c in
                                          (("valu_incl_mfma", "SQ_INSTS_VALU"), ("lds", "SQ_INSTS_LDS"), ("vmem_rd", "SQ_INSTS_VMEM_RD"),
                                           ("vmem_wr", "SQ_INSTS_VMEM_WR"), ("salu", "SQ_INSTS_SALU"), ("smem", "SQ_INSTS_SMEM")) if c in per}
        if "GRBM_GUI_ACTIVE" in per and avg_ns:
            e["clock_GHz"] = round(per["GRBM_GUI_ACTIVE"] / 8 / avg_ns, 3)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in per:
                e["mfma_busy_share"] = round(per["SQ_VALU_MFMA_BUSY_CYCLES"] / (per["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)
        if "SQ_LEVEL_WAVES" in per and per.get("SQ_BUSY_CYCLES"):
            e["level_waves_over_busy_cycles"] = round(per["SQ_LEVEL_WAVES"] / per["SQ_BUSY_CYCLES"], 3)
        if "SQ_LDS_IDX_ACTIVE" in per and per.get("SQ_BUSY_CU_CYCLES"):
            e["lds_active_over_busy_cu_cycles"] = round(per["SQ_LDS_IDX_ACTIVE"] / per["SQ_BUSY_CU_CYCLES"], 4)
        out[k] = e
    order = sorted(out, key=lambda k: -out[k]["avg_us_under_pmc"] * out[k]["launches_seen"])
    res = {"_note": __doc__.strip().splitlines()[0] + " Command per pass: OSI_NO_OVERLAP=1 rocprofv3 --kernel-trace --pmc <4 counters> -- "
                    "python3 bench.py --steps 2 --warmup 1 --windows 1 --no-cpu-baseline --no-profile", "kernels": {k: out[k] for k in order}}
    json.dump(res, open(os.path.join(src, "stall_breakdown.json"), "w"), indent=1)
    for k in order[:14]:
        e = out[k]
        print(f"{k[:60]:60s} n={e['launches_seen']:4d} {e['avg_us_under_pmc']:8.1f} us  mfma_busy {e.get('mfma_busy_share')}  "
              f"{e.get('share_of_wave_cycles')}  per-mfma {e.get('instructions_per_mfma')}")


if __name__ == "__main__":
    main()
