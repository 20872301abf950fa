"""Dev tool: turn the rocprofv3 output of tools/collect_profiles.sh (merged back under gpurun_out/) into the files under profiles/.

    python tools/summarize_profiles.py gpurun_out/prof_final r01

writes profiles/<tag>_bench_kernel_stats.csv, <tag>_bench_under_rocprof.json and <tag>_hbm_traffic_per_step.json.
HBM bytes follow MI355X_MICROARCH.md's recipe: FETCH_SIZE and WRITE_SIZE in separate passes, KiB units, fetch doubled on gfx950.
"""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict


def klass(name):
    # Winograd forms (round 5): the class is in the template arguments — k_wino<XF, EPI, ODD>, k_wino_fixup<EPI, ODD>, k_wino_weights<FLIP>
    if "k_wino_wgrad" in name:
        return "conv_wgrad"
    if "k_wino" in name:
        import re
        args = re.search(r"k_wino\w*<([^>]*)>", name)
        a = [x.strip() for x in args.group(1).split(",")] if args else []
        if "k_wino_weights" in name:
            epi = a[0] if a else "0"
        elif "k_wino_fixup" in name:
            epi = a[0] if a else "0"
        else:
            epi = a[1] if len(a) > 1 else "0"
        return "conv_dgrad" if epi == "1" else "conv_fwd"
    if "k_conv_fwd" in name or "k_stem_fwd" in name or "k_conv1x1_rows" in name:
        return "conv_fwd"
    if "k_conv_dgrad" in name:
        return "conv_dgrad"
    if "k_conv_wgrad" in name or "k_slab_reduce" in name or "k_stem" in name:
        return "conv_wgrad"
    if "k_bn_bwd" in name or "k_colsum2" in name:
        return "bn_bwd"
    if "k_bn_" in name:
        return "bn_fwd"
    if "k_adam" in name or "k_sgd" in name:
        return "optimizer"
    return "other"


def pmc_per_step(path):
    tot, steps = defaultdict(float), 0
    with open(path) as f:
        for row in csv.DictReader(f):
            n = row["Kernel_Name"]
            tot[klass(n)] += float(row["Counter_Value"])
            steps += "k_adam" in n
    return {k: v / steps for k, v in tot.items()}, steps


def mfma_util_per_class(path):
    """Duration-weighted mean of rocprofv3's derived MfmaUtil (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * SIMDs), in %)."""
    num, den = defaultdict(float), defaultdict(float)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != "MfmaUtil":
                continue
            dur = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
            k = klass(row["Kernel_Name"])
            num[k] += float(row["Counter_Value"]) * dur
            den[k] += dur
    return {k: round(num[k] / den[k], 2) for k in num if den[k] > 0}


def main():
    src, tag = sys.argv[1], sys.argv[2]
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
    shutil.copy(os.path.join(src, "stats", "s_kernel_stats.csv"), os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
    shutil.copy(os.path.join(src, "bench_under_rocprof.json"), os.path.join(dst, f"{tag}_bench_under_rocprof.json"))
    ser = os.path.join(src, "stats_serial", "s_kernel_stats.csv")
    if os.path.isfile(ser):
        shutil.copy(ser, os.path.join(dst, f"{tag}_bench_kernel_stats_serialized.csv"))
        shutil.copy(os.path.join(src, "bench_under_rocprof_serial.json"), os.path.join(dst, f"{tag}_bench_under_rocprof_serialized.json"))
    fetch, s1 = pmc_per_step(os.path.join(src, "FETCH_SIZE", "p_counter_collection.csv"))
    write, s2 = pmc_per_step(os.path.join(src, "WRITE_SIZE", "p_counter_collection.csv"))
    out = {"_note": "per training step (B=128, Protocol-2 workload), rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over "
                    f"`bench.py --steps 3 --warmup 1` ({s1}/{s2} steps seen, totals divided by the step count); counter units KiB; fetch doubled "
                    "per MI355X_MICROARCH.md (gfx950 counts 128-B wide reads as 64 B) - validated on the Adam kernel: 0.38 GB read / "
                    "0.285 GB written expected"}
    # provenance for bench.py's roofline.traffic: the workload / step time this profile belongs to (a plain, un-profiled run of the
    # same command in the same gpurun call) and the commit it was taken at
    plain = os.path.join(src, "bench_plain.json")
    if os.path.isfile(plain):
        line = [l for l in open(plain).read().splitlines() if l.startswith("{")][-1]
        b = json.loads(line)
        out["_meta"] = {"workload": b["config"].get("workload_key"), "batch": b["config"]["batch_per_gpu"], "ms_per_step": b["ms_per_step"],
                        "images_per_sec": b["value"], "commit": (open(os.path.join(src, "commit.txt")).read().strip()
                                                                  if os.path.isfile(os.path.join(src, "commit.txt")) else None)}
    for k in sorted(set(fetch) | set(write)):
        out[k] = {"fetch_GB_raw": round(fetch.get(k, 0) * 1024 / 1e9, 3),
                  "fetch_GB_x2_wide_read_correction": round(2 * fetch.get(k, 0) * 1024 / 1e9, 3),
                  "write_GB": round(write.get(k, 0) * 1024 / 1e9, 3)}
    # achieved HBM-side bandwidth per kernel class = PMC bytes / rocprofv3 kernel time of the serialized run (no co-running kernels)
    if os.path.isfile(ser):
        steps_ser, tms = 0, defaultdict(float)
        with open(ser) as fh:
            for row in csv.DictReader(fh):
                tms[klass(row["Name"])] += float(row["TotalDurationNs"]) / 1e6
                if "k_adam" in row["Name"]:
                    steps_ser = int(row["Calls"])
        for k in list(out):
            if not k.startswith("_") and steps_ser and tms.get(k):
                ms = tms[k] / steps_ser
                out[k]["kernel_ms_per_step_serialized"] = round(ms, 3)
                out[k]["achieved_TBps"] = round((out[k]["fetch_GB_x2_wide_read_correction"] + out[k]["write_GB"]) / ms, 3)
        out["_note"] += "; achieved_TBps = (2*fetch + write) / summed kernel time of that class in the serialized rocprofv3 run (peak 8.0 spec / ~6.3 measured)"
    mf = os.path.join(src, "MfmaUtil", "p_counter_collection.csv")
    if os.path.isfile(mf):
        util = mfma_util_per_class(mf)
        rec = {"_note": "rocprofv3 --pmc MfmaUtil over `bench.py --steps 3 --warmup 1` (kernels serialised by the counter "
                        "collection): duration-weighted mean per kernel class of 100 * SQ_VALU_MFMA_BUSY_CYCLES / "
                        "(GRBM_GUI_ACTIVE * SIMDs) — share of the launch during which the matrix pipe of a SIMD is busy",
               "_meta": out.get("_meta"), "mfma_busy_percent": util}
        # matrix FLOPs actually ISSUED per step: every cycle a SIMD's matrix pipe is busy retires 64 FLOP (v_mfma_f32_32x32x2_f32: 4096 FLOP
        # in 64 cycles; 16x16x4: 2048 in 32), so busy cycles x 64 counts the multiplies the kernels really executed — the Winograd layers'
        # 4/9, border tiles and ragged-tile padding included
        mi = os.path.join(src, "MfmaIssued", "p_counter_collection.csv")
        if os.path.isfile(mi):
            busy, insts, steps = defaultdict(float), defaultdict(float), 0
            seen = set()
            with open(mi) as fh:
                for row in csv.DictReader(fh):
                    k = klass(row["Kernel_Name"])
                    if row["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
                        busy[k] += float(row["Counter_Value"])
                    elif row["Counter_Name"] == "SQ_INSTS_MFMA":
                        insts[k] += float(row["Counter_Value"])
                    if "k_adam" in row["Kernel_Name"] and row["Dispatch_Id"] not in seen:
                        seen.add(row["Dispatch_Id"]); steps += 1
            if steps:
                rec["issued_gflop_per_step_measured"] = {k: round(busy[k] * 64 / steps / 1e9, 1) for k in ("conv_fwd", "conv_dgrad", "conv_wgrad")}
                rec["issued_gflop_per_step_measured"]["total"] = round(sum(rec["issued_gflop_per_step_measured"].values()), 1)
                rec["mfma_instructions_per_step"] = {k: round(insts[k] / steps) for k in ("conv_fwd", "conv_dgrad", "conv_wgrad")}
                rec["_note"] += ("; issued_gflop_per_step_measured = SQ_VALU_MFMA_BUSY_CYCLES x 64 FLOP per busy SIMD cycle, per step "
                                 f"({steps} steps seen): the matrix work really executed (Winograd 4/9, border tiles, tile padding included)")
        with open(os.path.join(dst, f"{tag}_mfma_util.json"), "w") as f:
            json.dump(rec, f, indent=1)
        print("MfmaUtil", util)
    with open(os.path.join(dst, f"{tag}_hbm_traffic_per_step.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
