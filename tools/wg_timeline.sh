#!/bin/bash
# Dev tool (GPU box): per-workgroup timelines (tools/wg_timeline.py, diagnostic stamps build) of the launches that own the step.
#   tools/wg_timeline.sh <tag>  -> gpurun_out/<tag>/*.json + summary.txt
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-wgtl}; mkdir -p "$OUT"
run() { name=$1; shift; timeout -k 10 120 python3 tools/wg_timeline.py "$@" --json "$OUT/$name.json" > "$OUT/$name.log" 2>&1 || echo "$name FAILED" >> "$OUT/summary.txt"; }
run fwd_act_3x3_256_h14      fwd_act 256 256 3 1 14
run fwd_act_1x1_256_1024_h14 fwd_act 256 1024 1 1 14
run fwd_1x1_1024_256_h14     fwd 1024 256 1 1 14
run fwd_act_1x1_64_256_h56   fwd_act 64 256 1 1 56
run fwd_act_3x3_64_h56       fwd_act 64 64 3 1 56
run fwd_1x1_256_64_h56       fwd 256 64 1 1 56
run fwd_act_3x3_128_h28      fwd_act 128 128 3 1 28
run fwd_act_3x3_512_h7       fwd_act 512 512 3 1 7
run dgrad_inblock_3x3_256_h14 dgrad_inblock 256 256 3 1 14
run dgrad_inblock_1x1_256_1024_h14 dgrad_inblock 256 1024 1 1 14
run dgrad_inblock_3x3_64_h56 dgrad_inblock 64 64 3 1 56
run wgrad_1x1_1024_256_h14   wgrad 1024 256 1 1 14
run wgrad_act_1x1_256_1024_h14 wgrad_act 256 1024 1 1 14
run wgrad_act_3x3_256_h14    wgrad_act 256 256 3 1 14
run wgrad_act_3x3_64_h56     wgrad_act 64 64 3 1 56
run wgrad_1x1_256_64_h56     wgrad 256 64 1 1 56
python3 - "$OUT" <<'PY' | tee -a "$OUT/summary.txt"
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    r = json.load(open(f)); s = r["share_of_cu_time"]; w = r["workgroup_us"]
    print(f"{os.path.basename(f)[:-5]:34s} {r['tflops_plain']:6.1f} TF/s {r['launch_us_plain']:7.1f} us  wg/cu {r['workgroups_per_cu']:6.2f}  work {s['mfma_work_at_2.4GHz']:.3f} ramp {s['ramp_before_first_loop']:.3f} tail {s['tail_after_last_loop']:.3f} between {s['between']:.3f} | "
          f"prologue {w['prologue']['median']:6.2f} loop {w['k_loop']['median']:7.2f} epilogue {w['epilogue']['median']:6.2f} us | cu finish {r['cu_finish_us']}")
    print("    in_k_loop/cu over time:", r["per_cu_over_time"]["in_k_loop"])
PY
