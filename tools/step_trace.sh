#!/bin/bash
# Dev tool (GPU box, via gpurun): per-dispatch kernel trace of ONE serialized training step (weight gradients on the main stream),
# in launch order: gpurun_out/<tag>_step_trace.txt = "start_us dur_us grid wg name".   tools/step_trace.sh [tag]
ROOT=$(pwd); TAG=${1:-r05}; OUT=$ROOT/gpurun_out/step_trace_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
OSI_NO_OVERLAP=1 rocprofv3 --kernel-trace -d "$OUT" -o t --output-format csv -- python3 "$ROOT/bench.py" --steps 2 --warmup 2 --no-cpu-baseline --sustained-steps 0 > "$OUT/bench.json" 2> "$OUT/err.txt"
python3 - "$OUT" "$ROOT/gpurun_out/${TAG}_step_trace.txt" <<'PY'
import csv, sys, glob, re
f = glob.glob(sys.argv[1] + "/**/t_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if "k_adam" in r["Kernel_Name"]]
a, b = ad[-2] + 1, ad[-1] + 1          # the last complete step: after one optimizer launch up to and including the next
t0 = int(rows[a]["Start_Timestamp"])
with open(sys.argv[2], "w") as o:
    for r in rows[a:b]:
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        n = re.sub(r"\(.*", "", n).replace("void ", "")
        o.write(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:10.1f} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:8.1f} '
                f'{r["Grid_Size_X"]:>8}x{r["Grid_Size_Y"]}x{r["Grid_Size_Z"]} {r["Workgroup_Size_X"]:>4} {n}\n')
print("wrote", sys.argv[2], b - a, "dispatches")
PY
rm -rf "$OUT"
