#!/bin/bash
# GPU box: effective shader clock per kernel class INSIDE the training step, from rocprofv3 --pmc GRBM_GUI_ACTIVE
# (GRBM_GUI_ACTIVE / 8 / dispatch duration; MI355X_MICROARCH.md 'DVFS give-back': reads high on dispatches under ~0.3 ms).
# Counter passes serialise the kernels, so this is the clock of each kernel running alone within a continuously busy step.
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${1:-clock_in_step}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d "$OUT/pmc" -o p --output-format csv -- python3 "$ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-profile --sustained-steps 0 > "$OUT/bench.json" 2> "$OUT/pmc.err"
f=$(find "$OUT/pmc" -name "*counter_collection.csv" | head -1)
python3 - "$f" > "$OUT/summary.txt" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
        continue
    dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    a = agg[name]
    a[0] += float(r["Counter_Value"]); a[1] += dur; a[2] += 1
print(f"{'kernel':60s} {'calls':>6s} {'avg us':>9s} {'GRBM_GUI_ACTIVE/8/dur GHz':>26s}")
for name, (cyc, dur, n) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"{name[:60]:60s} {n:6d} {dur / n / 1e3:9.1f} {cyc / 8 / dur:26.3f}")
PY
rm -rf "$OUT/pmc"
cat "$OUT/summary.txt"
