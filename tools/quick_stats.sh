#!/bin/bash
# Dev tool (GPU box): per-kernel durations of a short serialized bench run; prints the top kernels. tools/quick_stats.sh <tag> [filter]
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$1; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
OSI_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d "$OUT" -o s --output-format csv -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --windows 1 --no-cpu-baseline --no-profile --sustained-steps 0 > "$OUT/bench.json" 2> "$OUT/err.txt"
f=$(find "$OUT" -name "s_kernel_stats.csv" | head -1)
python3 - "$f" "${2:-}" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2]
for r in rows[:60] if not flt else rows:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    if flt and flt not in n: continue
    print(f"{n[:90]:90s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms {float(r['Percentage']):5.2f}%")
PY
