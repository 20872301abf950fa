#!/bin/bash
# Dev (GPU box): kernel statistics of the inference forward, both modes -> gpurun_out/eval_forward/{fused,topology}_stats.csv + a summary.
#   tools/eval_forward.sh [steps]     (then copy gpurun_out/eval_forward/summary.txt to profiles/)
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; OUT="$ROOT/gpurun_out/eval_forward"; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
STEPS=${1:-20}
for m in topology fused; do
  python3 "$ROOT/tools/eval_forward.py" $m $STEPS > "$OUT/${m}_plain.json" 2> "$OUT/${m}_plain.err" || exit 1
  rocprofv3 --kernel-trace --stats -d "$OUT/$m" -o s --output-format csv -- python3 "$ROOT/tools/eval_forward.py" $m $STEPS > "$OUT/${m}_rocprof.json" 2> "$OUT/${m}.err" || exit 1
  cp "$(find "$OUT/$m" -name 's_kernel_stats.csv' | head -1)" "$OUT/${m}_kernel_stats.csv"
done
python3 - "$OUT" $STEPS <<'PY' > "$OUT/summary.txt"
import csv, json, sys
out, steps = sys.argv[1], int(sys.argv[2])
print("Inference forward (validate() / get_arrays(), reference train.py:142-234) at B = 128, 224 x 224, C = 30; rocprofv3 --kernel-trace --stats")
for m in ("topology", "fused"):
    plain = json.load(open(f"{out}/{m}_plain.json")); prof = json.load(open(f"{out}/{m}_rocprof.json"))
    rows = list(csv.DictReader(open(f"{out}/{m}_kernel_stats.csv")))
    n = steps + 3
    tot = sum(float(r["TotalDurationNs"]) for r in rows) / n / 1e6
    calls = sum(int(r["Calls"]) for r in rows) / n
    print(f"\n== {m}: {plain['images_per_sec']:.0f} img/s, {plain['ms_per_batch']:.3f} ms per batch of {plain['batch']} (plain run; {prof['ms_per_batch']:.3f} ms under the tracer); "
          f"kernels {tot:.3f} ms and {calls:.0f} launches per forward; checksum {plain['logit_checksum']:.6f}")
    print(f"{'kernel':96s} {'calls/fwd':>9s} {'ms/fwd':>8s} {'avg us':>8s}")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
        print(f"{r['Name'][:96]:96s} {int(r['Calls']) / n:9.1f} {float(r['TotalDurationNs']) / n / 1e6:8.3f} {float(r['AverageNs']) / 1e3:8.1f}")
PY
cat "$OUT/summary.txt"
