#!/bin/bash
# Dev tool (GPU box): derived rocprofv3 counters for every conv launch of two bench steps, one counter per pass (kernel-trace only).
#   tools/pmc_layer_probe.sh <tag> "MemUnitStalled WriteUnitStalled ..."
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$1; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in $2; do
  OSI_NO_OVERLAP=1 rocprofv3 --kernel-trace --pmc $c -d "$OUT/$c" -o p --output-format csv -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --windows 1 --no-cpu-baseline --no-profile --sustained-steps 0 > "$OUT/bench_$c.json" 2> "$OUT/$c.err"
done
python3 - "$OUT" $2 <<'PY'
import csv, sys, os, glob
from collections import defaultdict
out = sys.argv[1]
for c in sys.argv[2:]:
    f = glob.glob(os.path.join(out, c, "**", "p_counter_collection.csv"), recursive=True)
    if not f: print(c, "no output"); continue
    acc = defaultdict(lambda: [0.0, 0.0, 0])
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != c: continue
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        if "k_conv" not in n and "k_bn_apply" not in n and "k_bn_bwd_apply" not in n: continue
        key = (n[:52], r["Grid_Size"])
        dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        a = acc[key]; a[0] += float(r["Counter_Value"]) * dur; a[1] += dur; a[2] += 1
    print("==", c)
    for k, a in sorted(acc.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f"{k[0]:52s} grid {k[1]:>9s} calls {a[2]:3d} avg_us {a[1]/a[2]/1e3:8.1f}  {c} {a[0]/a[1]:8.2f}")
PY
