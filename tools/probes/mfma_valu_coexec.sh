#!/bin/bash
# Dev (GPU box): the co-execution probe, plain (timing table) and under the SQ counters -> gpurun_out/<tag>/coexec.txt
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"; OUT="$ROOT/gpurun_out/${1:-coexec}"; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
BIN="$ROOT/tools/probes/bin/mfma_valu_coexec"
[ -x "$BIN" ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 "$ROOT/tools/probes/mfma_valu_coexec.hip" -o "$BIN" 2>/dev/null || exit 1
"$BIN" 20000 > "$OUT/timing.txt" 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -d "$OUT/pmc" -o p --output-format csv -- "$BIN" 2000 > "$OUT/pmc.out" 2> "$OUT/pmc.err" || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d "$OUT/pmc2" -o p --output-format csv -- "$BIN" 2000 > "$OUT/pmc2.out" 2> "$OUT/pmc2.err" || exit 1
python3 - "$OUT" <<'PY' > "$OUT/coexec.txt"
import csv, glob, re, sys
from collections import defaultdict
out = sys.argv[1]
print(open(f"{out}/timing.txt").read())
acc = defaultdict(lambda: defaultdict(list))
for d in ("pmc", "pmc2"):
    for f in glob.glob(f"{out}/{d}/**/p_counter_collection.csv", recursive=True):
        per = defaultdict(dict)
        for r in csv.DictReader(open(f)):
            m = re.search(r"k_probe<(\d+), (true|false)>", r["Kernel_Name"])
            if not m:
                continue
            key = (int(r["Dispatch_Id"]), int(m.group(1)), m.group(2) == "true", int(r["Workgroup_Size"]))
            per[key][r["Counter_Name"]] = per[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        seen = defaultdict(int)
        for key in sorted(per):
            k = key[1:]
            seen[k] += 1
            if seen[k] <= 2:        # the two warm-up launches
                continue
            for c, v in per[key].items():
                acc[k][c].append(v)
print("SQ counters of the timed launch (2000 steps x 4): matrix-pipe busy cycles per matrix instruction, co-execution cycles / busy cycles, issue-stall share")
print(f"{'mode':14s} {'waves':>5s} {'K':>3s} {'busy/mfma':>9s} {'coexec/busy':>11s} {'valu/mfma':>9s} {'stall/wave':>10s}")
for (K, bf, wg) in sorted(acc, key=lambda k: (k[1], k[2], k[0])):
    c = {n: sum(v) / len(v) for n, v in acc[(K, bf, wg)].items()}
    mf = c.get("SQ_INSTS_MFMA", 0) or float("nan")
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", float("nan"))
    print(f"{'bf16 32x32x16' if bf else 'f32 32x32x2':14s} {wg // 256:5d} {K:3d} {busy / mf:9.1f} {c.get('SQ_VALU_MFMA_COEXEC_CYCLES', float('nan')) / busy:11.3f} "
          f"{(c.get('SQ_INSTS_VALU', float('nan')) - mf) / mf:9.2f} {c.get('SQ_WAIT_INST_ANY', float('nan')) / c.get('SQ_WAVE_CYCLES', float('nan')):10.3f}")
PY
cat "$OUT/coexec.txt"
