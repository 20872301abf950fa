#!/bin/bash
# Dev tool (GPU box): SQ counter passes over single conv launches. usage: tools/pmc_conv.sh <tag> <one_conv.py args...>
set -e
ROOT=$(pwd); TAG=$1; shift
OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d "$OUT/p$i" -o p --output-format csv -- python3 "$ROOT/tools/one_conv.py" "$@" > "$OUT/p$i.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob(sys.argv[1] + "/p*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "k_conv" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
