#!/bin/bash
# Dev (GPU box): per-kernel SQ counters of the Winograd kernels (and their direct twins), layer by layer -> gpurun_out/<tag>/wino_pmc.txt
#   tools/wino_pmc.sh <tag> [B]
# One rocprofv3 --kernel-trace --pmc pass per counter group (<= 8 SQ counters / 2 GRBM per pass; the program directly after `--`).
# Units (MI355X_MICROARCH.md, "rocprofv3 PMC slots" / constants table): SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count QUAD-cycles summed
# over waves; SQ_BUSY_CYCLES quad-cycles per SE... ; SQ_VALU_MFMA_BUSY_CYCLES and SQ_VALU_MFMA_COEXEC_CYCLES count cycles per SIMD.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; OUT="$ROOT/gpurun_out/${1:-wino_pmc}"; B=${2:-128}; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d "$OUT/p$i" -o p --output-format csv -- python3 "$ROOT/tools/wino_pmc_run.py" "$OUT/labels_p$i.json" $B 4 > "$OUT/p$i.out" 2> "$OUT/p$i.err"
  echo "pass $i ($set): rc=$?"
done
python3 "$ROOT/tools/wino_pmc_summary.py" "$OUT" | tee "$OUT/wino_pmc.txt"
