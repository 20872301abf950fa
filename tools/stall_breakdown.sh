#!/bin/bash
# Dev tool (GPU box): where do the wave-cycles of the conv kernels go? SQ counter passes (4 counters each, kernel-trace only) over the
# serialized bench step, aggregated per kernel instantiation.
#   tools/stall_breakdown.sh <tag>        -> gpurun_out/<tag>/stall_breakdown.json (+ the raw CSVs)
# Units (MI355X_MICROARCH.md, rocprofv3 PMC slots): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_BUSY_CYCLES count quad-cycles summed
# over waves (resp. over SEs / CUs); SQ_VALU_MFMA_BUSY_CYCLES counts cycles. WAIT_ANY (parked at s_waitcnt / barrier) + WAIT_INST_ANY
# (issue stall: pipe busy, dependency) + ACTIVE_INST_ANY (issuing) ~= WAVE_CYCLES.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-stall}; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" \
           "SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_WAVES" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  OSI_NO_OVERLAP=1 rocprofv3 --kernel-trace --pmc $set -d "$OUT/p$i" -o p --output-format csv -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --windows 1 --no-cpu-baseline --no-profile --sustained-steps 0 > "$OUT/bench_p$i.json" 2> "$OUT/p$i.err"
  echo "pass $i ($set): rc=$?"
done
python3 "$ROOT/tools/stall_summary.py" "$OUT"
