#!/bin/bash
# GPU box: the effective shader clock of long fp32-MFMA dispatches by two independent methods
#   (a) in-kernel: s_memtime / s_memrealtime (printed by `conv_ablate long`)
#   (b) out-of-kernel: GRBM_GUI_ACTIVE / 8 / dispatch duration from rocprofv3 (MI355X_MICROARCH.md, 'DVFS give-back')
# usage: bash tools/clock_xcheck.sh <outdir under gpurun_out>
set -e
OUT=gpurun_out/${1:-clock_xcheck}
mkdir -p $OUT
export TMPDIR=/tmp
mkdir -p build && hipcc -O3 --offload-arch=gfx950 tools/probes/conv_ablate.hip -o build/conv_ablate 2>/dev/null
./build/conv_ablate long > $OUT/plain.txt
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -- ./build/conv_ablate long > $OUT/under_pmc.txt 2> $OUT/pmc.err
find $OUT/pmc -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $OUT/counters.csv
find $OUT/pmc -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $OUT/kernels.csv
rm -rf $OUT/pmc
cat $OUT/plain.txt $OUT/under_pmc.txt
