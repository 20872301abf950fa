#!/bin/bash
# Dev tool, run ON the GPU box (through gpurun) from the repo root: collects the rocprofv3 evidence bench.py's roofline cites.
#   1. --kernel-trace --stats of the default bench command (per-kernel durations)
#   2. --pmc FETCH_SIZE, --pmc WRITE_SIZE (HBM-side bytes) and --pmc MfmaUtil (matrix-pipe busy %) in separate passes, kernel-trace only
# Outputs land under gpurun_out/; tools/summarize_profiles.py turns them into the files committed under profiles/.
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${1:-prof_final}
mkdir -p "$OUT"
echo "${OSI_COMMIT:-unknown}" > "$OUT/commit.txt"
note() { echo "$(date +%T) $*" >> "$OUT/progress.log"; }   # a line per stage: the run is never silent for minutes
note start
python3 "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --sustained-steps 0 --eval-steps 0 > "$OUT/bench_plain.json" 2> "$OUT/bench_plain.err"
note "plain bench done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o s --output-format csv -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --sustained-steps 0 --eval-steps 0 > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
note "kernel stats done"
# the same with the weight gradients kept on the main stream (no co-running kernels): per-kernel averages comparable with the
# event-timed per-class figures of bench.py's roofline leg
OSI_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d "$OUT/stats_serial" -o s --output-format csv -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --sustained-steps 0 --eval-steps 0 > "$OUT/bench_under_rocprof_serial.json" 2> "$OUT/stats_serial.err"
note "serialized kernel stats done"
for c in FETCH_SIZE WRITE_SIZE MfmaUtil; do
  note "pmc $c"
  rocprofv3 --kernel-trace --pmc $c -d "$OUT/$c" -o p --output-format csv -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --sustained-steps 0 --eval-steps 0 > "$OUT/bench_$c.json" 2> "$OUT/$c.err"
done
# matrix instructions actually issued (the Winograd layers issue 4/9 of the direct convolution's multiplies): busy cycles x 64 FLOP per
# SIMD cycle = issued MFMA FLOPs, whatever the instruction shape (32x32x2 and 16x16x4 both run 64 FLOP per cycle and SIMD)
note "pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA -d "$OUT/MfmaIssued" -o p --output-format csv -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --sustained-steps 0 --eval-steps 0 > "$OUT/bench_MfmaIssued.json" 2> "$OUT/MfmaIssued.err"
# keep what tools/summarize_profiles.py reads; the per-dispatch traces of the stats runs are large
find "$OUT" -name "*_kernel_trace.csv" -path "*stats*" -delete 2>/dev/null || true
note done
echo done
