#!/bin/bash
# Dev (GPU box): bench.py with several builds of libosi_hip.so swapped in place, interleaved (A B C A B C ...), same box.
#   tools/ab_many.sh rounds libA.so libB.so [libC.so ...] [-- bench args...]
cd "$(dirname "$0")/.."
R=$1; shift
LIBS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ "$1" == "--" ] && shift
L=openset-imagenet_amd/csrc/libosi_hip.so
cp $L /tmp/libosi_hip_keep.so
for r in $(seq $R); do
  for v in "${LIBS[@]}"; do
    cp $v $L
    echo -n "$(basename $v)  "
    python bench.py --steps 20 --warmup 10 --no-cpu-baseline --sustained-steps 0 --eval-steps 0 "$@" 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'])" || { cp /tmp/libosi_hip_keep.so $L; exit 1; }
  done
done
cp /tmp/libosi_hip_keep.so $L
