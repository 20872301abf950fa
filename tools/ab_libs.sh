#!/bin/bash
# Dev (GPU box): bench.py with two builds of libosi_hip.so swapped in place, interleaved (A B A B ...), same box.
#   tools/ab_libs.sh path/to/libA.so path/to/libB.so [rounds] [bench args...]
cd "$(dirname "$0")/.."
A=$1; B=$2; R=${3:-3}; shift 3
L=openset-imagenet_amd/csrc/libosi_hip.so
cp $L /tmp/libosi_hip_keep.so
for r in $(seq $R); do
  for v in $A $B; do
    cp $v $L
    echo -n "$(basename $v)  "
    python bench.py --steps 20 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'])" || { cp /tmp/libosi_hip_keep.so $L; exit 1; }
  done
done
cp /tmp/libosi_hip_keep.so $L
