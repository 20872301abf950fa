#!/bin/bash
# Dev (GPU box): the stream-K cut of the Winograd units' ragged last round — knob wino_streamk 0 = never, 1 = both directions,
# 2 = forward only (default), 3 = input gradient only — interleaved on one box.   tools/ab_wino_streamk.sh [rounds] [bench args...]
cd "$(dirname "$0")/.."
R=${1:-3}; shift
for r in $(seq $R); do for k in 2 1 0 3; do
  echo -n "wino_streamk=$k  "
  OSI_WINO_STREAMK=$k python bench.py --steps 20 --warmup 10 --no-cpu-baseline --sustained-steps 0 --eval-steps 0 "$@" 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); pc = d['roofline']['per_class']; print(d['ms_per_step'], d['value'], 'fwd', pc['conv_fwd']['ms_per_step'], 'dgrad', pc['conv_dgrad']['ms_per_step'], 'wgrad', pc['conv_wgrad']['ms_per_step'])" || exit 1
done; done
