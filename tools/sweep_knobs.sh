#!/bin/bash
# Dev tool (GPU box): A/B the development knobs on the whole training step. usage: tools/sweep_knobs.sh <out-file> "ENV=val ENV2=val" "..." ...
OUT=$1; shift
: > "$OUT"
for cfg in "$@"; do
  for rep in 1 2; do
    env $cfg python3 bench.py --no-cpu-baseline --no-profile --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', 'rep$rep', b['value'], b['ms_per_step'])" >> "$OUT"
  done
done
cat "$OUT"
