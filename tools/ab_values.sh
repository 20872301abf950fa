#!/bin/bash
# One environment switch, several values, interleaved on ONE box: tools/ab_values.sh VAR "v1 v2 v3" [rounds]
var=$1; vals=$2; rounds=${3:-2}
for r in $(seq 1 $rounds); do
  for v in $vals; do
    env $var=$v python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); pc=d['roofline']['per_class']
print('$var=$v', 'ms/step', d['ms_per_step'], d['windows_ms_per_step'], 'fwd', pc['conv_fwd']['ms_per_step'], 'dgrad', pc['conv_dgrad']['ms_per_step'], 'wgrad', pc['conv_wgrad']['ms_per_step'], 'bn', pc['bn_fwd']['ms_per_step'], pc['bn_bwd']['ms_per_step'])"
  done
done
