#!/bin/bash
# A/B of one environment switch on ONE box, interleaved: tools/ab_bench.sh OSI_TAIL_SPLIT 0 1 [rounds]
# prints ms_per_step (median of three windows) and the serialized per-class times of every run
var=$1; a=$2; b=$3; rounds=${4:-2}
for r in $(seq 1 $rounds); do
  for v in $a $b; do
    env $var=$v python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); pc=d['roofline']['per_class']
print('$var=$v', 'ms/step', d['ms_per_step'], d['windows_ms_per_step'], 'fwd', pc['conv_fwd']['ms_per_step'], 'dgrad', pc['conv_dgrad']['ms_per_step'], 'wgrad', pc['conv_wgrad']['ms_per_step'], 'bn', pc['bn_fwd']['ms_per_step'], pc['bn_bwd']['ms_per_step'])"
  done
done
